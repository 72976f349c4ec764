// Inter-GPU transport of the distributed layer: one process per GPU, RCCL over
// xGMI between them.
//
// The reference is single-process: its only multi-GPU notion is "replica i uses
// GPU gpu_id" (/root/reference/wf/win_seq_gpu.hpp:231-237 picks a device per
// replica; SURVEY.md §2.6 lists no inter-GPU communication at all).  On MI355X
// nodes the GPUs are fully connected by point-to-point xGMI links, so the natural
// scale-out of a keyed stream is: every rank ingests its own sources, tuples are
// re-partitioned by key ACROSS ranks with one all-to-all per exchange round, and
// the keyed state of a key lives on exactly one GPU.  This header is the thin
// communication substrate of that design:
//
//   Transport        what the exchange operator needs (small host-visible
//                    all-gather, device all-to-all-v, all-reduce, barrier)
//   RcclTransport    ncclComm over xGMI; all-to-all-v = one grouped
//                    ncclSend/ncclRecv per peer pair, so every link carries its
//                    pair's slice concurrently (no ring: xGMI is point-to-point)
//   LocalTransport   ranks are threads of one process sharing a LocalFabric;
//                    used by the tests (several ranks on the one GPU of a test
//                    box) and by single-process multi-GPU graphs
#pragma once
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../gpu/hip_utils.hpp"

namespace wf {
namespace dist {

constexpr int MAX_WORLD = 16;

#define WF_RCCL_CHECK(expr)                                                                                         \
    do {                                                                                                            \
        ncclResult_t _r = (expr);                                                                                   \
        if (_r != ncclSuccess) {                                                                                    \
            std::fprintf(stderr, "WindFlow RCCL error: %s at %s:%d in %s\n", ncclGetErrorString(_r), __FILE__, __LINE__, #expr); \
            std::abort();                                                                                           \
        }                                                                                                           \
    } while (0)

// per-peer slices of one all-to-all-v (device pointers, byte counts)
struct PeerSlices {
    const void* send[MAX_WORLD] = {};
    size_t send_bytes[MAX_WORLD] = {};
    void* recv[MAX_WORLD] = {};
    size_t recv_bytes[MAX_WORLD] = {};
};

class Transport {
public:
    virtual ~Transport() = default;
    int rank() const { return rank_; }
    int world() const { return world_; }
    virtual const char* name() const = 0;
    // every rank contributes `words` 64-bit words; on return `out` (host memory,
    // world*words words, rank-major) holds everybody's contribution.  Ordered
    // after the work already enqueued on `s`.
    virtual void all_gather_host(const uint64_t* mine, uint64_t* out, size_t words, hipStream_t s) = 0;
    // device all-to-all with per-peer sizes, enqueued on `s` (returns without waiting)
    virtual void all_to_all_v(const PeerSlices& p, hipStream_t s) = 0;
    // element-wise sum of `n` doubles in device memory, in place, enqueued on `s`
    virtual void all_reduce_sum(double* dev, size_t n, hipStream_t s) = 0;
    virtual void barrier(hipStream_t s) = 0;

protected:
    int rank_ = 0, world_ = 1;
};

// ---- RCCL -------------------------------------------------------------------------
class RcclTransport : public Transport {
public:
    static constexpr size_t ID_BYTES = sizeof(ncclUniqueId);
    // rank 0 creates the id and ships it to the other ranks out of band
    // (torch.distributed / a file / the launcher's environment)
    static std::string create_id() {
        ncclUniqueId id;
        WF_RCCL_CHECK(ncclGetUniqueId(&id));
        return std::string(reinterpret_cast<const char*>(&id), sizeof(id));
    }
    RcclTransport(int rank, int world, const std::string& id_bytes, int device) : device_(device) {
        if (world < 1 || world > MAX_WORLD) gpu::failwith("RcclTransport: world size out of range");
        if (id_bytes.size() != sizeof(ncclUniqueId)) gpu::failwith("RcclTransport: malformed unique id");
        rank_ = rank;
        world_ = world;
        ncclUniqueId id;
        std::memcpy(&id, id_bytes.data(), sizeof(id));
        WF_HIP_CHECK(hipSetDevice(device_));
        WF_RCCL_CHECK(ncclCommInitRank(&comm_, world, id, rank));
        stage_dev_.resize(STAGE_WORDS * (size_t)(world + 1));
        stage_host_.resize(STAGE_WORDS * (size_t)(world + 1));
    }
    ~RcclTransport() override {
        if (comm_) (void)ncclCommDestroy(comm_);
    }
    const char* name() const override { return "rccl"; }

    void all_gather_host(const uint64_t* mine, uint64_t* out, size_t words, hipStream_t s) override {
        if (words > STAGE_WORDS) gpu::failwith("RcclTransport::all_gather_host: too many words");
        std::lock_guard<std::mutex> l(m_);
        uint64_t* h_send = stage_host_.data();
        uint64_t* h_recv = stage_host_.data() + STAGE_WORDS;
        uint64_t* d_send = stage_dev_.data();
        uint64_t* d_recv = stage_dev_.data() + STAGE_WORDS;
        std::memcpy(h_send, mine, words * sizeof(uint64_t));
        WF_HIP_CHECK(hipMemcpyAsync(d_send, h_send, words * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        WF_RCCL_CHECK(ncclAllGather(d_send, d_recv, words, ncclUint64, comm_, s));
        WF_HIP_CHECK(hipMemcpyAsync(h_recv, d_recv, words * sizeof(uint64_t) * (size_t)world_, hipMemcpyDeviceToHost, s));
        WF_HIP_CHECK(hipStreamSynchronize(s));
        std::memcpy(out, h_recv, words * sizeof(uint64_t) * (size_t)world_);
    }

    void all_to_all_v(const PeerSlices& p, hipStream_t s) override {
        std::lock_guard<std::mutex> l(m_);
        WF_RCCL_CHECK(ncclGroupStart());
        for (int peer = 0; peer < world_; peer++) {
            if (p.send_bytes[peer] > 0) WF_RCCL_CHECK(ncclSend(p.send[peer], p.send_bytes[peer], ncclChar, peer, comm_, s));
            if (p.recv_bytes[peer] > 0) WF_RCCL_CHECK(ncclRecv(p.recv[peer], p.recv_bytes[peer], ncclChar, peer, comm_, s));
        }
        WF_RCCL_CHECK(ncclGroupEnd());
    }

    void all_reduce_sum(double* dev, size_t n, hipStream_t s) override {
        std::lock_guard<std::mutex> l(m_);
        WF_RCCL_CHECK(ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, comm_, s));
    }

    void barrier(hipStream_t s) override {
        uint64_t x = 1, out[MAX_WORLD];
        all_gather_host(&x, out, 1, s);
    }

private:
    static constexpr size_t STAGE_WORDS = 64;
    int device_;
    ncclComm_t comm_ = nullptr;
    std::mutex m_;  // one collective at a time per communicator
    gpu::DeviceBuffer<uint64_t> stage_dev_;
    gpu::PinnedBuffer<uint64_t> stage_host_;
};

// rank 0 writes the id to `path` (atomically), the others poll for it: bootstrap for
// native programs started by a launcher that only provides RANK / WORLD_SIZE
inline std::string bootstrap_id_via_file(const std::string& path, int rank, int timeout_s = 120) {
    if (rank == 0) {
        const std::string id = RcclTransport::create_id();
        const std::string tmp = path + ".tmp";
        {
            std::ofstream f(tmp, std::ios::binary);
            f.write(id.data(), (std::streamsize)id.size());
        }
        std::rename(tmp.c_str(), path.c_str());
        return id;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s);
    while (std::chrono::steady_clock::now() < deadline) {
        std::ifstream f(path, std::ios::binary);
        if (f) {
            std::string id((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            if (id.size() == RcclTransport::ID_BYTES) return id;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    gpu::failwith("bootstrap_id_via_file: timed out waiting for " + path);
}

// ---- in-process fabric ---------------------------------------------------------------
// Shared by the `world` LocalTransports of one process.
class LocalFabric {
public:
    explicit LocalFabric(int world) : world_(world), gather_((size_t)world * 64), slices_(world), reduce_(world) {
        if (world < 1 || world > MAX_WORLD) gpu::failwith("LocalFabric: world size out of range");
    }
    int world() const { return world_; }
    // sense-reversing barrier
    void sync() {
        std::unique_lock<std::mutex> l(m_);
        const uint64_t gen = gen_;
        if (++arrived_ == world_) {
            arrived_ = 0;
            gen_++;
            cv_.notify_all();
        } else {
            cv_.wait(l, [&] { return gen_ != gen; });
        }
    }

private:
    friend class LocalTransport;
    int world_;
    std::mutex m_;
    std::condition_variable cv_;
    int arrived_ = 0;
    uint64_t gen_ = 0;
    std::vector<uint64_t> gather_;
    std::vector<PeerSlices> slices_;
    std::vector<std::pair<double*, size_t>> reduce_;
};

class LocalTransport : public Transport {
public:
    LocalTransport(std::shared_ptr<LocalFabric> fabric, int rank) : fabric_(std::move(fabric)) {
        rank_ = rank;
        world_ = fabric_->world();
    }
    const char* name() const override { return "local"; }

    void all_gather_host(const uint64_t* mine, uint64_t* out, size_t words, hipStream_t) override {
        if (words > 64) gpu::failwith("LocalTransport::all_gather_host: too many words");
        std::memcpy(&fabric_->gather_[(size_t)rank_ * 64], mine, words * sizeof(uint64_t));
        fabric_->sync();
        for (int r = 0; r < world_; r++) std::memcpy(out + (size_t)r * words, &fabric_->gather_[(size_t)r * 64], words * sizeof(uint64_t));
        fabric_->sync();
    }

    void all_to_all_v(const PeerSlices& p, hipStream_t s) override {
        WF_HIP_CHECK(hipStreamSynchronize(s));  // my slices are complete before anybody reads them
        fabric_->slices_[rank_] = p;
        fabric_->sync();
        for (int peer = 0; peer < world_; peer++) {
            const PeerSlices& q = fabric_->slices_[peer];
            if (q.send_bytes[rank_] != p.recv_bytes[peer]) gpu::failwith("LocalTransport::all_to_all_v: size mismatch between ranks");
            if (p.recv_bytes[peer] > 0) WF_HIP_CHECK(hipMemcpyAsync(p.recv[peer], q.send[rank_], p.recv_bytes[peer], hipMemcpyDefault, s));
        }
        WF_HIP_CHECK(hipStreamSynchronize(s));
        fabric_->sync();  // senders may now reuse their buffers
    }

    void all_reduce_sum(double* dev, size_t n, hipStream_t s) override {
        WF_HIP_CHECK(hipStreamSynchronize(s));
        fabric_->reduce_[rank_] = {dev, n};
        fabric_->sync();
        std::vector<double> acc(n, 0.0), tmp(n);
        for (int r = 0; r < world_; r++) {
            WF_HIP_CHECK(hipMemcpyAsync(tmp.data(), fabric_->reduce_[r].first, n * sizeof(double), hipMemcpyDefault, s));
            WF_HIP_CHECK(hipStreamSynchronize(s));
            for (size_t i = 0; i < n; i++) acc[i] += tmp[i];
        }
        fabric_->sync();  // everybody has read the inputs
        WF_HIP_CHECK(hipMemcpyAsync(dev, acc.data(), n * sizeof(double), hipMemcpyDefault, s));
        WF_HIP_CHECK(hipStreamSynchronize(s));
        fabric_->sync();
    }

    void barrier(hipStream_t) override { fabric_->sync(); }

private:
    std::shared_ptr<LocalFabric> fabric_;
};

}  // namespace dist
}  // namespace wf
