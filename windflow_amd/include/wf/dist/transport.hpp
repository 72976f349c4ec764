// Inter-GPU transport of the distributed layer: one process per GPU, RCCL over
// xGMI between them.
//
// The reference is single-process: its only multi-GPU notion is "replica i uses
// GPU gpu_id" (/root/reference/wf/win_seq_gpu.hpp:231-237 picks a device per
// replica; SURVEY.md §2.6 lists no inter-GPU communication at all).  On MI355X
// nodes the GPUs are fully connected by point-to-point xGMI links, so the natural
// scale-out of a keyed stream is: every rank ingests its own sources, tuples are
// re-partitioned by key ACROSS ranks with one all-to-all per exchange round, and
// the keyed state of a key lives on exactly one GPU.  This header is the
// communication substrate of that design, in two planes:
//
//   control plane   ControlBoard: a small table in memory shared by the ranks (POSIX shared memory between the
//                   processes of a node, plain heap between the threads of one process).  For every round each rank
//                   publishes one row - the sizes of the slices it will send, end-of-stream and punctuation flags,
//                   its watermark - and reads the others'.  Publishing and collecting are plain loads and stores:
//                   no stream is synchronised and no collective runs to agree on the sizes of an all-to-all-v, so a
//                   rank can publish several rounds ahead while earlier rounds are still moving data.
//   data plane      Transport::exchange(): the all-to-all-v of one round, enqueued on a stream and never waited for
//                   by the host.  RcclTransport: one grouped ncclSend/ncclRecv per peer pair, so every xGMI link
//                   carries its pair's slice concurrently (no ring: the links are point to point).  LocalTransport:
//                   ranks are threads of one process; a rank PULLS its slices from the peers' send buffers with
//                   device copies (peer copies when the ranks sit on different GPUs) - the direct-store / load path
//                   that needs no rendezvous at all.
#pragma once
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../gpu/hip_utils.hpp"
#include "control_board.hpp"

namespace wf {
namespace dist {

#define WF_RCCL_CHECK(expr)                                                                                         \
    do {                                                                                                            \
        ncclResult_t _r = (expr);                                                                                   \
        if (_r != ncclSuccess) {                                                                                    \
            std::fprintf(stderr, "WindFlow RCCL error: %s at %s:%d in %s\n", ncclGetErrorString(_r), __FILE__, __LINE__, #expr); \
            std::abort();                                                                                           \
        }                                                                                                           \
    } while (0)

// ---- RCCL -------------------------------------------------------------------------
// Which RCCL is this process talking to?  The library is compiled against the headers of /opt/rocm and linked with an
// RPATH to /opt/rocm/lib, so a native program runs with exactly that library.  A Python process that imported torch first
// already holds torch's bundled librccl.so under the same SONAME and the loader hands THAT one to the extension.  Both
// are fine for the calls used here (unique id, communicator, grouped send / recv, all-gather, all-reduce: unchanged since
// NCCL 2.7), but the choice must be visible and a library from another major version must not be used by accident: every
// transport reports the path and the version once, and refuses a major-version mismatch or a library older than 2.7.
struct RcclIdentity {
    int runtime_version = 0;                     // ncclGetVersion()
    int header_version = NCCL_VERSION_CODE;      // what the code was compiled against
    std::string path;                            // shared object that answered
};
inline const RcclIdentity& rccl_identity() {
    static const RcclIdentity id = [] {
        RcclIdentity r;
        if (ncclGetVersion(&r.runtime_version) != ncclSuccess) gpu::failwith("ncclGetVersion failed");
        Dl_info info{};
        if (dladdr(reinterpret_cast<void*>(&ncclCommInitRank), &info) != 0 && info.dli_fname != nullptr) r.path = info.dli_fname;
        const char* q = std::getenv("WF_QUIET");
        if (!(q && std::atoi(q) != 0) || std::getenv("WF_RCCL_VERBOSE"))
            std::fprintf(stderr, "[wf::dist] RCCL %d.%d.%d from %s (compiled against %d.%d.%d)\n", r.runtime_version / 10000,
                         (r.runtime_version / 100) % 100, r.runtime_version % 100, r.path.empty() ? "?" : r.path.c_str(), r.header_version / 10000,
                         (r.header_version / 100) % 100, r.header_version % 100);
        if (r.runtime_version / 10000 != r.header_version / 10000 || r.runtime_version < 20700)
            gpu::failwith("RCCL library " + r.path + " reports version " + std::to_string(r.runtime_version) + ", compiled against " +
                          std::to_string(r.header_version) + ": refusing a library of another major version (or older than 2.7)");
        return r;
    }();
    return id;
}

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
        (void)rccl_identity();  // reports the library in use, refuses an incompatible one
        WF_HIP_CHECK(hipSetDevice(device_));
        WF_RCCL_CHECK(ncclCommInitRank(&comm_, world, id, rank));
        stage_dev_.resize(STAGE_WORDS * (size_t)(world + 1));
        stage_host_.resize(STAGE_WORDS * (size_t)(world + 1));
        // the control board lives in POSIX shared memory named after the unique id (every rank of the job has it)
        shm_.reset(new ShmBoard(id_bytes, rank, world));
        board_ = shm_->take_board();
        for (auto& e : round_events_) e.reset(new gpu::Event());
    }
    ~RcclTransport() override {
        if (comm_) (void)ncclCommDestroy(comm_);
    }
    const char* name() const override { return "rccl"; }

    void exchange(uint64_t round, const PeerSlices& p, const ControlRow*, hipStream_t s) override {
        std::lock_guard<std::mutex> l(m_);
        bool any = false;
        for (int peer = 0; peer < world_; peer++) any = any || p.send_bytes[peer] > 0 || p.recv_bytes[peer] > 0;
        if (any) {
            WF_RCCL_CHECK(ncclGroupStart());
            for (int peer = 0; peer < world_; peer++) {
                if (p.send_bytes[peer] > 0) WF_RCCL_CHECK(ncclSend(p.send[peer], p.send_bytes[peer], ncclChar, peer, comm_, s));
                if (p.recv_bytes[peer] > 0) WF_RCCL_CHECK(ncclRecv(p.recv[peer], p.recv_bytes[peer], ncclChar, peer, comm_, s));
            }
            WF_RCCL_CHECK(ncclGroupEnd());
        }
        round_events_[round % ControlBoard::DEPTH]->record(s);
    }
    bool sends_done(uint64_t round) override { return round_events_[round % ControlBoard::DEPTH]->done(); }

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

    void all_reduce_sum(double* dev, size_t n, hipStream_t s) override {
        std::lock_guard<std::mutex> l(m_);
        WF_RCCL_CHECK(ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, comm_, s));
    }

private:
    static constexpr size_t STAGE_WORDS = 64;
    int device_;
    ncclComm_t comm_ = nullptr;
    std::mutex m_;  // one collective at a time per communicator
    gpu::DeviceBuffer<uint64_t> stage_dev_;
    gpu::PinnedBuffer<uint64_t> stage_host_;
    std::unique_ptr<ShmBoard> shm_;
    std::unique_ptr<gpu::Event> round_events_[ControlBoard::DEPTH];
};

// rank 0 writes the id to `path` (atomically), the others poll for it: bootstrap for
// native programs started by a launcher that only provides RANK / WORLD_SIZE
inline std::string bootstrap_id_via_file(const std::string& path_prefix, int rank, int timeout_s = 120) {
    // The ranks of one job share their parent (the launcher: torchrun's agent, a shell): its pid names the file, so a file
    // left behind by a crashed job is never read by the next one; rank 0 also removes whatever carries the name first.
    const std::string path = path_prefix + "." + std::to_string((long)getppid());
    if (rank == 0) {
        (void)std::remove(path.c_str());
        const std::string id = RcclTransport::create_id();
        const std::string tmp = path + ".tmp";
        {
            std::ofstream f(tmp, std::ios::binary);
            f.write(id.data(), (std::streamsize)id.size());
        }
        std::rename(tmp.c_str(), path.c_str());
        return id;
    }
    // a stale file of the same name (pid reuse) is older than this process: skip it until rank 0 has replaced it
    const auto started = std::chrono::system_clock::now() - std::chrono::seconds(30);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s);
    while (std::chrono::steady_clock::now() < deadline) {
        struct stat st {};
        if (::stat(path.c_str(), &st) == 0 && std::chrono::system_clock::from_time_t(st.st_mtime) >= started) {
            std::ifstream f(path, std::ios::binary);
            if (f) {
                std::string id((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
                if (id.size() == RcclTransport::ID_BYTES) return id;
            }
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    gpu::failwith("bootstrap_id_via_file: timed out waiting for " + path);
}
// rank 0, once every rank holds its communicator: the file has done its job
inline void bootstrap_id_cleanup(const std::string& path_prefix) {
    (void)std::remove((path_prefix + "." + std::to_string((long)getppid())).c_str());
}

// ---- in-process fabric ---------------------------------------------------------------
// Shared by the `world` LocalTransports of one process: the control board in plain memory + scratch for the blocking helpers.
class LocalFabric {
public:
    explicit LocalFabric(int world) : world_(world), shared_(new ControlBoard::Shared()), gather_((size_t)world * 64), reduce_(world) {
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
    std::unique_ptr<ControlBoard::Shared> shared_;  // value-initialised: all tags and counters zero
    std::mutex m_;
    std::condition_variable cv_;
    int arrived_ = 0;
    uint64_t gen_ = 0;
    std::vector<uint64_t> gather_;
    std::vector<std::pair<double*, size_t>> reduce_;
};

// A rank PULLS: the peers' rows carry the addresses of their send buffers, so the receiver copies its slices out of them
// with device copies on its own stream - no rendezvous, the sender only learns (through the board) when everybody is done
// with its buffer.  Between GPUs of one process these are peer copies over xGMI.
class LocalTransport : public Transport {
public:
    LocalTransport(std::shared_ptr<LocalFabric> fabric, int rank) : fabric_(std::move(fabric)) {
        rank_ = rank;
        world_ = fabric_->world();
        board_.reset(new ControlBoard(fabric_->shared_.get(), rank, world_));
    }
    const char* name() const override { return "local"; }
    bool pulls() const override { return true; }

    void exchange(uint64_t round, const PeerSlices& p, const ControlRow* rows, hipStream_t s) override {
        bool copied = false;
        for (int peer = 0; peer < world_; peer++) {
            if (p.recv_bytes[peer] == 0) continue;
            const char* src = reinterpret_cast<const char*>((uintptr_t)rows[peer].send_base) + p.pull_offset[peer];
            WF_HIP_CHECK(hipMemcpyAsync(p.recv[peer], src, p.recv_bytes[peer], hipMemcpyDefault, s));
            copied = true;
        }
        if (!copied) {  // nothing to wait for: the peers' buffers are free as far as this rank is concerned
            for (int peer = 0; peer < world_; peer++) board_->ack(round, peer);
            return;
        }
        std::unique_ptr<gpu::Event> e(new gpu::Event());
        e->record(s);
        pending_.push_back({round, std::move(e)});
        poll_acks();
    }
    bool sends_done(uint64_t round) override {
        poll_acks();
        return board_->acks(round) >= (uint64_t)world_;
    }
    void quiesce(hipStream_t s) override {
        WF_HIP_CHECK(hipStreamSynchronize(s));
        poll_acks();
    }

    void all_gather_host(const uint64_t* mine, uint64_t* out, size_t words, hipStream_t) override {
        if (words > 64) gpu::failwith("LocalTransport::all_gather_host: too many words");
        std::memcpy(&fabric_->gather_[(size_t)rank_ * 64], mine, words * sizeof(uint64_t));
        fabric_->sync();
        for (int r = 0; r < world_; r++) std::memcpy(out + (size_t)r * words, &fabric_->gather_[(size_t)r * 64], words * sizeof(uint64_t));
        fabric_->sync();
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
    // my copies of a round have completed: tell every peer its send buffer of that round is free (for me)
    void poll_acks() {
        while (!pending_.empty() && pending_.front().second->done()) {
            for (int peer = 0; peer < world_; peer++) board_->ack(pending_.front().first, peer);
            pending_.pop_front();
        }
    }
    std::shared_ptr<LocalFabric> fabric_;
    std::deque<std::pair<uint64_t, std::unique_ptr<gpu::Event>>> pending_;
};

// One PROCESS per rank, no collective library in the data path: a rank publishes the IPC handle of the device allocation
// that holds its send buffer, the peers map it once (hipIpcOpenMemHandle; the mapping is cached per allocation) and PULL
// their slices with device copies on their own streams - over xGMI when the ranks sit on different GPUs, inside the card
// when they share one (which RCCL refuses: "duplicate GPU").  The direct load/store path SURVEY.md par. 5.8 asks for, and the
// multi-process test vehicle of the control plane on a one-GPU box.
class IpcTransport : public Transport {
public:
    // `token`: any string all ranks of the job share and no other job on the host uses (names the control board)
    IpcTransport(int rank, int world, const std::string& token, int device) : device_(device) {
        if (world < 1 || world > MAX_WORLD) gpu::failwith("IpcTransport: world size out of range");
        rank_ = rank;
        world_ = world;
        WF_HIP_CHECK(hipSetDevice(device_));
        shm_.reset(new ShmBoard("ipc:" + token, rank, world));
        board_ = shm_->take_board();
    }
    ~IpcTransport() override {
        for (auto& m : mapped_) (void)hipIpcCloseMemHandle(m.ptr);
    }
    const char* name() const override { return "ipc"; }
    bool pulls() const override { return true; }

    void describe_send(const void* buf_base, const void* send, ControlRow& row) override {
        row.send_base = (uint64_t)(uintptr_t)send;
        row.ipc_offset = (uint64_t)((const char*)send - (const char*)buf_base);
        auto it = exported_.find(buf_base);
        if (it == exported_.end()) {
            hipIpcMemHandle_t h;
            WF_HIP_CHECK(hipIpcGetMemHandle(&h, const_cast<void*>(buf_base)));
            it = exported_.emplace(buf_base, h).first;
        }
        std::memset(row.ipc_handle, 0, sizeof(row.ipc_handle));
        std::memcpy(row.ipc_handle, &it->second, sizeof(hipIpcMemHandle_t));
    }

    void exchange(uint64_t round, const PeerSlices& p, const ControlRow* rows, hipStream_t s) override {
        bool copied = false;
        for (int peer = 0; peer < world_; peer++) {
            if (p.recv_bytes[peer] == 0) continue;
            const char* base = (peer == rank_) ? reinterpret_cast<const char*>((uintptr_t)rows[peer].send_base)
                                               : static_cast<const char*>(map_peer(peer, rows[peer].ipc_handle)) + rows[peer].ipc_offset;
            WF_HIP_CHECK(hipMemcpyAsync(p.recv[peer], base + p.pull_offset[peer], p.recv_bytes[peer], hipMemcpyDeviceToDevice, s));
            copied = true;
        }
        if (!copied) {
            for (int peer = 0; peer < world_; peer++) board_->ack(round, peer);
            return;
        }
        std::unique_ptr<gpu::Event> e(new gpu::Event());
        e->record(s);
        pending_.push_back({round, std::move(e)});
        poll_acks();
    }
    bool sends_done(uint64_t round) override {
        poll_acks();
        return board_->acks(round) >= (uint64_t)world_;
    }
    void quiesce(hipStream_t s) override {
        WF_HIP_CHECK(hipStreamSynchronize(s));
        poll_acks();
    }
    void all_gather_host(const uint64_t* mine, uint64_t* out, size_t words, hipStream_t) override {
        if (words > ControlBoard::GATHER_WORDS) gpu::failwith("IpcTransport::all_gather_host: too many words");
        board_->all_gather(mine, out, words);
    }
    // setup-time helper (a few words at a time through the control board)
    void all_reduce_sum(double* dev, size_t n, hipStream_t s) override {
        std::vector<double> host(n);
        WF_HIP_CHECK(hipMemcpyAsync(host.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost, s));
        WF_HIP_CHECK(hipStreamSynchronize(s));
        std::vector<uint64_t> all(ControlBoard::GATHER_WORDS * (size_t)world_);
        for (size_t i = 0; i < n; i += ControlBoard::GATHER_WORDS) {
            const size_t w = std::min(ControlBoard::GATHER_WORDS, n - i);
            board_->all_gather(reinterpret_cast<const uint64_t*>(host.data() + i), all.data(), w);
            for (size_t j = 0; j < w; j++) {
                double acc = 0.0;
                for (int r = 0; r < world_; r++) {
                    double v;
                    std::memcpy(&v, &all[(size_t)r * w + j], sizeof(v));
                    acc += v;
                }
                host[i + j] = acc;
            }
        }
        WF_HIP_CHECK(hipMemcpyAsync(dev, host.data(), n * sizeof(double), hipMemcpyHostToDevice, s));
        WF_HIP_CHECK(hipStreamSynchronize(s));
    }

private:
    struct Mapped {
        int peer;
        unsigned char handle[64];
        void* ptr;
    };
    void* map_peer(int peer, const unsigned char* handle) {
        for (const Mapped& m : mapped_)
            if (m.peer == peer && std::memcmp(m.handle, handle, 64) == 0) return m.ptr;
        hipIpcMemHandle_t h;
        std::memcpy(&h, handle, sizeof(h));
        Mapped m;
        m.peer = peer;
        std::memcpy(m.handle, handle, 64);
        WF_HIP_CHECK(hipIpcOpenMemHandle(&m.ptr, h, hipIpcMemLazyEnablePeerAccess));
        mapped_.push_back(m);
        return m.ptr;
    }
    void poll_acks() {
        while (!pending_.empty() && pending_.front().second->done()) {
            for (int peer = 0; peer < world_; peer++) board_->ack(pending_.front().first, peer);
            pending_.pop_front();
        }
    }
    int device_;
    std::unique_ptr<ShmBoard> shm_;
    std::map<const void*, hipIpcMemHandle_t> exported_;
    std::vector<Mapped> mapped_;
    std::deque<std::pair<uint64_t, std::unique_ptr<gpu::Event>>> pending_;
};

}  // namespace dist
}  // namespace wf
