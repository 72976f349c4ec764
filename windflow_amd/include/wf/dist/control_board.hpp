// Control plane of the distributed layer and the transport interface (no collective library needed here: the
// implementations - RCCL, in-process, hipIpc - are in transport.hpp).
//
//   control plane   ControlBoard: a small table in memory shared by the ranks (POSIX shared memory between the
//                   processes of a node, plain heap between the threads of one process).  For every round each rank
//                   publishes one row - the sizes of the slices it will send, end-of-stream and punctuation flags,
//                   its watermark - and reads the others'.  Publishing and collecting are plain loads and stores:
//                   no stream is synchronised and no collective runs to agree on the sizes of an all-to-all-v, so a
//                   rank can publish several rounds ahead while earlier rounds are still moving data.
//   data plane      Transport::exchange(): the all-to-all-v of one round, enqueued on a stream and never waited for
//                   by the host.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <thread>

#include "../gpu/hip_utils.hpp"

namespace wf {
namespace dist {

constexpr int MAX_WORLD = 16;
constexpr int MAX_LANES = 16;

// ---- control plane ---------------------------------------------------------------------------------
// One row per (round, rank).  counts[d] = tuples this rank sends to destination d = owner_rank * lanes + lane.
struct ControlRow {
    uint32_t counts[MAX_WORLD * MAX_LANES];
    uint64_t done;        // the rank's input stream has ended
    uint64_t step;        // greatest step punctuation every local lane has delivered
    uint64_t watermark;   // greatest timestamp of the rank's batch of this round
    uint64_t created_ns;  // when the rank's batch of this round was opened (latency accounting)
    uint64_t capacity;    // largest batch capacity the rank has seen (sizes the receive slabs)
    uint64_t send_base;   // pull transports: address of the rank's send buffer for this round (in the sender's address space)
    uint64_t ipc_offset;  // IpcTransport: offset of send_base inside the allocation named by ipc_handle
    uint64_t aux;         // DETERMINISTIC graphs: greatest timestamp the rank has sent so far (`watermark` is what it claims to be complete)
    uint64_t held;        // DETERMINISTIC graphs: tuples the rank has received but may not release yet (senders that run ahead slow down)
    unsigned char ipc_handle[64];  // IpcTransport: hipIpcMemHandle_t of the allocation that holds the send buffer
};
static_assert(sizeof(hipIpcMemHandle_t) <= 64, "ControlRow::ipc_handle too small");

// Rounds are numbered 0, 1, 2...; the board keeps DEPTH of them.  A rank may publish round r once every rank has
// collected round r - DEPTH (its slot is then free); tags make rows visible with release / acquire ordering.
class ControlBoard {
public:
    static constexpr uint64_t DEPTH = 8;
    static constexpr size_t GATHER_WORDS = 64;
    static constexpr int PROGRESS_LANES = 16;  // sources per rank with a progress counter on the board
    struct Slot {
        std::atomic<uint64_t> tag;  // round + 1 when the row is valid
        ControlRow row;
    };
    struct Shared {
        std::atomic<uint64_t> attached;                  // ranks that have mapped the board
        std::atomic<uint64_t> collected[MAX_WORLD];      // rounds the rank has collected (next round to collect)
        std::atomic<uint64_t> acked[DEPTH][MAX_WORLD];   // data plane: peers that are done with rank's send buffer of the round
        std::atomic<uint64_t> barrier_count, barrier_gen;
        uint64_t gather[MAX_WORLD][GATHER_WORDS];        // scratch of the blocking helpers (between two barriers)
        Slot slots[DEPTH][MAX_WORLD];
        // free-running progress counters of the ranks' sources (benchmark sources that read one time-ordered feed keep
        // their synthetic clocks together across the ranks with them: bench_common.hpp SkewBound)
        std::atomic<uint64_t> progress[MAX_WORLD][PROGRESS_LANES];
    };
    static_assert(std::atomic<uint64_t>::is_always_lock_free, "the control board needs lock-free 64-bit atomics");

    ControlBoard(Shared* sh, int rank, int world) : sh_(sh), rank_(rank), world_(world) {}
    std::atomic<uint64_t>* progress(int rank) { return sh_->progress[rank]; }

    bool can_publish(uint64_t round) const {
        if (round < DEPTH) return true;
        for (int r = 0; r < world_; r++)
            if (sh_->collected[r].load(std::memory_order_acquire) + DEPTH <= round) return false;
        return true;
    }
    // precondition: can_publish(round)
    void publish(uint64_t round, const ControlRow& row) {
        Slot& s = sh_->slots[round % DEPTH][rank_];
        s.row = row;
        sh_->acked[round % DEPTH][rank_].store(0, std::memory_order_relaxed);
        s.tag.store(round + 1, std::memory_order_release);
    }
    bool published(uint64_t round, int rank) const { return sh_->slots[round % DEPTH][rank].tag.load(std::memory_order_acquire) == round + 1; }
    // has any OTHER rank published this round?  (a rank without input answers such a round at once with an empty row)
    bool anyone_published(uint64_t round) const {
        for (int r = 0; r < world_; r++)
            if (r != rank_ && published(round, r)) return true;
        return false;
    }
    // all rows of the round, rank-major; false if some rank has not published yet
    bool try_collect(uint64_t round, ControlRow* out) {
        for (int r = 0; r < world_; r++)
            if (!published(round, r)) return false;
        for (int r = 0; r < world_; r++) out[r] = sh_->slots[round % DEPTH][r].row;
        sh_->collected[rank_].store(round + 1, std::memory_order_release);
        return true;
    }
    // data plane acknowledgements (pull transports): `rank`'s send buffer of `round` has been copied by one more peer
    void ack(uint64_t round, int rank) { sh_->acked[round % DEPTH][rank].fetch_add(1, std::memory_order_acq_rel); }
    uint64_t acks(uint64_t round) const { return sh_->acked[round % DEPTH][rank_].load(std::memory_order_acquire); }
    // host barrier over the ranks (setup / teardown only)
    void barrier() {
        const uint64_t gen = sh_->barrier_gen.load(std::memory_order_acquire);
        if (sh_->barrier_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint64_t)world_) {
            sh_->barrier_count.store(0, std::memory_order_relaxed);
            sh_->barrier_gen.store(gen + 1, std::memory_order_release);
        } else {
            while (sh_->barrier_gen.load(std::memory_order_acquire) == gen) std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    // every rank contributes `words` (<= GATHER_WORDS) 64-bit words; blocking
    void all_gather(const uint64_t* mine, uint64_t* out, size_t words) {
        std::memcpy(sh_->gather[rank_], mine, words * sizeof(uint64_t));
        barrier();
        for (int r = 0; r < world_; r++) std::memcpy(out + (size_t)r * words, sh_->gather[r], words * sizeof(uint64_t));
        barrier();
    }
    int rank() const { return rank_; }
    int world() const { return world_; }

private:
    Shared* sh_;
    int rank_, world_;
};

// The control board of a multi-process job: POSIX shared memory named after a token every rank of the job knows (a fresh
// segment is zero-filled, which is the board's initial state).  The constructor returns once every rank has mapped it.
class ShmBoard {
public:
    ShmBoard(const std::string& token, int rank, int world) {
        uint64_t h = 1469598103934665603ull;
        for (unsigned char c : token) h = (h ^ c) * 1099511628211ull;
        const std::string name = "/wf_board_" + std::to_string(h);
        const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0) gpu::failwith("ShmBoard: shm_open failed for the control board");
        if (ftruncate(fd, (off_t)sizeof(ControlBoard::Shared)) != 0) gpu::failwith("ShmBoard: ftruncate failed for the control board");
        void* p = mmap(nullptr, sizeof(ControlBoard::Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) gpu::failwith("ShmBoard: mmap failed for the control board");
        shared_ = static_cast<ControlBoard::Shared*>(p);
        board_.reset(new ControlBoard(shared_, rank, world));
        // nobody uses the board before everybody has mapped it; then the name can go (the mapping stays)
        shared_->attached.fetch_add(1, std::memory_order_acq_rel);
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
        while (shared_->attached.load(std::memory_order_acquire) < (uint64_t)world) {
            if (std::chrono::steady_clock::now() > deadline) gpu::failwith("ShmBoard: timed out waiting for the other ranks to map the control board");
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        board_->barrier();
        if (rank == 0) shm_unlink(name.c_str());
    }
    ~ShmBoard() {
        board_.reset();
        if (shared_) munmap(shared_, sizeof(ControlBoard::Shared));
    }
    ShmBoard(const ShmBoard&) = delete;
    ShmBoard& operator=(const ShmBoard&) = delete;
    std::unique_ptr<ControlBoard> take_board() { return std::move(board_); }

private:
    ControlBoard::Shared* shared_ = nullptr;
    std::unique_ptr<ControlBoard> board_;
};

// per-peer slices of one all-to-all-v (device pointers, byte counts)
struct PeerSlices {
    const void* send[MAX_WORLD] = {};
    size_t send_bytes[MAX_WORLD] = {};
    void* recv[MAX_WORLD] = {};
    size_t recv_bytes[MAX_WORLD] = {};
    // pull transports: where my slice starts inside peer p's send buffer (bytes from the peer's ControlRow::send_base)
    size_t pull_offset[MAX_WORLD] = {};
};

class Transport {
public:
    virtual ~Transport() = default;
    int rank() const { return rank_; }
    int world() const { return world_; }
    virtual const char* name() const = 0;
    ControlBoard& board() { return *board_; }
    // all-to-all-v of round `round`, enqueued on `s` (returns without waiting).  `rows` are the collected rows of the round.
    virtual void exchange(uint64_t round, const PeerSlices& p, const ControlRow* rows, hipStream_t s) = 0;
    // called while a round's row is filled: `buf_base` is the start of the device allocation that holds the send buffer,
    // `send` the first byte the peers will read
    virtual void describe_send(const void* buf_base, const void* send, ControlRow& row) {
        (void)buf_base;
        row.send_base = (uint64_t)(uintptr_t)send;
    }
    // true: peers read this rank's send buffer on their own streams, from the moment the round's row is on the board -
    // a row must not be published before the buffer is complete
    virtual bool pulls() const { return false; }
    // may the send buffers handed to exchange(round) be reused?  (non-blocking)
    virtual bool sends_done(uint64_t round) = 0;
    // end of stream: complete (and acknowledge) every transfer this rank still has in flight on `s`
    virtual void quiesce(hipStream_t) {}
    // element-wise sum of `n` doubles in device memory, in place, enqueued on `s`
    virtual void all_reduce_sum(double* dev, size_t n, hipStream_t s) = 0;
    // every rank contributes `words` 64-bit words (small, setup-time data); blocking
    virtual void all_gather_host(const uint64_t* mine, uint64_t* out, size_t words, hipStream_t s) = 0;
    virtual void barrier(hipStream_t) { board_->barrier(); }

protected:
    int rank_ = 0, world_ = 1;
    std::unique_ptr<ControlBoard> board_;
};

}  // namespace dist
}  // namespace wf
