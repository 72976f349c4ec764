// Nodes of Exchange_GPU (dist/exchange_gpu.hpp) and of the device KeyBy in front of keyed GPU operators.
//
// Exchange_GPU: KeyBy across GPUs.  Every rank (process or thread owning one GPU)
// places one Exchange_GPU in its MultiPipe; batches entering it are partitioned by
// key on the device and re-distributed with one all-to-all per round, so that a key
// is only ever seen by the rank that owns it - and, inside that rank, by the lane
// that owns it: the exchange delivers `lanes` key-disjoint streams, one per replica
// of the keyed operator behind it (Key_FFAT_GPU, Key_Farm_GPU, keyed Map_GPU ... built
// with withKeyDisjointLanes()), which then run unchanged on purely local state and
// need no second partitioning pass.  owner rank = bits 48-63 of mix64(key) mod world,
// lane = bits 24-47 mod lanes (gpu/keyby_gpu.hpp).
//
// No counterpart in the reference (single process, no inter-GPU traffic, SURVEY.md
// §2.6); this is the piece that turns "N replicas on N GPUs" into a scalable
// one-process-per-GPU deployment over xGMI.
//
//   lanes (stage 1, one thread per producer lane): ship the batch to the device if needed and pack it by destination
//           (rank, lane): ONE slab ordered by destination + the destination offsets in pinned memory.  Nothing waits:
//           the bundle carries the event of the pack kernels.
//   endpoint (stage 2, one thread per rank) runs ROUNDS, several in flight:
//           stage     every packed slab that is ready (up to one per local lane) joins the next round: their per-destination
//                     segments are merged into one send slab ordered by destination (one segmented-copy kernel), so a round
//                     under load carries `local lanes` batches per rank in ONE slice per peer - few, fat transfers per
//                     xGMI link instead of many thin ones - while a lone batch goes out as it is (no merge, low latency).
//           publish   the per-destination counts of the round become the rank's row on the control board
//                     (dist/transport.hpp).  A rank without input answers a round that some other rank has opened with an
//                     empty row at once; nobody idles on a timer.
//           collect   once every rank's row of the oldest open round is there, the sizes of the all-to-all-v are known:
//                     the peers' chunks (all the lanes of this rank, contiguous in their send slabs) land back to back in
//                     a staging slab, one segmented-copy kernel lays them out as batches per lane (per-key order = round,
//                     source rank, position: FIFO per producer is preserved; a lane that receives more than a lane batch
//                     holds - a hot key - gets several), the lane batches go downstream with their counts known on the
//                     host.  The host never waits for the transfer.
//           release   a send slab returns to its pool when the transport says the peers are done with it.
//   With one rank there is nothing to exchange: the endpoint merges the ready packed slabs straight into lane batches
//   (the device KeyBy, coalescing the batches of all local lanes).
//   Step punctuations (header.reserved[0], see Batch_Source) are delivered downstream once every lane of every rank has
//   reported them; the forwarded watermark is the minimum over the ranks still producing; at end of stream a rank keeps
//   answering rounds until every rank is done and a round moved nothing.
//
// The same two stages, without a transport, are the KeyBy INSIDE a GPU (device_keyby_stages below): keyed GPU operators
// whose input lanes are not key-disjoint put them in front of their replicas - every producer packs its batches by lane,
// one endpoint thread merges the packed slabs of all producers into one batch per replica.
#pragma once
#include <cstddef>
#include <deque>
#include <map>
#include <unordered_map>

#include "../gpu/batch_ops.hpp"
#include "../gpu/keyby_gpu.hpp"
#include "control_board.hpp"

namespace wf {

struct ExchangeConfig {
    size_t local_lanes = 0;        // producers feeding this exchange on this rank (punctuations to wait for per step); 0: as many as there are
    size_t out_lanes = 1;          // key-disjoint streams delivered downstream (= replicas of the keyed operator behind)
    uint64_t idle_round_us = 2000; // with no input and nobody else asking, still open a round this often (keeps watermarks moving)
    size_t max_batches_per_round = 0;  // input batches merged into one round (0: one per local lane, at most 16)
    // a round leaves when it is full (max_batches_per_round), when another rank is waiting for this one, or when its oldest
    // batch has waited this long: under load the rounds are fat (few large transfers and kernels), a lone batch is delayed by
    // at most this much
    uint64_t linger_us = 300;
    // DETERMINISTIC graphs (one rank): the endpoint is the ordering point of the batch lanes.  Every producer lane is in
    // timestamp order; the endpoint releases, per output lane, the tuples whose timestamp is <= the least of the producers'
    // latest watermarks, merged in timestamp order (ties: producer, then position) - the low-watermark rule of the tuple
    // world's ordering nodes (ordering_node.hpp) applied to batches.
    bool deterministic = false;
};

// counters of one rank's endpoint, readable after the run (Exchange_GPU::stats())
struct ExchangeStats {
    uint64_t rounds = 0, empty_rounds = 0;
    uint64_t bundles = 0;        // input batches that went into the rounds (bundles / rounds = how fat the rounds were)
    uint64_t lane_batches = 0;   // batches delivered downstream
    uint64_t tuples_sent = 0, tuples_received = 0, bytes_received = 0;
    uint64_t max_link_bytes_per_round = 0;  // xGMI is point to point: a round is bound by its busiest link
    uint64_t link_bytes_sent[dist::MAX_WORLD] = {};  // bytes this rank has put on the link to every peer (the busiest one bounds the job)
    uint64_t start_us = 0, elapsed_us = 0;
};

// What a partition lane hands to the rank's endpoint
struct ExchangeBundle {
    uint64_t arrived_us = 0;                // when the endpoint got it
    size_t consumed[dist::MAX_WORLD * dist::MAX_LANES] = {};  // deterministic mode: tuples of every destination segment already released
    gpu::Batch* packed = nullptr;           // the batch grouped by destination (rank, lane)
    volatile uint32_t* offsets = nullptr;   // pinned: world * lanes + 1 offsets into `packed`, valid once packed->ready has fired
    gpu::Batch* passthrough = nullptr;      // one rank, one lane: the batch itself
    gpu::BatchHeader hdr{};
    size_t capacity = 0;
};

namespace gpu {
// Segmented copy: segment y of the table is copied by the blocks (*, y); the first n_hdrs rows also write a batch header.
// The table lives in pinned host memory (read over the bus once per block: a few KB per launch, no staging copy).
struct SegCopy {
    const void* src;
    void* dst;
    uint64_t count;  // tuples
};
struct HdrWrite {
    BatchHeader value;
    BatchHeader* dst;
    uint64_t pad[7];
};
static_assert(sizeof(HdrWrite) == 128, "HdrWrite layout");
template <typename T>
__global__ void __launch_bounds__(BLOCK_THREADS) segcopy_kernel(const SegCopy* segs, uint32_t n_segs, const HdrWrite* hdrs, uint32_t n_hdrs) {
    using W = std::conditional_t<sizeof(T) % 8 == 0 && alignof(T) % 8 == 0, uint64_t,
                                 std::conditional_t<sizeof(T) % 4 == 0 && alignof(T) % 4 == 0, uint32_t, unsigned char>>;
    const uint32_t y = blockIdx.y;
    if (y < n_hdrs && blockIdx.x == 0 && threadIdx.x < sizeof(BatchHeader) / 8)
        reinterpret_cast<uint64_t*>(hdrs[y].dst)[threadIdx.x] = reinterpret_cast<const uint64_t*>(&hdrs[y].value)[threadIdx.x];
    if (y >= n_segs) return;
    const SegCopy sc = segs[y];
    const size_t words = (size_t)sc.count * (sizeof(T) / sizeof(W));
    const W* __restrict__ src = static_cast<const W*>(sc.src);
    W* __restrict__ dst = static_cast<W*>(sc.dst);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ring of pinned tables for segcopy launches; a slot is reused once the kernel that read it has completed
class SegTables {
public:
    explicit SegTables(size_t slots = 16) : slots_(slots) {}
    ~SegTables() {
        for (Slot& s : slots_)
            if (s.mem) (void)hipHostFree(s.mem);
    }
    std::vector<SegCopy> segs;
    std::vector<HdrWrite> hdrs;
    void clear() {
        segs.clear();
        hdrs.clear();
    }
    // copies of `segs` (and header writes of `hdrs`) enqueued on s
    template <typename T>
    void launch(hipStream_t s) {
        if (segs.empty() && hdrs.empty()) return;
        Slot& slot = slots_[next_];
        next_ = (next_ + 1) % slots_.size();
        slot.ev.host_sync();
        const size_t hb = hdrs.size() * sizeof(HdrWrite), sb = segs.size() * sizeof(SegCopy);
        if (slot.bytes < hb + sb) {
            if (slot.mem) WF_HIP_CHECK(hipHostFree(slot.mem));
            slot.bytes = std::max<size_t>(2 * (hb + sb), 8192);
            WF_HIP_CHECK(hipHostMalloc((void**)&slot.mem, slot.bytes, hipHostMallocDefault));
        }
        if (hb) std::memcpy(slot.mem, hdrs.data(), hb);
        if (sb) std::memcpy(slot.mem + hb, segs.data(), sb);
        uint64_t longest = 0;
        for (const SegCopy& c : segs) longest = std::max(longest, c.count);
        const size_t words = longest * ((sizeof(T) + 7) / 8);
        const unsigned gx = (unsigned)std::min<size_t>(128, std::max<size_t>(1, (words + 8 * BLOCK_THREADS - 1) / (8 * BLOCK_THREADS)));
        const unsigned gy = (unsigned)std::max(segs.size(), hdrs.size());
        hipLaunchKernelGGL((segcopy_kernel<T>), dim3(gx, gy), dim3(BLOCK_THREADS), 0, s, reinterpret_cast<const SegCopy*>(slot.mem + hb),
                           (uint32_t)segs.size(), reinterpret_cast<const HdrWrite*>(slot.mem), (uint32_t)hdrs.size());
        WF_HIP_CHECK(hipGetLastError());
        slot.ev.record(s);
    }

private:
    struct Slot {
        char* mem = nullptr;
        size_t bytes = 0;
        Event ev;
    };
    std::vector<Slot> slots_;
    size_t next_ = 0;
};
// ---- deterministic merge (ExchangeConfig::deterministic) ----
// thread per segment: how many tuples of a timestamp-ordered segment have ts <= W
template <typename T>
__global__ void __launch_bounds__(BLOCK_THREADS) det_count_le_kernel(const SegCopy* segs, uint32_t n_segs, uint64_t W, volatile uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_segs) return;
    const T* seg = static_cast<const T*>(segs[i].src);
    uint32_t lo = 0, hi = (uint32_t)segs[i].count;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if ((uint64_t)std::get<2>(seg[mid].getControlFields()) <= W)
            lo = mid + 1;
        else
            hi = mid;
    }
    out[i] = lo;
}
// The work buffer of a release holds the tuples of replica 0, then those of replica 1, ... (`off`: where each starts).
struct DetLaneOff {
    uint32_t off[dist::MAX_LANES + 1];
};
// sort keys of one 32-bit half of (ts - base); perm == nullptr: identity.  lane_shift >= 0: the replica the tuple goes to
// is put on top of the key (bits lane_shift and up), so that ONE sort orders every replica's tuples by timestamp
template <typename T>
__global__ void __launch_bounds__(BLOCK_THREADS)
    det_keys_kernel(const T* work, uint32_t n, uint64_t base, int shift, const uint32_t* perm, uint32_t* keys, uint32_t* idx, DetLaneOff lanes,
                    uint32_t n_lanes, int lane_shift) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t src = perm ? perm[i] : i;
        uint32_t k = (uint32_t)((((uint64_t)std::get<2>(work[src].getControlFields())) - base) >> shift);
        if (lane_shift >= 0) {
            uint32_t l = 0;
            while (l + 1 < n_lanes && src >= lanes.off[l + 1]) l++;
            k |= l << lane_shift;
        }
        keys[i] = k;
        idx[i] = src;
    }
}
__global__ void det_set_u32_kernel(uint32_t* p, uint32_t v) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *p = v;
}
template <typename T>
__global__ void __launch_bounds__(BLOCK_THREADS) det_gather_kernel(const T* work, const uint32_t* perm, T* out, uint32_t first, uint32_t count) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) out[i] = work[perm[first + i]];
}
// ---- deterministic merge across ranks ----
// Sort key of a pending tuple: (lane', timestamp - base, source rank).  lane' is the output lane of the tuple when it may be
// released (timestamp <= W) and n_lanes when it has to wait: ONE stable sort then leaves the released tuples of every lane in
// (timestamp, rank, arrival) order, followed by the tuples that stay pending in their arrival order.  `shift`: which 32 bits
// of the key are produced (keys wider than 32 bits are sorted least significant half first).
template <typename T>
__global__ void __launch_bounds__(BLOCK_THREADS)
    xdet_keys_kernel(const T* pend, const uint32_t* src_rank, uint32_t n, uint64_t W, uint64_t base, uint32_t n_lanes, int ts_bits, int rank_bits, int shift,
                     const uint32_t* perm, uint32_t* keys, uint32_t* idx) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t src = perm ? perm[i] : i;
        const T t = pend[src];
        const uint64_t ts = (uint64_t)std::get<2>(t.getControlFields());
        unsigned long long k;
        if (ts <= W) {
            const uint64_t h = mix64(gpu::key_hash64(std::get<0>(t.getControlFields())));
            const unsigned long long lane = ((h >> HASH_SHIFT_LANES) & 0xffffffull) % n_lanes;
            const unsigned long long rel = ts > base ? ts - base : 0ull;
            k = (lane << (ts_bits + rank_bits)) | (rel << rank_bits) | (unsigned long long)src_rank[src];
        } else {
            k = (unsigned long long)n_lanes << (ts_bits + rank_bits);
        }
        keys[i] = (uint32_t)(k >> shift);
        idx[i] = src;
    }
}
// where the lanes (and the pending rest) begin in the sorted order: offsets[l] = first position of lane' >= l, offsets[n_lanes + 1] = n;
// written straight to pinned host memory
template <typename T>
__global__ void __launch_bounds__(BLOCK_THREADS)
    xdet_offsets_kernel(const T* pend, const uint32_t* perm, uint32_t n, uint64_t W, uint32_t n_lanes, volatile uint32_t* offsets) {
    auto lane_of = [&](uint32_t pos) -> uint32_t {
        const T t = pend[perm[pos]];
        if ((uint64_t)std::get<2>(t.getControlFields()) > W) return n_lanes;
        const uint64_t h = mix64(gpu::key_hash64(std::get<0>(t.getControlFields())));
        return (uint32_t)(((h >> HASH_SHIFT_LANES) & 0xffffffull) % n_lanes);
    };
    if (n == 0) {
        if (blockIdx.x == 0)
            for (uint32_t x = threadIdx.x; x <= n_lanes + 1; x += blockDim.x) offsets[x] = 0;
        return;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t d = lane_of(i);
        const uint32_t first = (i == 0) ? 0u : lane_of(i - 1) + 1u;
        for (uint32_t x = first; x <= d; x++) offsets[x] = i;
        if (i == n - 1)
            for (uint32_t x = d + 1; x <= n_lanes + 1; x++) offsets[x] = n;
    }
}
struct XdetRankOff {
    uint32_t off[dist::MAX_WORLD + 1];
};
// source rank of the tuples that arrived in this round (chunk r of the staging area came from rank r)
__global__ void __launch_bounds__(BLOCK_THREADS) xdet_fill_ranks_kernel(uint32_t* ranks, XdetRankOff o, uint32_t world) {
    const uint32_t n = o.off[world];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t r = 0;
        while (r + 1 < world && i >= o.off[r + 1]) r++;
        ranks[i] = r;
    }
}
__global__ void __launch_bounds__(BLOCK_THREADS) xdet_gather_u32_kernel(const uint32_t* src, const uint32_t* perm, uint32_t* dst, uint32_t first, uint32_t count) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) dst[i] = src[perm[first + i]];
}
}  // namespace gpu

// ---- stage 1: partition lanes (parallel) ------------------------------------------------------
template <typename tuple_t>
class ExchangePart_Node : public GPU_Replica<tuple_t> {
public:
    ExchangePart_Node(int world, size_t out_lanes, GPUConfig cfg, edge_kind_t in_kind, std::string name, size_t par, size_t idx,
                      meta::closing_func_t closing, std::shared_ptr<StatsHub> hub, bool always_pack = false)
        : GPU_Replica<tuple_t>(cfg, in_kind, std::move(name), par, idx, std::move(closing), std::move(hub)), world_(world), lanes_(out_lanes),
          always_pack_(always_pack) {}
    ~ExchangePart_Node() override {
        for (auto& kv : mirrors_) (void)hipHostFree((void*)kv.second);
    }

protected:
    void gpu_init() override {
        // a packed slab is held until its round has been merged (or, for a lone batch, until the peers have read it)
        const size_t slabs = std::max<size_t>(3, this->cfg_.pool_slabs) + (world_ > 1 ? dist::ControlBoard::DEPTH : 2);
        if (world_ > 1 || lanes_ > 1 || always_pack_) part_.reset(new gpu::BatchPartitioner<tuple_t>(1, this->cfg_.gpu_id, slabs + (always_pack_ ? 8 : 0)));
    }
    void process_device(gpu::Batch* d) override {
        gpu::TraceRange trace("wf:exchange_partition");
        auto* b = new ExchangeBundle();
        b->hdr = d->host_hdr;
        b->capacity = d->capacity;
        hipStream_t s = this->stream();
        if (part_) {
            // the offsets mirror belongs to the slab the pack writes to: it is reused exactly when the slab is
            b->packed = part_->partition_packed(d, (size_t)world_, lanes_, [this](gpu::Batch* slab) { return mirror_of(slab); }, s, this->num_kernels_);
            b->offsets = mirror_of(b->packed);
            this->release_input(d);
            b->packed->ready.record(s);
        } else {
            d->ready.record(s);
            b->passthrough = d;
        }
        this->emit(b);
    }

private:
    volatile uint32_t* mirror_of(gpu::Batch* slab) {
        auto it = mirrors_.find(slab);
        if (it != mirrors_.end()) return it->second;
        uint32_t* m = nullptr;
        WF_HIP_CHECK(hipHostMalloc((void**)&m, (dist::MAX_WORLD * dist::MAX_LANES + 1) * sizeof(uint32_t), hipHostMallocDefault));
        mirrors_[slab] = m;
        return m;
    }
    int world_;
    size_t lanes_;
    bool always_pack_;
    std::unique_ptr<gpu::BatchPartitioner<tuple_t>> part_;
    std::unordered_map<gpu::Batch*, volatile uint32_t*> mirrors_;
};

// ---- stage 2: the rank's endpoint on the fabric (one thread) -------------------------------------
template <typename tuple_t>
class ExchangeEndpoint_Node : public Replica_Base {
public:
    ExchangeEndpoint_Node(std::shared_ptr<dist::Transport> tr, ExchangeConfig xcfg, GPUConfig cfg, std::string name, meta::closing_func_t closing,
                          std::shared_ptr<StatsHub> hub, std::shared_ptr<ExchangeStats> xstats, std::shared_ptr<size_t> n_producers = nullptr)
        : Replica_Base(std::move(name), 1, 0, std::move(closing), std::move(hub), false, true), tr_(std::move(tr)), xcfg_(xcfg), cfg_(cfg),
          xstats_(std::move(xstats)), n_producers_(std::move(n_producers)) {
        if (!xstats_) xstats_ = std::make_shared<ExchangeStats>();
        if (world() > dist::MAX_WORLD) gpu::failwith("Exchange_GPU: world size above MAX_WORLD");
        if (xcfg_.out_lanes == 0 || xcfg_.out_lanes > (size_t)dist::MAX_LANES) gpu::failwith("Exchange_GPU: between 1 and 16 output lanes");
        cfg_.gpu_id = cfg.device_for(0);
    }
    void svc_init() override {
        Replica_Base::svc_init();
        // the producers are known once the graph is wired
        if (xcfg_.local_lanes == 0) xcfg_.local_lanes = (n_producers_ && *n_producers_ > 0) ? *n_producers_ : 1;
        group_ = std::min<size_t>(xcfg_.local_lanes, MAX_GROUP);
        if (xcfg_.max_batches_per_round != 0) group_ = std::min(group_, xcfg_.max_batches_per_round);
        WF_HIP_CHECK(hipSetDevice(cfg_.gpu_id));
        stream_.reset(new gpu::Stream());
        tables_.reset(new gpu::SegTables());
        last_round_us_ = start_us_ = xstats_->start_us = current_time_usecs();
    }
    void svc(void* msg, int channel) override {
        auto* b = static_cast<ExchangeBundle*>(msg);
        b->arrived_us = current_time_usecs();
        if (xcfg_.deterministic) {
            det_arrive(b, channel);
            return;
        }
        pending_.push_back(b);
        pump();
    }
    void eosnotify(int channel) override {
        if (!xcfg_.deterministic || !stream_) return;
        det_init();
        det_ended_[(size_t)channel] = true;
        if (world() > 1)
            pump();
        else
            det_release(false);
    }
    bool wants_idle() const override { return !xcfg_.deterministic || world() > 1; }
    void on_idle() override {
        if (!finished_) pump();
    }
    void flush() override {
        // end of my stream: keep answering rounds until everybody is done
        done_ = true;
        if (!stream_) return;  // never started
        if (xcfg_.deterministic && world() == 1) {
            det_init();
            det_release(true);
            finished_ = true;
            return;
        }
        if (xcfg_.deterministic) det_init();
        while (!finished_) {
            if (!pump()) std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        // whatever this rank still reads from the peers' buffers completes here, so that they can finish too
        if (tr_) tr_->quiesce(stream_->get());
        while (!sent_.empty()) {
            if (!release_sent()) std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    void svc_end() override {
        if (stream_) stream_->sync();
        if (det_counts_) (void)hipHostFree((void*)det_counts_);
        if (det_segs_) (void)hipHostFree(det_segs_);
        if (xd_offsets_) (void)hipHostFree((void*)xd_offsets_);
        xd_offsets_ = nullptr;
        det_counts_ = nullptr;
        det_segs_ = nullptr;
        tables_.reset();
        if (xstats_) xstats_->elapsed_us = current_time_usecs() - start_us_;
        Replica_Base::svc_end();
    }

protected:
    void finalize_stats(Stats_Record& s) override { s.num_kernels = xstats_ ? xstats_->rounds : 0; }

private:
    int world() const { return tr_ ? tr_->world() : 1; }  // no transport: the KeyBy inside one GPU
    int rank() const { return tr_ ? tr_->rank() : 0; }
    static constexpr size_t MAX_GROUP = 16;                            // input batches merged into one round
    static constexpr uint64_t WINDOW = dist::ControlBoard::DEPTH - 2;  // rounds published but not yet collected
    using src_list_t = std::vector<std::pair<const tuple_t*, size_t>>;

    // a round prepared for publication: its row and the buffer the peers will read
    struct Staged {
        dist::ControlRow row{};
        gpu::Batch* buf = nullptr;
        std::unique_ptr<gpu::Event> merged;  // the merge kernel (pull transports publish only once the buffer is complete)
    };

    // ready bundles at the head of the queue, at most group_.  Unless `now`, a round that is not full yet waits for more
    // batches until its oldest one has lingered long enough.
    size_t take_ready(std::vector<std::unique_ptr<ExchangeBundle>>& out, bool now) {
        size_t ready = 0;
        while (ready < group_ && ready < pending_.size() && pending_[ready]->packed->ready.done()) ready++;
        if (ready == 0) return 0;
        if (ready < group_ && !now && !done_ && current_time_usecs() - pending_.front()->arrived_us < xcfg_.linger_us) return 0;
        for (size_t i = 0; i < ready; i++) {
            out.emplace_back(pending_.front());
            pending_.pop_front();
        }
        return ready;
    }

    bool stage_round(bool now) {
        std::vector<std::unique_ptr<ExchangeBundle>> take;
        if (take_ready(take, now) == 0) return false;
        const size_t D = (size_t)world() * xcfg_.out_lanes;
        hipStream_t s = stream_->get();
        staged_.emplace_back();
        Staged& st = staged_.back();
        for (auto& b : take) {
            my_cap_ = std::max(my_cap_, b->capacity);
            if (b->hdr.reserved[0] != 0) note_local_punctuation(b->hdr.reserved[0]);
            for (size_t d = 0; d < D; d++) st.row.counts[d] += b->offsets[d + 1] - b->offsets[d];
            st.row.watermark = std::max(st.row.watermark, b->hdr.watermark);
            if (b->hdr.created_ns != 0) st.row.created_ns = st.row.created_ns == 0 ? b->hdr.created_ns : std::min(st.row.created_ns, b->hdr.created_ns);
        }
        if (take.size() == 1) {
            st.buf = take[0]->packed;  // a lone batch goes out as it is
        } else {
            st.buf = acquire(send_pool_, send_cap_, group_ * my_cap_, dist::ControlBoard::DEPTH + 4);
            tables_->clear();
            size_t run = 0;
            for (size_t d = 0; d < D; d++)
                for (auto& b : take) {
                    const size_t lo = b->offsets[d], c = b->offsets[d + 1] - lo;
                    if (c == 0) continue;
                    tables_->segs.push_back({b->packed->template data<tuple_t>() + lo, st.buf->template data<tuple_t>() + run, c});
                    run += c;
                }
            tables_->template launch<tuple_t>(s);
            for (auto& b : take) b->packed->release(s);
            st.merged.reset(new gpu::Event());
            st.merged->record(s);
        }
        st.row.step = local_step_done_;
        st.row.capacity = my_cap_;
        tr_->describe_send(st.buf->slab, st.buf->template data<tuple_t>(), st.row);
        xstats_->bundles += take.size();
        return true;
    }

    // one step of the state machine; true if anything moved
    bool pump() {
        if (world() == 1) return pump_single();
        bool progress = false;
        dist::ControlBoard& board = tr_->board();
        // publish
        // (a round's slot on the board and in the transport is reused DEPTH rounds later: not before its send buffers are back)
        while (pub_ - col_ < WINDOW && board.can_publish(pub_) && (sent_.empty() || sent_.front().first + dist::ControlBoard::DEPTH > pub_)) {
            // a peer that has opened this round is waiting for this rank: answer with what is there
            if (staged_.size() < 2 && (xcfg_.deterministic ? xdet_stage() : stage_round(board.anyone_published(pub_)))) progress = true;
            dist::ControlRow row{};
            gpu::Batch* buf = nullptr;
            if (!staged_.empty()) {
                Staged& st = staged_.front();
                if (tr_->pulls() && st.merged && !st.merged->done()) break;  // the peers would read a half-written buffer
                row = st.row;
                buf = st.buf;
                staged_.pop_front();
            } else {
                if (!board.anyone_published(pub_) && current_time_usecs() - last_round_us_ < xcfg_.idle_round_us)
                    break;  // nothing to say yet, nobody is waiting for this rank, and the last round is recent
                row.step = local_step_done_;
                row.capacity = my_cap_;
                row.watermark = xd_sent_w_;
                row.aux = xd_max_ts_;
            }
            row.held = xd_held_;
            row.done = (done_ && pending_.empty() && staged_.empty() && xdet_drained()) ? 1 : 0;
            board.publish(pub_, row);
            open_.push_back(buf);
            pub_++;
            last_round_us_ = current_time_usecs();
            progress = true;
        }
        // collect
        dist::ControlRow rows[dist::MAX_WORLD];
        while (col_ < pub_ && board.try_collect(col_, rows)) {
            gpu::Batch* buf = open_.front();
            open_.pop_front();
            run_round(col_, rows, buf);
            col_++;
            progress = true;
        }
        if (release_sent()) progress = true;
        return progress;
    }

    void run_round(uint64_t round, const dist::ControlRow* rows, gpu::Batch* buf) {
        gpu::TraceRange trace("wf:exchange_round");
        const int world = tr_->world(), me = tr_->rank();
        const size_t L = xcfg_.out_lanes;
        hipStream_t s = stream_->get();
        // what I receive: chunk[r] = all my lanes from rank r, contiguous in r's send buffer
        size_t chunk[dist::MAX_WORLD], total = 0;
        bool all_done = true, any_traffic = false;
        uint64_t min_step = UINT64_MAX, created = 0;
        dist::PeerSlices sl;
        for (int r = 0; r < world; r++) {
            const dist::ControlRow& row = rows[r];
            chunk[r] = 0;
            size_t before = 0;
            for (size_t d = 0; d < (size_t)world * L; d++) {
                any_traffic = any_traffic || row.counts[d] != 0;
                if (d < (size_t)me * L) before += row.counts[d];
            }
            for (size_t l = 0; l < L; l++) chunk[r] += row.counts[(size_t)me * L + l];
            sl.pull_offset[r] = before * sizeof(tuple_t);
            total += chunk[r];
            all_done = all_done && row.done != 0;
            min_step = std::min(min_step, row.step);
            rank_wm_[r] = std::max(rank_wm_[r], row.watermark);
            rank_done_[r] = row.done != 0;
            rank_aux_[r] = std::max(rank_aux_[r], row.aux);
            if (row.step > rank_step_[r]) {  // everything of the rank's steps up to row.step has a timestamp <= what it has sent so far
                rank_step_[r] = row.step;
                rank_step_ts_[r] = rank_aux_[r];
            }
            xd_peer_held_ = r == 0 ? row.held : std::max<uint64_t>(xd_peer_held_, row.held);
            cap_ = std::max<size_t>(cap_, (size_t)row.capacity);
            if (row.created_ns != 0 && chunk[r] > 0) created = created == 0 ? row.created_ns : std::min(created, row.created_ns);
        }
        // what I forward is the LOW watermark: the minimum over the ranks still producing
        uint64_t wm = UINT64_MAX;
        for (int r = 0; r < world; r++)
            if (!rank_done_[r]) wm = std::min(wm, rank_wm_[r]);
        if (wm == UINT64_MAX)
            for (int r = 0; r < world; r++) wm = (r == 0) ? rank_wm_[r] : std::max(wm, rank_wm_[r]);
        const uint64_t step_out = xcfg_.deterministic ? 0 : take_global_step(min_step);
        gpu::Batch* stage = nullptr;
        tuple_t* land = nullptr;  // where the peers' chunks land, back to back
        if (xcfg_.deterministic) land = xdet_reserve(total);
        if (any_traffic) {
            if (total > 0 && !xcfg_.deterministic) {
                // the staging slab grows with the traffic (a rank that owns the hot keys receives more than its share)
                size_t want = std::max<size_t>(2 * group_ * std::max<size_t>(cap_, 1), 1);
                while (want < total) want *= 2;
                stage = acquire(stage_pool_, stage_cap_, want, cfg_.pool_slabs + 2);
                land = stage->template data<tuple_t>();
            }
            size_t off = 0, mine_before = 0;
            uint64_t sent = 0;
            for (int p = 0; p < world; p++) {
                size_t to_p = 0;
                for (size_t l = 0; l < L; l++) to_p += rows[me].counts[(size_t)p * L + l];
                if (to_p > 0) {
                    sl.send[p] = buf->template data<tuple_t>() + mine_before;
                    sl.send_bytes[p] = to_p * sizeof(tuple_t);
                    sent += to_p;
                }
                mine_before += to_p;
                if (chunk[p] > 0) {
                    sl.recv[p] = land + off;
                    sl.recv_bytes[p] = chunk[p] * sizeof(tuple_t);
                    off += chunk[p];
                }
                if (p != me) {
                    xstats_->max_link_bytes_per_round = std::max<uint64_t>(xstats_->max_link_bytes_per_round, std::max(sl.send_bytes[p], sl.recv_bytes[p]));
                    xstats_->link_bytes_sent[p] += sl.send_bytes[p];
                }
            }
            xstats_->tuples_sent += sent;
            xstats_->tuples_received += total;
            xstats_->bytes_received += total * sizeof(tuple_t);
        } else {
            xstats_->empty_rounds++;
        }
        // every rank calls exchange() every round, in round order (the collectives of a communicator must line up)
        tr_->exchange(round, sl, rows, s);
        sent_.push_back({round, buf});
        if (xcfg_.deterministic) {
            xdet_receive(chunk, total, wm, all_done, min_step, created, s);
            xstats_->rounds++;
            if (all_done && !any_traffic && xd_held_ == 0) finished_ = true;
            return;
        }
        if (total > 0 || step_out != 0) {
            src_list_t lane_src[dist::MAX_LANES];
            size_t src = 0;
            for (int r = 0; r < world; r++)
                for (size_t l = 0; l < L; l++) {
                    const size_t c = rows[r].counts[(size_t)me * L + l];
                    if (c > 0) lane_src[l].push_back({stage->template data<tuple_t>() + src, c});
                    src += c;
                }
            emit_lanes(lane_src, wm, created, step_out, s);
        }
        if (stage != nullptr) stage->release(s);
        xstats_->rounds++;
        if (all_done && !any_traffic) finished_ = true;
    }

    // lane_src[l]: the pieces of lane l's stream in delivery order.  They become batches of at most lane_cap_ tuples; a step
    // punctuation rides on the last batch of every lane (an empty one if the lane received nothing).
    void emit_lanes(src_list_t* lane_src, uint64_t wm, uint64_t created, uint64_t step_out, hipStream_t s) {
        const size_t L = xcfg_.out_lanes;
        if (lane_cap_ == 0) {
            // fixed from here on: the operators behind size their buffers from the first batch they see
            const size_t cap = std::max<size_t>(std::max(cap_, my_cap_), 1);
            lane_cap_ = std::max(cap, (2 * group_ * cap + L - 1) / L);
            lane_pool_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, cfg_.gpu_id, sizeof(tuple_t), lane_cap_, std::max<size_t>(cfg_.pool_slabs, 3) * L));
        }
        tables_->clear();
        std::vector<std::pair<gpu::Batch*, size_t>> outs;
        // One copy launch serves several lane batches, but never more than L of them are held back: a skewed round (every rank
        // sends its whole round to one lane of this rank) needs more batches than the bounded pool owns, and slabs only come back
        // once the batches acquired before them have been handed downstream.
        auto hand_over = [&]() {
            tables_->template launch<tuple_t>(s);
            for (auto& ol : outs) {
                ol.first->ready.record(s);
                emit_to(ol.first, ol.second);
            }
            xstats_->lane_batches += outs.size();
            outs.clear();
            tables_->clear();
        };
        for (size_t l = 0; l < L; l++) {
            size_t lane_total = 0;
            for (auto& pc : lane_src[l]) lane_total += pc.second;
            if (lane_total == 0 && step_out == 0) continue;
            size_t piece = 0, piece_off = 0, left = lane_total;
            do {
                const size_t n = std::min(left, lane_cap_);
                gpu::Batch* out = lane_pool_->acquire(s);
                out->host_hdr = gpu::BatchHeader{};
                out->host_hdr.count = (uint32_t)n;
                out->host_hdr.source_id = (uint32_t)rank();
                out->host_hdr.watermark = wm;
                out->host_hdr.created_ns = created;
                out->host_hdr.first_seq = seq_;
                out->host_hdr.reserved[0] = (left == n) ? step_out : 0;
                out->count_on_host = true;
                gpu::HdrWrite hw{};
                hw.value = out->host_hdr;
                hw.dst = out->header();
                tables_->hdrs.push_back(hw);
                size_t fill = 0;
                while (fill < n) {
                    const size_t c = std::min(n - fill, lane_src[l][piece].second - piece_off);
                    tables_->segs.push_back({lane_src[l][piece].first + piece_off, out->template data<tuple_t>() + fill, c});
                    fill += c;
                    piece_off += c;
                    if (piece_off == lane_src[l][piece].second) {
                        piece++;
                        piece_off = 0;
                    }
                }
                seq_ += n;
                left -= n;
                outs.push_back({out, l});
                if (outs.size() >= L) hand_over();
            } while (left > 0);
        }
        if (!outs.empty()) hand_over();
    }

    // send buffers whose round the peers are done with go back to their pool
    bool release_sent() {
        bool any = false;
        while (!sent_.empty() && tr_->sends_done(sent_.front().first)) {
            if (sent_.front().second != nullptr) sent_.front().second->release_host();
            sent_.pop_front();
            any = true;
        }
        return any;
    }

    // one rank: the exchange is the device KeyBy; the ready packed slabs of all local lanes are merged into lane batches
    bool pump_single() {
        bool progress = false;
        hipStream_t s = stream_->get();
        while (!pending_.empty()) {
            if (pending_.front()->passthrough != nullptr) {  // one lane: nothing to partition
                std::unique_ptr<ExchangeBundle> b(pending_.front());
                pending_.pop_front();
                if (b->hdr.reserved[0] != 0) note_local_punctuation(b->hdr.reserved[0]);
                b->passthrough->host_hdr.reserved[0] = take_global_step(local_step_done_);  // the host view of the header is what travels downstream
                emit_to(b->passthrough, 0);
                xstats_->bundles++;
                xstats_->lane_batches++;
            } else {
                std::vector<std::unique_ptr<ExchangeBundle>> take;
                if (take_ready(take, false) == 0) break;
                const size_t L = xcfg_.out_lanes;
                src_list_t lane_src[dist::MAX_LANES];
                uint64_t wm = 0, created = 0;
                size_t total = 0;
                for (auto& b : take) {
                    my_cap_ = std::max(my_cap_, b->capacity);
                    if (b->hdr.reserved[0] != 0) note_local_punctuation(b->hdr.reserved[0]);
                    wm = std::max(wm, b->hdr.watermark);
                    if (b->hdr.created_ns != 0) created = created == 0 ? b->hdr.created_ns : std::min(created, b->hdr.created_ns);
                    for (size_t l = 0; l < L; l++) {
                        const size_t lo = b->offsets[l], c = b->offsets[l + 1] - lo;
                        if (c > 0) lane_src[l].push_back({b->packed->template data<tuple_t>() + lo, c});
                        total += c;
                    }
                }
                const uint64_t step_out = take_global_step(local_step_done_);
                if (total > 0 || step_out != 0) emit_lanes(lane_src, wm, created, step_out, s);
                for (auto& b : take) b->packed->release(s);
                xstats_->bundles += take.size();
                xstats_->tuples_received += total;
            }
            xstats_->rounds++;
            progress = true;
        }
        if (done_ && pending_.empty()) finished_ = true;
        return progress;
    }

    // ---- deterministic mode ------------------------------------------------------------------------------------
    void det_init() {
        if (!det_q_.empty()) return;
        const size_t n = std::max<size_t>(num_in(), 1);
        det_q_.resize(n);
        det_wm_.assign(n, 0);
        det_seen_.assign(n, false);
        det_ended_.assign(n, false);
        WF_HIP_CHECK(hipHostMalloc((void**)&det_counts_, DET_MAX_SEGS * sizeof(uint32_t), hipHostMallocDefault));
        WF_HIP_CHECK(hipHostMalloc((void**)&det_segs_, DET_MAX_SEGS * sizeof(gpu::SegCopy), hipHostMallocDefault));
        det_n_dev_.resize(1);
    }
    void det_arrive(ExchangeBundle* b, int channel) {
        det_init();
        const size_t c = (size_t)channel;
        if (c >= det_q_.size()) gpu::failwith("Exchange_GPU: unexpected input channel");
        my_cap_ = std::max(my_cap_, b->capacity);
        det_q_[c].push_back(b);
        det_wm_[c] = std::max(det_wm_[c], b->hdr.watermark);
        det_seen_[c] = true;
        if (world() > 1)
            pump();
        else
            det_release(false);
    }
    // release what the watermark rule allows (everything at end of stream)
    void det_release(bool final) {
        const size_t L = xcfg_.out_lanes;
        hipStream_t s = stream_->get();
        uint64_t W = UINT64_MAX;
        bool any_open = false;
        for (size_t c = 0; c < det_q_.size(); c++) {
            if (det_ended_[c]) continue;
            any_open = true;
            if (!det_seen_[c]) return;  // a producer that has said nothing yet holds everything back
            W = std::min(W, det_wm_[c]);
        }
        if (final || !any_open) W = UINT64_MAX;
        if (W != UINT64_MAX && det_released_once_ && W <= det_last_w_) return;  // nothing new can be released
        // the (bundle, lane) segments not yet released; those of a bundle that may hold later tuples are cut by a binary search
        struct Piece {
            ExchangeBundle* b;
            size_t lane, off, len, take;
            int probe;  // index in the probe table, -1: the whole piece goes
        };
        std::vector<Piece> pieces;
        uint32_t n_probe = 0;
        uint64_t hi_ts = det_last_w_;
        for (auto& q : det_q_)
            for (ExchangeBundle* b : q) {
                b->packed->ready.host_sync();  // the offsets are on the host from here on
                // watermark = what the producer claims to have delivered completely; reserved[2] (when set) = greatest timestamp
                // the batch may hold, for batches that claim less than they carry (a release cut into several batches)
                const uint64_t upper = b->hdr.reserved[2] != 0 ? b->hdr.reserved[2] : b->hdr.watermark;
                const bool whole = upper <= W;
                for (size_t l = 0; l < L; l++) {
                    const size_t lo = b->offsets[l] + b->consumed[l], end = b->offsets[l + 1];
                    if (lo >= end) continue;
                    Piece pc{b, l, lo, end - lo, end - lo, -1};
                    if (!whole) {
                        if (n_probe >= DET_MAX_SEGS) gpu::failwith("Exchange_GPU: too many pending segments in deterministic mode");
                        det_segs_[n_probe] = gpu::SegCopy{b->packed->template data<tuple_t>() + lo, nullptr, end - lo};
                        pc.probe = (int)n_probe++;
                    }
                    pieces.push_back(pc);
                }
                hi_ts = std::max(hi_ts, std::min(W, upper));
            }
        if (n_probe > 0) {
            // on a stream of its own: the answer is needed on the host now, but the sorts and gathers of the previous release
            // (stream s) need not be over for it (the packed slabs it reads are complete: host_sync above)
            if (!det_probe_stream_) det_probe_stream_.reset(new gpu::Stream());
            hipStream_t ps = det_probe_stream_->get();
            hipLaunchKernelGGL((gpu::det_count_le_kernel<tuple_t>), dim3((n_probe + gpu::BLOCK_THREADS - 1) / gpu::BLOCK_THREADS), dim3(gpu::BLOCK_THREADS), 0, ps,
                               det_segs_, n_probe, W, det_counts_);
            WF_HIP_CHECK(hipStreamSynchronize(ps));
            for (Piece& pc : pieces)
                if (pc.probe >= 0) pc.take = det_counts_[pc.probe];
        }
        // per lane: concatenate the released pieces, sort them by timestamp (stable: ties keep producer, position order), cut
        // the result into lane batches
        uint64_t step_out = 0;
        std::vector<ExchangeBundle*> finished_bundles;
        for (auto& q : det_q_)
            while (!q.empty()) {
                ExchangeBundle* b = q.front();
                bool all = true;
                for (size_t l = 0; l < L; l++) {
                    size_t take = 0;
                    for (const Piece& pc : pieces)
                        if (pc.b == b && pc.lane == l) take = pc.take;
                    if (b->offsets[l] + b->consumed[l] + take < b->offsets[l + 1]) all = false;
                }
                if (!all) break;
                if (b->hdr.reserved[0] != 0) note_local_punctuation(b->hdr.reserved[0]);
                finished_bundles.push_back(b);
                q.pop_front();
            }
        step_out = take_global_step(local_step_done_);
        if (lane_cap_ == 0) {
            const size_t cap = std::max<size_t>(my_cap_, 1);
            lane_cap_ = cap;
            lane_pool_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, cfg_.gpu_id, sizeof(tuple_t), lane_cap_, std::max<size_t>(cfg_.pool_slabs, 3) * L));
        }
        const uint64_t base = det_last_w_;
        uint64_t created = 0;  // oldest tuple of the release (latency accounting downstream)
        for (const Piece& pc : pieces)
            if (pc.take > 0 && pc.b->hdr.created_ns != 0) created = created == 0 ? pc.b->hdr.created_ns : std::min(created, pc.b->hdr.created_ns);
        // All replicas' tuples of the release go through ONE sort: the work buffer is filled replica by replica (inside a
        // replica: producer by producer, in arrival order) and the replica index sits on top of the sort key.
        std::vector<size_t> lane_n(L, 0), lane_off(L + 1, 0), cursor(L, 0);
        for (const Piece& pc : pieces) lane_n[pc.lane] += pc.take;
        for (size_t l = 0; l < L; l++) lane_off[l + 1] = lane_off[l] + lane_n[l];
        const size_t total = lane_off[L];
        if (total > UINT32_MAX) gpu::failwith("Exchange_GPU: too many tuples in one deterministic release");
        const uint32_t* perm = nullptr;
        if (total > 0) {
            if (det_work_.size() < total) {
                WF_HIP_CHECK(hipStreamSynchronize(s));  // the old buffers may still be read
                det_work_.resize(total + total / 2);
                det_keys_.resize(total + total / 2);
                det_idx_.resize(total + total / 2);
            }
            tables_->clear();
            for (const Piece& pc : pieces) {
                if (pc.take == 0) continue;
                tables_->segs.push_back({pc.b->packed->template data<tuple_t>() + pc.off, det_work_.data() + lane_off[pc.lane] + cursor[pc.lane], pc.take});
                cursor[pc.lane] += pc.take;
            }
            tables_->template launch<tuple_t>(s);
            const uint32_t n32 = (uint32_t)total;
            hipLaunchKernelGGL(gpu::det_set_u32_kernel, dim3(1), dim3(1), 0, s, det_n_dev_.data(), n32);
            const uint64_t range = hi_ts >= base ? hi_ts - base : 0;
            int bits = 1, lane_bits = 0;
            while (bits < 64 && (range >> bits) != 0) bits++;
            while ((size_t(1) << lane_bits) < L) lane_bits++;
            gpu::DetLaneOff lo{};
            for (size_t l = 0; l <= L; l++) lo.off[l] = (uint32_t)lane_off[l];
            const unsigned grid = gpu::grid_for(total);
            if (bits + lane_bits <= 32) {
                hipLaunchKernelGGL((gpu::det_keys_kernel<tuple_t>), dim3(grid), dim3(gpu::BLOCK_THREADS), 0, s, det_work_.data(), n32, base, 0,
                                   (const uint32_t*)nullptr, det_keys_.data(), det_idx_.data(), lo, (uint32_t)L, lane_bits ? bits : -1);
                perm = det_sorter_.sort(det_keys_.data(), det_idx_.data(), det_n_dev_.data(), total, bits + lane_bits, s).vals;
            } else {
                // LSD: the low half of the timestamps, then a second stable sort on (replica, high half)
                const int hi_bits = bits > 32 ? bits - 32 : 0;  // (31 or 32 bits of time plus the replica index: no high half)
                if (hi_bits + lane_bits > 32) gpu::failwith("Exchange_GPU: timestamps of one deterministic release span more than 2^60");
                hipLaunchKernelGGL((gpu::det_keys_kernel<tuple_t>), dim3(grid), dim3(gpu::BLOCK_THREADS), 0, s, det_work_.data(), n32, base, 0,
                                   (const uint32_t*)nullptr, det_keys_.data(), det_idx_.data(), lo, (uint32_t)L, -1);
                auto sorted = det_sorter_.sort(det_keys_.data(), det_idx_.data(), det_n_dev_.data(), total, 32, s);
                if (det_keys2_.size() < det_keys_.size()) {
                    WF_HIP_CHECK(hipStreamSynchronize(s));  // the second round of an earlier release may still be reading them
                    det_keys2_.resize(det_keys_.size());
                    det_idx2_.resize(det_keys_.size());
                }
                hipLaunchKernelGGL((gpu::det_keys_kernel<tuple_t>), dim3(grid), dim3(gpu::BLOCK_THREADS), 0, s, det_work_.data(), n32, base, 32,
                                   (const uint32_t*)sorted.vals, det_keys2_.data(), det_idx2_.data(), lo, (uint32_t)L, lane_bits ? hi_bits : -1);
                perm = det_sorter2_.sort(det_keys2_.data(), det_idx2_.data(), det_n_dev_.data(), total, std::max(1, hi_bits + lane_bits), s).vals;
            }
        }
        for (size_t l = 0; l < L; l++) {
            const size_t n = lane_n[l];
            // a lane that receives nothing still learns the new watermark (an empty batch): the operator behind it is a
            // producer of the next ordering point, which would otherwise wait for it for ever
            if (n == 0 && step_out == 0 && hi_ts <= base) continue;
            size_t done = 0;
            do {
                const size_t cnt = std::min(n - done, lane_cap_);
                gpu::Batch* out = lane_pool_->acquire(s);
                out->host_hdr = gpu::BatchHeader{};
                out->host_hdr.count = (uint32_t)cnt;
                out->host_hdr.source_id = (uint32_t)rank();
                // only the last batch of a release may claim the new watermark: the ones before it hold a part of the release
                const bool last = done + cnt == n;
                out->host_hdr.watermark = last ? hi_ts : base;
                out->host_hdr.reserved[2] = last ? 0 : hi_ts;
                out->host_hdr.first_seq = seq_;
                out->host_hdr.reserved[0] = last ? step_out : 0;
                out->host_hdr.created_ns = created;
                out->count_on_host = true;
                // (the header travels by value in a pinned table; the work and sort buffers are reused by the next release in stream order)
                tables_->clear();
                tables_->hdrs.push_back(gpu::HdrWrite{out->host_hdr, out->header(), {}});
                tables_->template launch<tuple_t>(s);
                if (cnt > 0)
                    hipLaunchKernelGGL((gpu::det_gather_kernel<tuple_t>), dim3(gpu::grid_for(cnt)), dim3(gpu::BLOCK_THREADS), 0, s, det_work_.data(), perm,
                                       out->template data<tuple_t>(), (uint32_t)(lane_off[l] + done), (uint32_t)cnt);
                out->ready.record(s);
                emit_to(out, l);
                xstats_->lane_batches++;
                seq_ += cnt;
                done += cnt;
            } while (done < n);
            xstats_->tuples_received += n;
        }
        for (Piece& pc : pieces) pc.b->consumed[pc.lane] += pc.take;
        for (ExchangeBundle* b : finished_bundles) {
            b->packed->release(s);
            delete b;
            xstats_->bundles++;
        }
        xstats_->rounds++;
        if (W != UINT64_MAX) {
            det_last_w_ = W;
            det_released_once_ = true;
        } else {
            det_last_w_ = hi_ts;
        }
    }

    // ---- deterministic mode across ranks ------------------------------------------------------------------------
    // Two levels of the low-watermark rule (reference: wf/ordering_node.hpp:154-193, per input channel of ONE node):
    //  * sending side: W_local = least of the latest watermarks of this rank's producer lanes.  A round carries every pending
    //    tuple with timestamp <= W_local, grouped by destination (not sorted: the receiver sorts anyway), and its row claims
    //    W_local: "everything this rank will ever send with a timestamp <= W_local has been sent".
    //  * receiving side: W = least of the ranks' claims (ranks that are done do not count).  Everything received so far with a
    //    timestamp <= W is released, per output lane, in (timestamp, source rank, arrival) order with ONE radix sort; the rest
    //    stays in a pending buffer on the device - the chunks of a round land directly behind it.
    // Senders that run ahead of the slowest rank see the receivers' pending buffers grow (ControlRow::held) and hold their
    // data back (they keep publishing their watermark), so the buffers stay bounded.
    bool xdet_drained() const {
        if (!xcfg_.deterministic || world() == 1) return true;
        for (auto& q : det_q_)
            if (!q.empty()) return false;
        return true;
    }
    // stage the next round of this rank; true if a round was staged
    bool xdet_stage() {
        det_init();
        const size_t L = xcfg_.out_lanes, D = (size_t)world() * L;
        hipStream_t s = stream_->get();
        uint64_t W = UINT64_MAX;
        bool any_open = false;
        for (size_t c = 0; c < det_q_.size(); c++) {
            if (det_ended_[c]) continue;
            any_open = true;
            if (!det_seen_[c]) return false;  // a producer that has said nothing yet holds everything back
            W = std::min(W, det_wm_[c]);
        }
        if (done_ || !any_open) W = UINT64_MAX;
        const bool advanced = !xd_staged_once_ || W > xd_sent_w_;
        if (!advanced) return false;
        // a rank that is ahead while some receiver sits on a large pending buffer keeps its data for now
        if (W != UINT64_MAX && xd_peer_held_ > XD_HELD_LIMIT && xd_sent_w_ > xd_released_w_) return false;
        struct Piece {
            ExchangeBundle* b;
            size_t dest, off, take;
            int probe;
        };
        std::vector<Piece> pieces;
        uint32_t n_probe = 0;
        for (auto& q : det_q_)
            for (ExchangeBundle* b : q) {
                b->packed->ready.host_sync();  // the offsets are on the host from here on
                const uint64_t upper = b->hdr.reserved[2] != 0 ? b->hdr.reserved[2] : b->hdr.watermark;
                xd_max_ts_ = std::max(xd_max_ts_, W == UINT64_MAX ? upper : std::min(W, upper));
                const bool whole = upper <= W;
                for (size_t d = 0; d < D; d++) {
                    const size_t lo = b->offsets[d] + b->consumed[d], end = b->offsets[d + 1];
                    if (lo >= end) continue;
                    Piece pc{b, d, lo, end - lo, -1};
                    if (!whole) {
                        if (n_probe >= DET_MAX_SEGS) gpu::failwith("Exchange_GPU: too many pending segments in deterministic mode");
                        det_segs_[n_probe] = gpu::SegCopy{b->packed->template data<tuple_t>() + lo, nullptr, end - lo};
                        pc.probe = (int)n_probe++;
                    }
                    pieces.push_back(pc);
                }
            }
        if (n_probe > 0) {
            if (!det_probe_stream_) det_probe_stream_.reset(new gpu::Stream());
            hipStream_t ps = det_probe_stream_->get();
            hipLaunchKernelGGL((gpu::det_count_le_kernel<tuple_t>), dim3((n_probe + gpu::BLOCK_THREADS - 1) / gpu::BLOCK_THREADS), dim3(gpu::BLOCK_THREADS), 0, ps,
                               det_segs_, n_probe, W, det_counts_);
            WF_HIP_CHECK(hipStreamSynchronize(ps));
            for (Piece& pc : pieces)
                if (pc.probe >= 0) pc.take = det_counts_[pc.probe];
        }
        staged_.emplace_back();
        Staged& st = staged_.back();
        size_t total = 0;
        for (const Piece& pc : pieces) {
            st.row.counts[pc.dest] += (uint32_t)pc.take;
            total += pc.take;
            if (pc.take > 0 && pc.b->hdr.created_ns != 0)
                st.row.created_ns = st.row.created_ns == 0 ? pc.b->hdr.created_ns : std::min(st.row.created_ns, pc.b->hdr.created_ns);
        }
        for (auto& q : det_q_)
            for (ExchangeBundle* b : q) my_cap_ = std::max(my_cap_, b->capacity);
        if (total > 0) {
            size_t want = std::max<size_t>(std::max(send_cap_, group_ * std::max<size_t>(my_cap_, 1)), 1);
            while (want < total) want *= 2;
            st.buf = acquire(send_pool_, send_cap_, want, dist::ControlBoard::DEPTH + 4);
            tables_->clear();
            size_t run = 0;
            std::stable_sort(pieces.begin(), pieces.end(), [](const Piece& a, const Piece& b) { return a.dest < b.dest; });  // destination-major, arrival order inside
            for (const Piece& pc : pieces) {
                if (pc.take == 0) continue;
                tables_->segs.push_back({pc.b->packed->template data<tuple_t>() + pc.off, st.buf->template data<tuple_t>() + run, pc.take});
                run += pc.take;
            }
            tables_->template launch<tuple_t>(s);
            st.merged.reset(new gpu::Event());
            st.merged->record(s);
            tr_->describe_send(st.buf->slab, st.buf->template data<tuple_t>(), st.row);
        }
        for (Piece& pc : pieces) pc.b->consumed[pc.dest] += pc.take;
        // bundles that have given everything go back to their lanes
        for (auto& q : det_q_)
            while (!q.empty()) {
                ExchangeBundle* b = q.front();
                bool all = true;
                for (size_t d = 0; d < D && all; d++) all = b->offsets[d] + b->consumed[d] >= b->offsets[d + 1];
                if (!all) break;
                if (b->hdr.reserved[0] != 0) note_local_punctuation(b->hdr.reserved[0]);
                b->packed->release(s);
                delete b;
                q.pop_front();
                xstats_->bundles++;
            }
        xd_sent_w_ = W;
        xd_staged_once_ = true;
        st.row.watermark = W;
        st.row.aux = xd_max_ts_;
        st.row.step = local_step_done_;
        st.row.capacity = my_cap_;
        return true;
    }
    // room for `incoming` tuples behind the pending ones; returns where they land
    tuple_t* xdet_reserve(size_t incoming) {
        const size_t need = xd_held_ + incoming;
        if (need > xd_cap_) {
            WF_HIP_CHECK(hipStreamSynchronize(stream_->get()));  // the old buffers may still be read
            size_t cap = std::max<size_t>(xd_cap_, 1 << 16);
            while (cap < need) cap *= 2;
            gpu::DeviceBuffer<tuple_t> np(cap);
            gpu::DeviceBuffer<uint32_t> nr(cap);
            if (xd_held_ > 0) {
                WF_HIP_CHECK(hipMemcpyAsync(np.data(), xd_pend_[xd_cur_].data(), xd_held_ * sizeof(tuple_t), hipMemcpyDeviceToDevice, stream_->get()));
                WF_HIP_CHECK(hipMemcpyAsync(nr.data(), xd_rank_[xd_cur_].data(), xd_held_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_->get()));
                WF_HIP_CHECK(hipStreamSynchronize(stream_->get()));
            }
            xd_pend_[xd_cur_] = std::move(np);
            xd_rank_[xd_cur_] = std::move(nr);
            xd_pend_[1 - xd_cur_] = gpu::DeviceBuffer<tuple_t>(cap);
            xd_rank_[1 - xd_cur_] = gpu::DeviceBuffer<uint32_t>(cap);
            xd_keys_ = gpu::DeviceBuffer<uint32_t>(cap);
            xd_idx_ = gpu::DeviceBuffer<uint32_t>(cap);
            xd_keys2_ = gpu::DeviceBuffer<uint32_t>(cap);
            xd_idx2_ = gpu::DeviceBuffer<uint32_t>(cap);
            xd_cap_ = cap;
        }
        return xd_pend_[xd_cur_].data() + xd_held_;
    }
    // the chunks of a round have been enqueued behind the pending tuples: release what the ranks' claims allow
    void xdet_receive(const size_t* chunk, size_t total, uint64_t W, bool all_done, uint64_t min_step, uint64_t created, hipStream_t s) {
        const int world = tr_->world();
        const size_t L = xcfg_.out_lanes;
        if (all_done) W = UINT64_MAX;
        if (!det_n_dev_.data()) det_n_dev_.resize(1);
        if (xd_offsets_ == nullptr) WF_HIP_CHECK(hipHostMalloc((void**)&xd_offsets_, (dist::MAX_LANES + 2) * sizeof(uint32_t), hipHostMallocDefault));
        if (created != 0) xd_created_ = xd_created_ == 0 ? created : std::min(xd_created_, created);
        if (total > 0) {
            gpu::XdetRankOff ro{};
            size_t acc = 0;
            for (int r = 0; r < world; r++) {
                ro.off[r] = (uint32_t)acc;
                acc += chunk[r];
            }
            ro.off[world] = (uint32_t)acc;
            hipLaunchKernelGGL(gpu::xdet_fill_ranks_kernel, dim3(gpu::grid_for(total)), dim3(gpu::BLOCK_THREADS), 0, s, xd_rank_[xd_cur_].data() + xd_held_, ro,
                               (uint32_t)world);
            xstats_->tuples_received += 0;  // (counted by run_round)
        }
        const size_t n = xd_held_ + total;
        // which steps may be announced: every rank has completed them AND nothing of them can still be pending
        uint64_t step_ts = 0;
        for (int r = 0; r < world; r++) step_ts = std::max(step_ts, rank_step_ts_[r]);
        const bool nothing_new = n == 0 || (W != UINT64_MAX && xd_released_once_ && W <= xd_released_w_);
        if (nothing_new) {
            // (the watermark did not move: the pending tuples stay; a step can still complete when nothing is pending)
            xd_held_ = n;
            if (n == 0) xdet_announce(min_step, W == UINT64_MAX ? xd_released_w_ : std::max(W, xd_released_w_), s);
            return;
        }
        if (n > UINT32_MAX) gpu::failwith("Exchange_GPU: too many pending tuples in deterministic mode");
        const uint32_t n32 = (uint32_t)n;
        const uint64_t base = xd_released_w_;
        uint64_t hi = W;
        if (W == UINT64_MAX) {
            hi = base;
            for (int r = 0; r < world; r++) hi = std::max(hi, rank_aux_[r]);
        }
        const uint64_t range = hi >= base ? hi - base : 0;
        int ts_bits = 1, lane_bits = 1, rank_bits = 0;
        while (ts_bits < 64 && (range >> ts_bits) != 0) ts_bits++;
        while ((size_t(1) << lane_bits) < L + 1) lane_bits++;
        while ((1 << rank_bits) < world) rank_bits++;
        if (ts_bits + lane_bits + rank_bits > 64) gpu::failwith("Exchange_GPU: timestamps of one deterministic release span more than 2^50");
        const tuple_t* P = xd_pend_[xd_cur_].data();
        const uint32_t* PR = xd_rank_[xd_cur_].data();
        hipLaunchKernelGGL(gpu::det_set_u32_kernel, dim3(1), dim3(1), 0, s, det_n_dev_.data(), n32);
        const unsigned grid = gpu::grid_for(n);
        const int total_bits = ts_bits + lane_bits + rank_bits;
        const uint32_t* perm = nullptr;
        hipLaunchKernelGGL((gpu::xdet_keys_kernel<tuple_t>), dim3(grid), dim3(gpu::BLOCK_THREADS), 0, s, P, PR, n32, W, base, (uint32_t)L, ts_bits, rank_bits, 0,
                           (const uint32_t*)nullptr, xd_keys_.data(), xd_idx_.data());
        if (total_bits <= 32) {
            perm = det_sorter_.sort(xd_keys_.data(), xd_idx_.data(), det_n_dev_.data(), n, total_bits, s).vals;
        } else {
            auto low = det_sorter_.sort(xd_keys_.data(), xd_idx_.data(), det_n_dev_.data(), n, 32, s);
            hipLaunchKernelGGL((gpu::xdet_keys_kernel<tuple_t>), dim3(grid), dim3(gpu::BLOCK_THREADS), 0, s, P, PR, n32, W, base, (uint32_t)L, ts_bits, rank_bits,
                               32, (const uint32_t*)low.vals, xd_keys2_.data(), xd_idx2_.data());
            perm = det_sorter2_.sort(xd_keys2_.data(), xd_idx2_.data(), det_n_dev_.data(), n, total_bits - 32, s).vals;
        }
        hipLaunchKernelGGL((gpu::xdet_offsets_kernel<tuple_t>), dim3(grid), dim3(gpu::BLOCK_THREADS), 0, s, P, perm, n32, W, (uint32_t)L, xd_offsets_);
        WF_HIP_CHECK(hipStreamSynchronize(s));  // the lane counts size the batches
        uint32_t off[dist::MAX_LANES + 2];
        for (size_t l = 0; l <= L + 1; l++) off[l] = xd_offsets_[l];
        const size_t held = off[L + 1] - off[L];
        xdet_lane_pool();
        // may the step be announced with this release?  (decided before the batches are cut: the punctuation rides on the last one)
        uint64_t step_out = 0;
        if (min_step != UINT64_MAX && min_step > announced_step_ && (held == 0 || hi >= step_ts)) step_out = take_global_step(min_step);
        for (size_t l = 0; l < L; l++) {
            const size_t cnt_l = off[l + 1] - off[l];
            if (cnt_l == 0 && step_out == 0 && hi <= base) continue;
            size_t done = 0;
            do {
                const size_t cnt = std::min(cnt_l - done, lane_cap_);
                gpu::Batch* out = lane_pool_->acquire(s);
                out->host_hdr = gpu::BatchHeader{};
                out->host_hdr.count = (uint32_t)cnt;
                out->host_hdr.source_id = (uint32_t)rank();
                const bool last = done + cnt == cnt_l;
                out->host_hdr.watermark = last ? hi : base;
                out->host_hdr.reserved[2] = last ? 0 : hi;
                out->host_hdr.first_seq = seq_;
                out->host_hdr.reserved[0] = last ? step_out : 0;
                out->host_hdr.created_ns = xd_created_;
                out->count_on_host = true;
                tables_->clear();
                tables_->hdrs.push_back(gpu::HdrWrite{out->host_hdr, out->header(), {}});
                tables_->template launch<tuple_t>(s);
                if (cnt > 0)
                    hipLaunchKernelGGL((gpu::det_gather_kernel<tuple_t>), dim3(gpu::grid_for(cnt)), dim3(gpu::BLOCK_THREADS), 0, s, P, perm,
                                       out->template data<tuple_t>(), (uint32_t)(off[l] + done), (uint32_t)cnt);
                out->ready.record(s);
                emit_to(out, l);
                xstats_->lane_batches++;
                seq_ += cnt;
                done += cnt;
            } while (done < cnt_l);
        }
        // what stays pending moves to the front of the other buffer (arrival order kept)
        if (held > 0) {
            hipLaunchKernelGGL((gpu::det_gather_kernel<tuple_t>), dim3(gpu::grid_for(held)), dim3(gpu::BLOCK_THREADS), 0, s, P, perm, xd_pend_[1 - xd_cur_].data(),
                               off[L], (uint32_t)held);
            hipLaunchKernelGGL(gpu::xdet_gather_u32_kernel, dim3(gpu::grid_for(held)), dim3(gpu::BLOCK_THREADS), 0, s, PR, perm, xd_rank_[1 - xd_cur_].data(), off[L],
                               (uint32_t)held);
        } else {
            xd_created_ = 0;
        }
        xd_cur_ = 1 - xd_cur_;
        xd_held_ = held;
        xd_released_w_ = hi;
        xd_released_once_ = true;
    }
    void xdet_lane_pool() {
        if (lane_cap_ != 0) return;
        const size_t L = xcfg_.out_lanes;
        const size_t cap = std::max<size_t>(std::max(cap_, my_cap_), 1);
        lane_cap_ = std::max(cap, (2 * group_ * cap + L - 1) / L);
        lane_pool_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, cfg_.gpu_id, sizeof(tuple_t), lane_cap_, std::max<size_t>(cfg_.pool_slabs, 3) * L));
    }
    // a step completes although this round released nothing: empty batches carry the punctuation
    void xdet_announce(uint64_t min_step, uint64_t wm, hipStream_t s) {
        if (min_step == UINT64_MAX || min_step <= announced_step_) return;
        const uint64_t step_out = take_global_step(min_step);
        if (step_out == 0) return;
        xdet_lane_pool();
        for (size_t l = 0; l < xcfg_.out_lanes; l++) {
            gpu::Batch* out = lane_pool_->acquire(s);
            out->host_hdr = gpu::BatchHeader{};
            out->host_hdr.source_id = (uint32_t)rank();
            out->host_hdr.watermark = wm;
            out->host_hdr.first_seq = seq_;
            out->host_hdr.reserved[0] = step_out;
            out->count_on_host = true;
            tables_->clear();
            tables_->hdrs.push_back(gpu::HdrWrite{out->host_hdr, out->header(), {}});
            tables_->template launch<tuple_t>(s);
            out->ready.record(s);
            emit_to(out, l);
            xstats_->lane_batches++;
        }
    }

    gpu::Batch* acquire(std::unique_ptr<gpu::BatchPool>& pool, size_t& pool_cap, size_t need, size_t slabs) {
        if (!pool || pool_cap < need) {
            // batches of the old pool may still be in flight: keep it alive until the end
            if (pool) retired_pools_.push_back(std::move(pool));
            pool_cap = need;
            pool.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, cfg_.gpu_id, sizeof(tuple_t), pool_cap, slabs));
        }
        return pool->acquire(stream_->get());
    }

    // a step is locally complete when every local lane has punctuated it
    void note_local_punctuation(uint64_t step) {
        if (++punct_seen_[step] >= xcfg_.local_lanes) {
            local_step_done_ = std::max(local_step_done_, step);
            punct_seen_.erase(step);
        }
    }
    // returns the step id to announce downstream (0: none) given the minimum step all ranks completed
    uint64_t take_global_step(uint64_t min_step) {
        if (min_step != UINT64_MAX && min_step > announced_step_) {
            announced_step_ = min_step;
            return min_step;
        }
        return 0;
    }

    std::shared_ptr<dist::Transport> tr_;
    ExchangeConfig xcfg_;
    GPUConfig cfg_;
    std::shared_ptr<ExchangeStats> xstats_;
    std::shared_ptr<size_t> n_producers_;
    std::unique_ptr<gpu::Stream> stream_;
    std::unique_ptr<gpu::SegTables> tables_;
    std::unique_ptr<gpu::BatchPool> send_pool_, stage_pool_, lane_pool_;
    std::vector<std::unique_ptr<gpu::BatchPool>> retired_pools_;
    size_t group_ = 1;
    size_t my_cap_ = 0, cap_ = 0, send_cap_ = 0, stage_cap_ = 0, lane_cap_ = 0;
    std::deque<ExchangeBundle*> pending_;                  // bundles not yet part of a round
    std::deque<Staged> staged_;                            // rounds prepared, not yet published
    std::deque<gpu::Batch*> open_;                         // send buffer (or nullptr) of every published, not yet collected round
    std::deque<std::pair<uint64_t, gpu::Batch*>> sent_;    // rounds whose send buffers the peers may still be reading
    uint64_t pub_ = 0, col_ = 0;
    uint64_t seq_ = 0, last_round_us_ = 0, start_us_ = 0;
    uint64_t local_step_done_ = 0, announced_step_ = 0;
    uint64_t rank_wm_[dist::MAX_WORLD] = {};
    bool rank_done_[dist::MAX_WORLD] = {};
    std::map<uint64_t, size_t> punct_seen_;
    bool done_ = false, finished_ = false;
    // deterministic mode
    static constexpr uint32_t DET_MAX_SEGS = 16384;
    std::unique_ptr<gpu::Stream> det_probe_stream_;
    std::vector<std::deque<ExchangeBundle*>> det_q_;  // per producer channel, in arrival order
    std::vector<uint64_t> det_wm_;
    std::vector<bool> det_seen_, det_ended_;
    uint64_t det_last_w_ = 0;
    bool det_released_once_ = false;
    volatile uint32_t* det_counts_ = nullptr;
    gpu::SegCopy* det_segs_ = nullptr;
    gpu::DeviceBuffer<tuple_t> det_work_;
    gpu::DeviceBuffer<uint32_t> det_keys_, det_idx_, det_keys2_, det_idx2_, det_n_dev_;
    gpu::RadixSorter det_sorter_, det_sorter2_;
    // deterministic mode across ranks
    static constexpr uint64_t XD_HELD_LIMIT = 32u << 20;  // pending tuples at a receiver from which the ranks that are ahead slow down
    uint64_t xd_sent_w_ = 0, xd_max_ts_ = 0, xd_released_w_ = 0, xd_peer_held_ = 0, xd_created_ = 0;
    bool xd_staged_once_ = false, xd_released_once_ = false;
    uint64_t rank_aux_[dist::MAX_WORLD] = {}, rank_step_[dist::MAX_WORLD] = {}, rank_step_ts_[dist::MAX_WORLD] = {};
    gpu::DeviceBuffer<tuple_t> xd_pend_[2];
    gpu::DeviceBuffer<uint32_t> xd_rank_[2], xd_keys_, xd_idx_, xd_keys2_, xd_idx2_;
    int xd_cur_ = 0;
    size_t xd_held_ = 0, xd_cap_ = 0;
    volatile uint32_t* xd_offsets_ = nullptr;
};

// The two stages in front of a consumer with `lanes` replicas: (1) one packing node per producer, fused onto it;
// (2) the endpoint, whose `lanes` output ports feed the consumer's replicas one to one.  `tr` == nullptr: the KeyBy inside a GPU.
template <typename tuple_t>
inline std::vector<StageSpec> exchange_stages(rt::Graph& g, std::shared_ptr<dist::Transport> tr, const std::string& name, GPUConfig cfg, ExchangeConfig xcfg,
                                              edge_kind_t in_kind, std::shared_ptr<StatsHub> hub, std::shared_ptr<ExchangeStats> xstats = nullptr,
                                              meta::closing_func_t closing = nullptr) {
    cfg.gpu_id = cfg.device_for(0);
    cfg.gpu_ids.clear();
    const int world = tr ? tr->world() : 1;
    auto n_producers = std::make_shared<size_t>(0);
    StageSpec a;
    a.name = name + "_partition";
    a.routing = FORWARD;
    const size_t lanes = xcfg.out_lanes;
    a.per_producer = [=](size_t i, size_t n) -> rt::Node* {
        *n_producers = n;
        return new ExchangePart_Node<tuple_t>(world, lanes, cfg, in_kind, name + "_partition", n, i, nullptr, hub, xcfg.deterministic);
    };
    StageSpec b;
    b.name = name;
    b.routing = FORWARD;
    b.internal = true;
    rt::Node* ep = g.adopt(new ExchangeEndpoint_Node<tuple_t>(tr, xcfg, cfg, name, closing, hub, xstats, n_producers));
    b.replicas.push_back(ep);
    // one output port per lane: the operator behind sees `out_lanes` producers
    for (size_t l = 0; l < lanes; l++) b.outputs.push_back(ep);
    b.make_emitter = [](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<tuple_t>(1), 1); };
    return {a, b};
}

// KeyBy inside one GPU for a keyed operator with `lanes` replicas (see the top of this file)
template <typename tuple_t>
inline std::vector<StageSpec> device_keyby_stages(rt::Graph& g, const std::string& op_name, const GPUConfig& cfg, size_t lanes, edge_kind_t in_kind,
                                                  std::shared_ptr<StatsHub> hub, bool deterministic = false) {
    ExchangeConfig xcfg;
    xcfg.out_lanes = lanes;
    xcfg.deterministic = deterministic;
    return exchange_stages<tuple_t>(g, nullptr, op_name + "_keyby", cfg, xcfg, in_kind, std::move(hub));
}

}  // namespace wf
