// Shared pieces of the benchmark / flagship pipelines: the 40-byte sensor tuple of
// the reference benchmarks (/root/reference/src/GPU_Tests/new_tests/benchmarks/
// gpu_map_stateful.cpp:80-85), a fast synthetic generator, and the step gate /
// barrier that lets a driver time "K steps" on a persistent streaming graph.
#pragma once
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <map>
#include <memory>
#include <chrono>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

#include <wf/basic.hpp>
#include <wf/context.hpp>

// non-temporal store (clang builtin; the host-only unit tests are built with g++, which has none: plain store there)
#if defined(__clang__)
#define WFBENCH_NT_STORE(value, ptr) __builtin_nontemporal_store((value), (ptr))
#else
#define WFBENCH_NT_STORE(value, ptr) (*(ptr) = (value))
#endif

namespace wfbench {

// 40 bytes, same field mix as the reference benchmark tuple
struct tuple_t {
    uint64_t key;
    uint64_t id;
    uint64_t ts;
    double value;
    double aux;
    WF_HOST_DEVICE std::tuple<uint64_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    WF_HOST_DEVICE void setControlFields(uint64_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};
static_assert(sizeof(tuple_t) == 40, "benchmark tuple must be 40 bytes");

// 32-byte window result
struct result_t {
    uint64_t key;
    uint64_t id;
    uint64_t ts;
    double value;
    WF_HOST_DEVICE std::tuple<uint64_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    WF_HOST_DEVICE void setControlFields(uint64_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};
static_assert(sizeof(result_t) == 32, "benchmark result must be 32 bytes");

// lift / combine of the flagship sliding-window sum
struct LiftValue {
    WF_HOST_DEVICE void operator()(const tuple_t& t, result_t& r) const { r.value = t.value; }
};
struct CombSum {
    WF_HOST_DEVICE void operator()(const result_t& a, const result_t& b, result_t& out) const { out.value = a.value + b.value; }
};

// A driver releases `quota` tuples per step; the source closes the step with a
// punctuated batch and sleeps until the next step (or stop).
struct StepGate {
    std::mutex m;
    std::condition_variable cv;
    int64_t quota = 0;      // tuples still to generate in the current step
    uint64_t step_id = 0;   // id of the step the quota belongs to
    bool stop = false;

    void release(uint64_t step, int64_t tuples) {
        {
            std::lock_guard<std::mutex> l(m);
            step_id = step;
            quota = tuples;
        }
        cv.notify_all();
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> l(m);
            stop = true;
        }
        cv.notify_all();
    }
};

// Sources that emit the SAME keys stand for readers of one time-ordered feed: they advance together.  Without this the
// synthetic clocks (a timestamp is a sequence number) of two source threads drift apart by whatever the scheduler does to
// them, and a time-based window downstream sees tuples that are "late" by an arbitrary amount.  A source may be at most
// `slack` batches ahead of the slowest one.
struct SkewBound {
    SkewBound(size_t n, uint64_t slack_batches) : n(n), slack(slack_batches), own(new std::atomic<uint64_t>[n]) {
        for (size_t i = 0; i < n; i++) own[i].store(0, std::memory_order_relaxed);
        rows.push_back(own.get());
        mine = own.get();
    }
    // sources of several ranks: row r holds the counters of rank r's n sources (shared memory: the control board of the
    // ranks' transport); this process advances row `rank`
    SkewBound(std::vector<std::atomic<uint64_t>*> rank_rows, size_t rank, size_t n, uint64_t slack_batches)
        : n(n), slack(slack_batches), rows(std::move(rank_rows)), mine(rows[rank]) {}
    // before source i generates its next batch; `gate` is the source's own gate (a shutdown ends the wait)
    void wait_turn(size_t i, StepGate& gate) {
        const uint64_t me = mine[i].load(std::memory_order_relaxed);
        for (unsigned spins = 0;; spins++) {
            uint64_t slowest = UINT64_MAX;
            for (std::atomic<uint64_t>* row : rows)
                for (size_t j = 0; j < n; j++) slowest = std::min(slowest, row[j].load(std::memory_order_acquire));
            if (me < slowest + slack) return;
            if ((spins & 63) == 63) {
                std::lock_guard<std::mutex> l(gate.m);
                if (gate.stop) return;
            }
            // a batch takes a source ~1 ms: waiting sources sleep instead of competing with the threads they wait for
            if (spins < 16)
                std::this_thread::yield();
            else
                std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    void batch_done(size_t i) { mine[i].fetch_add(1, std::memory_order_release); }
    size_t n;
    uint64_t slack;
    std::unique_ptr<std::atomic<uint64_t>[]> own;
    std::vector<std::atomic<uint64_t>*> rows;
    std::atomic<uint64_t>* mine;
};

struct StepBarrier {
    std::mutex m;
    std::condition_variable cv;
    std::map<uint64_t, int> arrivals;
    void arrive(uint64_t step) {
        {
            std::lock_guard<std::mutex> l(m);
            arrivals[step]++;
        }
        cv.notify_all();
    }
    void wait(uint64_t step, int expected) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return arrivals[step] >= expected; });
        arrivals.erase(step);
    }
};

// How a generator draws the key of the next tuple.  The reference's studies use round-robin keys and, in some runs, a
// zipf generator (/root/reference/src/GPU_Tests/zipf.hpp:14-21, results.org:111); round-robin in ascending order is the
// friendliest stream there is (every batch arrives sorted, one tuple per key), so the unfriendly ones are first-class here.
enum key_dist_t { KEYS_ROUNDROBIN = 0, KEYS_UNIFORM = 1, KEYS_ZIPF = 2 };
struct KeySpace {
    size_t n_keys = 1;
    int dist = KEYS_ROUNDROBIN;
    double zipf_alpha = 1.3;
    bool shared = false;  // every source draws from the same key set (a KeyBy is then needed in front of keyed operators)
    // zipf: rank r has probability ~ 1 / (r+1)^alpha.  Quantile table with 2^16 buckets of equal probability; a bucket holds the
    // first rank that falls in it and how many ranks it spans: hot ranks own many buckets, the ranks of the tail share theirs and
    // are drawn uniformly inside it.  (512 KB, cache resident: a generator thread must not stall on DRAM for every tuple; the
    // mass of every bucket is exact, the shape inside a tail bucket is flattened.)
    struct Bucket {
        uint32_t first, span;
    };
    std::shared_ptr<std::vector<Bucket>> quantiles;
    static constexpr int QBITS = 16;
    void prepare() {
        if (dist != KEYS_ZIPF || quantiles) return;
        const size_t n = n_keys, q = size_t(1) << QBITS;
        std::vector<double> cdf(n);
        long double acc = 0;
        for (size_t r = 0; r < n; r++) {
            acc += std::pow((double)(r + 1), -zipf_alpha);
            cdf[r] = (double)acc;
        }
        auto tab = std::make_shared<std::vector<Bucket>>(q);
        size_t r = 0;
        for (size_t i = 0; i < q; i++) {
            // ranks whose cumulative probability interval meets bucket i = [i, i+1) / q
            const double lo = (double)i / (double)q * (double)acc, hi = (double)(i + 1) / (double)q * (double)acc;
            while (r + 1 < n && cdf[r] <= lo) r++;
            size_t e = r;
            while (e + 1 < n && cdf[e] < hi) e++;
            (*tab)[i] = Bucket{(uint32_t)r, (uint32_t)(e - r + 1)};
        }
        quantiles = tab;
    }
    // rank drawn by 64 random bits
    static WF_HOST_DEVICE inline uint64_t zipf_rank(const Bucket* tab, uint64_t rnd) {
        const Bucket b = tab[rnd >> (64 - QBITS)];
        return b.span == 1 ? b.first : b.first + (uint64_t)(((rnd & 0xffffffffull) * b.span) >> 32);
    }
    static KeySpace make(size_t n_keys, int dist = KEYS_ROUNDROBIN, double alpha = 1.3, bool shared = false) {
        KeySpace k;
        k.n_keys = n_keys ? n_keys : 1;
        k.dist = dist;
        k.zipf_alpha = alpha;
        k.shared = shared;
        k.prepare();
        return k;
    }
};

// Bulk generator functor for Batch_Source:  bool(tuple_t*, size_t cap, size_t& n, RuntimeContext&)
// round-robin keys receive in-order per-key ids; with random keys id is the per-source sequence number (count-based
// windows then number the tuples of a key by arrival on the device); ts is a monotone per-source microsecond clock.
class GatedGenerator {
public:
    // src_offset / src_stride: position of this process's sources among all the sources of the job
    // (rank * n_sources, world * n_sources), so keys stay disjoint across ranks as well
    GatedGenerator(StepGate* gates, size_t n_keys, uint64_t ts_step_us, size_t n_sources, size_t src_offset = 0, size_t src_stride = 0)
        : GatedGenerator(gates, KeySpace::make(n_keys), ts_step_us, n_sources, src_offset, src_stride) {}
    GatedGenerator(StepGate* gates, KeySpace keys, uint64_t ts_step_us, size_t n_sources, size_t src_offset = 0, size_t src_stride = 0)
        : gates_(gates), keys_(std::move(keys)), n_keys_(keys_.n_keys), ts_step_(ts_step_us), n_sources_(src_stride ? src_stride : n_sources),
          src_offset_(src_offset) {}

    bool operator()(tuple_t* buf, size_t cap, size_t& n, wf::RuntimeContext& ctx) {
        StepGate& g = gates_[ctx.getReplicaIndex()];
        punct_ = 0;
        int64_t take = 0;
        {
            std::unique_lock<std::mutex> l(g.m);
            g.cv.wait(l, [&] { return g.quota > 0 || g.stop; });
            if (g.quota <= 0 && g.stop) {
                n = 0;
                return false;
            }
            take = g.quota < (int64_t)cap ? g.quota : (int64_t)cap;
            g.quota -= take;
            if (g.quota == 0) punct_ = g.step_id;  // last batch of the step
        }
        if (skew_) skew_->wait_turn(ctx.getReplicaIndex(), g);
        opened_ns_ = wf::current_time_nsecs();
        const uint64_t src = ctx.getReplicaIndex() + src_offset_;
        uint64_t seq = seq_, kc = kc_, round = round_;
        const uint64_t nk = n_keys_, tstep = ts_step_;
        // disjoint key sets: key % n_sources == src, per-key order is then well defined without cross-source ordering
        const uint64_t kmul = keys_.shared ? 1 : n_sources_, kadd = keys_.shared ? 0 : src;
        const bool nt = nt_stores_;
        if (rng_ == 0) rng_ = 0x9E3779B97F4A7C15ull * (src + 1);
        uint64_t rng = rng_;
        const int dist = keys_.dist;
        const KeySpace::Bucket* quant = keys_.quantiles ? keys_.quantiles->data() : nullptr;
        for (int64_t i = 0; i < take; i++, seq++) {
            uint64_t k, id;
            if (dist == KEYS_ROUNDROBIN) {
                k = kc;
                id = round;
                if (++kc == nk) {
                    kc = 0;
                    round++;
                }
            } else {
                rng ^= rng << 13;  // xorshift64
                rng ^= rng >> 7;
                rng ^= rng << 17;
                k = dist == KEYS_UNIFORM ? (uint64_t)(((unsigned __int128)rng * nk) >> 64) : KeySpace::zipf_rank(quant, rng);
                id = seq;
            }
            if (nt) {
                // non-temporal stores: the slab is only read back by the DMA engine; not pulling its lines into the
                // cache first saves the read-for-ownership traffic when many ranks share the host memory system
                WFBENCH_NT_STORE(k * kmul + kadd, &buf[i].key);
                WFBENCH_NT_STORE(id, &buf[i].id);
                WFBENCH_NT_STORE(seq * tstep, &buf[i].ts);
                WFBENCH_NT_STORE((double)(seq & 1023), &buf[i].value);
                WFBENCH_NT_STORE(1.0, &buf[i].aux);
            } else {
                buf[i].key = k * kmul + kadd;
                buf[i].id = id;
                buf[i].ts = seq * tstep;
                buf[i].value = (double)(seq & 1023);
                buf[i].aux = 1.0;
            }
        }
        seq_ = seq;
        kc_ = kc;
        round_ = round;
        rng_ = rng;
        n = (size_t)take;
        if (skew_) skew_->batch_done(ctx.getReplicaIndex());
        return true;
    }
    // sources that share keys advance together (SkewBound); call before the generator is copied into the source replicas
    void bound_skew(std::shared_ptr<SkewBound> s) { skew_ = std::move(s); }
    // Batch_Source hook: non-zero closes a step
    uint64_t punctuation() const { return punct_; }
    // Batch_Source hook: when the first tuple of the batch was generated
    uint64_t opened_ns() const { return opened_ns_; }

private:
    std::shared_ptr<SkewBound> skew_;
    StepGate* gates_;
    KeySpace keys_;
    size_t n_keys_;
    uint64_t ts_step_;
    size_t n_sources_;
    size_t src_offset_ = 0;
    uint64_t seq_ = 0, kc_ = 0, round_ = 0, rng_ = 0;
    uint64_t punct_ = 0;
    uint64_t opened_ns_ = 0;
    // on by default (single GPU: same throughput, lower p99 latency; several ranks: less host memory traffic); WF_BENCH_NT_STORES=0 disables
    bool nt_stores_ = std::getenv("WF_BENCH_NT_STORES") == nullptr || std::atoi(std::getenv("WF_BENCH_NT_STORES")) != 0;
};

// ---- the same stream born on the device (Device_Source) -------------------------------------------------------------
// Generator evaluated by one GPU thread per tuple (and by the host for the watermark of a batch): a pure function of the
// source and of the source-local sequence number, producing what GatedGenerator produces for round-robin keys and an
// equivalent random stream for uniform / zipf keys.  With it the pipeline is fed at HBM speed: what is measured is the
// framework and the GPU, not the PCIe link or the host's memory system (bench.py --source device).
struct DeviceGenerator {
    uint64_t n_keys = 1, kmul = 1, src_offset = 0, ts_step = 1;
    int dist = KEYS_ROUNDROBIN;
    uint32_t shared = 0;
    const KeySpace::Bucket* q_dev = nullptr;   // zipf quantile table in device memory
    const KeySpace::Bucket* q_host = nullptr;  // the same table on the host
    static WF_HOST_DEVICE uint64_t mix(uint64_t x) {
        x += 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    }
    WF_HOST_DEVICE void operator()(uint64_t seq, uint32_t replica, tuple_t& t) const {
        const uint64_t src = (uint64_t)replica + src_offset;
        uint64_t k, id;
        if (dist == KEYS_ROUNDROBIN) {
            k = seq % n_keys;
            id = seq / n_keys;
        } else {
            const uint64_t rnd = mix(seq * 0x100000001B3ull + src);
            if (dist == KEYS_UNIFORM) {
                k = (uint64_t)(((unsigned __int128)rnd * n_keys) >> 64);
            } else {
#if defined(__HIP_DEVICE_COMPILE__)
                const KeySpace::Bucket* tab = q_dev;
#else
                const KeySpace::Bucket* tab = q_host;
#endif
                k = KeySpace::zipf_rank(tab, rnd);
            }
            id = seq;
        }
        t.key = shared ? k : k * kmul + src;
        t.id = id;
        t.ts = seq * ts_step;
        t.value = (double)(seq & 1023);
        t.aux = 1.0;
    }
};

// host side of a gated device source (Device_Source::withControl): releases the driver's quota batch by batch
struct GatedControl {
    StepGate* gates = nullptr;
    std::shared_ptr<SkewBound> skew;
    bool operator()(uint64_t, uint32_t replica, size_t cap, size_t& n, uint64_t& punct) const {
        StepGate& g = gates[replica];
        punct = 0;
        int64_t take = 0;
        {
            std::unique_lock<std::mutex> l(g.m);
            g.cv.wait(l, [&] { return g.quota > 0 || g.stop; });
            if (g.quota <= 0 && g.stop) {
                n = 0;
                return false;
            }
            take = g.quota < (int64_t)cap ? g.quota : (int64_t)cap;
            g.quota -= take;
            if (g.quota == 0) punct = g.step_id;
        }
        if (skew) {
            skew->wait_turn(replica, g);
            skew->batch_done(replica);
        }
        n = (size_t)take;
        return true;
    }
};

}  // namespace wfbench
