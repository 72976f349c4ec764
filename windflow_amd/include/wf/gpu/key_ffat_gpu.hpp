// Key_FFAT_GPU: keyed sliding-window aggregation (lift + combine) with every
// piece of keyed state resident in HBM.
//
// Capability parity with /root/reference/wf/key_ffat_gpu.hpp:68-247 +
// wf/win_seqffat_gpu.hpp:56-698 + wf/flatfat_gpu.hpp (count- and time-based
// sliding windows - and, beyond the reference, tumbling and hopping ones -, quantum = gcd(win, slide) panes for time-based windows,
// triggering delay, late tuples dropped, incomplete windows emitted at
// end-of-stream).  Design is ours:
//  * the reference does the windowing bookkeeping per tuple on the host, keeps one
//    cudaMalloc'ed tree per key and launches kernels per key; here a whole
//    mixed-key batch is processed on the device:
//      1. assign_slots_kernel   key -> slot through a device hash table
//      2. RadixSorter           stable sort of (slot, tuple index): the tuples of a
//                               key become one contiguous run in arrival order
//      3. ffat_update_kernel    one thread per run: lift, fold into panes, insert
//                               completed panes into the key's FlatFAT (a slab of
//                               trees in HBM), fire windows with an O(log n) ordered
//                               range query, append results to the output batch
//    so the number of launches per batch is constant whatever the number of keys,
//    and state for millions of keys stays on the device (288 GB HBM3E).
//  * lift and combine must both be device-callable (the reference lifts on the
//    host because its tuples arrive one by one on the host).
//  * end-of-stream runs ffat_flush_kernel on the device (the reference falls back
//    to the CPU: win_seqffat_gpu.hpp:543-660).
#pragma once
#include <algorithm>
#include <map>
#include <numeric>
#include <tuple>
#include <vector>

#include "batch_ops.hpp"
#include "flatfat_gpu.hpp"
#include "../dist/exchange_nodes.hpp"
#include "keyby_gpu.hpp"
#include "primitives.hpp"

namespace wf {
namespace gpu {

struct FFATGeometry {
    uint32_t is_tb;       // 1: time-based
    uint64_t pane_len;    // CB: tuples per pane; TB: quantum (time units)
    uint32_t win_panes;   // panes per window
    uint32_t slide_panes; // panes per slide
    uint64_t delay;       // TB triggering delay
    uint32_t tree_cap;    // leaves per tree (power of two > win_panes)
    uint32_t acc_cap;     // TB: open quanta ring (power of two)
    uint32_t out_cap;     // capacity of the output batch
    uint32_t skip_gaps;   // TB: fast-forward over long stretches of empty windows instead of emitting them
};

struct FFATKeyHeader {
    uint64_t panes;         // completed panes inserted so far
    uint64_t next_lwid;     // next window to fire
    uint64_t last_quantum;  // TB: oldest open quantum
    uint64_t max_quantum;   // TB: greatest quantum that received a tuple (+1), 0 = none
    uint32_t pcount;        // CB: tuples in the open pane
    uint32_t seen;          // key has received at least one tuple
};
static_assert(sizeof(FFATKeyHeader) == 40, "FFATKeyHeader is 40 bytes");
// the state of a key as the kernels hold it in registers
template <typename R>
struct FFATKeyState : FFATKeyHeader {
    R partial;              // CB: aggregate of the open pane
};

// Where the state of a key lives in HBM: ONE record per key,
//     [ header 40 B | batch stamp 4 B (primitives.hpp DupProbe) | flags of the ring (acc_cap bytes) | pad to 8 | ring of acc_cap results ]
// aligned to and padded to a multiple of 128 B (the line of the L2): the ring holds the open quanta of a time-based
// window, or - slot 0 - the aggregate of the open pane of a count-based one.  With the usual two open quanta and 40-byte
// results a record is exactly 128 B: a tuple that completes no pane reads one line and writes it back.  (Rounds 1 and 2
// kept the state, the ring and its flags in three slabs: 4-5 lines per tuple, 312 B read + 289 B written per tuple measured.)
// The tree of a key stays in a slab of its own: it is only touched when a pane completes.
template <typename R>
struct FFATRecords {
    char* base = nullptr;
    uint32_t stride = 0;    // bytes per key
    uint32_t ring_off = 0;  // offset of the ring in a record
    static constexpr uint32_t STAMP_OFF = sizeof(FFATKeyHeader);  // the probe for repeated keys stamps the record itself: the line
                                                                 // is on its way into the caches when the update kernel asks for it
    static constexpr uint32_t FLAGS_OFF = STAMP_OFF + 4;
    static size_t ring_offset(uint32_t acc_cap) { return (FLAGS_OFF + (size_t)acc_cap + 7) & ~size_t(7); }
    static size_t stride_for(uint32_t acc_cap) { return (ring_offset(acc_cap) + (size_t)acc_cap * sizeof(R) + 127) & ~size_t(127); }
    __device__ __forceinline__ char* rec(uint32_t slot) const { return base + (size_t)slot * stride; }
    __device__ __forceinline__ const FFATKeyHeader& header(uint32_t slot) const { return *reinterpret_cast<const FFATKeyHeader*>(rec(slot)); }
    __device__ __forceinline__ R* ring(uint32_t slot) const { return reinterpret_cast<R*>(rec(slot) + ring_off); }
    __device__ __forceinline__ uint8_t* used(uint32_t slot) const { return reinterpret_cast<uint8_t*>(rec(slot) + FLAGS_OFF); }
    // count-based windows keep the open pane's aggregate in ring slot 0; time-based ones have no use for `partial`
    __device__ __forceinline__ FFATKeyState<R> load(uint32_t slot, uint32_t is_tb) const {
        FFATKeyState<R> st;
        static_cast<FFATKeyHeader&>(st) = header(slot);
        if (!is_tb) st.partial = ring(slot)[0];
        return st;
    }
    __device__ __forceinline__ void store(uint32_t slot, const FFATKeyState<R>& st, uint32_t is_tb) const {
        *reinterpret_cast<FFATKeyHeader*>(rec(slot)) = static_cast<const FFATKeyHeader&>(st);
        if (!is_tb) ring(slot)[0] = st.partial;
    }
};

struct FFATCounters {
    uint32_t dropped;          // late tuples
    uint32_t overflow;         // results that did not fit the output batch
    uint32_t skipped_windows;  // TB: empty windows skipped by a time jump beyond the tree capacity
    uint32_t table_full;       // tuples whose key could not be inserted
    uint32_t huge_runs;        // runs spread over the chip by the huge-run kernels (ffat_huge_gpu.hpp)
};

// where fired windows go.  GlobalEmit: one returning atomic per result (fine when a thread fires a
// handful of windows per batch).  WaveEmit: lane 0 of a wave parks results in LDS and the whole wave
// flushes them with ONE atomic per chunk (a returning global atomic costs microseconds: it must not
// sit on the sequential path of a long run).
template <typename R>
struct GlobalEmit {
    R* out;
    uint32_t* out_count;
    uint32_t out_cap;
    FFATCounters* ctr;
    __device__ __forceinline__ void operator()(const R& r) {
        const uint32_t pos = atomicAdd(out_count, 1u);
        if (pos < out_cap)
            out[pos] = r;
        else
            atomicAdd(&ctr->overflow, 1u);
    }
};

template <typename R>
struct WaveEmit {
    R* buf;  // LDS, WAVE entries
    uint32_t n;
    R* out;
    uint32_t* out_count;
    uint32_t out_cap;
    FFATCounters* ctr;
    // called by lane 0 only
    __device__ __forceinline__ void operator()(const R& r) {
        if (n == (uint32_t)WAVE) {  // rare: more than 64 windows fired inside one chunk (long silent gaps)
            const uint32_t base = atomicAdd(out_count, n);
            for (uint32_t k = 0; k < n; k++) {
                if (base + k < out_cap)
                    out[base + k] = buf[k];
                else
                    atomicAdd(&ctr->overflow, 1u);
            }
            n = 0;
        }
        buf[n++] = r;
    }
    // called by the whole wave after lane 0 is done with a chunk
    __device__ __forceinline__ void flush_wave() {
        const uint32_t cnt = (uint32_t)__shfl((int)n, 0, WAVE);
        if (cnt == 0) return;
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(out_count, cnt);
        base = (uint32_t)__shfl((int)base, 0, WAVE);
        const uint32_t l = (uint32_t)lane_id();
        if (l < cnt) {
            if (base + l < out_cap)
                out[base + l] = buf[l];
            else
                atomicAdd(&ctr->overflow, 1u);
        }
        n = 0;
    }
};

// a pane is complete: insert it in the key's tree and fire the window it closes (if any)
template <typename R, typename C, typename K, typename E>
__device__ __forceinline__ void ffat_complete_pane(FFATKeyState<R>& st, R* tree, const FFATGeometry& g, const K& key, const R& pane, const C& comb,
                                                   E& emit) {
    fat_insert(tree, g.tree_cap, (uint32_t)(st.panes & (g.tree_cap - 1)), pane, comb);
    st.panes++;
    if (st.panes >= g.win_panes && (st.panes - g.win_panes) % g.slide_panes == 0) {
        R res;
        fat_query_circular(tree, g.tree_cap, (uint32_t)((st.panes - g.win_panes) & (g.tree_cap - 1)), g.win_panes, res, comb);
        res.setControlFields(key, st.next_lwid, cf_ts(res));
        st.next_lwid++;
        emit(res);
    }
}

constexpr int FFAT_PREFETCH = 4;
// runs of at least this many tuples are left to the wave-per-run kernel
constexpr uint32_t FFAT_LONG_RUN = 12;

// `lifted` is the aggregate of `len` consecutive tuples of a key that all belong to the same pane
// (CB) / quantum (TB): fold it into the open pane / quantum and complete what it closes.  With
// len = 1 this is the per-tuple step of the reference engine (win_seqffat.hpp CB :330-419, TB :422-513).
template <typename R, typename C, typename E>
__device__ __forceinline__ void ffat_fold(FFATKeyState<R>& st, R* tree, R* acc, uint8_t* used, const FFATGeometry& g, const R& lifted, uint32_t len,
                                          const C& comb, E& emit, FFATCounters* ctr) {
    const auto key = cf_key(lifted);
    const uint64_t ts = cf_ts(lifted);  // greatest timestamp folded in `lifted`
    st.seen = 1;
    if (!g.is_tb) {
        if (st.pcount == 0) {
            st.partial = lifted;
        } else {
            R tmp;
            fat_combine(comb, st.partial, lifted, tmp);
            st.partial = tmp;
        }
        st.pcount += len;
        if (st.pcount == g.pane_len) {
            ffat_complete_pane(st, tree, g, key, st.partial, comb, emit);
            st.pcount = 0;
        }
        return;
    }
    const uint64_t q = ts / g.pane_len;
    if (q < st.last_quantum) {
        atomicAdd(&ctr->dropped, len);
        return;
    }
    // quanta whose last instant + delay is older than ts are complete
    uint64_t done_upto = (ts > g.delay) ? (ts - g.delay) / g.pane_len : 0;  // quanta < done_upto are complete
    // the ring of open quanta must be able to hold q
    if (q >= done_upto + g.acc_cap) done_upto = q - g.acc_cap + 1;
    if (done_upto > st.last_quantum) {
        auto complete_until = [&](uint64_t upto) {
            while (st.last_quantum < upto) {
                const uint32_t a = (uint32_t)(st.last_quantum & (g.acc_cap - 1));
                R pane{};
                if (used[a]) {
                    pane = acc[a];
                    used[a] = 0;
                }
                pane.setControlFields(key, st.last_quantum, (st.last_quantum + 1) * g.pane_len - 1);
                ffat_complete_pane(st, tree, g, key, pane, comb, emit);
                st.last_quantum++;
            }
        };
        // A time jump much longer than the tree: flush the quanta that can hold data
        // (and the windows containing them), then fast-forward over the stretch in
        // which every window is empty instead of materialising each of them.
        const uint64_t keep = (uint64_t)g.tree_cap + g.acc_cap;
        if (g.skip_gaps && done_upto - st.last_quantum > 4 * keep) {
            complete_until(st.last_quantum + keep);
            const uint64_t skip = done_upto - keep - st.last_quantum;
            const uint64_t panes_after = st.panes + skip;
            const uint64_t fired_after = panes_after >= g.win_panes ? (panes_after - g.win_panes) / g.slide_panes + 1 : 0;
            if (fired_after > st.next_lwid) {
                atomicAdd(&ctr->skipped_windows, (uint32_t)(fired_after - st.next_lwid));
                st.next_lwid = fired_after;
            }
            st.panes = panes_after;
            st.last_quantum += skip;
        }
        complete_until(done_upto);
    }
    const uint32_t a = (uint32_t)(q & (g.acc_cap - 1));
    if (used[a]) {
        R tmp;
        fat_combine(comb, acc[a], lifted, tmp);
        acc[a] = tmp;
    } else {
        acc[a] = lifted;
        used[a] = 1;
    }
    if (q + 1 > st.max_quantum) st.max_quantum = q + 1;
}

template <typename T, typename R, typename L>
__device__ __forceinline__ R ffat_lift(const T& t, const L& lift) {
    R lifted{};
    lifted.setControlFields(cf_key(t), 0, cf_ts(t));
    lift(t, lifted);
    return lifted;
}

template <typename T, typename R, typename L, typename C, typename E>
__device__ __forceinline__ void ffat_process_tuple(FFATKeyState<R>& st, R* tree, R* acc, uint8_t* used, const FFATGeometry& g, const T& t,
                                                   const L& lift, const C& comb, E& emit, FFATCounters* ctr) {
    ffat_fold(st, tree, acc, used, g, ffat_lift<T, R, L>(t, lift), 1u, comb, emit, ctr);
}

// the value of `v` held by lane (lane - d); any trivially copyable struct made of 32-bit words
template <typename X>
__device__ __forceinline__ X wave_shfl_up(const X& v, int d) {
    static_assert(sizeof(X) % 4 == 0, "wave shuffles move 32-bit words");
    int w[sizeof(X) / 4];
    __builtin_memcpy(w, &v, sizeof(X));
#pragma unroll
    for (size_t k = 0; k < sizeof(X) / 4; k++) w[k] = __shfl_up(w[k], d, WAVE);
    X o;
    __builtin_memcpy(&o, w, sizeof(X));
    return o;
}

// Short runs: one thread per run of equal slots in the sorted (slot, index) arrays.  Heads of
// runs of FFAT_LONG_RUN tuples or more are only recorded in `long_heads` for the wave-per-run
// kernel below (streams with few keys would otherwise serialise thousands of dependent random
// loads in one thread).
template <typename T, typename R, typename L, typename C>
__global__ void __launch_bounds__(BLOCK_THREADS)
    ffat_update_kernel(const BatchHeader* hdr, const T* data, SortedSel sel, FFATRecords<R> recs, R* trees,
                       FFATGeometry g, R* out, uint32_t* out_count, FFATCounters* ctr, uint32_t* long_heads,
                       uint32_t* n_long, L lift, C comb, uint32_t huge_mode = 0, uint32_t huge_len = 0, uint32_t* huge_heads = nullptr,
                       uint32_t* n_huge = nullptr, volatile uint32_t* huge_seen = nullptr, uint32_t batch_no = 0, uint32_t long_run = FFAT_LONG_RUN) {
    const uint32_t n = hdr->count;
    // a batch without a second tuple of any key was not sorted: every tuple is a run of its own, in any order
    const uint32_t* sslot = sel.keys();
    const uint32_t* sidx = sel.vals();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = sslot[i];
        if (i > 0 && sslot[i - 1] == slot) continue;  // not the head of a run
        if (slot == 0xffffffffu) {
            atomicAdd(&ctr->table_full, 1u);
            continue;
        }
        if (i + long_run - 1 < n && sslot[i + long_run - 1] == slot) {
            // huge_mode 1: a run long enough to be spread over the chip is only reported (the host then switches the
            // huge-run kernels on for the batches that follow); 2: it is left to them (ffat_huge_gpu.hpp)
            const bool huge = huge_mode != 0 && i + huge_len - 1 < n && sslot[i + huge_len - 1] == slot;
            if (huge) *huge_seen = batch_no;
            if (huge && huge_mode == 2)
                huge_heads[atomicAdd(n_huge, 1u)] = i;
            else
                long_heads[atomicAdd(n_long, 1u)] = i;
            continue;
        }
        R* tree = trees + (size_t)slot * 2 * g.tree_cap;
        R* acc = g.is_tb ? recs.ring(slot) : nullptr;
        uint8_t* used = g.is_tb ? recs.used(slot) : nullptr;
        FFATKeyState<R> st = recs.load(slot, g.is_tb);
        GlobalEmit<R> emit{out, out_count, g.out_cap, ctr};
        // software prefetch: the tuples of a run are gathered FFAT_PREFETCH at a time so that
        // their (dependent, random) loads are in flight together
        bool run_open = true;
        for (uint32_t j0 = i; run_open; j0 += FFAT_PREFETCH) {
            T pre[FFAT_PREFETCH];
            bool ok[FFAT_PREFETCH];
#pragma unroll
            for (int u = 0; u < FFAT_PREFETCH; u++) {
                const uint32_t j = j0 + u;
                ok[u] = (j < n) && (sslot[j] == slot);
                if (ok[u]) pre[u] = data[sidx[j]];
            }
#pragma unroll
            for (int u = 0; u < FFAT_PREFETCH; u++) {
                if (!ok[u]) {
                    run_open = false;
                    break;
                }
                ffat_process_tuple(st, tree, acc, used, g, pre[u], lift, comb, emit, ctr);
            }
        }
        recs.store(slot, st, g.is_tb);
    }
}

// aggregate of one segment (tuples of a chunk that fall in the same pane / quantum), staged in LDS for
// the sequential per-segment fold of the long-run kernel
template <typename R>
struct FFATSegment {
    R agg;
    uint32_t len;
    uint32_t pad;
};

// leaves of the per-wave working tree of the bulk path: a chunk completes at most WAVE panes and a
// window needs the win_panes - 1 panes before them, so bulk mode needs win_panes + WAVE <= FFAT_WORK_LEAVES
constexpr uint32_t FFAT_WORK_LEAVES = 128;

// LDS layout of one wave of the long-run kernel
template <typename R>
struct FFATLongLayout {
    uint32_t tree_nodes;  // 0: the tree stays in global memory
    uint32_t acc_slots;   // 0: the ring of open quanta stays in global memory
    uint32_t bulk;        // 1: tree_nodes = 2 * FFAT_WORK_LEAVES, a working tree private to the run (bulk path);
                          // 2: bulk path on the key's own tree in global memory (windows of more than 64 panes)
    __host__ __device__ static size_t align16(size_t x) { return (x + 15) & ~size_t(15); }
    __host__ __device__ size_t segs_off() const { return 0; }
    __host__ __device__ size_t res_off() const { return align16(WAVE * sizeof(FFATSegment<R>)); }
    __host__ __device__ size_t tree_off() const { return res_off() + align16(WAVE * sizeof(R)); }
    __host__ __device__ size_t acc_off() const { return tree_off() + align16((size_t)tree_nodes * sizeof(R)); }
    __host__ __device__ size_t used_off() const { return acc_off() + align16((size_t)acc_slots * sizeof(R)); }
    __host__ __device__ size_t wave_bytes() const { return used_off() + align16(acc_slots); }
    // stage as much of the per-key state as fits `budget` bytes per wave
    static FFATLongLayout choose(const FFATGeometry& g, size_t budget) {
        const uint32_t acc = g.is_tb ? g.acc_cap : 0;
        FFATLongLayout l{0, 0, 0};
        FFATLongLayout with_acc{0, acc, 0};
        if (with_acc.wave_bytes() <= budget) l = with_acc;
        FFATLongLayout with_tree{2 * g.tree_cap, l.acc_slots, 0};
        if (with_tree.wave_bytes() <= budget) l = with_tree;
        // bulk paths: with the ring of open quanta staged when it fits (a long triggering delay makes it large), else in place
        for (const uint32_t a : {acc, 0u}) {
            if (l.bulk != 0) break;
            FFATLongLayout with_bulk{2 * FFAT_WORK_LEAVES, a, 1};
            if (g.win_panes + WAVE <= FFAT_WORK_LEAVES && with_bulk.wave_bytes() <= budget) l = with_bulk;
            FFATLongLayout global_bulk{0, a, 2};
            if (l.bulk == 0 && g.win_panes + WAVE <= g.tree_cap && global_bulk.wave_bytes() <= budget) l = global_bulk;
        }
        return l;
    }
};

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the value of `v` held by lane `src`
template <typename X>
__device__ __forceinline__ X wave_bcast(const X& v, int src) {
    static_assert(sizeof(X) % 4 == 0, "wave shuffles move 32-bit words");
    int w[sizeof(X) / 4];
    __builtin_memcpy(w, &v, sizeof(X));
#pragma unroll
    for (size_t k = 0; k < sizeof(X) / 4; k++) w[k] = __shfl(w[k], src, WAVE);
    X o;
    __builtin_memcpy(&o, w, sizeof(X));
    return o;
}

// recompute every internal node of a tree from its leaves, one level per step, a lane per node
template <typename R, typename C>
__device__ __forceinline__ void fat_rebuild_wave(R* tree, uint32_t cap, const C& comb) {
    for (uint32_t width = cap >> 1; width >= 1; width >>= 1) {
        for (uint32_t j = (uint32_t)lane_id(); j < width; j += WAVE) {
            const uint32_t node = width + j;
            R tmp;
            fat_combine(comb, tree[2 * node], tree[2 * node + 1], tmp);
            tree[node] = tmp;
        }
        wave_sync_lds();
    }
}

// recompute the ancestors of `count` consecutive leaves that start at circular position `first`: one level per step, a lane
// per dirty node (count / 2 + 1 nodes on the first level, half of that on the next, ... then the path to the root)
template <typename R, typename C>
__device__ __forceinline__ void fat_rebuild_range_wave(R* tree, uint32_t cap, uint32_t first, uint32_t count, const C& comb) {
    if (count == 0) return;
    if (count >= cap) {
        fat_rebuild_wave(tree, cap, comb);
        return;
    }
    first &= cap - 1;
    // the circular range is one or two linear ranges of leaves; [lo, hi] inclusive, n2 == 0: no second range
    uint32_t lo1 = first, hi1 = (first + count <= cap ? first + count : cap) - 1;
    uint32_t lo2 = 0, hi2 = 0, n2 = first + count > cap ? 1u : 0u;
    if (n2) hi2 = first + count - cap - 1;
    for (uint32_t width = cap >> 1; width >= 1; width >>= 1) {
        lo1 >>= 1;
        hi1 >>= 1;
        lo2 >>= 1;
        hi2 >>= 1;
        const uint32_t c1 = hi1 - lo1 + 1, c2 = n2 ? hi2 - lo2 + 1 : 0;  // (nodes the two ranges share are computed twice: same value)
        for (uint32_t j = (uint32_t)lane_id(); j < c1 + c2; j += WAVE) {
            const uint32_t node = width + (j < c1 ? lo1 + j : lo2 + (j - c1));
            R tmp;
            fat_combine(comb, tree[2 * node], tree[2 * node + 1], tmp);
            tree[node] = tmp;
        }
        wave_sync_lds();
    }
}

// Long runs: one wave64 per run, exploiting the associativity of the combine.
//  * all chunks: the 64 lanes gather and LIFT the next 64 tuples of the run together; lanes whose tuples
//    fall in the same pane (CB) / quantum (TB) form a segment and an order-preserving segmented wave scan
//    (6 shuffle + combine steps) leaves each segment's aggregate in its last lane;
//  * bulk path (window of at most 64 panes, in-order chunk, no triggering delay): the panes completed by
//    the chunk are written side by side as leaves of a 128-leaf working tree private to the run (LDS),
//    the tree is rebuilt a level per step, and every window fired by the chunk is evaluated by a lane of
//    its own - nothing is sequential per tuple, per pane or per window;
//  * bulk path on the key's own tree (GBULK; windows of more than 64 panes): the same, but the completed panes go straight
//    to the leaves of the key's tree in global memory, only the ancestors of the new leaves are recomputed (a lane per
//    dirty node and level) and every fired window is a lane's O(log) query of that tree;
//  * otherwise lane 0 folds the segments one by one (same rules as ffat_fold), on the staged tree.
// The windows fired by a chunk leave with one atomic.
template <bool GBULK, typename T, typename R, typename L, typename C>
__global__ void __launch_bounds__(BLOCK_THREADS)
    ffat_update_long_kernel(const BatchHeader* hdr, const T* data, const uint32_t* sslot, const uint32_t* sidx, FFATRecords<R> recs, R* trees,
                            FFATGeometry g, R* out, uint32_t* out_count, FFATCounters* ctr,
                            const uint32_t* long_heads, const uint32_t* n_long, FFATLongLayout<R> lay, L lift, C comb,
                            volatile uint32_t* disorder_seen = nullptr, uint32_t batch_no = 0) {
    extern __shared__ __align__(16) char ffat_smem[];
    const uint32_t n = hdr->count;
    const uint32_t runs = *n_long;
    const int lane = lane_id();
    char* my = ffat_smem + (size_t)wave_id() * lay.wave_bytes();
    FFATSegment<R>* segs = reinterpret_cast<FFATSegment<R>*>(my + lay.segs_off());
    R* ltree = reinterpret_cast<R*>(my + lay.tree_off());
    R* lacc = reinterpret_cast<R*>(my + lay.acc_off());
    uint8_t* lused = reinterpret_cast<uint8_t*>(my + lay.used_off());
    WaveEmit<R> emit{reinterpret_cast<R*>(my + lay.res_off()), 0u, out, out_count, g.out_cap, ctr};
    const bool bulk = lay.bulk != 0;
    // leaves of the tree the bulk path works on
    const uint32_t CW = GBULK ? g.tree_cap : FFAT_WORK_LEAVES;
    // geometry seen by the sequential rules when they work on the private tree
    FFATGeometry gw = g;
    if (bulk) gw.tree_cap = CW;
    const uint32_t waves_total = gridDim.x * (BLOCK_THREADS / WAVE);
    for (uint32_t r = blockIdx.x * (BLOCK_THREADS / WAVE) + wave_id(); r < runs; r += waves_total) {
        const uint32_t head = long_heads[r];
        const uint32_t slot = sslot[head];
        FFATKeyState<R> st = recs.load(slot, g.is_tb);  // bulk: identical in every lane; otherwise lane 0's copy is the live one
        R* gtree = trees + (size_t)slot * 2 * g.tree_cap;
        // the tree of the bulk path (one instantiation per address space: ds_* or global_* instructions, not flat ones)
        R* wtree;
        if constexpr (GBULK)
            wtree = gtree;
        else
            wtree = ltree;
        if (bulk && !GBULK) {
            // import the panes a future window can still need
            const uint64_t lo = st.panes > g.win_panes ? st.panes - g.win_panes : 0;
            for (uint64_t idx = lo + lane; idx < st.panes; idx += WAVE)
                ltree[CW + (uint32_t)(idx & (CW - 1))] = gtree[g.tree_cap + (uint32_t)(idx & (g.tree_cap - 1))];
            wave_sync_lds();
            fat_rebuild_wave(ltree, CW, comb);
        } else if (lay.tree_nodes != 0) {
            for (uint32_t k = lane; k < 2 * g.tree_cap; k += WAVE) ltree[k] = gtree[k];
        }
        R* gacc = g.is_tb ? recs.ring(slot) : nullptr;
        uint8_t* gused = g.is_tb ? recs.used(slot) : nullptr;
        const bool acc_staged = g.is_tb && lay.acc_slots != 0;
        // the ring as the bulk path sees it (generic pointers: it touches a few slots per chunk)
        R* racc = acc_staged ? lacc : gacc;
        uint8_t* rused = acc_staged ? lused : gused;
        if (acc_staged) {
            for (uint32_t k = lane; k < g.acc_cap; k += WAVE) {
                lacc[k] = gacc[k];
                lused[k] = gused[k];
            }
        }
        wave_sync_lds();
        // bulk path: panes written to the working tree whose windows have not been evaluated yet.  Rebuilding the
        // tree and evaluating windows costs the same for 1 or 64 new panes, so chunks that complete few panes
        // (long panes) are accumulated until 64 panes are pending.
        uint32_t pend = 0;
        decltype(cf_key(st.partial)) run_key{};
        auto fire_pending = [&]() {
            if (pend == 0) return;
            if constexpr (GBULK)
                fat_rebuild_range_wave(wtree, CW, (uint32_t)((st.panes - pend) & (CW - 1)), pend, comb);
            else
                fat_rebuild_wave(wtree, CW, comb);
            // windows k in [next_lwid, k_end), window k = panes [k*slide, k*slide + win)
            const uint64_t Pn = st.panes;
            const uint64_t k_end = Pn >= g.win_panes ? (Pn - g.win_panes) / g.slide_panes + 1 : 0;
            const uint32_t nf = k_end > st.next_lwid ? (uint32_t)(k_end - st.next_lwid) : 0;  // <= pend <= 64
            // results are parked in LDS and leave with one atomic when the 64 slots are full (or at the end of the
            // run): a returning global atomic per chunk would dominate chunks that fire few windows
            if (emit.n + nf > (uint32_t)WAVE) {
                emit.flush_wave();
                wave_sync_lds();
            }
            if ((uint32_t)lane < nf) {
                const uint64_t k = st.next_lwid + (uint64_t)lane;
                R res;
                fat_query_circular(wtree, CW, (uint32_t)((k * g.slide_panes) & (CW - 1)), g.win_panes, res, comb);
                res.setControlFields(run_key, k, cf_ts(res));
                emit.buf[emit.n + (uint32_t)lane] = res;
            }
            emit.n += nf;
            if (k_end > st.next_lwid) st.next_lwid = k_end;
            pend = 0;
            wave_sync_lds();
        };
        // software pipeline: chunk c+1 is loaded into registers before chunk c is reduced
        uint32_t j = head + lane;
        bool ok = (j < n) && (sslot[j] == slot);
        T reg{};
        if (ok) reg = data[sidx[j]];
        while (true) {
            const unsigned long long okmask = __ballot(ok);  // the valid lanes are a prefix: the arrays are sorted
            const uint32_t cnt = (uint32_t)__popcll(okmask);
            // 1. lift + segment key
            R val{};
            uint64_t segkey = 0;
            const uint32_t pc0 = (uint32_t)__shfl((int)st.pcount, 0, WAVE);  // tuples already in the open pane (CB)
            if (ok) {
                val = ffat_lift<T, R, L>(reg, lift);
                segkey = g.is_tb ? cf_ts(reg) / g.pane_len : ((uint64_t)pc0 + (uint64_t)lane) / g.pane_len;
            }
            // next chunk's loads go out now and complete while this chunk is reduced
            T nreg{};
            bool nok = false;
            if (cnt == WAVE) {
                j += WAVE;
                nok = (j < n) && (sslot[j] == slot);
                if (nok) nreg = data[sidx[j]];
            }
            // 2. segment index = number of boundaries at or before this lane
            const uint32_t lo32 = (uint32_t)(segkey & 0xffffffffu), hi32 = (uint32_t)(segkey >> 32);
            const uint32_t plo = (uint32_t)__shfl_up((int)lo32, 1, WAVE), phi = (uint32_t)__shfl_up((int)hi32, 1, WAVE);
            const uint64_t prevkey = ((uint64_t)phi << 32) | plo;
            const bool boundary = ok && (lane == 0 || prevkey != segkey);
            const unsigned long long bmask = __ballot(boundary);
            const unsigned long long upto_me = bmask & (lanemask_lt() | (1ull << lane));
            const int seg = __popcll(upto_me) - 1;
            // 3. order-preserving segmented inclusive scan; segments are contiguous, so once no lane finds a member
            //    of its segment d lanes below, no larger distance can either (short panes stop after 2-3 steps)
            for (int d = 1; d < WAVE; d <<= 1) {
                const int oseg = __shfl_up(seg, d, WAVE);
                const bool take = ok && lane >= d && oseg == seg;
                if (__ballot(take) == 0) break;
                const R other = wave_shfl_up(val, d);
                if (take) {
                    R tmp;
                    fat_combine(comb, other, val, tmp);
                    val = tmp;
                }
            }
            const bool tail = ok && ((lane + 1 == (int)cnt) || ((bmask >> (lane + 1)) & 1ull));
            const uint32_t n_segs = (uint32_t)__popcll(bmask);
            // 4. can the whole chunk be handled in bulk?
            bool fast = bulk && cnt > 0;
            // time-based: what the sequential rules would do with the chunk, computed for all its tuples at once.
            //   done_i     quanta below it are complete once tuple i has been seen (its timestamp minus the triggering delay; and
            //              the ring of open quanta must be able to hold the tuple's own quantum)
            //   L_i        oldest open quantum when tuple i arrives = max(last_quantum, done_j of the tuples before it)
            //   late_i     the tuple's quantum is already complete: dropped
            //   Lq_new     oldest open quantum after the chunk; the m quanta below it are completed by the chunk
            uint64_t Lq_new = 0, q_max = 0;
            bool late = false, simple = true;
            uint32_t n_late = 0;
            if (fast && g.is_tb) {
                const uint64_t Lq = st.last_quantum;
                const bool disorder = __ballot(ok && lane > 0 && segkey < prevkey) != 0;
                // (the host then orders the following batches by (key, timestamp) before they get here: primitives.hpp TsPresorter)
                if (disorder && lane == 0 && disorder_seen != nullptr) *disorder_seen = batch_no;
                const uint64_t q_first = wave_bcast(segkey, 0), q_last = wave_bcast(segkey, (int)cnt - 1);
                auto done_of = [&](uint64_t ts, uint64_t q) {
                    uint64_t d = ts > g.delay ? (ts - g.delay) / g.pane_len : 0;
                    if (q >= d + g.acc_cap) d = q - g.acc_cap + 1;
                    return d;
                };
                if (!disorder && q_first >= Lq) {
                    // the common case, a chunk in timestamp order that starts at an open quantum: done_i never exceeds the
                    // quantum of the tuple after it, so nothing is late and the last tuple alone decides what is complete
                    const uint64_t ts_last = wave_bcast((uint64_t)cf_ts(reg), (int)cnt - 1);
                    Lq_new = done_of(ts_last, q_last);
                    if (Lq_new < Lq) Lq_new = Lq;
                    q_max = q_last;
                } else {
                    const uint64_t done = ok ? done_of(cf_ts(reg), segkey) : 0;
                    uint64_t pm = done;  // inclusive prefix maximum
                    for (int d = 1; d < WAVE; d <<= 1) {
                        const uint64_t o = wave_shfl_up(pm, d);
                        if (lane >= d && o > pm) pm = o;
                    }
                    uint64_t before = wave_shfl_up(pm, 1);
                    if (lane == 0 || before < Lq) before = Lq;
                    late = ok && segkey < before;
                    const unsigned long long late_mask = __ballot(late);
                    n_late = (uint32_t)__popcll(late_mask);
                    Lq_new = wave_bcast(pm, WAVE - 1);
                    if (Lq_new < Lq) Lq_new = Lq;
                    uint64_t qm = ok && !late ? segkey : 0;  // newest quantum that received a tuple
                    for (int d = 1; d < WAVE; d <<= 1) {
                        const uint64_t o = wave_shfl_up(qm, d);
                        if (lane >= d && o > qm) qm = o;
                    }
                    q_max = wave_bcast(qm, WAVE - 1);
                    simple = !disorder && late_mask == 0;  // the segments of the scan above are the chunk's quanta
                }
                // (the quanta the chunk completes are taken 64 at a time, see below; a jump in time beyond what the ring of open
                // quanta can hold is an empty stretch: the sequential rules know how to skip it)
                fast = Lq_new - Lq <= (uint64_t)(g.acc_cap > (uint32_t)WAVE ? g.acc_cap : (uint32_t)WAVE);
            }
            if (fast) {
                const auto key = wave_bcast(cf_key(val), 0);
                const uint64_t P = st.panes;
                uint32_t m;        // panes completed by this chunk
                R open_val = val;  // value of my segment's pane, with what was already open folded in
                if (!g.is_tb) {
                    const uint32_t total = pc0 + cnt;
                    m = (uint32_t)(total / g.pane_len);
                    if (pend + m > (uint32_t)WAVE) fire_pending();
                    if (tail) {
                        if (seg == 0 && pc0 > 0) fat_combine(comb, st.partial, val, open_val);
                        if ((uint32_t)seg < m) wtree[CW + (uint32_t)((P + (uint64_t)seg) & (CW - 1))] = open_val;
                    }
                    const uint32_t rem = (uint32_t)(total % g.pane_len);
                    const R last_val = wave_bcast(open_val, (int)cnt - 1);  // the last segment: the new open pane when rem > 0
                    st.pcount = rem;
                    if (rem > 0) st.partial = last_val;
                    st.panes = P + m;
                    pend += m;
                } else {
                    const uint64_t Lq = st.last_quantum;
                    const uint32_t amask = g.acc_cap - 1;
                    run_key = key;  // (fire_pending stamps the windows with it)
                    // one lane per quantum the chunk touches holds the fold of the chunk's tuples of that quantum, in arrival order
                    bool qtail = tail;
                    R qval = val;
                    if (!simple) {
                        // tuples out of timestamp order (or late ones in between): the tuples of a quantum are not neighbours.
                        // One ordered fold over the member lanes per distinct quantum (chunks usually touch two or three).
                        const R mine = ok ? ffat_lift<T, R, L>(reg, lift) : R{};
                        qtail = false;
                        unsigned long long rem = __ballot(ok && !late);
                        while (rem != 0) {
                            const int f = __ffsll((long long)rem) - 1;
                            const uint64_t qs = wave_bcast(segkey, f);
                            bool has = ok && !late && segkey == qs;
                            const unsigned long long members = __ballot(has);
                            R fold = mine;
                            for (int d = 1; d < WAVE; d <<= 1) {
                                const R other = wave_shfl_up(fold, d);
                                const bool ohas = __shfl_up((int)has, d, WAVE) != 0 && lane >= d;
                                if (ohas) {
                                    if (has) {
                                        R tmp;
                                        fat_combine(comb, other, fold, tmp);
                                        fold = tmp;
                                    } else {
                                        fold = other;
                                        has = true;
                                    }
                                }
                            }
                            if (lane == 63 - __builtin_clzll(members)) {
                                qval = fold;
                                qtail = true;
                            }
                            rem &= ~members;
                        }
                    }
                    // The quanta below Lq_new are complete, 64 of them at a time (a chunk usually completes a handful; a chunk that
                    // follows a jump in time - the piece of a lane that runs ahead of the others - completes what the triggering
                    // delay had kept open, thousands of quanta: lane 0 used to walk them one by one, milliseconds per run).
                    for (uint64_t b = Lq; b < Lq_new;) {
                        const uint32_t ms = (uint32_t)(Lq_new - b < (uint64_t)WAVE ? Lq_new - b : (uint64_t)WAVE);
                        if (pend + ms > (uint32_t)WAVE) fire_pending();
                        const uint64_t Pb = st.panes;
                        const uint64_t off = b - Lq;  // (the ring holds the acc_cap quanta from Lq on)
                        // every completed quantum starts from what the ring of open quanta holds for it (or as an empty pane) ...
                        if ((uint32_t)lane < ms) {
                            const uint64_t qq = b + (uint64_t)lane;
                            R e{};
                            if (off + (uint64_t)lane < (uint64_t)g.acc_cap && rused[(uint32_t)(qq & amask)]) e = racc[(uint32_t)(qq & amask)];
                            e.setControlFields(key, qq, (qq + 1) * g.pane_len - 1);
                            wtree[CW + (uint32_t)((Pb + (uint64_t)lane) & (CW - 1))] = e;
                        }
                        wave_sync_lds();
                        // ... the chunk's tuples of a completed quantum join it ...
                        if (qtail && segkey >= b && segkey < b + ms) {
                            const uint32_t x = (uint32_t)(segkey - b), a = (uint32_t)(segkey & amask);
                            R v = qval;
                            if (segkey - Lq < (uint64_t)g.acc_cap && rused[a]) fat_combine(comb, racc[a], qval, v);
                            v.setControlFields(key, segkey, (segkey + 1) * g.pane_len - 1);
                            wtree[CW + (uint32_t)((Pb + (uint64_t)x) & (CW - 1))] = v;
                        }
                        wave_sync_lds();
                        // ... their slots in the ring are free again ...
                        if ((uint32_t)lane < ms && off + (uint64_t)lane < (uint64_t)g.acc_cap) rused[(uint32_t)((b + (uint64_t)lane) & amask)] = 0;
                        wave_sync_lds();
                        st.panes = Pb + ms;
                        pend += ms;
                        b += ms;
                    }
                    // ... and the tuples of the quanta that stay open are folded into the ring
                    if (qtail && segkey >= Lq_new) {
                        const uint32_t a = (uint32_t)(segkey & amask);
                        if (rused[a]) {
                            R tmp;
                            fat_combine(comb, racc[a], qval, tmp);
                            racc[a] = tmp;
                        } else {
                            racc[a] = qval;
                            rused[a] = 1;
                        }
                    }
                    if (n_late != 0 && lane == 0) atomicAdd(&ctr->dropped, n_late);
                    st.last_quantum = Lq_new;
                    if (n_late < cnt && q_max + 1 > st.max_quantum) st.max_quantum = q_max + 1;
                }
                st.seen = 1;
                run_key = key;
                wave_sync_lds();
            } else {
                // sequential rules, one step per segment, by lane 0 (on a consistent tree: fire what is pending first)
                fire_pending();
                if (tail) {
                    const int first = 63 - __builtin_clzll(upto_me);  // lane of my segment's boundary
                    segs[seg].agg = val;
                    segs[seg].len = (uint32_t)(lane - first + 1);
                }
                wave_sync_lds();
                if (lane == 0) {
                    // one copy of the loop per staging combination, so that the compiler knows which pointers
                    // are LDS (ds_* instructions) and which are global instead of emitting flat accesses
                    auto fold_all = [&](R* tr, R* ac, uint8_t* us, const FFATGeometry& geo) {
                        for (uint32_t u = 0; u < n_segs; u++) ffat_fold(st, tr, ac, us, geo, segs[u].agg, segs[u].len, comb, emit, ctr);
                    };
                    if (lay.tree_nodes != 0 && acc_staged)
                        fold_all(ltree, lacc, lused, gw);
                    else if (lay.tree_nodes != 0)
                        fold_all(ltree, gacc, gused, gw);
                    else if (acc_staged)
                        fold_all(gtree, lacc, lused, g);
                    else
                        fold_all(gtree, gacc, gused, g);
                }
                wave_sync_lds();
                if (bulk) {  // the bulk path keeps the state (and the number of parked results) identical in every lane
                    st = wave_bcast(st, 0);
                    emit.n = (uint32_t)__shfl((int)emit.n, 0, WAVE);
                } else {
                    emit.flush_wave();
                    wave_sync_lds();
                }
            }
            if (cnt < WAVE) break;
            reg = nreg;
            ok = nok;
        }
        fire_pending();
        emit.flush_wave();  // whatever the last chunks parked
        wave_sync_lds();
        if (bulk && !GBULK) {
            // export the panes a future window (or the end-of-stream flush) can still need, rebuild the key's tree
            const uint64_t lo = st.panes > g.win_panes ? st.panes - g.win_panes : 0;
            for (uint64_t idx = lo + lane; idx < st.panes; idx += WAVE)
                gtree[g.tree_cap + (uint32_t)(idx & (g.tree_cap - 1))] = ltree[CW + (uint32_t)(idx & (CW - 1))];
            wave_sync_lds();
            fat_rebuild_wave(gtree, g.tree_cap, comb);
        } else if (lay.tree_nodes != 0) {
            for (uint32_t k = lane; k < 2 * g.tree_cap; k += WAVE) gtree[k] = ltree[k];
        }
        if (acc_staged) {
            for (uint32_t k = lane; k < g.acc_cap; k += WAVE) {
                gacc[k] = lacc[k];
                gused[k] = lused[k];
            }
        }
        if (lane == 0) recs.store(slot, st, g.is_tb);
        wave_sync_lds();
    }
}

// end of stream: one thread per table slot; completes what is open and emits the
// remaining (incomplete) windows, as the CPU engine does (win_seqffat.hpp flush)
// (the slots that hold a key that received tuples, compacted first: a round of the flush then never walks over empty slots -
// with a long triggering delay a round holds three or four keys, and the table of a replica may be mostly empty)
template <typename R, typename K>
__global__ void __launch_bounds__(BLOCK_THREADS)
    ffat_flush_list_kernel(KeyTableView<K> table, FFATRecords<R> recs, uint32_t table_cap, uint32_t* list, uint32_t* count) {
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < table_cap; slot += gridDim.x * blockDim.x)
        if (table.occupied(slot) && recs.header(slot).seen) list[atomicAdd(count, 1u)] = slot;
}

template <typename R, typename C, typename K>
__global__ void __launch_bounds__(BLOCK_THREADS)
    ffat_flush_kernel(KeyTableView<K> table, const uint32_t* list, uint32_t lo, uint32_t hi, FFATRecords<R> recs, R* trees,
                      FFATGeometry g, R* out, uint32_t* out_count, FFATCounters* ctr, C comb) {
    for (uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += gridDim.x * blockDim.x) {
        const uint32_t slot = list[i];
        FFATKeyState<R> st = recs.load(slot, g.is_tb);
        R* tree = trees + (size_t)slot * 2 * g.tree_cap;
        GlobalEmit<R> emit{out, out_count, g.out_cap, ctr};
        const auto key = table.key_at(slot);
        if (!g.is_tb) {
            if (st.pcount > 0) {
                ffat_complete_pane(st, tree, g, key, st.partial, comb, emit);
                st.pcount = 0;
            }
        } else {
            R* acc = recs.ring(slot);
            uint8_t* used = recs.used(slot);
            while (st.last_quantum < st.max_quantum) {
                const uint32_t a = (uint32_t)(st.last_quantum & (g.acc_cap - 1));
                R pane{};
                if (used[a]) {
                    pane = acc[a];
                    used[a] = 0;
                }
                pane.setControlFields(key, st.last_quantum, (st.last_quantum + 1) * g.pane_len - 1);
                ffat_complete_pane(st, tree, g, key, pane, comb, emit);
                st.last_quantum++;
            }
        }
        // incomplete windows: window lwid covers panes [lwid*slide_panes, +win_panes)
        while (st.next_lwid * g.slide_panes < st.panes) {
            const uint64_t first = st.next_lwid * g.slide_panes;
            const uint32_t cnt = (uint32_t)(st.panes - first);
            R res;
            fat_query_circular(tree, g.tree_cap, (uint32_t)(first & (g.tree_cap - 1)), cnt < g.win_panes ? cnt : g.win_panes, res, comb);
            res.setControlFields(key, st.next_lwid, cf_ts(res));
            st.next_lwid++;
            emit(res);
        }
        recs.store(slot, st, g.is_tb);
    }
}

// clamp the result count to the capacity and mirror it for the host
__global__ void ffat_finalize_kernel(BatchHeader* out_hdr, uint32_t cap, volatile uint32_t* mirror, uint32_t* n_long = nullptr,
                                     uint32_t* n_huge = nullptr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (n_long) *n_long = 0;  // ready for the next batch (same stream)
        if (n_huge) *n_huge = 0;
        uint32_t c = out_hdr->count;
        if (c > cap) c = cap;
        out_hdr->count = c;
        *mirror = c;
    }
}

}  // namespace gpu
}  // namespace wf
#include "ffat_huge_gpu.hpp"
namespace wf {
namespace gpu {

// The engine proper: owns the device state, processes device batches of tuples and
// produces device batches of results.  Single stream, not thread safe.
template <typename T, typename R, typename L, typename C>
class KeyedFFATEngine {
    using key_t = key_of_t<T>;
    static_assert(is_device_key_v<key_t>, "Key_FFAT_GPU: keys must be integers or fixed-size keys (wf::FixedKey; primitives.hpp is_device_key_v)");
    static_assert(std::is_same_v<key_t, key_of_t<R>>, "Key_FFAT_GPU: tuples and results must have the same key type");

public:
    KeyedFFATEngine(L lift, C comb, uint64_t win, uint64_t slide, uint64_t delay, win_type_t type, size_t max_keys, size_t batch_capacity,
                    size_t out_capacity, int device, size_t pool_slabs, bool skip_gaps = false)
        : lift_(lift), comb_(comb), device_(device) {
        g_.skip_gaps = skip_gaps ? 1u : 0u;
        if (win == 0 || slide == 0) failwith("Key_FFAT_GPU: window length or slide cannot be zero");
        // Tumbling (slide == win) and hopping (slide > win) windows are served by the same machinery: a pane is
        // gcd(win, slide), a tumbling window is exactly one pane, and the panes that fall in the gaps between hopping windows
        // are folded into the tree like any other but never queried.  (The reference refuses them, key_ffat.hpp:132-135,
        // although the Yahoo benchmark's 10 s tumbling count per campaign is precisely such a query.)
        const uint64_t pane = std::gcd(win, slide);
        g_.is_tb = (type == TB);
        g_.pane_len = pane;
        g_.win_panes = (uint32_t)(win / pane);
        g_.slide_panes = (uint32_t)(slide / pane);
        g_.delay = delay;
        uint32_t cap = 2;
        while (cap <= g_.win_panes) cap <<= 1;
        // long windows: the long-run kernel appends up to 64 panes to the key's tree before it evaluates the windows they fire
        if (g_.win_panes > WAVE && cap < g_.win_panes + WAVE) cap <<= 1;
        g_.tree_cap = cap;
        uint32_t acc = 2;
        const uint64_t need = (type == TB) ? delay / pane + 2 : 1;
        while (acc < need) acc <<= 1;
        g_.acc_cap = acc;
        g_.out_cap = (uint32_t)out_capacity;
        table_cap_ = 2;
        while (table_cap_ < 2 * max_keys) table_cap_ <<= 1;
        slot_bits_ = 0;
        while ((1ull << slot_bits_) < table_cap_) slot_bits_++;
        batch_cap_ = batch_capacity;
        WF_HIP_CHECK(hipSetDevice(device_));
        // never the legacy (null) stream: its implicit dependencies would break stream captures running in other threads
        Stream init;
        table_.init(table_cap_, init.get());
        rec_stride_ = FFATRecords<R>::stride_for(g_.acc_cap);
        records_.resize(table_cap_ * rec_stride_);
        WF_HIP_CHECK(hipMemsetAsync(records_.data(), 0, records_.bytes(), init.get()));
        trees_.resize((size_t)table_cap_ * 2 * g_.tree_cap);
        WF_HIP_CHECK(hipMemsetAsync(trees_.data(), 0, trees_.bytes(), init.get()));
        slot_.resize(batch_cap_);
        idx_.resize(batch_cap_);
        counters_.resize(1);
        WF_HIP_CHECK(hipMemsetAsync(counters_.data(), 0, sizeof(FFATCounters), init.get()));
        if (const char* e = std::getenv("WF_FFAT_LONG_RUN")) long_run_ = (uint32_t)std::max(2, std::atoi(e));
        long_heads_.resize(batch_cap_ / long_run_ + 2);
        n_long_.resize(2);  // [0] long runs, [1] huge runs of the batch in flight
        WF_HIP_CHECK(hipMemsetAsync(n_long_.data(), 0, 2 * sizeof(uint32_t), init.get()));
        // huge runs (ffat_huge_gpu.hpp): count-based windows, or time-based ones whose ring of open quanta is not enormous
        // (a long triggering delay keeps many quanta open: the scratch of a run spans all of them.  Sized for the 288 GB of an
        // MI355X: up to 2^18 open quanta per key as long as the scratch stays below 6 GB; beyond that the wave-per-run kernel)
        huge_ok_ = env_flag("WF_FFAT_HUGE", true) && batch_cap_ >= FFAT_HUGE_RUN;
        if (huge_ok_) {
            huge_max_runs_ = (uint32_t)(batch_cap_ / FFAT_HUGE_RUN + 1);
            const size_t leaves = batch_cap_ + (size_t)huge_max_runs_ * (g_.win_panes + 70 + (g_.is_tb ? g_.acc_cap : 0));
            if (g_.is_tb && (g_.acc_cap > (1u << 18) || leaves * (3 * sizeof(R) + 1) > (6ull << 30) || leaves >= (1ull << 31))) huge_ok_ = false;
            huge_leaves_cap_ = (uint32_t)leaves;
        }
        if (huge_ok_) {
            huge_heads_.resize(huge_max_runs_);
            huge_runs_.resize(huge_max_runs_);
            huge_ctl_.resize(1);
            huge_parts_.resize(2 * (batch_cap_ / WAVE + huge_max_runs_));
            huge_parts2_.resize(2 * (2 * (batch_cap_ / WAVE + huge_max_runs_) / WAVE + huge_max_runs_));
            huge_leaves_.resize(huge_leaves_cap_);
            if (g_.is_tb) huge_has_.resize(huge_leaves_cap_);
            if (g_.win_panes > FFAT_HUGE_DIRECT_WIN) {
                huge_pre_.resize(huge_leaves_cap_);
                huge_suf_.resize(huge_leaves_cap_);
            }
            WF_HIP_CHECK(hipHostMalloc((void**)&huge_seen_, 64, hipHostMallocDefault));
            // so few keys that a full batch MUST hold huge runs: the huge-run kernels are on from the first batch (otherwise the
            // batches enqueued before the device's first report has reached the host fold their runs with one wave each)
            *huge_seen_ = max_keys * (size_t)FFAT_HUGE_RUN <= batch_cap_ ? 1u : 0u;
        }
        // the long-run kernel stages the tree and the ring of open quanta of a key in LDS when they fit the LDS the device
        // offers a block of 4 waves (160 KB on gfx950, less 4 KB of slack; hip_utils.hpp max_lds_per_block)
        const size_t lds_budget = max_lds_per_block(device_) - 4096;
        long_layout_ = FFATLongLayout<R>::choose(g_, lds_budget / (BLOCK_THREADS / WAVE));
        long_lds_bytes_ = (BLOCK_THREADS / WAVE) * long_layout_.wave_bytes();
        if (long_lds_bytes_ > lds_budget) failwith("Key_FFAT_GPU: result type too large for the LDS staging buffers");
        allow_dynamic_lds(&ffat_update_long_kernel<true, T, R, L, C>, long_lds_bytes_);
        allow_dynamic_lds(&ffat_update_long_kernel<false, T, R, L, C>, long_lds_bytes_);
        // the engine's kernels run on another stream: wait for the initialisation here
        dups_.use_external_stamps(reinterpret_cast<uint32_t*>(records_.data() + FFATRecords<R>::STAMP_OFF), (uint32_t)rec_stride_, init.get());
        init.sync();
        sorter_.reserve(batch_cap_);
        out_pool_.reset(new BatchPool(MemKind::DEVICE, device_, sizeof(R), out_capacity, pool_slabs));
    }

    ~KeyedFFATEngine() {
        for (auto& kv : graphs_) (void)hipGraphExecDestroy(kv.second);
        if (huge_seen_) (void)hipHostFree((void*)huge_seen_);
    }
    KeyedFFATEngine(const KeyedFFATEngine&) = delete;
    KeyedFFATEngine& operator=(const KeyedFFATEngine&) = delete;

    const FFATGeometry& geometry() const { return g_; }
    size_t device_bytes() const { return table_.bytes() + records_.bytes() + trees_.bytes(); }

    // enqueue the processing of `in` on `s`; returns the batch of fired windows (count on device).
    // With WF_HIP_GRAPHS=1 the per-batch sequence (header clear, slot assignment, 3 kernels per sort
    // pass, the two update kernels, finalize: 10-16 launches) is captured ONCE per (input slab, output
    // slab, size) into a hipGraph and replayed with a single hipGraphLaunch; slabs come from small
    // bounded pools, so a handful of graphs covers the steady state.  Measured on MI355X / ROCm 7.0
    // (profiles/r1_latency_sweep/hip_graphs.txt) replay is NOT faster than the plain launches - 16 Ki
    // batches: 324-353 vs 362 M tuples/s, 256 Ki: 1035-1105 vs 1130 - so it is off by default.
    Batch* process(Batch* in, hipStream_t s, uint64_t& kernels) {
        const size_t upper = in->count_on_host ? in->host_hdr.count : in->capacity;
        if (upper > batch_cap_) failwith("Key_FFAT_GPU: input batch larger than the configured batch capacity");
        Batch* out = out_pool_->acquire(s);
        out->host_hdr = in->host_hdr;
        out->count_on_host = false;
        kernels += 1 + (upper > 0 ? 3 + sorter_.kernels_per_sort(slot_bits_ + 1) : 0);
        if (use_graphs_) {
            const GraphKey key{in->slab, out->slab, upper};
            auto it = graphs_.find(key);
            if (it == graphs_.end() && graphs_.size() < MAX_GRAPHS) {
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                // captured on a stream of its own that never carries real work or event dependencies
                if (!capture_stream_) capture_stream_.reset(new Stream());
                WF_HIP_CHECK(hipStreamBeginCapture(capture_stream_->get(), hipStreamCaptureModeThreadLocal));
                enqueue(in, out, upper, capture_stream_->get());
                WF_HIP_CHECK(hipStreamEndCapture(capture_stream_->get(), &graph));
                WF_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                WF_HIP_CHECK(hipGraphDestroy(graph));
                it = graphs_.emplace(key, exec).first;
            }
            if (it != graphs_.end()) {
                WF_HIP_CHECK(hipGraphLaunch(it->second, s));
                graph_launches_++;
                return out;
            }
        }
        enqueue(in, out, upper, s);
        return out;
    }
    uint64_t graph_launches() const { return graph_launches_; }
    uint64_t presorted_batches() const { return presorted_batches_; }
    uint64_t probed_batches() const { return dups_.probed_batches(); }

private:
    void enqueue(Batch* in, Batch* out, size_t upper, hipStream_t s) {
        if (upper == 0) WF_HIP_CHECK(hipMemsetAsync(out->header(), 0, sizeof(BatchHeader), s));
        if (upper > 0) {
            const unsigned grid = grid_for(upper);
            // also clears the header of the output batch
            // time-based windows fed by merged, unsynchronised lanes: runs out of timestamp order are folded sequentially by the
            // kernels below; once they report such runs the batches are ordered by (slot, timestamp) first, so that the parallel
            // in-order paths (bulk, huge runs) apply
            batch_no_++;
            const bool presort = g_.is_tb && !use_graphs_ && disorder_.decide(batch_no_);
            // a batch that holds no key twice skips the sort (decided on the device: primitives.hpp DupProbe)
            const DupProbe probe = (presort || use_graphs_) ? DupProbe{} : dups_.probe(batch_no_);
            const SortGate gate = dups_.gate(probe);
            hipLaunchKernelGGL((assign_slots_kernel<T, key_t>), dim3(grid), dim3(BLOCK_THREADS), 0, s, in->header(), in->template data<T>(),
                               table_.view(), slot_.data(), idx_.data(), &counters_.data()->table_full, out->header(), (uint32_t*)nullptr, probe);
            const auto sorted = presort ? presorter_.sort(in->header(), in->template data<T>(), in->device_count_ptr(), upper, slot_.data(), slot_bits_ + 1,
                                                         disorder_.ts_bits(), g_.pane_len, disorder_.clamped(), batch_no_, s)
                                        : sorter_.sort(slot_.data(), idx_.data(), in->device_count_ptr(), upper, slot_bits_ + 1, s, gate);
            const SortedSel sel{sorted.keys, sorted.vals, slot_.data(), idx_.data(), gate};
            if (presort) presorted_batches_++;
            // huge runs: the four extra launches are only made while the stream shows such runs (the short-run kernel reports
            // them through a pinned word; a run that shows up while they are off is handled by the wave-per-run kernel)
            uint32_t huge_mode = 0;
            if (huge_ok_ && !use_graphs_) {
                const uint32_t seen = *huge_seen_;
                huge_mode = (seen != 0 && batch_no_ - seen <= HUGE_LINGER_BATCHES) ? 2u : 1u;
            }
            hipLaunchKernelGGL((ffat_update_kernel<T, R, L, C>), dim3(grid), dim3(BLOCK_THREADS), 0, s, in->header(), in->template data<T>(),
                               sel, recs(), trees_.data(), g_,
                               out->template data<R>(), out->device_count_ptr(), counters_.data(), long_heads_.data(), n_long_.data(), lift_,
                               comb_, huge_mode, FFAT_HUGE_RUN, huge_heads_.data(), n_long_.data() + 1, huge_seen_, batch_no_, long_run_);
            if (huge_mode == 2) {
                hipLaunchKernelGGL((ffat_huge_setup_kernel<T, R>), dim3(1), dim3(BLOCK_THREADS), 0, s, in->header(), in->template data<T>(), sorted.keys,
                                   sorted.vals, recs(), g_, huge_heads_.data(), n_long_.data() + 1, huge_runs_.data(), huge_ctl_.data(),
                                   huge_max_runs_, huge_leaves_cap_);
                const unsigned cgrid = (unsigned)std::min<size_t>(2048, std::max<size_t>(1, (upper / WAVE + huge_max_runs_) / (BLOCK_THREADS / WAVE) + 1));
                hipLaunchKernelGGL((ffat_huge_chunks_kernel<T, R, L, C>), dim3(cgrid), dim3(BLOCK_THREADS), 0, s, in->template data<T>(), sorted.vals,
                                   g_.pane_len, g_.is_tb, g_.win_panes, huge_runs_.data(), huge_ctl_.data(), huge_parts_.data(), huge_leaves_.data(),
                                   huge_has_.data(), lift_, comb_);
                const unsigned sgrid = (unsigned)std::min<size_t>(1024, std::max<size_t>(1, (upper / WAVE / 32 + huge_max_runs_) / (BLOCK_THREADS / WAVE) + 1));
                hipLaunchKernelGGL((ffat_huge_stitch_kernel<R, C>), dim3(sgrid), dim3(BLOCK_THREADS), 0, s, g_, huge_runs_.data(), huge_ctl_.data(),
                                   huge_parts_.data(), huge_parts2_.data(), huge_leaves_.data(), huge_has_.data(), table_.view(), comb_);
                const unsigned fgrid = (unsigned)std::min<size_t>(1024, std::max<size_t>(1, huge_max_runs_));  // a workgroup per run
                hipLaunchKernelGGL((ffat_huge_fix_kernel<R, C>), dim3(fgrid), dim3(BLOCK_THREADS), 0, s, recs(), trees_.data(), g_, huge_runs_.data(), huge_ctl_.data(), huge_parts2_.data(), huge_leaves_.data(),
                                   huge_has_.data(), out->device_count_ptr(), long_heads_.data(), n_long_.data(), table_.view(), counters_.data(), comb_,
                                   disorder_.seen(), batch_no_);
                const bool hgw = g_.win_panes > FFAT_HUGE_DIRECT_WIN;
                if (hgw) {
                    // two work items per block of win_panes leaves, a wave each
                    const size_t items = 2 * ((size_t)huge_leaves_cap_ / g_.win_panes + huge_max_runs_);
                    const unsigned hgrid = (unsigned)std::min<size_t>(1024, std::max<size_t>(1, items / (BLOCK_THREADS / WAVE) + 1));
                    hipLaunchKernelGGL((ffat_huge_hgw_kernel<R, C>), dim3(hgrid), dim3(BLOCK_THREADS), 0, s, g_, huge_runs_.data(), huge_ctl_.data(),
                                       huge_leaves_.data(), huge_pre_.data(), huge_suf_.data(), comb_);
                }
                hipLaunchKernelGGL((ffat_huge_windows_kernel<R, C>), dim3(grid), dim3(BLOCK_THREADS), 0, s, g_, huge_runs_.data(), huge_ctl_.data(),
                                   huge_leaves_.data(), hgw ? huge_pre_.data() : (const R*)nullptr, hgw ? huge_suf_.data() : (const R*)nullptr,
                                   out->template data<R>(), counters_.data(), table_.view(), comb_);
            }
            // at most upper / FFAT_LONG_RUN long runs; a wave each, 4 waves per block (exits at once when there is none)
            const unsigned lgrid = (unsigned)std::min<size_t>(512, std::max<size_t>(1, upper / long_run_ / (BLOCK_THREADS / WAVE) + 1));
            if (long_layout_.bulk == 2)
                hipLaunchKernelGGL((ffat_update_long_kernel<true, T, R, L, C>), dim3(lgrid), dim3(BLOCK_THREADS), long_lds_bytes_, s, in->header(),
                                   in->template data<T>(), sorted.keys, sorted.vals, recs(), trees_.data(),
                                   g_, out->template data<R>(), out->device_count_ptr(), counters_.data(), long_heads_.data(), n_long_.data(),
                                   long_layout_, lift_, comb_, disorder_.seen(), batch_no_);
            else
                hipLaunchKernelGGL((ffat_update_long_kernel<false, T, R, L, C>), dim3(lgrid), dim3(BLOCK_THREADS), long_lds_bytes_, s, in->header(),
                                   in->template data<T>(), sorted.keys, sorted.vals, recs(), trees_.data(),
                                   g_, out->template data<R>(), out->device_count_ptr(), counters_.data(), long_heads_.data(), n_long_.data(),
                                   long_layout_, lift_, comb_, disorder_.seen(), batch_no_);
        }
        // (Folding this into the last workgroup of the long-run kernel - a ticket - was tried and measured slower: device-fed
        // flagship 2.31-2.49 G with the fold, 2.55-2.63 G with the launch; 16 Ki batches 720 vs 745 M tuples/s.)
        hipLaunchKernelGGL(ffat_finalize_kernel, dim3(1), dim3(64), 0, s, out->header(), g_.out_cap, out->count_mirror, n_long_.data(),
                           n_long_.data() + 1);
    }

public:
    // end of stream: emit every incomplete window.  The table is flushed in slot
    // ranges sized so that the windows of a range always fit one output batch;
    // the caller hands each batch downstream before asking for the next one, so
    // the bounded output pool never deadlocks.
    // flush one slot range at a time (bounded number of output batches in flight)
    bool flush_range(size_t& cursor, hipStream_t s, uint64_t& kernels, Batch*& out) {
        if (cursor == 0) {
            // the keys to flush, in slot order (the order of the results of different keys then does not depend on the atomics)
            flush_list_.resize(table_cap_ + 1);
            uint32_t* count = flush_list_.data() + table_cap_;
            WF_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(uint32_t), s));
            hipLaunchKernelGGL((ffat_flush_list_kernel<R>), dim3(grid_for(table_cap_)), dim3(BLOCK_THREADS), 0, s, table_.view(), recs(),
                               (uint32_t)table_cap_, flush_list_.data(), count);
            kernels++;
            uint32_t n = 0;
            WF_HIP_CHECK(hipMemcpyAsync(&n, count, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            WF_HIP_CHECK(hipStreamSynchronize(s));
            flush_n_ = n;
            if (n > 1) {
                std::vector<uint32_t> h(n);
                WF_HIP_CHECK(hipMemcpyAsync(h.data(), flush_list_.data(), n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                WF_HIP_CHECK(hipStreamSynchronize(s));
                std::sort(h.begin(), h.end());
                WF_HIP_CHECK(hipMemcpyAsync(flush_list_.data(), h.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
                WF_HIP_CHECK(hipStreamSynchronize(s));
            }
        }
        if (cursor >= flush_n_) return false;
        const size_t per_key = (size_t)g_.win_panes / g_.slide_panes + 2 + (g_.is_tb ? g_.acc_cap : 0);
        const size_t slots_per_round = std::max<size_t>(1, g_.out_cap / per_key);
        const size_t lo = cursor, hi = std::min(flush_n_, lo + slots_per_round);
        cursor = hi;
        out = out_pool_->acquire(s);
        out->host_hdr = BatchHeader{};
        out->count_on_host = false;
        WF_HIP_CHECK(hipMemsetAsync(out->header(), 0, sizeof(BatchHeader), s));
        hipLaunchKernelGGL((ffat_flush_kernel<R, C>), dim3(grid_for(hi - lo)), dim3(BLOCK_THREADS), 0, s, table_.view(), flush_list_.data(), (uint32_t)lo,
                           (uint32_t)hi, recs(), trees_.data(), g_, out->template data<R>(), out->device_count_ptr(),
                           counters_.data(), comb_);
        hipLaunchKernelGGL(ffat_finalize_kernel, dim3(1), dim3(64), 0, s, out->header(), g_.out_cap, out->count_mirror);
        kernels += 2;
        return true;
    }

    FFATCounters read_counters(hipStream_t s) {
        FFATCounters c{};
        WF_HIP_CHECK(hipMemcpyAsync(&c, counters_.data(), sizeof(c), hipMemcpyDeviceToHost, s));
        WF_HIP_CHECK(hipStreamSynchronize(s));
        return c;
    }

private:
    L lift_;
    C comb_;
    int device_;
    FFATGeometry g_{};
    size_t table_cap_ = 0, batch_cap_ = 0;
    int slot_bits_ = 0;
    KeyTable<key_t> table_;
    DeviceBuffer<char> records_;  // one record per table slot: header, ring flags, ring (FFATRecords)
    size_t rec_stride_ = 0;
    FFATRecords<R> recs() const { return FFATRecords<R>{records_.data(), (uint32_t)rec_stride_, (uint32_t)FFATRecords<R>::ring_offset(g_.acc_cap)}; }
    DeviceBuffer<R> trees_;
    DeviceBuffer<uint32_t> slot_, idx_;
    DeviceBuffer<uint32_t> flush_list_;  // end of stream: the slots to flush, then their number
    size_t flush_n_ = 0;
    DeviceBuffer<FFATCounters> counters_;
    DeviceBuffer<uint32_t> long_heads_, n_long_;
    uint32_t long_run_ = FFAT_LONG_RUN;  // (WF_FFAT_LONG_RUN: experiments)
    FFATLongLayout<R> long_layout_{0, 0, 0};
    // huge runs
    static constexpr uint32_t HUGE_LINGER_BATCHES = 256;  // batches without a huge run after which their kernels are switched off
    bool huge_ok_ = false;
    uint32_t huge_max_runs_ = 0, huge_leaves_cap_ = 0, batch_no_ = 0;
    volatile uint32_t* huge_seen_ = nullptr;  // pinned: number of the last batch in which the device saw a huge run
    DeviceBuffer<uint32_t> huge_heads_;
    DeviceBuffer<FFATHugeRun> huge_runs_;
    DeviceBuffer<FFATHugeCtl> huge_ctl_;
    DeviceBuffer<FFATPartial<R>> huge_parts_, huge_parts2_;
    DeviceBuffer<R> huge_leaves_, huge_pre_, huge_suf_;
    DeviceBuffer<uint8_t> huge_has_;
    // hipGraph cache
    struct GraphKey {
        const void* in;
        const void* out;
        size_t n;
        bool operator<(const GraphKey& o) const { return std::tie(in, out, n) < std::tie(o.in, o.out, o.n); }
    };
    static constexpr size_t MAX_GRAPHS = 256;
    std::map<GraphKey, hipGraphExec_t> graphs_;
    std::unique_ptr<Stream> capture_stream_;
    bool use_graphs_ = env_flag("WF_HIP_GRAPHS", false);
    uint64_t graph_launches_ = 0;
    static bool env_flag(const char* name, bool dflt) {
        const char* v = std::getenv(name);
        return v ? std::atoi(v) != 0 : dflt;
    }
    size_t long_lds_bytes_ = 0;
    RadixSorter sorter_;
    TsPresorter presorter_;
    DisorderMonitor disorder_;
    DupMonitor dups_;
    uint64_t presorted_batches_ = 0;
    std::unique_ptr<BatchPool> out_pool_;
};

}  // namespace gpu

// ---- operator ------------------------------------------------------------------
struct KeyFFATGPUConfig {
    uint64_t win_len = 2, slide_len = 1, delay = 0;
    win_type_t type = CB;
    size_t max_keys = 1 << 16;   // distinct keys per replica the device table is sized for
    size_t out_capacity = 0;     // results per output batch (0: twice the input batch capacity)
    // time-based windows: a key silent for much longer than a window normally produces one
    // empty result per slide when it speaks again; true = count those windows instead of
    // emitting them (FFATCounters::skipped_windows)
    bool skip_empty_gaps = false;
    // batch input only: lane i already carries exactly the keys of replica i (key-partitioned
    // source or upstream exchange); otherwise batches go through the device KeyBy emitter
    bool lanes_are_key_disjoint = false;
};

template <typename tuple_t, typename result_t, typename L, typename C>
class KeyFFATGPU_Node : public GPU_Replica<tuple_t> {
public:
    KeyFFATGPU_Node(L lift, C comb, KeyFFATGPUConfig wcfg, GPUConfig cfg, edge_kind_t in_kind, std::string name, size_t par, size_t idx,
                    meta::closing_func_t closing, std::shared_ptr<StatsHub> hub, std::shared_ptr<std::vector<gpu::FFATCounters>> counters)
        : GPU_Replica<tuple_t>(cfg, in_kind, std::move(name), par, idx, std::move(closing), std::move(hub), true), lift_(lift), comb_(comb),
          wcfg_(wcfg), counters_(std::move(counters)) {}

protected:
    void process_device(gpu::Batch* d) override {
        if (!engine_) {
            // (a user's output capacity grows with input batches that coalesce several of the configured length: GPU_Replica)
            size_t out_cap = this->scaled_out_capacity(wcfg_.out_capacity, d->capacity);
            if (out_cap == 0) out_cap = 2 * d->capacity;
            engine_.reset(new gpu::KeyedFFATEngine<tuple_t, result_t, L, C>(lift_, comb_, wcfg_.win_len, wcfg_.slide_len, wcfg_.delay, wcfg_.type,
                                                                            wcfg_.max_keys, d->capacity, out_cap, this->cfg_.gpu_id,
                                                                            this->cfg_.pool_slabs, wcfg_.skip_empty_gaps));
        }
        gpu::Batch* out = engine_->process(d, this->stream(), this->num_kernels_);
        d->release(this->stream());
        this->emit_device(out);
        // an unbounded stream never reaches gpu_flush(): lost results / unassigned keys are also looked for periodically
        if (++since_check_ >= 256) {
            since_check_ = 0;
            check_counters(engine_->read_counters(this->stream()));
        }
    }
    void check_counters(const gpu::FFATCounters& c) {
        // losing results silently is not an option: fail loudly and say which knob to turn
        if (c.overflow > 0)
            fatal_error("Key_FFAT_GPU: " + std::to_string(c.overflow) + " window results did not fit the output batches; raise the output capacity "
                        "(KeyFFATGPU_Builder::withOutputCapacity)");
        if (c.table_full > 0)
            fatal_error("Key_FFAT_GPU: tuples without a state slot - the device key table is full (raise KeyFFATGPU_Builder::withMaxKeys) or a "
                        "tuple carries the reserved key 2^64-1");
    }
    uint32_t since_check_ = 0;
    void gpu_flush() override {
        if (!engine_) return;
        size_t cursor = 0;
        gpu::Batch* out = nullptr;
        while (engine_->flush_range(cursor, this->stream(), this->num_kernels_, out)) this->emit_device(out);
        const gpu::FFATCounters c = engine_->read_counters(this->stream());
        if (counters_) (*counters_)[this->context.getReplicaIndex()] = c;
        dropped_ = c.dropped;
        check_counters(c);
    }

    void finalize_stats(Stats_Record& s) override {
        GPU_Replica<tuple_t>::finalize_stats(s);
        s.inputs_ignored = dropped_;  // late tuples, counted on the device
    }

private:
    uint64_t dropped_ = 0;
    L lift_;
    C comb_;
    KeyFFATGPUConfig wcfg_;
    std::shared_ptr<std::vector<gpu::FFATCounters>> counters_;
    std::unique_ptr<gpu::KeyedFFATEngine<tuple_t, result_t, L, C>> engine_;
};

// Keyed operator.  With tuple input the MultiPipe routes by key on the host
// (KF_Emitter) and every replica batches its own keys; with batch input replica i
// consumes lane i directly, so the producer lanes must carry disjoint key sets
// (true by construction after a KeyBy exchange or a key-partitioned source).
template <typename tuple_t, typename result_t>
class Key_FFAT_GPU : public Basic_Operator {
public:
    template <typename L, typename C>
    Key_FFAT_GPU(L lift, C comb, KeyFFATGPUConfig wcfg, size_t parallelism, std::string name, GPUConfig cfg = GPUConfig(),
                 meta::closing_func_t closing = nullptr, routing_func_t routing = default_routing)
        : name_(std::move(name)), par_(parallelism), wcfg_(wcfg), routing_(std::move(routing)), kind_(std::make_shared<edge_kind_t>(EDGE_TUPLE)),
          counters_(std::make_shared<std::vector<gpu::FFATCounters>>(parallelism)), slabs_(cfg.pool_slabs), gpu_id_(cfg.gpu_id), cfg_(cfg) {
        if (par_ == 0) fatal_error("Key_FFAT_GPU has parallelism zero");
        auto hub = hub_;
        auto kind = kind_;
        auto counters = counters_;
        const std::string n = name_;
        const size_t N = par_;
        factory_ = [=](rt::Graph& g) {
            std::vector<rt::Node*> reps;
            for (size_t i = 0; i < N; i++)
                reps.push_back(g.adopt(new KeyFFATGPU_Node<tuple_t, result_t, L, C>(lift, comb, wcfg, cfg, *kind, n, N, i, closing, hub, counters)));
            return reps;
        };
    }
    std::string getName() const override { return name_; }
    size_t getParallelism() const override { return par_; }
    routing_modes_t getRoutingMode() const override { return *kind_ == EDGE_TUPLE ? COMPLEX : FORWARD; }
    bool isWindowed() const override { return true; }
    bool isGPUOperator() const override { return true; }
    const std::type_info& input_type() const override { return typeid(tuple_t); }
    const std::type_info& output_type() const override { return typeid(result_t); }
    edge_kind_t output_kind() const override { return EDGE_DEVICE_BATCH; }
    bool accepts_input_kind(edge_kind_t) const override { return true; }
    void set_input_kind(edge_kind_t k) override { *kind_ = k; }
    std::function<rt::Node*(edge_kind_t)> output_adapter_factory() const override { return batch_adapter_factory<result_t>(slabs_); }
    // per-replica device counters, valid after the end of the run
    std::vector<gpu::FFATCounters> getDeviceCounters() const { return *counters_; }

    bool merges_batch_lanes_in_order() const override { return !wcfg_.lanes_are_key_disjoint; }
    bool takes_batch_lanes_one_to_one() const override { return wcfg_.lanes_are_key_disjoint; }
    // tuple producers fill batches themselves (count-based windows keep the reference's rule: DETERMINISTIC graphs only, which
    // the tuple path enforces)
    std::function<rt::Node*(size_t, size_t)> input_batcher_factory(Mode m) const override {
        if (wcfg_.lanes_are_key_disjoint || (wcfg_.type == CB && m != Mode::DETERMINISTIC)) return nullptr;
        return batcher_factory<tuple_t>(cfg_);
    }
    std::vector<StageSpec> build_stages(rt::Graph& g, Mode mode) override {
        StageSpec s;
        s.name = name_;
        const size_t N = par_;
        const edge_kind_t kind = *kind_;
        const bool det = mode == Mode::DETERMINISTIC;
        if (kind != EDGE_TUPLE && !wcfg_.lanes_are_key_disjoint && (N > 1 || det)) {
            // device KeyBy (dist/exchange_nodes.hpp): every producer packs its batches by replica, one endpoint thread merges the
            // packed slabs of all producers into one device batch per replica
            *kind_ = EDGE_DEVICE_BATCH;
            GPUConfig kcfg;
            kcfg.gpu_id = gpu_id_;
            kcfg.pool_slabs = slabs_;
            std::vector<StageSpec> stages = device_keyby_stages<tuple_t>(g, name_, kcfg, N, kind, hub_, det);
            s.replicas = factory_(g);
            s.routing = FORWARD;  // lane i of the endpoint -> replica i
            s.make_emitter = [N](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<tuple_t>(N), N); };
            stages.push_back(s);
            return stages;
        }
        s.replicas = factory_(g);
        if (*kind_ == EDGE_TUPLE) {
            s.routing = COMPLEX;
            routing_func_t rf = routing_;
            s.make_emitter = [rf, N](rt::Graph& gr, size_t) { return single_emitter(gr, new KF_Emitter<tuple_t>(rf, N), N); };
            const win_type_t type = wcfg_.type;
            s.make_ordering = [type](Mode m) -> rt::Node* {
                // count-based windows count arrivals per key on the device: only the order matters
                if (m == Mode::DETERMINISTIC) return new Ordering_Node<tuple_t>(type == CB ? TS_RENUMBERING : TS);
                return nullptr;
            };
        } else {
            s.routing = FORWARD;  // lane i -> replica i, or round-robin when parallelism differs
            s.make_emitter = [N](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<tuple_t>(N), N); };
        }
        return {s};
    }

private:
    std::string name_;
    size_t par_;
    KeyFFATGPUConfig wcfg_;
    routing_func_t routing_;
    std::shared_ptr<edge_kind_t> kind_;
    std::shared_ptr<std::vector<gpu::FFATCounters>> counters_;
    size_t slabs_;
    int gpu_id_;
    GPUConfig cfg_;
    std::function<std::vector<rt::Node*>(rt::Graph&)> factory_;
};

}  // namespace wf
