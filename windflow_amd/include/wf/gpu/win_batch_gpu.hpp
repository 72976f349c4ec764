// WinBatchEngine: the device-resident, batch-fed engine of the NON-incremental GPU
// window operators (Win_Seq_GPU, Key_Farm_GPU, Win_Farm_GPU and the GPU stages of
// Pane_Farm_GPU / Win_MapReduce_GPU).  The user supplies
//     __host__ __device__ void(uint64_t gwid, const tuple_t* data, size_t n,
//                              result_t* res, char* scratch, size_t scratch_size)
// (/root/reference/API:102-152) or, for associative queries, the
// wave-parallel form wave_reduce(lift, combine).
//
// What it replaces: /root/reference/wf/win_seq_gpu.hpp:83-669.  There the window
// bookkeeping runs per tuple on the host (archive, open windows, start/end
// offsets), fired windows are batched PER KEY, five cudaMemcpyAsync ship offsets
// and a pageable archive per batch, one batch is in flight, and K1
// (ComputeBatch_Kernel, :56-80) gives one thread a whole window.  Here the host
// only moves batches; everything else happens on the device, for all the keys of a
// mixed batch at once:
//   1. assign_slots + radix sort      key -> slot, tuples of a key become a run
//   2. wb_runs_kernel                 one thread per run: run table (slot, start, length, carried tuples)
//   3. scan                           offsets of the runs' segments in the work buffer
//   4. wb_gather_kernel               segment = [tuples carried from earlier batches | run], contiguous and
//                                     ordered: this is the reference's per-key archive, rebuilt per batch
//                                     only for the keys the batch touches
//   5. wb_plan_kernel + scan          windows fired by the batch, per key (window assignment of
//                                     SURVEY.md A.4 evaluated on the device), offsets of their results
//   6. wb_desc_kernel                 one thread per fired window: tuple range by binary search on the
//                                     ordinals (id / timestamp), result pre-stamped (key, gwid, ts)
//   7. wb_eval_kernel                 K1: the user's function, one LANE per window with an adaptive
//                                     number of active lanes per wave64 (opaque functors), or one WAVE
//                                     per window with coalesced loads and a shuffle reduction (wave_reduce)
//   8. wb_commit_kernel               tuples still needed by open windows go back to the key's carry slab
// so launches per batch are constant whatever the number of keys, several batches are in flight
// (stream-ordered pools), and end-of-stream runs on the device too (the reference computes the
// leftover windows on the CPU, :562-620).  The same engine serves every role of the composite
// operators (SEQ, PLQ, WLQ, MAP, REDUCE: SURVEY.md A.4, A.8, A.9), fed by device / host batches or
// by tuples staged into pinned batches by GPU_Replica.
#pragma once
#include <algorithm>
#include <memory>
#include <numeric>
#include <type_traits>
#include <vector>

#include "../win_seq.hpp"
#include "batch_ops.hpp"
#include "flatfat_gpu.hpp"
#include "keyby_gpu.hpp"
#include "primitives.hpp"

namespace wf {
namespace gpu {

// ---- associative window functions: wave-parallel evaluation --------------------------------
// wave_reduce(lift, combine): the window result is combine(...combine(lift(t0), lift(t1))..., lift(tn-1)) with an
// associative (not necessarily commutative) combine; the engine evaluates a window with a whole wave64.
template <typename L, typename C>
struct WaveReduce {
    static constexpr bool is_wave_reduce = true;
    L lift;
    C comb;
};
template <typename L, typename C>
inline WaveReduce<L, C> wave_reduce(L lift, C comb) {
    return WaveReduce<L, C>{lift, comb};
}
template <typename F, typename = void>
struct is_wave_reduce_t : std::false_type {};
template <typename F>
struct is_wave_reduce_t<F, std::void_t<decltype(F::is_wave_reduce)>> : std::true_type {};

// ---- map + reduce over one window (Win_MapReduce_GPU on batches) ---------------------------------
// map_reduce(map, reduce, M): the window is cut into M stripes (the tuples at window positions m, m + M, m + 2M ... form
// stripe m - on the CPU the tuples of a key are dealt round-robin to M MAP replicas, /root/reference/wf/wm_nodes.hpp:126-134),
// `map` computes a partial result per stripe and `reduce` the window's result from the M partials; both are device functors
// with the 6-argument signature.  One workgroup per window: its 256 threads lay the window out stripe by stripe in a scratch
// area, M of them run `map`, one runs `reduce`.
template <typename MapF, typename RedF>
struct MapReduceFunc {
    static constexpr bool is_map_reduce = true;
    MapF map;
    RedF red;
    uint32_t stripes;
};
template <typename MapF, typename RedF>
inline MapReduceFunc<MapF, RedF> map_reduce(MapF map, RedF red, size_t stripes) {
    return MapReduceFunc<MapF, RedF>{map, red, (uint32_t)(stripes < 1 ? 1 : stripes)};
}
template <typename F, typename = void>
struct is_map_reduce_t : std::false_type {};
template <typename F>
struct is_map_reduce_t<F, std::void_t<decltype(F::is_map_reduce)>> : std::true_type {};
// map_only(map, M): the MAP stage alone - the engine emits the M partial results of every window (ids M * k + m for the k-th
// window of a key, as the M MAP replicas of the host operator number them: wf/win_mapreduce.hpp), the REDUCE stage runs
// elsewhere (Win_MapReduce_GPU with a host REDUCE function)
struct NoReduce {};
template <typename MapF>
inline MapReduceFunc<MapF, NoReduce> map_only(MapF map, size_t stripes) {
    return MapReduceFunc<MapF, NoReduce>{map, NoReduce{}, (uint32_t)(stripes < 1 ? 1 : stripes)};
}
template <typename F>
struct is_map_only_t : std::false_type {};
template <typename MapF>
struct is_map_only_t<MapReduceFunc<MapF, NoReduce>> : std::true_type {};

// ---- parameters -----------------------------------------------------------------------------
struct WinGeometry {
    uint64_t win, slide, delay;
    uint32_t is_tb;
    uint32_t role;      // role_t
    uint32_t renumber;  // CB: ordinals are arrival indexes counted per key on the device (the ids of the tuples are ignored)
    uint32_t pad0;
    uint64_t id_outer, n_outer, slide_outer, id_inner, n_inner, slide_inner;  // WinOperatorConfig
    uint64_t map_first, map_stride;
    uint32_t carry_cap;  // tuples a key can carry from one batch to the next
    uint32_t out_cap;    // results per output batch
    uint32_t work_cap;   // tuples in the work buffer
    uint32_t max_keys;
};

struct WinAssignD {
    uint64_t first_gwid_key, initial_id, gwid_stride, a_inner;
};
// SURVEY.md A.4 (reference win_seq.hpp:328-336); `hashcode` is what std::hash gives on the host (primitives.hpp key_hash64:
// the identity for integers, FixedKey::hash() for fixed-size keys)
__host__ __device__ inline WinAssignD win_assign(const WinGeometry& g, uint64_t hashcode) {
    WinAssignD a;
    a.a_inner = (g.id_inner + g.n_inner - (hashcode % g.n_inner)) % g.n_inner;
    const uint64_t a_outer = (g.id_outer + g.n_outer - (hashcode % g.n_outer)) % g.n_outer;
    a.first_gwid_key = a.a_inner * g.n_outer + a_outer;
    const uint64_t initial_outer = a_outer * g.slide_outer, initial_inner = a.a_inner * g.slide_inner;
    a.initial_id = (g.role == (uint32_t)WLQ || g.role == (uint32_t)REDUCE) ? initial_inner : initial_outer + initial_inner;
    a.gwid_stride = g.n_outer * g.n_inner;
    return a;
}
// windows of a key whose end (+ triggering delay) has been reached by ordinal M: lwid < fired_upto
__host__ __device__ inline uint64_t win_fired_upto(const WinGeometry& g, uint64_t initial, uint64_t M) {
    const uint64_t first_end = initial + g.win + (g.is_tb ? g.delay : 0);
    return M >= first_end ? (M - first_end) / g.slide + 1 : 0;
}
// windows of a key opened by ordinal M (sliding, tumbling and hopping alike: floor((M - initial) / slide) + 1)
__host__ __device__ inline uint64_t win_opened_upto(const WinGeometry& g, uint64_t initial, uint64_t M) {
    return M >= initial ? (M - initial) / g.slide + 1 : 0;
}

struct WinKeyState {
    uint64_t next_lwid;     // next window of the key to fire (local id)
    uint64_t max_ord;       // greatest ordinal received (tuples and end-of-stream markers)
    uint64_t count;         // tuples received (the next arrival index when renumbering)
    uint64_t emit_counter;  // results emitted (PLQ / MAP renumbering of the result ids)
    uint32_t carry_len;     // tuples in the carry slab
    uint32_t flags;         // bit 0: something received, bit 1: dense id assigned
    uint32_t dense;         // index of the key's carry area
    uint32_t cur_run;       // run of the key in the batch being processed
};

struct WinRun {
    uint32_t slot, start, len, carry_len;
    uint64_t seg_off;     // first tuple of the segment, relative to the base pointer the kernels receive
    uint32_t seg_len;     // carry_len + len
    uint32_t unsorted;    // the segment is not ordered by ordinal: wb_sortfix_kernel repairs it
    uint64_t max_ord;     // greatest ordinal of the key so far
    uint64_t first_lwid;  // first window fired by this batch
    uint64_t keep_from;   // state before the batch: ordinals below belong to no open window
    uint64_t base_count;  // state before the batch: tuples received
    uint64_t initial;     // first ordinal of the key's first window on this replica
    uint32_t dense;
    uint32_t pad;
};

struct WinDesc {
    uint64_t gwid;
    uint64_t off;  // first tuple of the window, relative to the base pointer
    uint32_t len;
    uint32_t pad;
};

struct WinCounters {
    uint32_t dropped;         // tuples too old for every open window of their key
    uint32_t deferred;        // windows whose firing was postponed because the output batch was full (not an error)
    uint32_t table_full;      // tuples without a slot (table full, more than max_keys keys, reserved key)
    uint32_t carry_overflow;  // tuples an open window needed but the key's carry area could not hold
    uint32_t work_overflow;   // tuples that did not fit the work buffer
    uint32_t pad[3];
};

struct WinCtl {
    uint32_t n_runs, work_total, win_total, n_keys;
    uint32_t remaining;  // end-of-stream flush: keys of the range that still have windows to fire after this round
    uint32_t pad[3];
};

template <typename T>
__host__ __device__ __forceinline__ uint64_t win_ord(const T& t, uint32_t is_tb) {
    return is_tb ? cf_ts(t) : cf_id(t);
}

// first position in [0, n) of seg whose ordinal is >= x (seg ordered by ordinal)
template <typename T>
__device__ __forceinline__ uint32_t win_lower_bound(const T* seg, uint32_t n, uint64_t x, uint32_t is_tb) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (win_ord(seg[mid], is_tb) < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// ---- 2. run table ------------------------------------------------------------------------------
// Short runs that fire no window (many keys, a few tuples of each per batch, a window every so many batches) do not enter the
// table: their tuples are APPENDED to the key's carry area by the thread that found the run.  Only a batch that completes a
// window of the key pays for gathering the carried tuples into a contiguous segment (and for writing the tail back).
constexpr uint32_t WB_QUIET_RUN = 16;       // longest run appended in place
constexpr uint32_t WB_NO_RUN = 0xffffffffu;  // WinKeyState::cur_run of a key whose run was appended

template <typename T, typename K>
__global__ void __launch_bounds__(BLOCK_THREADS)
    wb_runs_kernel(const BatchHeader* hdr, const T* data, SortedSel sel, KeyTableView<K> table,
                   WinKeyState* states, WinRun* runs, uint32_t* work_len, T* carry, WinCtl* ctl, WinCounters* ctr, WinGeometry g, uint32_t quiet_ok,
                   uint32_t* unsorted_stamp, uint32_t batch_no, volatile uint32_t* disorder_seen) {
    const uint32_t n = hdr->count;
    // (a batch that holds no key twice was not sorted: every tuple is a run of its own, primitives.hpp DupProbe)
    const uint32_t* sslot = sel.keys();
    const uint32_t* sidx = sel.vals();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = sslot[i];
        if (i > 0 && sslot[i - 1] == slot) {  // not the head of a run
            // is the run in ordinal order?  (Runs that are get their tuples MERGED with the carried ones by the gather; the
            // others are placed in arrival order and repaired by wb_sortfix_kernel - and reported, so that the host orders
            // the following batches by (slot, timestamp) before they get here.)
            if (!g.renumber && slot != 0xffffffffu && win_ord(data[sidx[i - 1]], g.is_tb) > win_ord(data[sidx[i]], g.is_tb)) {
                unsorted_stamp[slot] = batch_no;
                if (disorder_seen != nullptr && i + 32 < n && sslot[i + 32] == slot) *disorder_seen = batch_no;  // (short runs are cheap to repair)
            }
            continue;
        }
        // end of the run: gallop, then bisect (sslot is sorted)
        uint32_t lo = i, step = 1;
        while (lo + step < n && sslot[lo + step] == slot) {
            lo += step;
            step <<= 1;
        }
        uint32_t a = lo + 1, b = lo + step < n ? lo + step : n;
        while (a < b) {
            const uint32_t m = (a + b) >> 1;
            if (sslot[m] == slot)
                a = m + 1;
            else
                b = m;
        }
        const uint32_t len = a - i;
        if (slot == 0xffffffffu) {
            atomicAdd(&ctr->table_full, len);
            continue;
        }
        WinKeyState st = states[slot];
        if (!(st.flags & 2u)) {
            st.dense = atomicAdd(&ctl->n_keys, 1u);
            st.flags |= 2u;
            states[slot].dense = st.dense;
            states[slot].flags = st.flags;
        }
        if (st.dense >= g.max_keys) {  // more distinct keys than carry areas
            atomicAdd(&ctr->table_full, len);
            continue;
        }
        const WinAssignD wa = win_assign(g, table.hash_at(slot));
        if (quiet_ok && len <= WB_QUIET_RUN && st.carry_len + len <= g.carry_cap && g.win >= g.slide) {
            // in order (also against what is carried), not late, and no window of the key ends with these tuples?
            T* area = carry + (size_t)st.dense * g.carry_cap;
            const uint64_t keep_from = wa.initial_id + st.next_lwid * g.slide;
            uint64_t prev = st.carry_len > 0 ? win_ord(area[st.carry_len - 1], g.is_tb) : ((st.next_lwid > 0) ? keep_from : 0);
            bool quiet = true;
            if (g.renumber) {
                prev = st.count + len - 1;
            } else {
                for (uint32_t k = 0; k < len && quiet; k++) {
                    const uint64_t o = win_ord(data[sidx[i + k]], g.is_tb);
                    quiet = o >= prev;
                    prev = o;
                }
            }
            const uint64_t max_ord = (st.flags & 1u) && st.max_ord > prev ? st.max_ord : prev;
            if (quiet && win_fired_upto(g, wa.initial_id, max_ord) <= st.next_lwid) {
                for (uint32_t k = 0; k < len; k++) {
                    T t = data[sidx[i + k]];
                    if (g.renumber) t.setControlFields(cf_key(t), st.count + k, cf_ts(t));
                    area[st.carry_len + k] = t;
                }
                WinKeyState* ks = &states[slot];
                ks->max_ord = max_ord;
                ks->count = st.count + len;
                ks->carry_len = st.carry_len + len;
                ks->flags = st.flags | 1u;
                ks->cur_run = WB_NO_RUN;
                continue;
            }
        }
        const uint32_t r = atomicAdd(&ctl->n_runs, 1u);
        states[slot].cur_run = r;
        WinRun run;
        run.slot = slot;
        run.start = i;
        run.len = len;
        run.carry_len = st.carry_len;
        run.seg_off = 0;
        run.seg_len = st.carry_len + len;
        run.unsorted = 0;
        run.max_ord = (st.flags & 1u) ? st.max_ord : 0;
        run.first_lwid = st.next_lwid;
        run.keep_from = wa.initial_id + st.next_lwid * g.slide;
        run.base_count = st.count;
        run.initial = wa.initial_id;
        run.dense = st.dense;
        run.pad = 0;
        runs[r] = run;
        work_len[r] = run.seg_len;
    }
}

// ---- 4. segments = carried tuples + run --------------------------------------------------------
// number of elements of the sorted sequence seq[0..n) (read through `at`) whose ordinal is < x (strict) / <= x
template <typename At>
__device__ __forceinline__ uint32_t wb_count_below(At at, uint32_t n, uint64_t x, bool or_equal) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint64_t o = at(mid);
        if (o < x || (or_equal && o == x))
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

constexpr uint32_t WB_GATHER_CHUNK = 1024;  // carried tuples copied by one wave

template <typename T>
__global__ void __launch_bounds__(BLOCK_THREADS)
    wb_gather_kernel(const BatchHeader* hdr, const T* data, SortedSel sel, const WinKeyState* states, WinRun* runs,
                     const uint32_t* work_off, const T* carry, T* work, const WinCtl* ctl, WinCounters* ctr, WinGeometry g,
                     const uint32_t* unsorted_stamp, uint32_t batch_no) {
    const uint32_t n = hdr->count;
    const uint32_t* sslot = sel.keys();
    const uint32_t* sidx = sel.vals();
    const uint32_t n_runs = ctl->n_runs;
    // A run that is in ordinal order (the common case; always after the (slot, timestamp) pre-sort) is MERGED with the carried
    // tuples of its key, which are in order too: the position of every tuple in the segment is its own index plus the number
    // of tuples of the other sequence that go before it (carried tuples first on ties) - one comparison when the run begins
    // after the last carried tuple, a binary search in the overlap otherwise.  No thread moves more than one tuple.
    // (a) the carried tuples, WB_GATHER_CHUNK of them per wave (a long triggering delay makes a key carry tens of thousands
    //     of tuples: one wave per run took 0.5 ms per batch for 16 keys x 20 000 carried tuples, profiles/r3_two_stage_hybrid)
    const uint32_t waves_total = gridDim.x * (BLOCK_THREADS / WAVE);
    const uint32_t chunks = (g.carry_cap + WB_GATHER_CHUNK - 1) / WB_GATHER_CHUNK;
    const uint64_t units = (uint64_t)n_runs * chunks;
    for (uint64_t u = blockIdx.x * (BLOCK_THREADS / WAVE) + wave_id(); u < units; u += waves_total) {
        const uint32_t r = (uint32_t)(u / chunks), c = (uint32_t)(u % chunks);
        const uint32_t off = work_off[r], clen = runs[r].carry_len, start = runs[r].start, len = runs[r].len;
        if (c == 0 && lane_id() == 0) runs[r].seg_off = off;
        const uint32_t k0 = c * WB_GATHER_CHUNK;
        if (k0 >= clen) continue;
        const uint32_t k1 = clen - k0 < WB_GATHER_CHUNK ? clen : k0 + WB_GATHER_CHUNK;
        const T* src = carry + (size_t)runs[r].dense * g.carry_cap;
        const bool merge = !g.renumber && len > 0 && unsorted_stamp[runs[r].slot] != batch_no;
        const uint64_t r_first = merge ? win_ord(data[sidx[start]], g.is_tb) : 0;
        for (uint32_t k = k0 + lane_id(); k < k1; k += WAVE) {
            uint32_t before = 0;
            if (merge) {
                const uint64_t o = win_ord(src[k], g.is_tb);
                if (o > r_first) before = wb_count_below([&](uint32_t m) { return win_ord(data[sidx[start + m]], g.is_tb); }, len, o, false);
            }
            if (off + k + before < g.work_cap) work[off + k + before] = src[k];
        }
    }
    // (b) the tuples of the batch, one thread per tuple
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t slot = sslot[j];
        if (slot == 0xffffffffu) continue;
        const WinKeyState* st = &states[slot];
        if (st->dense >= g.max_keys) continue;
        const uint32_t r = st->cur_run;
        if (r == WB_NO_RUN) continue;  // appended to the key's carry area by wb_runs_kernel
        const uint32_t start = runs[r].start, clen = runs[r].carry_len;
        T t = data[sidx[j]];
        uint64_t ord;
        uint32_t dst = work_off[r] + clen + (j - start);
        if (g.renumber) {
            ord = runs[r].base_count + (uint64_t)(j - start);
            t.setControlFields(cf_key(t), ord, cf_ts(t));
        } else {
            ord = win_ord(t, g.is_tb);
            const T* area = carry + (size_t)runs[r].dense * g.carry_cap;
            if (unsorted_stamp[slot] != batch_no) {
                // merge position: the carried tuples whose ordinal is <= mine go before me
                if (clen > 0) {
                    uint32_t before = clen;
                    if (ord < win_ord(area[clen - 1], g.is_tb)) before = wb_count_below([&](uint32_t m) { return win_ord(area[m], g.is_tb); }, clen, ord, true);
                    dst = work_off[r] + before + (j - start);
                }
            } else {
                // arrival order, repaired afterwards (wb_sortfix_kernel); the greatest ordinal can be anywhere in the run
                runs[r].unsorted = 1;
                atomicMax((unsigned long long*)&runs[r].max_ord, (unsigned long long)ord);
            }
            // late: older than every window still open (tuples in the gaps between hopping windows are not late, just unused)
            if (ord < runs[r].keep_from && runs[r].first_lwid > 0 &&
                (g.win >= g.slide || (ord >= runs[r].initial && (ord - runs[r].initial) % g.slide < g.win)))
                atomicAdd(&ctr->dropped, 1u);
        }
        if (j == start + runs[r].len - 1) atomicMax((unsigned long long*)&runs[r].max_ord, (unsigned long long)ord);
        if (dst >= g.work_cap) {
            atomicAdd(&ctr->work_overflow, 1u);
            continue;
        }
        work[dst] = t;
    }
}

// out-of-order arrivals (time-based windows with a triggering delay): restore the order of the segment.  Rare and
// short by assumption: one thread per affected run, stable insertion sort.
template <typename T>
__global__ void __launch_bounds__(BLOCK_THREADS) wb_sortfix_kernel(WinRun* runs, T* work, const WinCtl* ctl, WinGeometry g) {
    const uint32_t n_runs = ctl->n_runs;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_runs; r += gridDim.x * blockDim.x) {
        if (!runs[r].unsorted) continue;
        T* seg = work + runs[r].seg_off;
        const uint32_t len = runs[r].seg_off + runs[r].seg_len <= g.work_cap ? runs[r].seg_len : 0;
        for (uint32_t k = 1; k < len; k++) {
            const uint64_t o = win_ord(seg[k], g.is_tb);
            if (win_ord(seg[k - 1], g.is_tb) <= o) continue;
            const T t = seg[k];
            uint32_t q = k;
            while (q > 0 && win_ord(seg[q - 1], g.is_tb) > o) {
                seg[q] = seg[q - 1];
                q--;
            }
            seg[q] = t;
        }
    }
}

// ---- 5. windows fired by the batch ---------------------------------------------------------------
template <typename K>
__global__ void __launch_bounds__(BLOCK_THREADS)
    wb_plan_kernel(KeyTableView<K> table, const WinRun* runs, uint32_t* n_fire, const WinCtl* ctl, WinGeometry g) {
    const uint32_t n_runs = ctl->n_runs;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_runs; r += gridDim.x * blockDim.x) {
        const WinAssignD wa = win_assign(g, table.hash_at(runs[r].slot));
        const uint64_t upto = win_fired_upto(g, wa.initial_id, runs[r].max_ord);
        const uint64_t nf = upto > runs[r].first_lwid ? upto - runs[r].first_lwid : 0;
        // windows beyond the capacity of the output batch are deferred anyway; the clamp also keeps the 32-bit scan exact
        uint64_t clamp = (uint64_t)g.out_cap + 1;
        if (clamp > 0xffffffffull / n_runs) clamp = 0xffffffffull / n_runs;
        n_fire[r] = (uint32_t)(nf > clamp ? clamp : nf);
    }
}

// end of stream: every window opened so far fires, at most `quota` per key and round; the "runs" are the slots of
// [slot_lo, slot_hi) and their segments are the carry areas themselves
template <typename K>
__global__ void __launch_bounds__(BLOCK_THREADS)
    wb_flush_plan_kernel(KeyTableView<K> table, uint32_t slot_lo, uint32_t slot_hi, WinKeyState* states, WinRun* runs, uint32_t* n_fire,
                         WinCtl* ctl, WinGeometry g, uint32_t quota) {
    for (uint32_t slot = slot_lo + blockIdx.x * blockDim.x + threadIdx.x; slot < slot_hi; slot += gridDim.x * blockDim.x) {
        const uint32_t r = slot - slot_lo;
        WinRun run{};
        run.slot = slot;
        uint32_t nf32 = 0;
        if (table.occupied(slot) && (states[slot].flags & 1u) && states[slot].dense < g.max_keys) {
            const WinKeyState st = states[slot];
            const WinAssignD wa = win_assign(g, table.hash_at(slot));
            const uint64_t upto = win_opened_upto(g, wa.initial_id, st.max_ord);
            uint64_t nf = upto > st.next_lwid ? upto - st.next_lwid : 0;
            if (nf > quota) {
                nf = quota;
                atomicAdd(&ctl->remaining, 1u);
            }
            nf32 = (uint32_t)nf;
            run.seg_off = (uint64_t)st.dense * g.carry_cap;
            run.seg_len = st.carry_len;
            run.first_lwid = st.next_lwid;
            run.max_ord = st.max_ord;
            run.dense = st.dense;
            states[slot].next_lwid = st.next_lwid + nf;
            states[slot].emit_counter = st.emit_counter + nf;
        }
        runs[r] = run;
        n_fire[r] = nf32;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->n_runs = slot_hi - slot_lo;
}

// ---- 6. window descriptors -----------------------------------------------------------------------
// EMIT_BASE: where the emit counter of the key stood before these windows (states are only updated by the commit)
template <typename T, typename R, typename K>
__global__ void __launch_bounds__(BLOCK_THREADS)
    wb_desc_kernel(KeyTableView<K> table, const WinKeyState* states, const WinRun* runs, const uint32_t* win_off, const uint32_t* n_fire,
                   const T* base, WinDesc* desc, R* out, const WinCtl* ctl, WinGeometry g, uint32_t flushing) {
    const uint32_t n_runs = ctl->n_runs;
    uint32_t n_win = ctl->win_total;
    if (n_win > g.out_cap) n_win = g.out_cap;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_win; w += gridDim.x * blockDim.x) {
        // the run of window w: last r with win_off[r] <= w (runs that fire nothing share the offset of their successor)
        uint32_t lo = 0, hi = n_runs;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (win_off[mid] <= w)
                lo = mid + 1;
            else
                hi = mid;
        }
        const uint32_t r = lo - 1;
        const WinRun run = runs[r];
        const uint64_t k = w - win_off[r];
        const uint64_t lwid = run.first_lwid + k;
        const WinAssignD wa = win_assign(g, table.hash_at(run.slot));
        const uint64_t start = wa.initial_id + lwid * g.slide, end = start + g.win;
        const T* seg = base + run.seg_off;
        uint32_t p0, p1;
        const uint32_t sl = run.seg_len;
        const uint64_t first_ord = sl ? win_ord(seg[0], g.is_tb) : 0;
        if (sl && !g.is_tb && !run.unsorted && win_ord(seg[sl - 1], 0) - first_ord == (uint64_t)(sl - 1)) {
            // dense progressive ids: positions follow from the ordinals
            p0 = start <= first_ord ? 0u : (start - first_ord >= sl ? sl : (uint32_t)(start - first_ord));
            p1 = end <= first_ord ? 0u : (end - first_ord >= sl ? sl : (uint32_t)(end - first_ord));
        } else {
            p0 = win_lower_bound(seg, sl, start, g.is_tb);
            p1 = p0 + win_lower_bound(seg + p0, sl - p0, end, g.is_tb);
        }
        const uint64_t gwid = wa.first_gwid_key + lwid * wa.gwid_stride;
        WinDesc d;
        d.gwid = gwid;
        d.off = run.seg_off + p0;
        d.len = p1 - p0;
        d.pad = 0;
        desc[w] = d;
        // pre-stamped result (reference window.hpp:162-167, :202): count-based results carry the timestamp of the
        // last tuple of the window, time-based ones the last instant of the window
        uint64_t ts = g.is_tb ? end - 1 : (p1 > p0 ? cf_ts(seg[p1 - 1]) : 0);
        uint64_t id = gwid;
        // during a flush the state already holds the counter after these windows
        const uint64_t emitted = states[run.slot].emit_counter - (flushing ? (uint64_t)n_fire[r] : 0) + k;
        if (g.role == (uint32_t)MAP)
            id = g.map_first + emitted * g.map_stride;
        else if (g.role == (uint32_t)PLQ)
            id = wa.a_inner + emitted * g.n_inner;
        R res{};
        res.setControlFields((decltype(cf_key(res)))table.key_at(run.slot), id, ts);
        out[w] = res;
    }
}

// ---- 7. K1: evaluation ------------------------------------------------------------------------------
// Opaque functor: a window is a sequential loop of one lane.  How many lanes of a wave64 work depends on how many
// windows the batch fired: few windows (long windows, few keys) are spread one per wave over the whole chip - a lane
// alone in its wave has the wave's load bandwidth and cache to itself -, many windows fill the waves (reference:
// the "adaptive" worker size of gpu_map_stateful.cpp:610-618, re-derived for wave64 and evaluated on the device,
// where the number of windows is known).  Consecutive lanes take consecutive windows of a key: their ranges overlap
// by win - slide tuples, which the CU's vector cache serves.
template <typename T, typename R, typename F>
__global__ void __launch_bounds__(BLOCK_THREADS)
    wb_eval_kernel(const T* base, const WinDesc* desc, R* out, const WinCtl* ctl, uint32_t out_cap, char* scratch, size_t scratch_size, F f) {
    uint32_t n_win = ctl->win_total;
    if (n_win > out_cap) n_win = out_cap;
    const uint32_t waves_total = gridDim.x * (BLOCK_THREADS / WAVE);
    uint32_t active = 1;
    while (active < (uint32_t)WAVE && (uint64_t)active * waves_total < n_win) active <<= 1;
    const uint32_t lane = (uint32_t)lane_id();
    if (lane >= active) return;
    const uint32_t gw = blockIdx.x * (BLOCK_THREADS / WAVE) + wave_id();
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    char* my_scratch = scratch_size ? scratch + (size_t)tid * scratch_size : nullptr;
    for (uint64_t w = (uint64_t)gw * active + lane; w < n_win; w += (uint64_t)waves_total * active) {
        const WinDesc d = desc[w];
        f(d.gwid, base + d.off, (size_t)d.len, &out[w], my_scratch, scratch_size);
    }
}

// LDS-tiled form of the same evaluation, used whenever a window fits the tile: consecutive windows of a key overlap by
// win - slide tuples, so a workgroup takes `group` consecutive windows, stages the span they cover -
// (group - 1) * slide + win tuples - in LDS ONCE with all its threads (coalesced 8-byte loads), and `group` threads then
// run the user's function on the staged copy: every tuple is fetched once per group instead of once per window it belongs
// to, and the sequential loop of an opaque functor reads LDS.  (The reference's K1 reads every window from global memory,
// win_seq_gpu.hpp:56-80.)  A group whose span does not fit (windows of different keys far apart, the end-of-stream flush
// over scattered carry areas) falls back to global pointers.
//  * The tile is sized from the LDS the device offers (hip_utils.hpp max_lds_per_block: 160 KB on gfx950 - win 1024 / slide
//    128 of 40-byte tuples: 25 windows per workgroup, 5 with the 64 KB of older parts) and a workgroup is 16 waves.
//  * The function is called from two branches - staged window / window in global memory - so that the compiler knows the
//    address space of `data` in each: ds_read / global_load instructions, unrolled and issued ahead of their uses, instead
//    of one flat_load + full wait per tuple (measured: the dependent-load loop of an opaque sum was 127 cycles per tuple).
//  * Which thread takes which window: the windows of a group are `slide` tuples apart, so lanes of one wave that read the
//    same position of consecutive windows hit LDS banks slide * sizeof(T) / 4 words apart - the same bank for every lane
//    when that is a multiple of 64 (slide 128 x 40 B).  The contract (const T* data, contiguous) rules out a skewed layout,
//    so the host picks the thread -> window map with fewer lanes per busy bank: consecutive windows on consecutive lanes
//    (`by_wave` = 0), or dealt round-robin to the 16 waves (1: window k on wave k % 16, lane k / 16).
constexpr int WB_TILE_THREADS = 1024;
constexpr int WB_TILE_WAVES = WB_TILE_THREADS / WAVE;
constexpr unsigned WB_TILE_GRID = 1024;
constexpr size_t WB_TILE_STATIC_LDS = 0;  // the kernel keeps no static LDS (see s_lohi)
template <typename T, typename R, typename F>
__global__ void __launch_bounds__(WB_TILE_THREADS)
    wb_eval_tile_kernel(const T* base, const WinDesc* desc, R* out, const WinCtl* ctl, uint32_t out_cap, uint32_t group, uint32_t tile_tuples,
                        uint32_t by_wave, char* scratch, size_t scratch_size, F f) {
    extern __shared__ __align__(16) char wb_tile_smem[];
    // the bounds of the group's span are reduced in the first 16 bytes of the tile itself, before the span is staged over
    // them: the kernel has no static LDS, so a window of exactly the device's LDS size (4096 tuples of 40 bytes in 160 KB) fits
    unsigned long long* s_lohi = reinterpret_cast<unsigned long long*>(wb_tile_smem);
    // staged in the widest words the tuple type is aligned to
    typedef std::conditional_t<alignof(T) % 8 == 0, uint64_t, std::conditional_t<alignof(T) % 4 == 0, uint32_t, uint8_t>> raw_word_t;
    typedef raw_word_t __attribute__((may_alias)) word_t;
    T* tile = reinterpret_cast<T*>(wb_tile_smem);
    uint32_t n_win = ctl->win_total;
    if (n_win > out_cap) n_win = out_cap;
    const uint32_t n_groups = (n_win + group - 1) / group;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    char* my_scratch = scratch_size ? scratch + (size_t)tid * scratch_size : nullptr;
    // the window of a group this thread evaluates
    const uint32_t k = by_wave ? (uint32_t)lane_id() * WB_TILE_WAVES + (uint32_t)wave_id() : threadIdx.x;
    for (uint32_t gi = blockIdx.x; gi < n_groups; gi += gridDim.x) {
        const uint32_t w0 = gi * group;
        const uint32_t cnt = n_win - w0 < group ? n_win - w0 : group;
        if (threadIdx.x == 0) {
            s_lohi[0] = ~0ull;
            s_lohi[1] = 0ull;
        }
        __syncthreads();
        WinDesc d{};
        const bool mine = k < cnt;
        if (mine) {
            d = desc[w0 + k];
            atomicMin(&s_lohi[0], (unsigned long long)d.off);
            atomicMax(&s_lohi[1], (unsigned long long)(d.off + d.len));
        }
        __syncthreads();
        const uint64_t lo = s_lohi[0], hi = s_lohi[1];
        __syncthreads();  // everyone has the bounds: the tile may be written
        const bool fits = hi >= lo && hi - lo <= (uint64_t)tile_tuples;
        if (fits) {
            const word_t* src = reinterpret_cast<const word_t*>(base + lo);
            word_t* dst = reinterpret_cast<word_t*>(tile);
            const uint32_t words = (uint32_t)(hi - lo) * (uint32_t)(sizeof(T) / sizeof(raw_word_t));
            for (uint32_t x = threadIdx.x; x < words; x += WB_TILE_THREADS) dst[x] = src[x];
        }
        __syncthreads();
        if (mine) {
            if (fits)
                f(d.gwid, static_cast<const T*>(tile + (d.off - lo)), (size_t)d.len, &out[w0 + k], my_scratch, scratch_size);
            else
                f(d.gwid, base + d.off, (size_t)d.len, &out[w0 + k], my_scratch, scratch_size);
        }
        __syncthreads();
    }
}

// wave_reduce(lift, combine): one wave64 per window.  Every lane lifts and folds WR_PER_LANE consecutive tuples (the wave
// reads 64 * WR_PER_LANE consecutive tuples per round: coalesced), an order-preserving shuffle tree folds the 64 partial
// results, rounds are folded left to right: the combine only has to be associative.
constexpr uint32_t WR_PER_LANE = 4;
template <typename T, typename R, typename L, typename C>
__global__ void __launch_bounds__(BLOCK_THREADS)
    wb_eval_reduce_kernel(const T* base, const WinDesc* desc, R* out, const WinCtl* ctl, uint32_t out_cap, L lift, C comb) {
    uint32_t n_win = ctl->win_total;
    if (n_win > out_cap) n_win = out_cap;
    const uint32_t waves_total = gridDim.x * (BLOCK_THREADS / WAVE);
    const uint32_t lane = (uint32_t)lane_id();
    for (uint32_t w = blockIdx.x * (BLOCK_THREADS / WAVE) + wave_id(); w < n_win; w += waves_total) {
        const WinDesc d = desc[w];
        const T* win = base + d.off;
        R acc{};
        bool has_acc = false;  // identical in every lane
        for (uint32_t r0 = 0; r0 < d.len; r0 += WAVE * WR_PER_LANE) {
            R part{};
            bool has = false;
            const uint32_t first = r0 + lane * WR_PER_LANE;
#pragma unroll
            for (uint32_t q = 0; q < WR_PER_LANE; q++) {
                if (first + q < d.len) {
                    R v{};
                    v.setControlFields(cf_key(win[first + q]), 0, cf_ts(win[first + q]));
                    lift(win[first + q], v);
                    if (has) {
                        R tmp;
                        fat_combine(comb, part, v, tmp);
                        part = tmp;
                    } else {
                        part = v;
                        has = true;
                    }
                }
            }
            // ordered fold over the lanes: the lanes that hold a value are a prefix of the wave
            for (int s = 1; s < WAVE; s <<= 1) {
                int wds[sizeof(R) / 4];
                __builtin_memcpy(wds, &part, sizeof(R));
#pragma unroll
                for (size_t x = 0; x < sizeof(R) / 4; x++) wds[x] = __shfl_down(wds[x], s, WAVE);
                R other;
                __builtin_memcpy(&other, wds, sizeof(R));
                const bool other_has = __shfl_down((int)has, s, WAVE) != 0 && (int)lane + s < WAVE;
                if ((lane & (2 * s - 1)) == 0 && has && other_has) {
                    R tmp;
                    fat_combine(comb, part, other, tmp);
                    part = tmp;
                }
            }
            // lane 0 holds the round's aggregate: broadcast and fold
            int wds[sizeof(R) / 4];
            __builtin_memcpy(wds, &part, sizeof(R));
#pragma unroll
            for (size_t x = 0; x < sizeof(R) / 4; x++) wds[x] = __shfl(wds[x], 0, WAVE);
            R round;
            __builtin_memcpy(&round, wds, sizeof(R));
            if (has_acc) {
                R tmp;
                fat_combine(comb, acc, round, tmp);
                acc = tmp;
            } else {
                acc = round;
                has_acc = true;
            }
        }
        if (lane == 0) {
            R res = out[w];  // pre-stamped
            const auto key = cf_key(res);
            const uint64_t id = cf_id(res), ts = cf_ts(res);
            if (has_acc) res = acc;
            res.setControlFields(key, id, ts);
            out[w] = res;
        }
    }
}

// map + reduce evaluation (see MapReduceFunc): one workgroup per window.  `area`: per-workgroup scratch of `area_tuples` tuples;
// a window that does not fit is evaluated as a single stripe straight from the work buffer (any partition of the window is a
// legal one).  `partials`: per-workgroup array of `stripes` results.
template <typename T, typename R, typename MapF, typename RedF>
__global__ void __launch_bounds__(BLOCK_THREADS)
    wb_eval_mapreduce_kernel(const T* base, const WinDesc* desc, const R* stamps, R* out, const WinCtl* ctl, uint32_t out_cap, T* area, uint32_t area_tuples,
                             R* partials, uint32_t stripes, char* scratch, size_t scratch_size, MapF map, RedF red) {
    uint32_t n_win = ctl->win_total;
    if (n_win > out_cap) n_win = out_cap;
    T* my_area = area + (size_t)blockIdx.x * area_tuples;
    R* my_part = partials + (size_t)blockIdx.x * stripes;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    char* my_scratch = scratch_size ? scratch + (size_t)tid * scratch_size : nullptr;
    for (uint32_t w = blockIdx.x; w < n_win; w += gridDim.x) {
        const WinDesc d = desc[w];
        const T* src = base + d.off;
        const uint32_t len = d.len;
        const uint32_t M = len <= area_tuples ? stripes : 1u;
        const uint32_t q = len / M, rem = len % M;  // stripe m holds q tuples, one more when m < rem
        if (M > 1)
            for (uint32_t p = threadIdx.x; p < len; p += BLOCK_THREADS) {
                const uint32_t m = p % M;
                my_area[(size_t)m * q + (m < rem ? m : rem) + p / M] = src[p];
            }
        __syncthreads();
        const R stamped = stamps[w];  // (map + reduce: the pre-stamped result itself, stamps == out)
        if constexpr (std::is_same_v<RedF, NoReduce>) {
            // MAP stage alone: `stripes` partial results per window, whatever the partition (a stripe may be empty)
            if (threadIdx.x < stripes) {
                const uint32_t m = threadIdx.x;
                const uint64_t id = cf_id(stamped) + m;  // (role MAP: the window's stamp carries stripes * k)
                R part{};
                part.setControlFields(cf_key(stamped), id, cf_ts(stamped));
                const T* stripe = M > 1 ? my_area + (size_t)m * q + (m < rem ? m : rem) : src;
                const size_t cnt = m < M ? (size_t)(q + (m < rem ? 1u : 0u)) : 0;
                map(d.gwid, stripe, cnt, &part, my_scratch, scratch_size);
                part.setControlFields(cf_key(stamped), id, cf_ts(stamped));
                out[(size_t)w * stripes + m] = part;
            }
            __syncthreads();
            continue;
        }
        if (threadIdx.x < M) {
            const uint32_t m = threadIdx.x;
            R part{};
            part.setControlFields(cf_key(stamped), (uint64_t)m, cf_ts(stamped));
            const T* stripe = M > 1 ? my_area + (size_t)m * q + (m < rem ? m : rem) : src;
            map(d.gwid, stripe, (size_t)(q + (m < rem ? 1u : 0u)), &part, my_scratch, scratch_size);
            part.setControlFields(cf_key(stamped), (uint64_t)m, cf_ts(stamped));
            my_part[m] = part;
        }
        __syncthreads();
        if constexpr (!std::is_same_v<RedF, NoReduce>) {
            if (threadIdx.x == 0) {
                R res = stamped;
                red(d.gwid, static_cast<const R*>(my_part), (size_t)M, &res, my_scratch, scratch_size);
                res.setControlFields(cf_key(stamped), cf_id(stamped), cf_ts(stamped));
                out[w] = res;
            }
        }
        __syncthreads();
    }
}

// ---- 8. carry + state --------------------------------------------------------------------------------
// A run's carry (the tail of its segment that open windows still need) is copied by `chunks` waves, WB_COMMIT_CHUNK words
// each, so that a few long carries (few keys, long windows) spread over the whole chip; wave 0 of a run writes its state.
constexpr uint32_t WB_COMMIT_CHUNK = 512;  // 8-byte words per wave

template <typename T, typename K>
__global__ void __launch_bounds__(BLOCK_THREADS)
    wb_commit_kernel(KeyTableView<K> table, WinKeyState* states, const WinRun* runs, const uint32_t* n_fire, const uint32_t* win_off,
                     const T* work, T* carry, const WinCtl* ctl, WinCounters* ctr, WinGeometry g, uint32_t chunks) {
    using W = std::conditional_t<sizeof(T) % 8 == 0 && alignof(T) % 8 == 0, uint64_t,
                                 std::conditional_t<sizeof(T) % 4 == 0 && alignof(T) % 4 == 0, uint32_t, unsigned char>>;
    constexpr uint32_t WPT = sizeof(T) / sizeof(W);
    const uint32_t n_runs = ctl->n_runs;
    const uint32_t waves_total = gridDim.x * (BLOCK_THREADS / WAVE);
    const uint32_t lane = (uint32_t)lane_id();
    const uint64_t units = (uint64_t)n_runs * chunks;
    for (uint64_t u = blockIdx.x * (BLOCK_THREADS / WAVE) + wave_id(); u < units; u += waves_total) {
        const uint32_t r = (uint32_t)(u / chunks), c = (uint32_t)(u % chunks);
        const WinRun run = runs[r];
        const WinAssignD wa = win_assign(g, table.hash_at(run.slot));
        // windows that did not fit the output batch are not fired now: they stay open (their tuples are carried) and fire
        // with the next batch that holds the key, or at end of stream
        const uint32_t fired = win_off[r] >= g.out_cap ? 0u : (n_fire[r] < g.out_cap - win_off[r] ? n_fire[r] : g.out_cap - win_off[r]);
        const uint64_t next_lwid = run.first_lwid + fired;
        const uint64_t keep_from = wa.initial_id + next_lwid * g.slide;  // first ordinal of the next window to fire
        const bool seg_ok = run.seg_off + run.seg_len <= g.work_cap;
        const T* seg = work + run.seg_off;
        const uint32_t sl = seg_ok ? run.seg_len : 0;
        const uint32_t p0 = win_lower_bound(seg, sl, keep_from, g.is_tb);
        T* dst = carry + (size_t)run.dense * g.carry_cap;
        uint32_t kept = 0;
        if (g.win < g.slide) {
            // hopping windows: tuples in the gaps belong to no window; compacted by one wave
            if (c != 0) continue;
            for (uint32_t c0 = p0; c0 < sl; c0 += WAVE) {
                const uint32_t p = c0 + lane;
                bool keep = p < sl;
                if (keep) {
                    const uint64_t o = win_ord(seg[p], g.is_tb);
                    keep = o >= wa.initial_id && (o - wa.initial_id) % g.slide < g.win;
                }
                const unsigned long long m = __ballot(keep);
                const uint32_t pos = kept + (uint32_t)__popcll(m & lanemask_lt());
                if (keep) {
                    if (pos < g.carry_cap)
                        dst[pos] = seg[p];
                    else
                        atomicAdd(&ctr->carry_overflow, 1u);
                }
                kept += (uint32_t)__popcll(m);
            }
        } else {
            kept = sl - p0;
            const uint32_t fit = kept < g.carry_cap ? kept : g.carry_cap;
            const uint32_t words = fit * WPT;
            const uint32_t lo = c * WB_COMMIT_CHUNK, hi = lo + WB_COMMIT_CHUNK < words ? lo + WB_COMMIT_CHUNK : words;
            const W* __restrict__ src_w = reinterpret_cast<const W*>(seg + p0);
            W* __restrict__ dst_w = reinterpret_cast<W*>(dst);
            for (uint32_t w = lo + lane; w < hi; w += WAVE) dst_w[w] = src_w[w];
            if (c == 0 && lane == 0 && kept > g.carry_cap) atomicAdd(&ctr->carry_overflow, kept - g.carry_cap);
        }
        if (c == 0 && lane == 0) {
            WinKeyState* st = &states[run.slot];
            st->next_lwid = next_lwid;
            st->max_ord = run.max_ord;
            st->count = run.base_count + run.len;
            st->emit_counter += fired;
            st->carry_len = kept < g.carry_cap ? kept : g.carry_cap;
            st->flags |= 1u;
        }
    }
}

// end-of-stream markers: they only advance the state of their key
template <typename T, typename K>
__global__ void __launch_bounds__(BLOCK_THREADS)
    wb_markers_kernel(const BatchHeader* hdr, const T* data, KeyTableView<K> table, WinKeyState* states, WinCtl* ctl,
                      WinCounters* ctr, WinGeometry g) {
    const uint32_t n = hdr->count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = table.slot_of(cf_key(data[i]));
        if (slot == 0xffffffffu) {
            atomicAdd(&ctr->table_full, 1u);
            continue;
        }
        const uint32_t old = atomicOr(&states[slot].flags, 3u);
        if (!(old & 2u)) {
            const uint32_t dense = atomicAdd(&ctl->n_keys, 1u);
            states[slot].dense = dense;
            if (dense >= g.max_keys) atomicAdd(&ctr->table_full, 1u);
        }
        atomicMax((unsigned long long*)&states[slot].max_ord, (unsigned long long)win_ord(data[i], g.is_tb));
    }
}

// clamp the result count to the capacity (the windows beyond it were deferred, see wb_commit_kernel) and mirror it for the host
// (`per_window`: results per fired window - the MAP stage alone emits one per stripe)
__global__ void wb_finalize_kernel(BatchHeader* out_hdr, const WinCtl* ctl, uint32_t cap, volatile uint32_t* mirror, WinCounters* ctr,
                                   uint32_t per_window = 1) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t c = ctl->win_total;
        if (c > cap) {
            atomicAdd(&ctr->deferred, c - cap);
            c = cap;
        }
        c *= per_window;
        out_hdr->count = c;
        *mirror = c;
    }
}

struct WinGPUOptions {
    size_t max_keys = 1024;       // distinct keys per replica (each owns a carry area)
    size_t carry_capacity = 0;    // tuples a key may carry between batches (0: count-based = window length, time-based = 1024)
    size_t out_capacity = 0;      // results per output batch (0: twice the input batch capacity)
    bool renumber = false;        // count-based windows: count arrivals per key on the device instead of trusting the ids
    bool lanes_are_key_disjoint = false;  // batch input: lane i already carries exactly the keys of replica i
    bool expect_disorder = false;  // the input merges unsynchronised lanes (DEFAULT graph): order every batch by (key, timestamp) first
};

// The engine proper: owns the device state, turns device batches of tuples into device batches of results.
// Single stream, not thread safe.
template <typename T, typename R, typename F>
class WinBatchEngine {
    using key_t = key_of_t<T>;
    static_assert(is_device_key_v<key_t>, "GPU window operators: keys must be integers or fixed-size keys (wf::FixedKey; primitives.hpp is_device_key_v)");
    static_assert(std::is_same_v<key_t, key_of_t<R>>, "GPU window operators: tuples and results must have the same key type");
    static_assert(std::is_trivially_copyable_v<T> && std::is_trivially_copyable_v<R>, "GPU window operators move tuples and results as raw bytes");

public:
    WinBatchEngine(F func, const WinParams& p, const WinGPUOptions& opt, size_t batch_capacity, int device, size_t pool_slabs, size_t scratch_size)
        : func_(func), device_(device), scratch_size_(scratch_size) {
        if (p.win_len == 0 || p.slide_len == 0) failwith("GPU window operator: window length or slide cannot be zero");
        g_.win = p.win_len;
        g_.slide = p.slide_len;
        g_.delay = p.triggering_delay;
        g_.is_tb = p.winType == TB;
        g_.role = (uint32_t)p.role;
        g_.renumber = (opt.renumber && p.winType == CB) ? 1u : 0u;
        g_.pad0 = 0;
        g_.id_outer = p.config.id_outer;
        g_.n_outer = p.config.n_outer;
        g_.slide_outer = p.config.slide_outer;
        g_.id_inner = p.config.id_inner;
        g_.n_inner = p.config.n_inner;
        g_.slide_inner = p.config.slide_inner;
        g_.map_first = p.map_first;
        g_.map_stride = p.map_stride;
        batch_cap_ = std::max<size_t>(batch_capacity, 1);
        size_t carry = opt.carry_capacity;
        if (carry == 0) carry = g_.is_tb ? 1024 : (size_t)std::min<uint64_t>(g_.win, 1ull << 26);
        g_.carry_cap = (uint32_t)std::max<size_t>(carry, 1);
        g_.max_keys = (uint32_t)std::max<size_t>(opt.max_keys, 1);
        g_.out_cap = (uint32_t)(opt.out_capacity ? opt.out_capacity : std::max<size_t>(2 * batch_cap_, 8192));
        // worst case of the work buffer: every run of a batch brings a full carry area along
        const size_t runs_max = std::min<size_t>(batch_cap_, g_.max_keys);
        const size_t work = batch_cap_ + runs_max * (size_t)g_.carry_cap;
        if (work >= (1ull << 31)) failwith("GPU window operator: batch length x carry capacity too large for one work buffer; lower withBatch / withMaxKeys");
        g_.work_cap = (uint32_t)work;
        table_cap_ = 2;
        while (table_cap_ < 2 * (size_t)g_.max_keys) table_cap_ <<= 1;
        while ((1ull << slot_bits_) < table_cap_) slot_bits_++;
        WF_HIP_CHECK(hipSetDevice(device_));
        size_t free_b = 0, total_b = 0;
        WF_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        const size_t need = (size_t)g_.max_keys * g_.carry_cap * sizeof(T) + work * sizeof(T);
        if (need > free_b)
            failwith("GPU window operator: " + std::to_string(need >> 20) + " MiB of device memory needed for " + std::to_string(g_.max_keys) +
                     " keys x " + std::to_string(g_.carry_cap) + " carried tuples; lower withMaxKeys / withArchiveCapacity");
        Stream init;  // never the legacy stream
        table_.init(table_cap_, init.get());
        states_.resize(table_cap_);
        WF_HIP_CHECK(hipMemsetAsync(states_.data(), 0, states_.bytes(), init.get()));
        unsorted_stamp_.resize(table_cap_);
        WF_HIP_CHECK(hipMemsetAsync(unsorted_stamp_.data(), 0, unsorted_stamp_.bytes(), init.get()));
        disorder_.set_sticky(true);  // (the repair of an unsorted run is a sequential insertion sort: never go back to it)
        if (opt.expect_disorder) disorder_.expect_disorder();
        carry_.resize((size_t)g_.max_keys * g_.carry_cap);
        work_.resize(work);
        slot_.resize(batch_cap_);
        idx_.resize(batch_cap_);
        n_tab_ = std::max<size_t>(runs_max, std::min<size_t>(table_cap_, 4096));  // the end-of-stream flush walks the table in chunks of this size
        runs_.resize(n_tab_);
        work_len_.resize(n_tab_);
        work_off_.resize(n_tab_);
        n_fire_.resize(n_tab_);
        win_off_.resize(n_tab_);
        desc_.resize(g_.out_cap);
        scan_ws_.resize(scan_workspace_words(n_tab_));
        ctl_.resize(1);
        WF_HIP_CHECK(hipMemsetAsync(ctl_.data(), 0, sizeof(WinCtl), init.get()));
        counters_.resize(1);
        WF_HIP_CHECK(hipMemsetAsync(counters_.data(), 0, sizeof(WinCounters), init.get()));
        if (scratch_size_ > 0) scratch_.resize(std::max((size_t)EVAL_GRID * BLOCK_THREADS, (size_t)WB_TILE_GRID * WB_TILE_THREADS) * scratch_size_);
        if constexpr (is_map_reduce_t<F>::value) {
            if (func_.stripes > BLOCK_THREADS) failwith("Win_MapReduce_GPU: at most 256 map stripes per window");
            // scratch of a workgroup: a whole window (count-based), or what a key can hold between two batches plus a batch (time-based)
            size_t longest = g_.is_tb ? (size_t)g_.carry_cap + batch_cap_ : (size_t)std::min<uint64_t>(g_.win, (uint64_t)g_.carry_cap + batch_cap_);
            mr_grid_ = EVAL_GRID;
            while (mr_grid_ > 64 && (size_t)mr_grid_ * longest * sizeof(T) > (2ull << 30)) mr_grid_ /= 2;
            while ((size_t)mr_grid_ * longest * sizeof(T) > (2ull << 30) && longest > 1024) longest /= 2;  // (longer windows: one stripe, see the kernel)
            mr_area_tuples_ = (uint32_t)longest;
            mr_area_.resize((size_t)mr_grid_ * longest);
            mr_partials_.resize((size_t)mr_grid_ * func_.stripes);
        }
        // windows per workgroup of the LDS-tiled evaluation: as many consecutive windows as share one tile of the LDS the device
        // offers (0: a window alone does not fit, evaluate from global memory).  Time-based windows are sized as if a key ticked
        // once per time unit, bounded by what a key can carry; a group that turns out larger than the tile falls back to global
        // memory by itself.
        {
            const size_t lds = max_lds_per_block(device_);
            // (whatever static LDS the compiler gave the kernel after all is taken off the budget)
            size_t fixed = WB_TILE_STATIC_LDS;
            if constexpr (!is_map_reduce_t<F>::value && !is_wave_reduce_t<F>::value) {
                hipFuncAttributes fa{};
                if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&wb_eval_tile_kernel<T, R, F>)) == hipSuccess) fixed = fa.sharedSizeBytes;
            }
            const uint64_t tile_max = lds > fixed ? (lds - fixed) / sizeof(T) : 0;
            uint64_t w = g_.win, sl = g_.slide;
            if (g_.is_tb && w > g_.carry_cap) {
                sl = std::max<uint64_t>(1, sl * g_.carry_cap / w);
                w = g_.carry_cap;
            }
            if (w <= tile_max) {
                // As many windows per workgroup as the LDS holds - but no more than it takes to give every CU two workgroups of
                // the windows a full batch fires: the evaluation of a batch lasts as long as ONE workgroup needs (stage its
                // span, then the sequential loop of the functor), so few fat workgroups on a few CUs are slower than many
                // lean ones (measured, one key, win 1024 / slide 128: 25 windows x 41 workgroups 49 us, 5 x 205: 25 us).
                int cus = 0;
                if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_) != hipSuccess || cus <= 0) cus = 256;
                const uint64_t by_lds = (tile_max - w) / sl + 1;
                const uint64_t per_batch = std::max<uint64_t>(1, batch_cap_ / std::max<uint64_t>(1, g_.slide));  // windows a full batch fires (count-based)
                const uint64_t balanced = std::max<uint64_t>(4, per_batch / (2 * (uint64_t)cus));
                tile_group_ = (uint32_t)std::min<uint64_t>(WB_TILE_THREADS, std::min(by_lds, g_.is_tb ? by_lds : balanced));
                tile_cus_ = (uint32_t)cus;
                tile_lds_total_ = lds;
                // count-based windows: exactly the span of the group; time-based ones: everything there is (the estimate is loose)
                tile_tuples_ = (uint32_t)(g_.is_tb ? tile_max : std::min<uint64_t>(tile_max, (uint64_t)(tile_group_ - 1) * sl + w));
                // lanes per busy LDS bank under the two thread -> window maps (see the kernel)
                auto lanes_per_bank = [&](uint64_t lane_stride_windows, uint32_t active) {
                    const uint64_t words = (sl * lane_stride_windows * sizeof(T) / 4) % 64;
                    const uint32_t banks = words == 0 ? 1u : (uint32_t)(64 / std::gcd<uint64_t>(words, 64));
                    return (double)active / (double)std::min(active, banks);
                };
                const uint32_t act_a = std::min<uint32_t>(WAVE, tile_group_), act_b = (tile_group_ + WB_TILE_WAVES - 1) / WB_TILE_WAVES;
                tile_by_wave_ = lanes_per_bank(WB_TILE_WAVES, act_b) < lanes_per_bank(1, act_a) ? 1u : 0u;
            }
            if (const char* e = std::getenv("WF_WIN_TILE")) {
                if (std::atoi(e) == 0) tile_group_ = 0;
            }
            if (const char* e = std::getenv("WF_WIN_QUIET_RUNS")) quiet_runs_ = std::atoi(e) != 0;
            if constexpr (!is_map_reduce_t<F>::value && !is_wave_reduce_t<F>::value) {
                if (tile_group_ > 0) allow_dynamic_lds(&wb_eval_tile_kernel<T, R, F>, (size_t)tile_tuples_ * sizeof(T));
            }
        }
        dups_.reserve(table_cap_, init.get());
        init.sync();
        sorter_.reserve(batch_cap_);
        if constexpr (is_map_only_t<F>::value) {
            per_window_ = func_.stripes;
            if ((uint64_t)g_.out_cap * per_window_ >= (1ull << 31)) failwith("Win_MapReduce_GPU: output capacity x map parallelism too large for one batch");
            stamps_.resize(g_.out_cap);
        }
        out_pool_.reset(new BatchPool(MemKind::DEVICE, device_, sizeof(R), (size_t)g_.out_cap * per_window_, pool_slabs));
    }
    WinBatchEngine(const WinBatchEngine&) = delete;
    WinBatchEngine& operator=(const WinBatchEngine&) = delete;

    const WinGeometry& geometry() const { return g_; }
    uint64_t presorted_batches() const { return presorted_batches_; }
    size_t device_bytes() const { return table_.bytes() + states_.bytes() + carry_.bytes() + work_.bytes(); }

    // enqueue the processing of `in` on `s`; returns the batch of fired windows (count on the device)
    Batch* process(Batch* in, hipStream_t s, uint64_t& kernels) {
        const size_t upper = in->count_on_host ? in->host_hdr.count : in->capacity;
        if (upper > batch_cap_) failwith("GPU window operator: input batch larger than the configured batch capacity");
        Batch* out = out_pool_->acquire(s);
        out->host_hdr = in->host_hdr;
        out->count_on_host = false;
        if (upper == 0) {
            WF_HIP_CHECK(hipMemsetAsync(out->header(), 0, sizeof(BatchHeader), s));
            WF_HIP_CHECK(hipMemsetAsync(&ctl_.data()->win_total, 0, sizeof(uint32_t), s));
            hipLaunchKernelGGL(wb_finalize_kernel, dim3(1), dim3(64), 0, s, out->header(), ctl_.data(), g_.out_cap, out->count_mirror, counters_.data(), per_window_);
            kernels += 1;
            return out;
        }
        const unsigned grid = grid_for(upper);
        const size_t runs_upper = std::min<size_t>(upper, n_tab_);
        const unsigned rgrid = grid_for(runs_upper);
        // ordinals that are not assigned here (timestamps, user ids) may arrive out of order: once the device has reported runs
        // that are, or when the caller knows its input is a merge of unsynchronised lanes, the batch is ordered by (slot,
        // timestamp) instead of (slot, arrival)
        batch_no_++;
        const bool presort = g_.is_tb && !g_.renumber && disorder_.decide(batch_no_);
        // a batch that holds no key twice skips the sort (decided on the device: primitives.hpp DupProbe)
        const DupProbe probe = presort ? DupProbe{} : dups_.probe(batch_no_);
        const SortGate gate = dups_.gate(probe);
        // also clears the header of the output batch and the run counter
        hipLaunchKernelGGL((assign_slots_kernel<T, key_t>), dim3(grid), dim3(BLOCK_THREADS), 0, s, in->header(), in->template data<T>(), table_.view(),
                           slot_.data(), idx_.data(), &counters_.data()->table_full, out->header(), &ctl_.data()->n_runs, probe);
        const auto sorted = presort ? presorter_.sort(in->header(), in->template data<T>(), in->device_count_ptr(), upper, slot_.data(), slot_bits_ + 1, 32,
                                                     1, disorder_.clamped(), batch_no_, s)
                                    : sorter_.sort(slot_.data(), idx_.data(), in->device_count_ptr(), upper, slot_bits_ + 1, s, gate);
        const SortedSel sel{sorted.keys, sorted.vals, slot_.data(), idx_.data(), gate};
        if (presort) presorted_batches_++;
        hipLaunchKernelGGL((wb_runs_kernel<T>), dim3(grid), dim3(BLOCK_THREADS), 0, s, in->header(), in->template data<T>(), sel,
                           table_.view(), states_.data(), runs_.data(), work_len_.data(), carry_.data(), ctl_.data(), counters_.data(), g_,
                           quiet_runs_ ? 1u : 0u, unsorted_stamp_.data(), batch_no_, disorder_.seen());
        exclusive_scan_u32(work_len_.data(), work_off_.data(), runs_upper, &ctl_.data()->work_total, scan_ws_.data(), s, &ctl_.data()->n_runs);
        hipLaunchKernelGGL((wb_gather_kernel<T>), dim3(grid), dim3(BLOCK_THREADS), 0, s, in->header(), in->template data<T>(), sel,
                           states_.data(), runs_.data(), work_off_.data(), carry_.data(), work_.data(), ctl_.data(), counters_.data(), g_,
                           unsorted_stamp_.data(), batch_no_);
        if (!g_.renumber)
            hipLaunchKernelGGL((wb_sortfix_kernel<T>), dim3(rgrid), dim3(BLOCK_THREADS), 0, s, runs_.data(), work_.data(), ctl_.data(), g_);
        hipLaunchKernelGGL((wb_plan_kernel<key_t>), dim3(rgrid), dim3(BLOCK_THREADS), 0, s, table_.view(), runs_.data(), n_fire_.data(), ctl_.data(), g_);
        exclusive_scan_u32(n_fire_.data(), win_off_.data(), runs_upper, &ctl_.data()->win_total, scan_ws_.data(), s, &ctl_.data()->n_runs);
        evaluate(work_.data(), out, s, 0);
        {
            const size_t wpt = (sizeof(T) + 7) / 8;
            const uint32_t chunks = (uint32_t)std::max<size_t>(1, (g_.carry_cap * wpt + WB_COMMIT_CHUNK - 1) / WB_COMMIT_CHUNK);
            const unsigned cgrid = (unsigned)std::min<size_t>(4096, std::max<size_t>(1, (runs_upper * chunks + BLOCK_THREADS / WAVE - 1) / (BLOCK_THREADS / WAVE)));
            hipLaunchKernelGGL((wb_commit_kernel<T>), dim3(cgrid), dim3(BLOCK_THREADS), 0, s, table_.view(), states_.data(), runs_.data(), n_fire_.data(),
                               win_off_.data(), work_.data(), carry_.data(), ctl_.data(), counters_.data(), g_, chunks);
        }
        hipLaunchKernelGGL(wb_finalize_kernel, dim3(1), dim3(64), 0, s, out->header(), ctl_.data(), g_.out_cap, out->count_mirror, counters_.data(), per_window_);
        kernels += 10 + sorter_.kernels_per_sort(slot_bits_ + 1);
        return out;
    }

    // a batch of end-of-stream markers: they only advance the state of their keys
    void apply_markers(Batch* in, hipStream_t s, uint64_t& kernels) {
        const size_t upper = in->count_on_host ? in->host_hdr.count : in->capacity;
        if (upper == 0) return;
        hipLaunchKernelGGL((wb_markers_kernel<T>), dim3(grid_for(upper)), dim3(BLOCK_THREADS), 0, s, in->header(), in->template data<T>(), table_.view(),
                           states_.data(), ctl_.data(), counters_.data(), g_);
        kernels += 1;
    }

    // end of stream: every window still open is computed on its carried tuples and emitted.  One call = one output batch:
    // a range of table slots, at most out_cap / range windows of each; a range is revisited until its keys have nothing left
    // (a key can owe any number of windows: a triggering delay longer than the stream holds all of them back).
    bool flush_range(size_t& cursor, hipStream_t s, uint64_t& kernels, Batch*& out) {
        if (cursor >= table_cap_) return false;
        const size_t slots_per_round = std::min<size_t>(n_tab_, g_.out_cap);
        const size_t lo = cursor, hi = std::min(table_cap_, lo + slots_per_round);
        const uint32_t quota = (uint32_t)std::max<size_t>(1, g_.out_cap / (hi - lo));
        out = out_pool_->acquire(s);
        out->host_hdr = BatchHeader{};
        out->count_on_host = false;
        WF_HIP_CHECK(hipMemsetAsync(out->header(), 0, sizeof(BatchHeader), s));
        WF_HIP_CHECK(hipMemsetAsync(&ctl_.data()->remaining, 0, sizeof(uint32_t), s));
        const unsigned rgrid = grid_for(hi - lo);
        hipLaunchKernelGGL((wb_flush_plan_kernel<key_t>), dim3(rgrid), dim3(BLOCK_THREADS), 0, s, table_.view(), (uint32_t)lo, (uint32_t)hi, states_.data(),
                           runs_.data(), n_fire_.data(), ctl_.data(), g_, quota);
        exclusive_scan_u32(n_fire_.data(), win_off_.data(), hi - lo, &ctl_.data()->win_total, scan_ws_.data(), s);
        evaluate(carry_.data(), out, s, 1);
        hipLaunchKernelGGL(wb_finalize_kernel, dim3(1), dim3(64), 0, s, out->header(), ctl_.data(), g_.out_cap, out->count_mirror, counters_.data(), per_window_);
        kernels += 6;
        WinCtl c{};
        WF_HIP_CHECK(hipMemcpyAsync(&c, ctl_.data(), sizeof(c), hipMemcpyDeviceToHost, s));
        WF_HIP_CHECK(hipStreamSynchronize(s));
        if (c.remaining == 0) cursor = hi;
        return true;
    }

    WinCounters read_counters(hipStream_t s) {
        WinCounters c{};
        WF_HIP_CHECK(hipMemcpyAsync(&c, counters_.data(), sizeof(c), hipMemcpyDeviceToHost, s));
        WF_HIP_CHECK(hipStreamSynchronize(s));
        return c;
    }

private:
    static constexpr unsigned EVAL_GRID = 1024;  // 4096 waves: 4 per SIMD on 256 CUs

    void evaluate(const T* base, Batch* out, hipStream_t s, uint32_t flushing) {
        // at most out_cap windows are described and evaluated; the grid-stride loops read the exact count on the device
        // (the MAP stage alone writes several results per window: the windows' stamps then live in an array of their own)
        R* stamped = is_map_only_t<F>::value ? stamps_.data() : out->template data<R>();
        hipLaunchKernelGGL((wb_desc_kernel<T, R>), dim3(grid_for(g_.out_cap)), dim3(BLOCK_THREADS), 0, s, table_.view(), states_.data(), runs_.data(),
                           win_off_.data(), n_fire_.data(), base, desc_.data(), stamped, ctl_.data(), g_, flushing);
        if constexpr (is_map_reduce_t<F>::value) {
            hipLaunchKernelGGL((wb_eval_mapreduce_kernel<T, R, decltype(func_.map), decltype(func_.red)>), dim3(mr_grid_), dim3(BLOCK_THREADS), 0, s, base,
                               desc_.data(), (const R*)stamped, out->template data<R>(), ctl_.data(), g_.out_cap, mr_area_.data(), mr_area_tuples_,
                               mr_partials_.data(), func_.stripes, scratch_.data(), scratch_size_, func_.map, func_.red);
        } else if constexpr (is_wave_reduce_t<F>::value) {
            hipLaunchKernelGGL((wb_eval_reduce_kernel<T, R, decltype(func_.lift), decltype(func_.comb)>), dim3(EVAL_GRID), dim3(BLOCK_THREADS), 0, s, base,
                               desc_.data(), out->template data<R>(), ctl_.data(), g_.out_cap, func_.lift, func_.comb);
        } else if (tile_group_ > 0) {
            // the workgroups loop over the groups of windows: no more of them than the chip holds at once (those without work
            // leave at once, but each would still wait for its share of LDS)
            const size_t tile_bytes = std::max<size_t>(16, (size_t)tile_tuples_ * sizeof(T));
            const unsigned resident = tile_cus_ * (unsigned)std::max<size_t>(1, tile_lds_total_ / (tile_bytes + WB_TILE_STATIC_LDS));
            const unsigned groups = (unsigned)((g_.out_cap + tile_group_ - 1) / tile_group_);
            hipLaunchKernelGGL((wb_eval_tile_kernel<T, R, F>), dim3(std::min(std::min(groups, WB_TILE_GRID), resident)), dim3(WB_TILE_THREADS), tile_bytes,
                               s, base, desc_.data(), out->template data<R>(), ctl_.data(), g_.out_cap, tile_group_, tile_tuples_, tile_by_wave_,
                               scratch_.data(), scratch_size_, func_);
        } else {
            hipLaunchKernelGGL((wb_eval_kernel<T, R, F>), dim3(EVAL_GRID), dim3(BLOCK_THREADS), 0, s, base, desc_.data(), out->template data<R>(),
                               ctl_.data(), g_.out_cap, scratch_.data(), scratch_size_, func_);
        }
    }

    F func_;
    int device_;
    size_t scratch_size_;
    WinGeometry g_{};
    size_t table_cap_ = 0, batch_cap_ = 0, n_tab_ = 0;
    int slot_bits_ = 0;
    uint32_t tile_group_ = 0, tile_tuples_ = 0, tile_by_wave_ = 0, tile_cus_ = 256;
    size_t tile_lds_total_ = 64 * 1024;
    bool quiet_runs_ = true;  // short runs that fire no window are appended to the carry areas in place (wb_runs_kernel)
    KeyTable<key_t> table_;
    DeviceBuffer<WinKeyState> states_;
    DeviceBuffer<T> carry_, work_;
    DeviceBuffer<uint32_t> slot_, idx_, work_len_, work_off_, n_fire_, win_off_, unsorted_stamp_;
    TsPresorter presorter_;
    DisorderMonitor disorder_;
    DupMonitor dups_;
    uint32_t batch_no_ = 0;
    uint64_t presorted_batches_ = 0;
    DeviceBuffer<WinRun> runs_;
    DeviceBuffer<WinDesc> desc_;
    DeviceBuffer<unsigned long long> scan_ws_;
    DeviceBuffer<WinCtl> ctl_;
    DeviceBuffer<WinCounters> counters_;
    DeviceBuffer<char> scratch_;
    DeviceBuffer<T> mr_area_;      // map + reduce evaluation: per-workgroup window laid out stripe by stripe
    DeviceBuffer<R> mr_partials_;
    DeviceBuffer<R> stamps_;       // MAP stage alone: the pre-stamped result of every fired window
    uint32_t per_window_ = 1;      // results per fired window
    unsigned mr_grid_ = 0;
    uint32_t mr_area_tuples_ = 0;
    RadixSorter sorter_;
    std::unique_ptr<BatchPool> out_pool_;
};

}  // namespace gpu

// ---- replica ----------------------------------------------------------------------------------------
constexpr uint64_t WF_MARKER_BATCH = 0x4d41524b45525321ull;  // BatchHeader::reserved[1] of a batch of end-of-stream markers

template <typename tuple_t, typename result_t, typename F>
class WinBatchGPU_Node : public GPU_Replica<tuple_t> {
public:
    WinBatchGPU_Node(F func, WinParams p, gpu::WinGPUOptions opt, GPUConfig cfg, edge_kind_t in_kind, std::string name, size_t par, size_t idx,
                     meta::closing_func_t closing, std::shared_ptr<StatsHub> hub, std::shared_ptr<std::vector<gpu::WinCounters>> counters = nullptr)
        : GPU_Replica<tuple_t>(cfg, in_kind, std::move(name), par, idx, std::move(closing), std::move(hub), true), func_(func), p_(p), opt_(opt),
          counters_(std::move(counters)) {
        if (p_.win_len == 0 || p_.slide_len == 0) fatal_error("window length or slide cannot be zero");
        if (this->cfg_.batch_len == 0) fatal_error("batch length cannot be zero");
    }

    void svc(void* msg, int ch) override {
        if (this->in_kind_ == EDGE_TUPLE && is_marker(msg)) {
            tuple_t* t = untag<tuple_t>(msg);
            markers_.push_back(*t);
            tuple_delete(t);
            return;
        }
        GPU_Replica<tuple_t>::svc(msg, ch);
    }

protected:
    void process_device(gpu::Batch* d) override {
        ensure_engine(d->capacity);
        if (d->host_hdr.reserved[1] == WF_MARKER_BATCH) {
            engine_->apply_markers(d, this->stream(), this->num_kernels_);
            this->release_input(d);
            return;
        }
        gpu::Batch* out = engine_->process(d, this->stream(), this->num_kernels_);
        this->release_input(d);
        this->emit_device(out);
        if (++since_check_ >= 256) {
            since_check_ = 0;
            check_counters(engine_->read_counters(this->stream()));
        }
    }
    void gpu_flush() override {
        if (!markers_.empty()) {
            // the markers travel like a batch of tuples, flagged in the header
            ensure_engine(std::max<size_t>(this->cfg_.batch_len, 1));
            hipStream_t s = this->stream();
            const size_t cap = std::max<size_t>(this->cfg_.batch_len, 1);
            if (!marker_pool_) {
                marker_pool_.reset(new gpu::BatchPool(gpu::MemKind::PINNED_HOST, this->cfg_.gpu_id, sizeof(tuple_t), cap, 2));
                marker_dev_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, this->cfg_.gpu_id, sizeof(tuple_t), cap, 2));
            }
            for (size_t off = 0; off < markers_.size(); off += cap) {
                const size_t n = std::min(cap, markers_.size() - off);
                gpu::Batch* h = marker_pool_->acquire_host();
                h->host_hdr.count = (uint32_t)n;
                h->host_hdr.reserved[1] = WF_MARKER_BATCH;
                std::memcpy(h->template data<tuple_t>(), markers_.data() + off, n * sizeof(tuple_t));
                *h->header() = h->host_hdr;
                gpu::Batch* d = marker_dev_->acquire(s);
                WF_HIP_CHECK(hipMemcpyAsync(d->slab, h->slab, h->bytes_used(), hipMemcpyHostToDevice, s));
                d->host_hdr = h->host_hdr;
                d->count_on_host = true;
                h->release(s);
                engine_->apply_markers(d, s, this->num_kernels_);
                d->release(s);
            }
            markers_.clear();
        }
        if (!engine_) return;
        size_t cursor = 0;
        gpu::Batch* out = nullptr;
        while (engine_->flush_range(cursor, this->stream(), this->num_kernels_, out)) this->emit_device(out);
        const gpu::WinCounters c = engine_->read_counters(this->stream());
        if (counters_) (*counters_)[this->context.getReplicaIndex()] = c;
        dropped_ = c.dropped;
        check_counters(c);
    }
    void finalize_stats(Stats_Record& s) override {
        GPU_Replica<tuple_t>::finalize_stats(s);
        s.inputs_ignored = dropped_;
    }

private:
    void ensure_engine(size_t batch_capacity) {
        if (engine_) return;
        gpu::WinGPUOptions o = opt_;
        o.out_capacity = this->scaled_out_capacity(opt_.out_capacity, batch_capacity);  // (coalesced input batches: GPU_Replica)
        engine_.reset(new gpu::WinBatchEngine<tuple_t, result_t, F>(func_, p_, o, batch_capacity, this->cfg_.gpu_id, this->cfg_.pool_slabs,
                                                                   this->cfg_.scratchpad_size));
    }
    void check_counters(const gpu::WinCounters& c) {
        // losing tuples or results silently is not an option: fail loudly and say which knob to turn
        if (c.table_full > 0)
            fatal_error("GPU window operator: tuples without a state slot - more keys than withMaxKeys(), or a tuple carries the reserved key 2^64-1");
        if (c.carry_overflow > 0)
            fatal_error("GPU window operator: " + std::to_string(c.carry_overflow) +
                        " tuples needed by open windows did not fit their key's archive; raise withArchiveCapacity()");
        if (c.work_overflow > 0) fatal_error("GPU window operator: work buffer overflow (internal sizing error)");
    }
    F func_;
    WinParams p_;
    gpu::WinGPUOptions opt_;
    std::shared_ptr<std::vector<gpu::WinCounters>> counters_;
    std::unique_ptr<gpu::WinBatchEngine<tuple_t, result_t, F>> engine_;
    std::vector<tuple_t> markers_;
    std::unique_ptr<gpu::BatchPool> marker_pool_, marker_dev_;
    uint64_t dropped_ = 0;
    uint32_t since_check_ = 0;
};

}  // namespace wf
