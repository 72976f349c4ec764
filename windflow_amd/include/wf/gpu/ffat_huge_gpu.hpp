// Huge runs of the keyed FlatFAT engine: a key that owns thousands of tuples of a batch (a hot key of a skewed stream, or a
// stream with a handful of keys) is spread over the whole chip instead of being folded by one wave.
//
// The per-key rules of key_ffat_gpu.hpp are sequential in the number of panes, but the combine is associative, so the run
// is cut in chunks of 64 tuples that are reduced independently:
//   1. ffat_huge_setup_kernel    one thread per huge run: its length (binary search in the sorted slots), where its chunks,
//                                partial aggregates and pane leaves live in the scratch buffers, the state it starts from
//   2. ffat_huge_chunks_kernel   one wave per chunk, all chunks of all runs at once: gather + lift + order-preserving segmented
//                                wave scan by pane.  Panes that begin and end inside the chunk are final: they are written as
//                                LEAVES of the run (a flat array of pane aggregates, the run's FlatFAT leaf level); the first
//                                and the last segment of a chunk may continue in its neighbours: they are left as PARTIALS
//   3. ffat_huge_stitch_kernel   one wave per 64 partials (two per chunk, in run order): the same segmented scan joins the
//                                pieces of a pane that straddles chunks; panes that end inside the 64 are final (leaves),
//                                the first and the last are again partials, of the next level (64 times fewer)
//   4. ffat_huge_fix_kernel      one workgroup per run: stitches the second-level partials with what was open before the run,
//                                fills the panes no tuple fell into (time-based windows), appends the panes to the key's
//                                tree in HBM, advances the key's state and reserves room in the output batch for the
//                                windows the run fires
//   5. ffat_huge_hgw_kernel      windows of more than FFAT_HUGE_DIRECT_WIN panes only: the leaves are cut in blocks of W = panes per
//                                window, and one wave per block computes the block's inclusive prefix folds, another its suffix
//                                folds (van Herk / Gil-Werman): a window that starts at leaf s is suffix[s] (+) prefix[s + W - 1],
//                                one combine whatever W is, and the order of the panes is kept (the combine need not commute)
//   6. ffat_huge_windows_kernel  one thread per fired window: that one combine, or for short windows a fold over the W leaves
//                                (old panes of the key in front, new ones behind)
// Used for count-based windows and for time-based windows whose run is in timestamp order, not late and not sparse in time
// (with a triggering delay the newest quanta of the run stay open in the key's ring instead of becoming leaves); everything
// else goes to the wave-per-run kernel, which implements the full rules.  The engine switches these five launches on when a batch shows a huge run and off again after
// a while without one (key_ffat_gpu.hpp), so streams without hot keys do not pay for them.
//
// Reference: the reference engine (wf/win_seqffat_gpu.hpp:369-396 + wf/flatfat_gpu.hpp:251-434) handles ONE key per replica
// and rebuilds its tree level by level, one launch per level; it has no notion of a batch that mixes hot and cold keys.
// (included by key_ffat_gpu.hpp, after the per-key rules it builds on)
#pragma once
#include "flatfat_gpu.hpp"
#include "primitives.hpp"

namespace wf {
namespace gpu {

constexpr uint32_t FFAT_HUGE_RUN = 2048;    // tuples of one key in one batch from which the run is spread over the chip
constexpr uint32_t FFAT_HUGE_DIRECT_WIN = 12;  // panes per window up to which a fired window is folded directly from the leaves

template <typename R>
struct FFATPartial {
    R val;
    uint32_t pane;   // relative to the first pane the run can complete
    uint32_t valid;  // 1: a value; 2: a place holder without value (the unit has a single segment: `pane` is that segment's pane)
};

struct FFATHugeCtl {
    uint32_t n_runs, total_chunks, total_units, pad1;
};

struct FFATHugeRun {
    uint32_t head, len, slot, n_chunks;
    uint32_t n_units, unit_base;  // second level: a unit stitches 64 first-level partials (32 chunks)
    uint32_t chunk_base;  // first chunk of the run in the batch-wide chunk numbering
    uint32_t leaf_base;   // first leaf of the run in the scratch: win_panes - 1 old panes, then the panes of the run
    uint32_t max_panes, pc0;
    uint64_t P0, Lq, first_lwid;
    uint64_t Lq_new;  // time-based: oldest quantum that stays open after the run (the quanta below it are completed by it)
    uint32_t m, nf, out_base, fallback;
};

template <typename T, typename R>
__global__ void __launch_bounds__(BLOCK_THREADS)
    ffat_huge_setup_kernel(const BatchHeader* hdr, const T* data, const uint32_t* sslot, const uint32_t* sidx, FFATRecords<R> recs,
                           FFATGeometry g, const uint32_t* huge_heads, const uint32_t* n_huge, FFATHugeRun* runs, FFATHugeCtl* ctl, uint32_t max_runs,
                           uint32_t leaves_cap) {
    const uint64_t pane_len = g.pane_len;
    const uint32_t is_tb = g.is_tb, win_panes = g.win_panes;
    const uint32_t n = hdr->count;
    uint32_t nr = *n_huge;
    if (nr > max_runs) nr = max_runs;
    for (uint32_t r = threadIdx.x; r < nr; r += blockDim.x) {
        FFATHugeRun hr{};
        hr.head = huge_heads[r];
        hr.slot = sslot[hr.head];
        uint32_t lo = hr.head + FFAT_HUGE_RUN - 1, hi = n;  // sslot[lo] == slot (that is what made the run huge)
        while (lo + 1 < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (sslot[mid] == hr.slot)
                lo = mid;
            else
                hi = mid;
        }
        hr.len = lo + 1 - hr.head;
        hr.n_chunks = (hr.len + WAVE - 1) / WAVE;
        const FFATKeyHeader st = recs.header(hr.slot);
        hr.P0 = st.panes;
        hr.first_lwid = st.next_lwid;
        if (!is_tb) {
            hr.pc0 = st.pcount;
            hr.max_panes = (uint32_t)(((uint64_t)hr.pc0 + hr.len) / pane_len) + 1;
        } else {
            hr.Lq = st.last_quantum;
            const uint64_t q0 = cf_ts(data[sidx[hr.head]]) / pane_len, q1 = cf_ts(data[sidx[hr.head + hr.len - 1]]) / pane_len;
            // late tuples, disorder across the run, or a run that is sparse in time (more panes than tuples: most windows are
            // empty): the sequential rules handle them
            if (q0 < hr.Lq || q1 < q0 || q1 - hr.Lq > (uint64_t)hr.len + WAVE + g.acc_cap) {
                hr.fallback = q1 < q0 ? 3 : 1;  // bit 1: out of timestamp order (reported to the host, which then pre-sorts the batches)
            } else {
                hr.max_panes = (uint32_t)(q1 - hr.Lq) + 1;
                // the run is in timestamp order (the chunks check it): its last tuple decides what is complete afterwards
                const uint64_t ts1 = cf_ts(data[sidx[hr.head + hr.len - 1]]);
                uint64_t done = ts1 > g.delay ? (ts1 - g.delay) / pane_len : 0;
                if (q1 >= done + g.acc_cap) done = q1 - g.acc_cap + 1;  // the ring of open quanta must hold the newest one
                hr.Lq_new = done > hr.Lq ? done : hr.Lq;
            }
        }
        runs[r] = hr;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t chunks = 0, leaves = 0, units = 0;
        for (uint32_t r = 0; r < nr; r++) {
            FFATHugeRun& hr = runs[r];
            const uint32_t need = hr.fallback ? 0 : (win_panes - 1) + hr.max_panes;
            if (leaves + need > leaves_cap) hr.fallback = 1;  // cannot happen with the sizes the engine allocates; stay safe
            hr.chunk_base = chunks;
            hr.unit_base = units;
            hr.leaf_base = leaves;
            if (!hr.fallback) {
                hr.n_units = (2 * hr.n_chunks + WAVE - 1) / WAVE;
                chunks += hr.n_chunks;
                units += hr.n_units;
                leaves += need;
            } else {
                hr.n_chunks = 0;
                hr.n_units = 0;
            }
        }
        ctl->n_runs = nr;
        ctl->total_chunks = chunks;
        ctl->total_units = units;
    }
}

template <typename T, typename R, typename L, typename C>
__global__ void __launch_bounds__(BLOCK_THREADS)
    ffat_huge_chunks_kernel(const T* data, const uint32_t* sidx, uint64_t pane_len, uint32_t is_tb, uint32_t win_panes, FFATHugeRun* runs,
                            const FFATHugeCtl* ctl, FFATPartial<R>* parts, R* leaves, uint8_t* leaf_has, L lift, C comb) {
    const uint32_t total = ctl->total_chunks, nr = ctl->n_runs;
    const uint32_t waves_total = gridDim.x * (BLOCK_THREADS / WAVE);
    const int lane = lane_id();
    for (uint32_t u = blockIdx.x * (BLOCK_THREADS / WAVE) + wave_id(); u < total; u += waves_total) {
        // the run this chunk belongs to: last run whose chunk_base <= u (runs without chunks share their successor's base)
        uint32_t lo = 0, hi = nr;
        while (lo + 1 < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (runs[mid].chunk_base <= u)
                lo = mid;
            else
                hi = mid;
        }
        while (runs[lo].n_chunks == 0 && lo > 0) lo--;  // (fallback runs hold no chunk)
        const FFATHugeRun hr = runs[lo];
        const uint32_t c = u - hr.chunk_base;
        const uint32_t pos = c * WAVE + (uint32_t)lane;
        const bool ok = pos < hr.len;
        const uint32_t cnt = hr.len - c * WAVE < (uint32_t)WAVE ? hr.len - c * WAVE : (uint32_t)WAVE;
        R val{};
        uint64_t pane = 0;
        if (ok) {
            const T t = data[sidx[hr.head + pos]];
            val = ffat_lift<T, R, L>(t, lift);
            pane = is_tb ? cf_ts(t) / pane_len - hr.Lq : ((uint64_t)hr.pc0 + pos) / pane_len;
        }
        const uint32_t lo32 = (uint32_t)(pane & 0xffffffffu), hi32 = (uint32_t)(pane >> 32);
        const uint32_t plo = (uint32_t)__shfl_up((int)lo32, 1, WAVE), phi = (uint32_t)__shfl_up((int)hi32, 1, WAVE);
        uint64_t prev = ((uint64_t)phi << 32) | plo;
        if (is_tb) {
            // panes must not decrease along the run (and never fall before the oldest open one: unsigned wrap-around)
            if (lane == 0 && pos > 0 && ok) prev = cf_ts(data[sidx[hr.head + pos - 1]]) / pane_len - hr.Lq;
            const bool bad = ok && ((pos > 0 && pane < prev) || pane >= (uint64_t)hr.max_panes);
            if (__ballot(bad) != 0) {
                const bool disorder = __ballot(ok && pos > 0 && pane < prev) != 0;
                if (lane == 0) atomicOr(&runs[lo].fallback, disorder ? 3u : 1u);
                continue;
            }
        }
        const bool boundary = ok && (lane == 0 || prev != pane);
        const unsigned long long bmask = __ballot(boundary);
        const unsigned long long upto_me = bmask & (lanemask_lt() | (1ull << lane));
        const int seg = __popcll(upto_me) - 1;
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
        const int n_segs = __popcll(bmask);
        FFATPartial<R>* my_parts = parts + 2 * (size_t)u;
        if (tail) {
            if (seg == 0) {
                my_parts[0].val = val;
                my_parts[0].pane = (uint32_t)pane;
                my_parts[0].valid = 1;
            } else if (seg == n_segs - 1) {
                my_parts[1].val = val;
                my_parts[1].pane = (uint32_t)pane;
                my_parts[1].valid = 1;
            } else {
                R v = val;
                if (is_tb) v.setControlFields(cf_key(val), hr.Lq + pane, (hr.Lq + pane + 1) * pane_len - 1);
                leaves[(size_t)hr.leaf_base + (win_panes - 1) + (uint32_t)pane] = v;
                if (is_tb) leaf_has[(size_t)hr.leaf_base + (win_panes - 1) + (uint32_t)pane] = 1;
            }
        }
        if (lane == 0 && n_segs < 2) {
            my_parts[1].pane = (uint32_t)pane;  // (lane 0 sits in the only segment)
            my_parts[1].valid = 2;
        }
        if (is_tb) {
            // quanta between two segments of this chunk that received nothing: empty panes
            const uint64_t gap = boundary && lane > 0 ? pane - prev - 1 : 0;
            if (__ballot(gap != 0) != 0 && gap != 0) {
                for (uint64_t q = prev + 1; q < pane; q++) {
                    R e{};
                    e.setControlFields(cf_key(val), hr.Lq + q, (hr.Lq + q + 1) * pane_len - 1);
                    leaves[(size_t)hr.leaf_base + (win_panes - 1) + (uint32_t)q] = e;
                    leaf_has[(size_t)hr.leaf_base + (win_panes - 1) + (uint32_t)q] = 0;
                }
            }
        }
    }
}

// Segmented, order-preserving scan over up to 64 consecutive partials (lane l holds one; `ok`: the lane holds an entry,
// `has`: the entry carries a value).  Entries of one pane are contiguous.  Afterwards the tail lane of every segment holds
// the segment's aggregate.
template <typename R>
struct FFATStitched {
    R val;
    bool has, boundary, tail;
    int seg, n_segs;
    uint32_t prev_pane;  // pane of the entry in the lane below (undefined for lane 0)
    uint32_t cnt;        // lanes that hold an entry
};
template <typename R, typename C>
__device__ __forceinline__ FFATStitched<R> ffat_stitch_wave(bool ok, uint32_t pane, bool has, const R& v, const C& comb) {
    const int lane = lane_id();
    FFATStitched<R> o;
    o.val = v;
    o.has = has;
    o.prev_pane = (uint32_t)__shfl_up((int)pane, 1, WAVE);
    o.boundary = ok && (lane == 0 || o.prev_pane != pane);
    const unsigned long long okmask = __ballot(ok);  // a prefix of the lanes
    o.cnt = (uint32_t)__popcll(okmask);
    const unsigned long long bmask = __ballot(o.boundary);
    o.seg = __popcll(bmask & (lanemask_lt() | (1ull << lane))) - 1;
    o.n_segs = __popcll(bmask);
    for (int d = 1; d < WAVE; d <<= 1) {
        const int oseg = __shfl_up(o.seg, d, WAVE);
        const bool take = ok && lane >= d && oseg == o.seg;
        if (__ballot(take) == 0) break;
        const R other = wave_shfl_up(o.val, d);
        const bool other_has = __shfl_up((int)o.has, d, WAVE) != 0;
        if (take && other_has) {
            if (o.has) {
                R tmp;
                fat_combine(comb, other, o.val, tmp);
                o.val = tmp;
            } else {
                o.val = other;
                o.has = true;
            }
        }
    }
    o.tail = ok && ((lane + 1 == (int)o.cnt) || ((bmask >> (lane + 1)) & 1ull));
    return o;
}

template <typename R, typename C, typename K>
__global__ void __launch_bounds__(BLOCK_THREADS)
    ffat_huge_stitch_kernel(FFATGeometry g, const FFATHugeRun* runs, const FFATHugeCtl* ctl, const FFATPartial<R>* parts, FFATPartial<R>* parts2,
                            R* leaves, uint8_t* leaf_has, KeyTableView<K> table, C comb) {
    const uint32_t total = ctl->total_units, nr = ctl->n_runs;
    const uint32_t waves_total = gridDim.x * (BLOCK_THREADS / WAVE);
    const int lane = lane_id();
    for (uint32_t u = blockIdx.x * (BLOCK_THREADS / WAVE) + wave_id(); u < total; u += waves_total) {
        uint32_t lo = 0, hi = nr;
        while (lo + 1 < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (runs[mid].unit_base <= u)
                lo = mid;
            else
                hi = mid;
        }
        while (runs[lo].n_units == 0 && lo > 0) lo--;
        const FFATHugeRun& hr = runs[lo];
        if (hr.fallback) continue;
        const uint32_t e = (u - hr.unit_base) * WAVE + (uint32_t)lane, n_entries = 2 * hr.n_chunks;
        const bool ok = e < n_entries;
        FFATPartial<R> pe{};
        if (ok) pe = parts[2 * (size_t)hr.chunk_base + e];
        const FFATStitched<R> st = ffat_stitch_wave(ok, pe.pane, ok && pe.valid == 1, pe.val, comb);
        using key_t = decltype(cf_key(pe.val));
        const key_t key = table.key_at(hr.slot);
        R* lv = leaves + (size_t)hr.leaf_base + (g.win_panes - 1);
        uint8_t* lh = leaf_has + (size_t)hr.leaf_base + (g.win_panes - 1);
        FFATPartial<R>* mine = parts2 + 2 * (size_t)u;
        if (st.tail) {
            if (st.seg == 0 || st.seg == st.n_segs - 1) {
                FFATPartial<R>& dst = mine[st.seg == 0 ? 0 : 1];
                dst.val = st.val;
                dst.pane = pe.pane;
                dst.valid = 1;
            } else {
                R x = st.val;
                if (g.is_tb) x.setControlFields(key, hr.Lq + pe.pane, (hr.Lq + pe.pane + 1) * g.pane_len - 1);
                lv[pe.pane] = x;
                if (g.is_tb) lh[pe.pane] = 1;
            }
        }
        if (lane == 0 && st.n_segs < 2) {
            mine[1].pane = pe.pane;
            mine[1].valid = 2;
        }
        // time-based: an even entry is the first partial of its chunk; the panes between it and the entry before it belong to
        // no chunk and received nothing (the panes between the two partials of one chunk are the chunk's own)
        if (g.is_tb && st.boundary && lane > 0 && (e & 1u) == 0) {
            for (uint32_t p = st.prev_pane + 1; p < pe.pane; p++) {
                R x{};
                x.setControlFields(key, hr.Lq + p, (hr.Lq + p + 1) * g.pane_len - 1);
                lv[p] = x;
                lh[p] = 0;
            }
        }
    }
}

// One workgroup per run.  The stitch of the second-level partials is a carry chain (wave 0, a handful of steps); everything
// that is proportional to the number of panes - joining the run's panes with the key's ring of open quanta, handing the quanta
// that stay open back to it, appending the complete panes to the key's tree - is done by all the waves of the group.
// Time-based windows with a long triggering delay: the run's first pane sits far behind the oldest open quantum (the ring holds
// many batches' worth of quanta).  Only the quanta the run completes (below Lq_new) and the ones it touches (from its first pane
// on) pass through the leaves; what lies between stays in the ring untouched - the work is proportional to the run, not to the
// ring (64 keys shared by 8 unsynchronised lanes, delay of 16 batches: 2.4 ms per run with one wave walking the ring, see
// profiles/r3_shared_keys).
template <typename R, typename C, typename K>
__global__ void __launch_bounds__(BLOCK_THREADS)
    ffat_huge_fix_kernel(FFATRecords<R> recs, R* trees, FFATGeometry g, FFATHugeRun* runs,
                         const FFATHugeCtl* ctl, const FFATPartial<R>* parts2, R* leaves, uint8_t* leaf_has, uint32_t* out_count, uint32_t* long_heads,
                         uint32_t* n_long, KeyTableView<K> table, FFATCounters* ctr, C comb, volatile uint32_t* disorder_seen = nullptr,
                         uint32_t batch_no = 0) {
    const uint32_t nr = ctl->n_runs;
    const int lane = lane_id();
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const bool w0 = wave_id() == 0;
    const uint32_t W = g.win_panes;
    __shared__ uint32_t sh_cpane, sh_pfirst, sh_m;
    for (uint32_t r = blockIdx.x; r < nr; r += gridDim.x) {
        FFATHugeRun hr = runs[r];
        if (hr.fallback) {
            if (tid == 0) {
                long_heads[atomicAdd(n_long, 1u)] = hr.head;
                if ((hr.fallback & 2u) && disorder_seen != nullptr) *disorder_seen = batch_no;
            }
            continue;
        }
        R* gtree = trees + (size_t)hr.slot * 2 * g.tree_cap;
        R* lv = leaves + (size_t)hr.leaf_base + (W - 1);  // lv[p] = pane P0 + p
        uint8_t* lh = leaf_has + (size_t)hr.leaf_base + (W - 1);  // time-based: the pane received tuples (else it is a place holder)
        using key_t = decltype(cf_key(lv[0]));
        const key_t key = table.key_at(hr.slot);
        // the panes a window fired by this run can still need, in front of the new ones
        for (uint32_t j = tid; j + 1 < W; j += nt) {
            const uint64_t back = (uint64_t)(W - 1 - j);  // pane P0 - back
            // (the key's first windows: the panes before pane 0 do not exist; no window reaches them, the block scans may)
            lv[-(int64_t)back] = hr.P0 >= back ? gtree[g.tree_cap + (uint32_t)((hr.P0 - back) & (g.tree_cap - 1))] : R{};
        }
        R* acc = g.is_tb ? recs.ring(hr.slot) : nullptr;
        uint8_t* used = g.is_tb ? recs.used(hr.slot) : nullptr;
        auto put_leaf = [&](uint32_t p, const R& v) {
            R x = v;
            if (g.is_tb) {
                x.setControlFields(key, hr.Lq + p, (hr.Lq + p + 1) * g.pane_len - 1);
                lh[p] = 1;
            }
            lv[p] = x;
        };
        auto put_empty = [&](uint32_t p) {
            R e{};
            e.setControlFields(key, hr.Lq + p, (hr.Lq + p + 1) * g.pane_len - 1);
            lv[p] = e;
            lh[p] = 0;
        };
        const uint32_t m_tb = g.is_tb ? (uint32_t)(hr.Lq_new - hr.Lq) : 0u;  // time-based: quanta completed by the run
        FFATKeyState<R> st{};
        // the carry of the stitch (kept identical in every lane of wave 0).  Count-based: what was open before the run is the
        // head of pane 0.  Time-based: the quanta that were open before the run (the key's ring) are joined afterwards.
        bool c_has = false;
        uint32_t c_pane = 0;
        R c_val{};
        if (w0) {
            st = recs.load(hr.slot, g.is_tb);
            if (!g.is_tb && hr.pc0 > 0) {
                c_has = true;
                c_val = st.partial;
            }
            // time-based: panes [from, to) received nothing
            auto fill_gap = [&](uint32_t from, uint32_t to) {
                for (uint32_t p = from + (uint32_t)lane; p < to; p += WAVE) put_empty(p);
            };
            uint32_t p_first = 0;
            const uint32_t n_entries = 2 * hr.n_units;
            for (uint32_t e0 = 0; e0 < n_entries; e0 += WAVE) {
                const uint32_t e = e0 + (uint32_t)lane;
                const bool ok = e < n_entries;
                FFATPartial<R> pe{};
                if (ok) pe = parts2[2 * (size_t)hr.unit_base + e];
                FFATStitched<R> sw = ffat_stitch_wave(ok, pe.pane, ok && pe.valid == 1, pe.val, comb);
                // the carry joins the first segment, or is final
                const uint32_t pane0 = (uint32_t)__shfl((int)pe.pane, 0, WAVE);
                if (c_has && c_pane == pane0) {
                    if (sw.tail && sw.seg == 0) {
                        R tmp;
                        fat_combine(comb, c_val, sw.val, tmp);
                        sw.val = tmp;
                    }
                } else {
                    if (c_has && lane == 0) put_leaf(c_pane, c_val);
                    if (g.is_tb) {  // (lane 0 always holds the first partial of a unit)
                        if (c_has)
                            fill_gap(c_pane + 1, pane0);
                        else
                            p_first = pane0;  // (the quanta in front of the run's first pane: filled by the whole group below)
                    }
                }
                if (sw.tail && sw.seg < sw.n_segs - 1) put_leaf(pe.pane, sw.val);
                if (g.is_tb && sw.boundary && lane > 0 && (e & 1u) == 0)
                    for (uint32_t p = sw.prev_pane + 1; p < pe.pane; p++) put_empty(p);
                // the last segment is the new carry
                const int last = (int)sw.cnt - 1;
                c_val = wave_bcast(sw.val, last);
                c_pane = (uint32_t)__shfl((int)pe.pane, last, WAVE);
                c_has = true;
            }
            if (lane == 0) {
                if (g.is_tb) put_leaf(c_pane, c_val);
                sh_cpane = c_pane;
                sh_pfirst = p_first;
            }
        }
        __syncthreads();
        if (g.is_tb) {
            // The run's panes 0 .. c_pane are quanta Lq .. Lq + c_pane.  Join them with what the key's ring held for these quanta,
            // then the quanta below Lq_new are complete (leaves) and the others go (back) to the ring.
            const uint32_t amask = g.acc_cap - 1, n_p = sh_cpane + 1, p_first = sh_pfirst;
            for (uint32_t p = tid; p < p_first && p < m_tb; p += nt) put_empty(p);
            __syncthreads();
            auto join = [&](uint32_t from, uint32_t to) {
                if (to > g.acc_cap) to = g.acc_cap;
                for (uint32_t p = from + tid; p < to; p += nt) {
                    const uint32_t a = (uint32_t)((hr.Lq + p) & amask);
                    if (used[a]) {
                        R x = acc[a];
                        if (lh[p]) fat_combine(comb, acc[a], lv[p], x);
                        put_leaf(p, x);
                    }
                }
            };
            if (p_first <= m_tb) {
                join(0, n_p);
            } else {
                join(0, m_tb);
                join(p_first, n_p);
            }
            __syncthreads();
            for (uint32_t p = tid; p < m_tb && p < g.acc_cap; p += nt) used[(uint32_t)((hr.Lq + p) & amask)] = 0;
            __syncthreads();
            // (at most acc_cap quanta: see the set-up kernel; the quanta between Lq_new and the run's first pane keep their ring entries)
            for (uint32_t p = (p_first > m_tb ? p_first : m_tb) + tid; p < n_p; p += nt) {
                const uint32_t a = (uint32_t)((hr.Lq + p) & amask);
                if (lh[p]) acc[a] = lv[p];
                used[a] = lh[p];
            }
        }
        // the last pane of the run: complete (count-based, when the run ends on a pane boundary) or left open
        if (tid == 0) {
            uint32_t m = 0;
            if (!g.is_tb) {
                const uint64_t total = (uint64_t)hr.pc0 + hr.len;
                m = (uint32_t)(total / g.pane_len);
                const uint32_t rem = (uint32_t)(total % g.pane_len);
                if (rem == 0) {
                    put_leaf(c_pane, c_val);
                    st.pcount = 0;
                } else {
                    st.partial = c_val;
                    st.pcount = rem;
                }
            } else {
                m = m_tb;  // the quanta from Lq_new on stay open
                const uint64_t q1 = hr.Lq + c_pane;
                st.last_quantum = hr.Lq_new;
                if (q1 + 1 > st.max_quantum) st.max_quantum = q1 + 1;
            }
            st.seen = 1;
            st.panes = hr.P0 + m;
            const uint64_t k_end = st.panes >= W ? (st.panes - W) / g.slide_panes + 1 : 0;
            hr.nf = k_end > st.next_lwid ? (uint32_t)(k_end - st.next_lwid) : 0;
            hr.first_lwid = st.next_lwid;
            hr.out_base = hr.nf ? atomicAdd(out_count, hr.nf) : 0;
            if (k_end > st.next_lwid) st.next_lwid = k_end;
            hr.m = m;
            runs[r] = hr;
            recs.store(hr.slot, st, g.is_tb);
            atomicAdd(&ctr->huge_runs, 1u);
            sh_m = m;
        }
        __syncthreads();
        // append the new panes to the key's tree (the newest tree_cap of them) and rebuild it
        const uint32_t m = sh_m;
        const uint64_t Pn = hr.P0 + m;
        uint64_t lo = Pn > g.tree_cap ? Pn - g.tree_cap : 0;
        if (lo < hr.P0) lo = hr.P0;
        for (uint64_t idx = lo + (uint64_t)tid; idx < Pn; idx += nt) gtree[g.tree_cap + (uint32_t)(idx & (g.tree_cap - 1))] = lv[idx - hr.P0];
        __syncthreads();
        if (w0 && m > 0) fat_rebuild_wave(gtree, g.tree_cap, comb);
        __syncthreads();
    }
}

// Block-wise prefix and suffix folds of the leaves of every huge run (see 5. above).  Array coordinates: a = 0 is the oldest
// of the W - 1 old panes, the run's own panes start at a = W - 1, `(W - 1) + m` entries are complete.  Work item 2 * b is
// the prefix scan of block b, 2 * b + 1 its suffix scan; a wave scans 64 entries per step and carries the fold along.
template <typename R, typename C>
__global__ void __launch_bounds__(BLOCK_THREADS)
    ffat_huge_hgw_kernel(FFATGeometry g, const FFATHugeRun* runs, const FFATHugeCtl* ctl, const R* leaves, R* pre, R* suf, C comb) {
    const uint32_t nr = ctl->n_runs;
    const uint32_t W = g.win_panes;
    const uint32_t waves_total = gridDim.x * (BLOCK_THREADS / WAVE);
    const uint32_t gw = blockIdx.x * (BLOCK_THREADS / WAVE) + wave_id();
    const int lane = lane_id();
    for (uint32_t r = 0; r < nr; r++) {
        const FFATHugeRun& hr = runs[r];
        if (hr.fallback || hr.nf == 0) continue;
        const uint32_t n_valid = (W - 1) + hr.m;
        const uint32_t items = 2 * ((n_valid + W - 1) / W);
        const R* arr = leaves + (size_t)hr.leaf_base;
        for (uint32_t it = gw; it < items; it += waves_total) {
            const uint32_t lo = (it >> 1) * W, hi = lo + W < n_valid ? lo + W : n_valid;
            const bool backwards = (it & 1u) != 0;
            R* dst = (backwards ? suf : pre) + (size_t)hr.leaf_base;
            bool c_has = false;
            R c_val{};
            for (uint32_t done = 0; lo + done < hi; done += WAVE) {
                const uint32_t left = hi - lo - done, cnt = left < (uint32_t)WAVE ? left : (uint32_t)WAVE;
                const bool ok = (uint32_t)lane < cnt;
                // lane order = fold order: ascending entries for the prefix, descending ones for the suffix
                const uint32_t a = backwards ? hi - 1 - done - (uint32_t)lane : lo + done + (uint32_t)lane;
                R val{};
                if (ok) val = arr[a];
                for (int d = 1; d < WAVE; d <<= 1) {
                    const R other = wave_shfl_up(val, d);
                    if (ok && lane >= d) {
                        R tmp;
                        if (backwards)
                            fat_combine(comb, val, other, tmp);  // the lower lanes hold the LATER panes
                        else
                            fat_combine(comb, other, val, tmp);
                        val = tmp;
                    }
                }
                if (c_has && ok) {
                    R tmp;
                    if (backwards)
                        fat_combine(comb, val, c_val, tmp);
                    else
                        fat_combine(comb, c_val, val, tmp);
                    val = tmp;
                }
                if (ok) dst[a] = val;
                c_val = wave_bcast(val, (int)cnt - 1);
                c_has = true;
            }
        }
    }
}

template <typename R, typename C, typename K>
__global__ void __launch_bounds__(BLOCK_THREADS)
    ffat_huge_windows_kernel(FFATGeometry g, const FFATHugeRun* runs, const FFATHugeCtl* ctl, const R* leaves, const R* pre, const R* suf, R* out,
                             FFATCounters* ctr, KeyTableView<K> table, C comb) {
    const uint32_t nr = ctl->n_runs;
    const uint32_t W = g.win_panes;
    for (uint32_t r = 0; r < nr; r++) {
        const FFATHugeRun& hr = runs[r];
        if (hr.fallback || hr.nf == 0) continue;
        using key_t = decltype(cf_key(out[0]));
        const key_t key = table.key_at(hr.slot);
        const R* lv = leaves + (size_t)hr.leaf_base + (W - 1);
        for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < hr.nf; w += gridDim.x * blockDim.x) {
            const uint64_t k = hr.first_lwid + w;
            const int64_t first = (int64_t)(k * g.slide_panes) - (int64_t)hr.P0;  // >= -(W - 1)
            R res;
            if (pre != nullptr) {
                const size_t a = (size_t)hr.leaf_base + (size_t)(first + (int64_t)(W - 1));  // array coordinate of the first pane
                res = suf[a];
                if ((first + (int64_t)(W - 1)) % (int64_t)W != 0) {  // not a whole block: the rest is a prefix of the next one
                    R tmp;
                    fat_combine(comb, res, pre[a + W - 1], tmp);
                    res = tmp;
                }
            } else {
                res = lv[first];
                for (uint32_t p = 1; p < W; p++) {
                    R tmp;
                    fat_combine(comb, res, lv[first + (int64_t)p], tmp);
                    res = tmp;
                }
            }
            res.setControlFields(key, k, cf_ts(res));
            const uint32_t pos = hr.out_base + w;
            if (pos < g.out_cap)
                out[pos] = res;
            else
                atomicAdd(&ctr->overflow, 1u);
        }
    }
}

}  // namespace gpu
}  // namespace wf
