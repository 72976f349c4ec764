// Device primitives shared by the keyed GPU operators: single-pass exclusive
// scan, key -> slot hash table, stable LSD radix sort of (slot, index) pairs and
// one-pass multi-way partition.
//
// What they replace in the reference:
//   K12 map_to_target / prescan / gather_sums + host recursion with
//       cudaDeviceSynchronize (wf/gpu_utils.hpp:322-401)     -> exclusive_scan_u32
//   K13 create_sub_batch, one scan+scatter pass per destination
//       (wf/standard_nodes_gpu.hpp:52-58, :172-212)           -> partition_by_dest
//   B11 Thrust sort_by_key / reduce_by_key key scatter
//       (src/GPU_Tests/scattering/test_v4.cpp:244-293)        -> radix_sort_pairs
//   host-built per-key linked lists of the stateful benchmarks
//       (gpu_map_stateful.cpp:311-328)                        -> DeviceKeyTable + sort
// All kernels are wave64: in-wave ranks come from __ballot + popcount, per-wave
// totals are combined through LDS, cross-tile offsets through decoupled look-back
// (scan) or a digit-major histogram scan (sort / partition).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <tuple>
#include <type_traits>
#include <utility>

#include "hip_utils.hpp"
#include "kernels_basic.hpp"

namespace wf {
namespace gpu {

// ---------------------------------------------------------------------------
// exclusive scan of uint32 (single pass, decoupled look-back)
// ws: [0] ticket, [1..] descriptors; must be zeroed before the launch
// ---------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK_THREADS * SCAN_ITEMS;
inline size_t scan_workspace_words(size_t n) { return 2 + (n + SCAN_TILE - 1) / SCAN_TILE; }

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t u = __shfl_up(v, o, WAVE);
        if (lane_id() >= o) v += u;
    }
    return v;
}

// n_dev (optional): the number of valid elements lives in device memory (at most n_upper, which sizes the grid)
__global__ void __launch_bounds__(BLOCK_THREADS)
    exclusive_scan_u32_kernel(const uint32_t* in, uint32_t* out, uint32_t n_upper, uint32_t* total, unsigned long long* ws,
                              const uint32_t* n_dev = nullptr) {
    uint32_t n = n_upper;
    if (n_dev != nullptr) {
        const uint32_t nd = *n_dev;
        n = nd < n_upper ? nd : n_upper;
    }
    __shared__ uint32_t s_wave[BLOCK_THREADS / WAVE];
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_base;
    // ws == nullptr: the caller guarantees a single tile (n_upper <= SCAN_TILE): no ticket, no look-back, no workspace to clear
    if (threadIdx.x == 0) s_tile = ws ? (uint32_t)atomicAdd(&ws[0], 1ull) : blockIdx.x;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (tile >= n_tiles) {
        if (n_tiles == 0 && tile == 0 && threadIdx.x == 0 && total) *total = 0;
        return;
    }
    unsigned long long* desc = ws + 1;
    const int lane = lane_id(), wave = wave_id();
    // blocked arrangement: thread t owns SCAN_ITEMS consecutive elements
    const uint32_t base = tile * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        sum += v[k];
    }
    const uint32_t inc = wave_inclusive_scan(sum);
    if (lane == WAVE - 1) s_wave[wave] = inc;
    __syncthreads();
    uint32_t wave_off = 0, tile_total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK_THREADS / WAVE; w++) {
        if (w < wave) wave_off += s_wave[w];
        tile_total += s_wave[w];
    }
    if (wave == 0) {
        uint32_t exclusive = 0;
        if (tile == 0) {
            if (lane == 0 && ws) __hip_atomic_store(&desc[0], DESC_INC | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&desc[tile], DESC_AGG | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int look = (int)tile - 1;
            while (true) {
                const int p = look - lane;
                unsigned long long d = DESC_INC;
                if (p >= 0) {
                    do {
                        d = __hip_atomic_load(&desc[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((d >> 32) == 0);
                }
                const unsigned long long inc_mask = __ballot((d >> 32) == 2);
                const int stop = inc_mask ? (int)__builtin_ctzll(inc_mask) : WAVE;
                uint32_t x = (lane <= stop) ? (uint32_t)(d & 0xffffffffu) : 0u;
#pragma unroll
                for (int o = WAVE / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, WAVE);
                exclusive += x;
                if (inc_mask) break;
                look -= WAVE;
            }
            if (lane == 0)
                __hip_atomic_store(&desc[tile], DESC_INC | (unsigned long long)(exclusive + tile_total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_base = exclusive;
            if (tile == n_tiles - 1 && total) *total = exclusive + tile_total;
        }
    }
    __syncthreads();
    uint32_t run = s_base + wave_off + (inc - sum);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

inline void exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total, unsigned long long* ws, hipStream_t s,
                               const uint32_t* n_dev = nullptr) {
    if (n <= (size_t)SCAN_TILE) {  // one tile: nothing to coordinate
        hipLaunchKernelGGL(exclusive_scan_u32_kernel, dim3(1), dim3(BLOCK_THREADS), 0, s, in, out, (uint32_t)n, total, (unsigned long long*)nullptr, n_dev);
        return;
    }
    WF_HIP_CHECK(hipMemsetAsync(ws, 0, scan_workspace_words(n) * sizeof(unsigned long long), s));
    const unsigned grid = (unsigned)((n + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(exclusive_scan_u32_kernel, dim3(grid ? grid : 1), dim3(BLOCK_THREADS), 0, s, in, out, (uint32_t)n, total, ws, n_dev);
}

// ---------------------------------------------------------------------------
// key -> slot table (open addressing, linear probing, insert-or-find)
// ---------------------------------------------------------------------------
constexpr unsigned long long KEY_EMPTY = ~0ull;

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

// returns the slot (table position) of `key`, inserting it if absent; 0xffffffff if the table is full or the key is
// the reserved value 2^64-1 (the empty-slot sentinel: accepting it would let two keys share a slot).  Callers count
// such tuples and the operators fail loudly on them.
__device__ __forceinline__ uint32_t key_table_slot(unsigned long long* table, uint32_t mask, unsigned long long key) {
    if (key == KEY_EMPTY) return 0xffffffffu;
    uint32_t h = (uint32_t)mix64(key) & mask;
    for (uint32_t probes = 0; probes <= mask; probes++) {
        unsigned long long cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return h;
        if (cur == KEY_EMPTY) {
            const unsigned long long prev = atomicCAS(&table[h], KEY_EMPTY, key);
            if (prev == KEY_EMPTY || prev == key) return h;
        }
        h = (h + 1) & mask;
    }
    return 0xffffffffu;
}

// ---- keys of any device-friendly type ------------------------------------------------------------------------------------
// A key is an integer (the table word IS the key) or a fixed-size value type with hash() and == (basic.hpp FixedKey; the
// reference's GPU programs key by std::string, src/mp_test_gpu/mp_common_string.hpp:44-84): then the table word is a 64-bit
// fingerprint of the key and the key itself lives in a slab next to the table; a slot is claimed with one CAS on the
// fingerprint, the winner publishes the key (`ready`), everybody else compares the key, so two keys with the same fingerprint
// simply occupy two slots.
template <typename K>
struct is_fixed_key {
    template <typename X>
    static auto test(int) -> decltype((uint64_t)std::declval<const X&>().hash(), (bool)(std::declval<const X&>() == std::declval<const X&>()), std::true_type{});
    template <typename>
    static std::false_type test(...);
    static constexpr bool value = decltype(test<K>(0))::value && std::is_trivially_copyable_v<K> && sizeof(K) <= 32 && sizeof(K) % 4 == 0;
};
template <typename K>
constexpr bool is_device_key_v = std::is_integral_v<K> || is_fixed_key<K>::value;

// what std::hash gives for the key on the host (the identity for integers): window assignment and routing derive from it
template <typename K>
__host__ __device__ __forceinline__ uint64_t key_hash64(const K& k) {
    if constexpr (std::is_integral_v<K>)
        return (uint64_t)k;
    else
        return (uint64_t)k.hash();
}
template <typename T>
__host__ __device__ __forceinline__ auto tuple_key(const T& t) {
    return std::get<0>(t.getControlFields());
}

template <typename K>
struct KeyTableView {
    static constexpr bool direct = std::is_integral_v<K>;
    unsigned long long* words = nullptr;  // direct: the keys; otherwise fingerprints
    K* keys = nullptr;                    // fixed-size keys only
    uint32_t* ready = nullptr;            // fixed-size keys only: the key of the slot has been published
    uint32_t mask = 0;

    // slot of `key`, inserting it if absent; 0xffffffff: table full (or, for integers, the reserved key 2^64-1)
    __device__ __forceinline__ uint32_t slot_of(const K& key) const {
        if constexpr (direct) {
            return key_table_slot(words, mask, (unsigned long long)key);
        } else {
            unsigned long long fp = mix64(key_hash64(key));
            if (fp == KEY_EMPTY) fp = 0;
            uint32_t h = (uint32_t)fp & mask;
            uint32_t probes = 0;
            // (a lane that finds the fingerprint of a key not yet published goes round the loop again instead of spinning: the
            // publisher may be a lane of the same wave, which completes its branch of the iteration before the next one begins)
            while (probes <= mask) {
                unsigned long long cur = __hip_atomic_load(&words[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == KEY_EMPTY) {
                    const unsigned long long prev = atomicCAS(&words[h], KEY_EMPTY, fp);
                    if (prev == KEY_EMPTY) {
                        keys[h] = key;
                        __hip_atomic_store(&ready[h], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        return h;
                    }
                    cur = prev;
                }
                if (cur == fp) {
                    // the flag and the key are read with agent-scope atomic loads: coherent across the dies without an acquire
                    // fence per probe (which would invalidate the caches of the whole compute unit for every tuple)
                    if (__hip_atomic_load(&ready[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) continue;
                    if (key_equals_coherent(&keys[h], key)) return h;
                }
                h = (h + 1) & mask;
                probes++;
            }
            return 0xffffffffu;
        }
    }
    __device__ static __forceinline__ bool key_equals_coherent(const K* stored, const K& key) {
        uint32_t mine[sizeof(K) / 4];
        __builtin_memcpy(mine, &key, sizeof(K));
        bool eq = true;
#pragma unroll
        for (size_t w = 0; w < sizeof(K) / 4; w++)
            eq = eq && __hip_atomic_load(reinterpret_cast<const uint32_t*>(stored) + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine[w];
        return eq;
    }
    __device__ __forceinline__ bool occupied(uint32_t slot) const { return words[slot] != KEY_EMPTY; }
    __device__ __forceinline__ K key_at(uint32_t slot) const {
        if constexpr (direct)
            return (K)words[slot];
        else
            return keys[slot];
    }
    // the key's host-side hash code (the window a replica starts at depends on it)
    __device__ __forceinline__ uint64_t hash_at(uint32_t slot) const {
        if constexpr (direct)
            return (uint64_t)words[slot];
        else
            return key_hash64(keys[slot]);
    }
};

template <typename K>
class KeyTable {
    static_assert(is_device_key_v<K>, "keys of GPU operators: an integer, or a trivially copyable type of at most 32 bytes with hash() and == (wf::FixedKey)");

public:
    void init(size_t capacity, hipStream_t s) {
        cap_ = capacity;
        words_.resize(capacity);
        WF_HIP_CHECK(hipMemsetAsync(words_.data(), 0xff, words_.bytes(), s));
        if constexpr (!KeyTableView<K>::direct) {
            keys_.resize(capacity);
            ready_.resize(capacity);
            WF_HIP_CHECK(hipMemsetAsync(keys_.data(), 0, keys_.bytes(), s));
            WF_HIP_CHECK(hipMemsetAsync(ready_.data(), 0, ready_.bytes(), s));
        }
    }
    KeyTableView<K> view() const {
        KeyTableView<K> v;
        v.words = words_.data();
        if constexpr (!KeyTableView<K>::direct) {
            v.keys = keys_.data();
            v.ready = ready_.data();
        }
        v.mask = (uint32_t)(cap_ - 1);
        return v;
    }
    size_t capacity() const { return cap_; }
    size_t bytes() const { return words_.bytes() + keys_.bytes() + ready_.bytes(); }

private:
    size_t cap_ = 0;
    DeviceBuffer<unsigned long long> words_;
    DeviceBuffer<K> keys_;
    DeviceBuffer<uint32_t> ready_;
};

// ---- batches in which no key occurs twice need no sort -------------------------------------------------------------
// The keyed engines sort (slot, index) pairs so that the tuples of a key become a run in arrival order.  A batch that holds
// at most one tuple of every key (many keys, round-robin or sparse arrivals: the flagship stream) is already "a run per
// tuple" in any order.  assign_slots_kernel finds out on the fly: every slot carries the number of the last batch that
// touched it (`stamp`), a tuple that finds the current batch number there is a second tuple of its key.  A per-block LDS
// filter catches the repeats inside a block first, so streams with few keys do not pile their atomics on a handful of words.
// The sort kernels are launched with a SortGate and return at once unless the batch reported a repeat; consumers pick the
// unsorted arrays in that case (SortedSel).
struct DupProbe {
    uint32_t* stamp = nullptr;               // stamp of slot 0; nullptr: no probing (the batch is always sorted)
    uint32_t stamp_stride = 4;               // bytes from one slot's stamp to the next (4: an array of its own; more: the stamps live
                                             // inside per-key records, so that the probe pulls in the line the update will need)
    uint32_t* dup_batch = nullptr;           // device word: number of the last batch in which a key occurred twice
    volatile uint32_t* dup_seen = nullptr;   // pinned copy of it for the host (DupMonitor)
    uint32_t batch_no = 0;
};
struct SortGate {
    const uint32_t* word = nullptr;  // nullptr: always run
    uint32_t value = 0;
    uint32_t rescue = 0;  // 1: the gate is expected to stay closed - the whole sort is ONE launch (a single workgroup does it if it must)
    __device__ __forceinline__ bool closed() const { return word != nullptr && *word != value; }
};
// where the (slot, index) pairs of the batch are: sorted if the gate let the sort run, as assigned otherwise
struct SortedSel {
    const uint32_t* sorted_keys;
    const uint32_t* sorted_vals;
    const uint32_t* raw_keys;
    const uint32_t* raw_vals;
    SortGate gate;
    __device__ __forceinline__ bool is_sorted() const { return !gate.closed(); }
    __device__ __forceinline__ const uint32_t* keys() const { return gate.closed() ? raw_keys : sorted_keys; }
    __device__ __forceinline__ const uint32_t* vals() const { return gate.closed() ? raw_vals : sorted_vals; }
};
constexpr uint32_t DUP_LDS_SLOTS = 1024;

// slot[i] = slot of tuple i's key; idx[i] = i.
// First kernel of a keyed batch: it also clears what the later kernels of the batch accumulate into (the header
// of the output batch, a counter word) - one launch instead of a memset each.
template <typename T, typename K>
__global__ void __launch_bounds__(BLOCK_THREADS)
    assign_slots_kernel(const BatchHeader* hdr, const T* data, KeyTableView<K> table, uint32_t* slot, uint32_t* idx,
                        uint32_t* n_failed, BatchHeader* clear_hdr, uint32_t* clear_word, DupProbe probe = DupProbe{}) {
    __shared__ uint32_t s_seen[DUP_LDS_SLOTS];
    const uint32_t n = hdr->count;
    if (blockIdx.x == 0 && threadIdx.x < sizeof(BatchHeader) / sizeof(uint64_t)) {
        if (clear_hdr) reinterpret_cast<uint64_t*>(clear_hdr)[threadIdx.x] = 0;
        if (clear_word && threadIdx.x == 0) *clear_word = 0;
    }
    const bool probing = probe.stamp != nullptr;
    if (probing) {
        for (uint32_t k = threadIdx.x; k < DUP_LDS_SLOTS; k += BLOCK_THREADS) s_seen[k] = 0xffffffffu;
        __syncthreads();
    }
    bool dup = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t s = table.slot_of((K)tuple_key(data[i]));
        if (s == 0xffffffffu) atomicAdd(n_failed, 1u);
        slot[i] = s;
        idx[i] = i;
        if (probing && s != 0xffffffffu) {
            // (the same slot twice under one LDS word = a repeat inside this block; a different slot just evicts)
            if (atomicExch(&s_seen[(s * 2654435761u) >> 22], s) == s)
                dup = true;
            else if (atomicMax(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(probe.stamp) + (size_t)s * probe.stamp_stride), probe.batch_no) ==
                     probe.batch_no)
                dup = true;
        }
    }
    if (probing && __ballot(dup) != 0 && lane_id() == 0) {
        if (atomicExch(probe.dup_batch, probe.batch_no) != probe.batch_no && probe.dup_seen != nullptr) *probe.dup_seen = probe.batch_no;
    }
}

// Host side of the probe: batches are probed unless repeats keep showing - then the probe is switched off (its atomics buy
// nothing, every batch is sorted) and tried again on one batch after `LINGER` batches.
class DupMonitor {
public:
    DupMonitor() {
        WF_HIP_CHECK(hipHostMalloc((void**)&seen_, 64, hipHostMallocDefault));
        *seen_ = 0;
        if (const char* e = std::getenv("WF_SORT_SKIP")) enabled_ = std::atoi(e) != 0;
    }
    ~DupMonitor() {
        if (seen_) (void)hipHostFree((void*)seen_);
    }
    DupMonitor(const DupMonitor&) = delete;
    DupMonitor& operator=(const DupMonitor&) = delete;
    void reserve(size_t table_slots, hipStream_t s) {
        stamp_.resize(table_slots);
        stamp0_ = stamp_.data();
        stride_ = 4;
        word_.resize(1);
        WF_HIP_CHECK(hipMemsetAsync(stamp_.data(), 0, stamp_.bytes(), s));
        WF_HIP_CHECK(hipMemsetAsync(word_.data(), 0, sizeof(uint32_t), s));
    }
    // the stamps live in the caller's per-key records (zeroed by the caller): `first` = stamp of slot 0, `stride` bytes apart
    void use_external_stamps(uint32_t* first, uint32_t stride, hipStream_t s) {
        stamp0_ = first;
        stride_ = stride;
        word_.resize(1);
        WF_HIP_CHECK(hipMemsetAsync(word_.data(), 0, sizeof(uint32_t), s));
    }
    // called once per batch before its kernels are enqueued; batch numbers start at 1
    DupProbe probe(uint32_t batch_no) {
        if (!enabled_ || stamp0_ == nullptr) return DupProbe{};
        const uint32_t seen = *seen_;
        if (seen != last_seen_) {
            last_seen_ = seen;
            off_until_ = batch_no + LINGER;
        }
        if (batch_no < off_until_) {
            clean_from_ = 0;
            return DupProbe{};
        }
        if (clean_from_ == 0) clean_from_ = batch_no;
        probed_++;
        return DupProbe{stamp0_, stride_, word_.data(), seen_, batch_no};
    }
    // The gate of the sort launches behind a probed batch.  After CONFIDENT probed batches without a report (reports lag by the
    // few batches in flight) the stream is taken to hold no repeats: the six gated launches of a sort shrink to one, whose single
    // workgroup does the whole sort in the unlikely case that it is needed after all (slow, but only until the report of that
    // batch has switched the probing off: a handful of batches at a change of regime).
    SortGate gate(const DupProbe& p) const {
        if (!p.stamp) return SortGate{};
        const bool confident = rescue_ && clean_from_ != 0 && p.batch_no - clean_from_ >= CONFIDENT;
        return SortGate{p.dup_batch, p.batch_no, confident ? 1u : 0u};
    }
    uint64_t probed_batches() const { return probed_; }

private:
    static constexpr uint32_t LINGER = 64, CONFIDENT = 32;
    volatile uint32_t* seen_ = nullptr;
    bool enabled_ = true;
    bool rescue_ = std::getenv("WF_SORT_RESCUE") == nullptr || std::atoi(std::getenv("WF_SORT_RESCUE")) != 0;
    uint32_t last_seen_ = 0, off_until_ = 0, clean_from_ = 0;
    uint64_t probed_ = 0;
    uint32_t* stamp0_ = nullptr;
    uint32_t stride_ = 4;
    DeviceBuffer<uint32_t> stamp_, word_;
};

// ---------------------------------------------------------------------------
// stable LSD radix sort of (key32, value32) pairs, 8 bits per pass
// ---------------------------------------------------------------------------
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = BLOCK_THREADS * RS_ITEMS;  // 2048 pairs per block
constexpr int RS_BINS = 256;

__global__ void __launch_bounds__(BLOCK_THREADS)
    radix_hist_kernel(const uint32_t* keys, const uint32_t* n_ptr, uint32_t n_tiles, int shift, uint32_t* hist /*[bin][tile]*/, SortGate gate) {
    __shared__ uint32_t s_hist[RS_BINS];
    if (gate.closed()) return;
    const uint32_t n = *n_ptr;
    const uint32_t tile = blockIdx.x;
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = tile * RS_TILE;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const uint32_t i = base + k * BLOCK_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&s_hist[(keys[i] >> shift) & 0xff], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * n_tiles + tile] = s_hist[threadIdx.x];
}

// RAW: `hist_scanned` holds the raw per-tile counts and `bin_totals` is not read - every block derives the bin
// totals and its own offsets inside the bins from them.  For batches of a few tiles that is cheaper than the
// launch of the scan kernel between the histogram and the scatter (RadixSorter::FUSE_TILES).
template <bool RAW>
__global__ void __launch_bounds__(BLOCK_THREADS)
    radix_scatter_kernel(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, const uint32_t* n_ptr,
                         uint32_t n_tiles, int shift, const uint32_t* hist_scanned /*[bin][tile], exclusive within the bin*/,
                         const uint32_t* bin_totals /*[bin]*/, SortGate gate) {
    constexpr int NW = BLOCK_THREADS / WAVE;
    __shared__ uint32_t s_cnt[RS_ITEMS * NW][RS_BINS];  // per (item round, wave) digit counts -> 32 KiB
    __shared__ uint32_t s_bin_base[RS_BINS];
    __shared__ uint32_t s_wsum[NW];
    if (gate.closed()) return;
    uint32_t tile_off = 0;  // tuples of this thread's digit in the tiles before this one
    {
        // exclusive scan of the 256 bin totals (one value per thread)
        uint32_t v = 0;
        if constexpr (RAW) {
            for (uint32_t t = 0; t < n_tiles; t++) {
                const uint32_t c = hist_scanned[threadIdx.x * n_tiles + t];
                if (t < blockIdx.x) tile_off += c;
                v += c;
            }
        } else {
            v = bin_totals[threadIdx.x];
            tile_off = hist_scanned[threadIdx.x * n_tiles + blockIdx.x];
        }
        const uint32_t inc = wave_inclusive_scan(v);
        if (lane_id() == WAVE - 1) s_wsum[wave_id()] = inc;
        __syncthreads();
        uint32_t off = 0;
        for (int w = 0; w < wave_id(); w++) off += s_wsum[w];
        s_bin_base[threadIdx.x] = off + inc - v;
    }
    const uint32_t n = *n_ptr;
    const uint32_t tile = blockIdx.x;
    const int lane = lane_id(), wave = wave_id();
    for (int r = 0; r < RS_ITEMS * NW; r++) s_cnt[r][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = tile * RS_TILE;
    uint32_t k_[RS_ITEMS], v_[RS_ITEMS], rank[RS_ITEMS];
    bool ok[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const uint32_t i = base + k * BLOCK_THREADS + threadIdx.x;
        ok[k] = i < n;
        k_[k] = ok[k] ? keys_in[i] : 0xffffffffu;
        v_[k] = ok[k] ? vals_in[i] : 0u;
        const uint32_t digit = (k_[k] >> shift) & 0xff;
        // lanes of this wave holding the same digit (match-any with 8 ballots)
        unsigned long long m = __ballot(ok[k]);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long bal = __ballot((digit >> b) & 1);
            m &= ((digit >> b) & 1) ? bal : ~bal;
        }
        rank[k] = (uint32_t)__popcll(m & lanemask_lt());
        if (ok[k] && rank[k] == 0) s_cnt[k * NW + wave][digit] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    // per digit (one thread per digit): exclusive prefix over the (round, wave) sequence + global base
    {
        uint32_t run = s_bin_base[threadIdx.x] + tile_off;
        for (int r = 0; r < RS_ITEMS * NW; r++) {
            const uint32_t c = s_cnt[r][threadIdx.x];
            s_cnt[r][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        if (ok[k]) {
            const uint32_t digit = (k_[k] >> shift) & 0xff;
            const uint32_t dst = s_cnt[k * NW + wave][digit] + rank[k];
            keys_out[dst] = k_[k];
            vals_out[dst] = v_[k];
        }
    }
    (void)lane;
}

// per-bin exclusive scan of the digit-major histogram: block b scans hist[b][0..n_tiles)
// and publishes the bin total; the scatter kernel adds the prefix over bins itself.
// (256 independent blocks instead of one serial block over 256*n_tiles entries.)
__global__ void __launch_bounds__(BLOCK_THREADS) radix_bin_scan_kernel(uint32_t* hist, uint32_t n_tiles, uint32_t* bin_totals, SortGate gate) {
    __shared__ uint32_t s_wave[BLOCK_THREADS / WAVE];
    __shared__ uint32_t s_carry;
    if (gate.closed()) return;
    uint32_t* row = hist + (size_t)blockIdx.x * n_tiles;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    for (uint32_t base = 0; base < n_tiles; base += BLOCK_THREADS) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = (i < n_tiles) ? row[i] : 0u;
        const uint32_t inc = wave_inclusive_scan(v);
        if (lane == WAVE - 1) s_wave[wave] = inc;
        __syncthreads();
        uint32_t off = s_carry;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        if (i < n_tiles) row[i] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == BLOCK_THREADS - 1) s_carry = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) bin_totals[blockIdx.x] = s_carry;
}

// ---- wide digits: 11 bits per pass (2048 bins) -------------------------------------------------
// Every pass costs three launches whose duration is latency, not bandwidth, for the batch sizes of a
// streaming pipeline; slots of a 2^18-entry key table need three 8-bit passes but only two 11-bit
// ones.  2048 bins do not fit the (round, wave) counter matrix of the 8-bit scatter, so here a tile
// is blocked by wave (wave w owns elements [w*512, (w+1)*512) of the tile) and each wave keeps ONE
// running counter per digit across its 8 rounds: order (wave, round, lane) is the input order, so the
// pass is stable.
constexpr int RSW_BITS = 11;
constexpr int RSW_BINS = 1 << RSW_BITS;

__global__ void __launch_bounds__(BLOCK_THREADS)
    radix_hist_wide_kernel(const uint32_t* keys, const uint32_t* n_ptr, uint32_t n_tiles, int shift, uint32_t* hist /*[bin][tile]*/, SortGate gate) {
    __shared__ uint32_t s_hist[RSW_BINS];
    if (gate.closed()) return;
    const uint32_t n = *n_ptr;
    const uint32_t tile = blockIdx.x;
    for (int b = threadIdx.x; b < RSW_BINS; b += BLOCK_THREADS) s_hist[b] = 0;
    __syncthreads();
    const uint32_t base = tile * RS_TILE;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const uint32_t i = base + k * BLOCK_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&s_hist[(keys[i] >> shift) & (RSW_BINS - 1)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < RSW_BINS; b += BLOCK_THREADS) hist[(size_t)b * n_tiles + tile] = s_hist[b];
}

template <bool RAW>  // see radix_scatter_kernel
__global__ void __launch_bounds__(BLOCK_THREADS)
    radix_scatter_wide_kernel(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, const uint32_t* n_ptr,
                              uint32_t n_tiles, int shift, const uint32_t* hist_scanned /*[bin][tile], exclusive within the bin*/,
                              const uint32_t* bin_totals /*[bin]*/, SortGate gate) {
    constexpr int NW = BLOCK_THREADS / WAVE;
    constexpr int PER_THREAD = RSW_BINS / BLOCK_THREADS;  // bins whose global base a thread computes
    __shared__ uint32_t s_cnt[NW][RSW_BINS];               // per-wave running digit counters, then offsets (32 KiB)
    __shared__ uint32_t s_bin_base[RSW_BINS];
    __shared__ uint32_t s_tile_off[RAW ? RSW_BINS : 1];    // RAW: tuples of a digit in the tiles before this one
    __shared__ uint32_t s_wsum[NW];
    if (gate.closed()) return;
    const int lane = lane_id(), wave = wave_id();
    {
        // exclusive scan of the 2048 bin totals: 8 consecutive bins per thread
        uint32_t local[PER_THREAD];
        uint32_t sum = 0;
#pragma unroll
        for (int q = 0; q < PER_THREAD; q++) {
            const uint32_t bin = threadIdx.x * PER_THREAD + q;
            if constexpr (RAW) {
                uint32_t tot = 0, before = 0;
                for (uint32_t t = 0; t < n_tiles; t++) {
                    const uint32_t c = hist_scanned[(size_t)bin * n_tiles + t];
                    if (t < blockIdx.x) before += c;
                    tot += c;
                }
                local[q] = tot;
                s_tile_off[bin] = before;
            } else {
                local[q] = bin_totals[bin];
            }
            sum += local[q];
        }
        const uint32_t inc = wave_inclusive_scan(sum);
        if (lane == WAVE - 1) s_wsum[wave] = inc;
        __syncthreads();
        uint32_t off = inc - sum;
        for (int w = 0; w < wave; w++) off += s_wsum[w];
#pragma unroll
        for (int q = 0; q < PER_THREAD; q++) {
            s_bin_base[threadIdx.x * PER_THREAD + q] = off;
            off += local[q];
        }
    }
    for (int b = threadIdx.x; b < RSW_BINS; b += BLOCK_THREADS) {
#pragma unroll
        for (int w = 0; w < NW; w++) s_cnt[w][b] = 0;
    }
    __syncthreads();
    const uint32_t n = *n_ptr;
    const uint32_t tile = blockIdx.x;
    const uint32_t base = tile * RS_TILE + (uint32_t)wave * (RS_ITEMS * WAVE);
    uint32_t k_[RS_ITEMS], v_[RS_ITEMS], rank[RS_ITEMS];
    bool ok[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const uint32_t i = base + k * WAVE + lane;
        ok[k] = i < n;
        k_[k] = ok[k] ? keys_in[i] : 0xffffffffu;
        v_[k] = ok[k] ? vals_in[i] : 0u;
        const uint32_t digit = (k_[k] >> shift) & (RSW_BINS - 1);
        // lanes of this wave holding the same digit (match-any with 11 ballots)
        unsigned long long m = __ballot(ok[k]);
#pragma unroll
        for (int b = 0; b < RSW_BITS; b++) {
            const unsigned long long bal = __ballot((digit >> b) & 1);
            m &= ((digit >> b) & 1) ? bal : ~bal;
        }
        const uint32_t in_match = (uint32_t)__popcll(m & lanemask_lt());
        uint32_t prev = 0;
        if (ok[k] && in_match == 0) {  // the first lane of every group advances the wave's counter of that digit
            prev = s_cnt[wave][digit];
            s_cnt[wave][digit] = prev + (uint32_t)__popcll(m);
        }
        const int leader = ok[k] ? (int)__builtin_ctzll(m) : lane;
        prev = (uint32_t)__shfl((int)prev, leader, WAVE);
        rank[k] = prev + in_match;
        // the next round of this wave must see the counters written by this one
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    // per digit: exclusive prefix over the waves + global base of the digit + offset of this tile inside the digit
    for (int d = threadIdx.x; d < RSW_BINS; d += BLOCK_THREADS) {
        uint32_t run = s_bin_base[d];
        if constexpr (RAW)
            run += s_tile_off[d];
        else
            run += hist_scanned[(size_t)d * n_tiles + tile];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t c = s_cnt[w][d];
            s_cnt[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        if (ok[k]) {
            const uint32_t digit = (k_[k] >> shift) & (RSW_BINS - 1);
            const uint32_t dst = s_cnt[wave][digit] + rank[k];
            keys_out[dst] = k_[k];
            vals_out[dst] = v_[k];
        }
    }
}

// ---- small batches: the whole sort in ONE workgroup ------------------------------------------------------------------------
// A pass of the radix sort is three launches, a sort two or three passes: for a batch of a few thousand pairs that is 6-9
// launches of a few microseconds each, on a stream whose other kernels take as long - latency-oriented pipelines (the
// reference's own batch of 10 000, results.org:126) spend their time launching.  Up to SMALL_SORT_MAX pairs are sorted by one
// 1024-thread workgroup instead: the pairs become 64-bit words (key << 32 | position) in LDS - 16 Ki of them are 128 KB of the
// 160 KB a gfx950 CU has -, a bitonic network orders them (the position in the low half makes the order stable and the words
// distinct), and the values are gathered by position on the way out.
constexpr uint32_t SMALL_SORT_MAX = 16384;
constexpr int SMALL_SORT_THREADS = 1024;
__global__ void __launch_bounds__(SMALL_SORT_THREADS)
    small_sort_kernel(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, const uint32_t* n_ptr, uint32_t n_pow2,
                      SortGate gate) {
    extern __shared__ __align__(16) unsigned long long ss_words[];
    if (gate.closed()) return;
    uint32_t n = *n_ptr;
    if (n > n_pow2) n = n_pow2;
    // the network only has to cover the smallest power of two that holds the batch
    uint32_t N = 2;
    while (N < n) N <<= 1;
    for (uint32_t i = threadIdx.x; i < N; i += SMALL_SORT_THREADS)
        ss_words[i] = i < n ? ((unsigned long long)keys_in[i] << 32) | (unsigned long long)i : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= N; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (N >> 1); t += SMALL_SORT_THREADS) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // the lower element of pair t at distance j
                const unsigned long long a = ss_words[i], b = ss_words[i + j];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    ss_words[i] = b;
                    ss_words[i + j] = a;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < n; i += SMALL_SORT_THREADS) {
        const unsigned long long w = ss_words[i];
        keys_out[i] = (uint32_t)(w >> 32);
        vals_out[i] = vals_in[(uint32_t)(w & 0xffffffffull)];
    }
}

// ---- any batch: the whole radix sort in ONE launch (rescue path) ---------------------------------------------------------
// Behind a gate that is expected to stay closed (SortGate::rescue) the sort is a single launch that returns at once.  If the
// batch does hold a repeated key after all, its one workgroup runs every pass by itself: stable LSD, 8-bit digits, a histogram
// over the batch, then rounds of 1024 pairs in input order - in-wave ranks from ballots, per-wave digit counts in LDS, running
// offsets per digit carried from round to round.  Some hundred microseconds for 128 Ki pairs instead of tens - for the few
// batches between a change of regime and the host noticing it (DupMonitor).
constexpr int RESCUE_THREADS = 1024;
constexpr int RESCUE_WAVES = RESCUE_THREADS / WAVE;
__global__ void __launch_bounds__(RESCUE_THREADS)
    rescue_sort_kernel(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, const uint32_t* n_ptr, uint32_t n_upper, int n_pass,
                       SortGate gate) {
    __shared__ uint32_t s_cnt[RESCUE_WAVES][RS_BINS];  // per-wave digit counts of a round, then their offsets
    __shared__ uint32_t s_off[RS_BINS];                // where the next pair of every digit goes
    __shared__ uint32_t s_wsum[RESCUE_WAVES];
    if (gate.closed()) return;
    uint32_t n = *n_ptr;
    if (n > n_upper) n = n_upper;
    const int lane = lane_id(), wave = wave_id();
    uint32_t *kin = keys_a, *vin = vals_a, *kout = keys_b, *vout = vals_b;
    for (int p = 0; p < n_pass; p++) {
        const int shift = 8 * p;
        // histogram of the pass
        if (threadIdx.x < RS_BINS) s_off[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += RESCUE_THREADS) atomicAdd(&s_off[(kin[i] >> shift) & 0xff], 1u);
        __syncthreads();
        // exclusive scan of the 256 bins (threads 0..255 = waves 0..3)
        uint32_t v = 0, inc = 0;
        if (threadIdx.x < RS_BINS) {
            v = s_off[threadIdx.x];
            inc = wave_inclusive_scan(v);
            if (lane == WAVE - 1) s_wsum[wave] = inc;
        }
        __syncthreads();
        if (threadIdx.x < RS_BINS) {
            uint32_t off = inc - v;
            for (int w = 0; w < wave; w++) off += s_wsum[w];
            s_off[threadIdx.x] = off;
        }
        __syncthreads();
        // rounds of 1024 pairs in input order
        for (uint32_t base = 0; base < n; base += RESCUE_THREADS) {
            for (int w = 0; w < RESCUE_WAVES; w += RESCUE_THREADS / RS_BINS) s_cnt[w + (int)(threadIdx.x / RS_BINS)][threadIdx.x % RS_BINS] = 0;
            __syncthreads();
            const uint32_t i = base + threadIdx.x;
            const bool ok = i < n;
            const uint32_t k = ok ? kin[i] : 0xffffffffu;
            const uint32_t val = ok ? vin[i] : 0u;
            const uint32_t digit = (k >> shift) & 0xff;
            unsigned long long m = __ballot(ok);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const unsigned long long bal = __ballot((digit >> b) & 1);
                m &= ((digit >> b) & 1) ? bal : ~bal;
            }
            const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt());
            if (ok && rank == 0) s_cnt[wave][digit] = (uint32_t)__popcll(m);
            __syncthreads();
            if (threadIdx.x < RS_BINS) {
                uint32_t run = s_off[threadIdx.x];
                for (int w = 0; w < RESCUE_WAVES; w++) {
                    const uint32_t c = s_cnt[w][threadIdx.x];
                    s_cnt[w][threadIdx.x] = run;
                    run += c;
                }
                s_off[threadIdx.x] = run;
            }
            __syncthreads();
            if (ok) {
                const uint32_t dst = s_cnt[wave][digit] + rank;
                kout[dst] = k;
                vout[dst] = val;
            }
            __syncthreads();
        }
        // the next pass reads what this one wrote: make the writes of all waves visible to all (one workgroup: a barrier and a
        // workgroup-scope fence do)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
        uint32_t* t = kin;
        kin = kout;
        kout = t;
        t = vin;
        vin = vout;
        vout = t;
    }
}

// Workspace + driver for sorting up to `capacity` pairs.  `bits` = number of
// significant key bits.  n is read from device memory so a data-dependent count
// never has to come back to the host.
class RadixSorter {
public:
    void reserve(size_t capacity) {
        if (capacity <= cap_) return;
        if (cap_ == 0) {
            // the one-workgroup sort keeps the whole batch in LDS: as many pairs as the device's LDS holds (WF_SMALL_SORT=0: off)
            int dev = 0;
            (void)hipGetDevice(&dev);
            small_lds_pairs_ = max_lds_per_block(dev) / sizeof(unsigned long long);
            if (const char* e = std::getenv("WF_SMALL_SORT")) small_sort_ = std::atoi(e) != 0;
            if (small_sort_) allow_dynamic_lds(small_sort_kernel, std::min<size_t>(small_lds_pairs_, SMALL_SORT_MAX) * sizeof(unsigned long long));
        }
        cap_ = capacity;
        tiles_ = (capacity + RS_TILE - 1) / RS_TILE;
        keys_tmp_.resize(capacity);
        vals_tmp_.resize(capacity);
        hist_.resize((size_t)RSW_BINS * tiles_);
        bin_totals_.resize(RSW_BINS);
    }
    struct Sorted {
        uint32_t* keys;
        uint32_t* vals;
    };
    // digit width of the passes: 11-bit digits when that saves a pass, 8-bit ones otherwise
    static int digit_bits(int bits) {
        const int p8 = std::max(1, (bits + 7) / 8), p11 = std::max(1, (bits + RSW_BITS - 1) / RSW_BITS);
        return p11 < p8 ? RSW_BITS : 8;
    }
    static int passes(int bits) {
        const int d = digit_bits(bits);
        return std::max(1, (bits + d - 1) / d);
    }
    // sorts the pairs (ping-pong between the caller's arrays and internal buffers) and
    // returns where the sorted pairs live; n_upper bounds the grid, the exact count is
    // read from device memory
    Sorted sort(uint32_t* keys, uint32_t* vals, const uint32_t* n_dev, size_t n_upper, int bits, hipStream_t s, SortGate gate = SortGate{}) {
        reserve(n_upper);
        const uint32_t tiles = (uint32_t)((n_upper + RS_TILE - 1) / RS_TILE);
        uint32_t* kin = keys;
        uint32_t* vin = vals;
        if (tiles == 0) return {kin, vin};
        uint32_t* kout = keys_tmp_.data();
        uint32_t* vout = vals_tmp_.data();
        // Only behind a gate that is expected to stay closed (the caller's DupProbe is on because recent batches held no key
        // twice): one launch that returns at once instead of six.  A network of log^2 stages in one workgroup is slower than the
        // multi-workgroup radix passes when it does have to run (measured: keyed map, 10 000-tuple batches full of repeats,
        // 258 -> 126 M tuples/s), so streams whose batches repeat keys never take it.
        if (small_sort_ && gate.word != nullptr && n_upper <= (size_t)SMALL_SORT_MAX && n_upper <= small_lds_pairs_) {
            uint32_t n_pow2 = 2;
            while (n_pow2 < n_upper) n_pow2 <<= 1;
            hipLaunchKernelGGL(small_sort_kernel, dim3(1), dim3(SMALL_SORT_THREADS), (size_t)n_pow2 * sizeof(unsigned long long), s, kin, vin, kout, vout,
                               n_dev, n_pow2, gate);
            return {kout, vout};
        }
        if (gate.word != nullptr && gate.rescue != 0) {
            const int n_pass8 = std::max(1, (bits + 7) / 8);
            hipLaunchKernelGGL(rescue_sort_kernel, dim3(1), dim3(RESCUE_THREADS), 0, s, kin, vin, kout, vout, n_dev, (uint32_t)n_upper, n_pass8, gate);
            return (n_pass8 & 1) ? Sorted{kout, vout} : Sorted{kin, vin};
        }
        const int d = digit_bits(bits), n_pass = passes(bits);
        // small batches (latency-oriented pipelines) are launch-bound: two launches per pass instead of three
        const bool fused = tiles <= FUSE_TILES;
        for (int p = 0; p < n_pass; p++) {
            const int shift = d * p;
            if (d == 8) {
                hipLaunchKernelGGL(radix_hist_kernel, dim3(tiles), dim3(BLOCK_THREADS), 0, s, kin, n_dev, tiles, shift, hist_.data(), gate);
                if (fused) {
                    hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3(tiles), dim3(BLOCK_THREADS), 0, s, kin, vin, kout, vout, n_dev, tiles,
                                       shift, hist_.data(), bin_totals_.data(), gate);
                } else {
                    hipLaunchKernelGGL(radix_bin_scan_kernel, dim3(RS_BINS), dim3(BLOCK_THREADS), 0, s, hist_.data(), tiles, bin_totals_.data(), gate);
                    hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3(tiles), dim3(BLOCK_THREADS), 0, s, kin, vin, kout, vout, n_dev, tiles,
                                       shift, hist_.data(), bin_totals_.data(), gate);
                }
            } else {
                hipLaunchKernelGGL(radix_hist_wide_kernel, dim3(tiles), dim3(BLOCK_THREADS), 0, s, kin, n_dev, tiles, shift, hist_.data(), gate);
                if (fused) {
                    hipLaunchKernelGGL(radix_scatter_wide_kernel<true>, dim3(tiles), dim3(BLOCK_THREADS), 0, s, kin, vin, kout, vout, n_dev,
                                       tiles, shift, hist_.data(), bin_totals_.data(), gate);
                } else {
                    hipLaunchKernelGGL(radix_bin_scan_kernel, dim3(RSW_BINS), dim3(BLOCK_THREADS), 0, s, hist_.data(), tiles, bin_totals_.data(), gate);
                    hipLaunchKernelGGL(radix_scatter_wide_kernel<false>, dim3(tiles), dim3(BLOCK_THREADS), 0, s, kin, vin, kout, vout, n_dev,
                                       tiles, shift, hist_.data(), bin_totals_.data(), gate);
                }
            }
            std::swap(kin, kout);
            std::swap(vin, vout);
        }
        return {kin, vin};
    }
    int kernels_per_sort(int bits) const { return 3 * passes(bits); }  // upper bound (2 per pass for small batches, 1 for the smallest)
    // batches of at most this many tiles (2048 pairs each) skip the scan kernel: the scatter blocks read the raw counts
    static constexpr uint32_t FUSE_TILES = 16;

private:
    size_t cap_ = 0, tiles_ = 0, small_lds_pairs_ = 0;
    bool small_sort_ = true;
    DeviceBuffer<uint32_t> keys_tmp_, vals_tmp_, hist_, bin_totals_;
};

// ---------------------------------------------------------------------------
// (slot, timestamp) order for batches whose keys arrive OUT of timestamp order
// ---------------------------------------------------------------------------
// The keyed engines give every key a contiguous run in ARRIVAL order (stable sort by slot).  A batch that merges several
// unsynchronised lanes (the device KeyBy of a DEFAULT graph, an exchange round) is a concatenation of sorted pieces: every
// run is out of timestamp order, and the per-run repairs (an insertion sort in the non-incremental engine, a sequential
// fold in the incremental one) are quadratic / serial in the run length.  Instead the whole batch is ordered by
// (slot, timestamp) with two stable radix sorts - by the low bits of (ts - base), then by slot - after which every run is in
// timestamp order and the engines' parallel in-order paths apply.  Time is counted in `unit`s (1: timestamps; the pane length:
// the tuples of a pane keep their arrival order, which is what an order-sensitive combine folds them in); `bits` of it are
// sorted (base = batch watermark - (2^bits - 1)); tuples outside that span are clamped and reported (the engines then fall
// back to their repairs for the affected runs, and the caller widens the span).  Ties keep their arrival order.
// Semantics: the tuples of ONE batch are then treated as if they had arrived in (unit-granular) timestamp order, so none of
// them is late with respect to another tuple of the same batch; lateness against earlier batches follows the usual rules.
template <typename T>
__global__ void __launch_bounds__(BLOCK_THREADS)
    ts_keys_kernel(const BatchHeader* hdr, const T* data, uint32_t bits, uint64_t unit, uint32_t* keys, uint32_t* idx, volatile uint32_t* clamp_seen,
                   uint32_t batch_no) {
    const uint32_t n = hdr->count;
    const uint64_t wm = hdr->watermark / unit;
    const uint64_t span = bits >= 32 ? 0xffffffffull : ((1ull << bits) - 1ull);
    const uint64_t base = wm > span ? wm - span : 0ull;
    bool clamped = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t ts = (uint64_t)std::get<2>(data[i].getControlFields()) / unit;
        uint64_t k = ts > base ? ts - base : 0ull;
        if (ts < base || k > span) clamped = true;
        if (k > span) k = span;
        keys[i] = (uint32_t)k;
        idx[i] = i;
    }
    if (__ballot(clamped) != 0 && lane_id() == 0 && clamp_seen != nullptr) *clamp_seen = batch_no;
}
__global__ void __launch_bounds__(BLOCK_THREADS)
    gather_u32_kernel(const BatchHeader* hdr, const uint32_t* src, const uint32_t* perm, uint32_t* dst) {
    const uint32_t n = hdr->count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[perm[i]];
}

class TsPresorter {
public:
    void reserve(size_t capacity) {
        if (capacity <= cap_) return;
        cap_ = capacity;
        keys_.resize(capacity);
        idx_.resize(capacity);
        slot2_.resize(capacity);
        by_ts_.reserve(capacity);
        by_slot_.reserve(capacity);
    }
    // `slot[i]` = slot of tuple i (arrival order).  Returns (sorted slots, permutation): position p of the result holds tuple
    // perm[p]; slots ascending, timestamps ascending inside a slot (as far as `bits` of time reach).
    template <typename T>
    RadixSorter::Sorted sort(const BatchHeader* hdr, const T* data, const uint32_t* n_dev, size_t n_upper, const uint32_t* slot, int slot_bits, int ts_bits,
                             uint64_t unit, volatile uint32_t* clamp_seen, uint32_t batch_no, hipStream_t s) {
        reserve(n_upper);
        const unsigned grid = grid_for(std::max<size_t>(n_upper, 1));
        hipLaunchKernelGGL((ts_keys_kernel<T>), dim3(grid), dim3(BLOCK_THREADS), 0, s, hdr, data, (uint32_t)ts_bits, unit ? unit : 1, keys_.data(), idx_.data(),
                           clamp_seen, batch_no);
        const auto by_ts = by_ts_.sort(keys_.data(), idx_.data(), n_dev, n_upper, ts_bits, s);
        hipLaunchKernelGGL(gather_u32_kernel, dim3(grid), dim3(BLOCK_THREADS), 0, s, hdr, slot, by_ts.vals, slot2_.data());
        return by_slot_.sort(slot2_.data(), by_ts.vals, n_dev, n_upper, slot_bits, s);
    }
    int kernels(int slot_bits, int ts_bits) const { return 2 + by_ts_.kernels_per_sort(ts_bits) + by_slot_.kernels_per_sort(slot_bits); }

private:
    size_t cap_ = 0;
    DeviceBuffer<uint32_t> keys_, idx_, slot2_;
    RadixSorter by_ts_, by_slot_;
};

// When to pre-sort: a pinned word the engine's kernels write the batch number to whenever they meet a run that is out of
// timestamp order (`seen`), another one for clamped timestamps (`clamped`).  The pre-sort stays on while disorder keeps
// showing; it is switched off for ONE probing batch after `linger` quiet batches (the pre-sort hides the disorder it cures),
// and every time the probe finds disorder again the next linger doubles.
class DisorderMonitor {
public:
    DisorderMonitor() {
        WF_HIP_CHECK(hipHostMalloc((void**)&words_, 64, hipHostMallocDefault));
        words_[0] = words_[1] = 0;
        if (const char* e = std::getenv("WF_TS_PRESORT")) mode_ = std::atoi(e);  // 0: never, 1: adaptive (default), 2: always
    }
    ~DisorderMonitor() {
        if (words_) (void)hipHostFree((void*)words_);
    }
    DisorderMonitor(const DisorderMonitor&) = delete;
    DisorderMonitor& operator=(const DisorderMonitor&) = delete;
    volatile uint32_t* seen() const { return words_; }
    volatile uint32_t* clamped() const { return words_ + 1; }
    // called once per batch, before its kernels are enqueued: pre-sort this batch?
    // sticky: once disorder has been seen the pre-sort stays on (engines whose repair of an unsorted run is quadratic)
    void set_sticky(bool b) { sticky_ = b; }
    // the caller knows its input is a merge of unsynchronised lanes: pre-sort from the first batch on
    void expect_disorder() {
        if (mode_ == 1) mode_ = 2;
    }
    bool decide(uint32_t batch_no) {
        if (mode_ == 0) return false;
        if (mode_ == 2) return true;
        if (sticky_ && on_since_ != 0) return true;
        const uint32_t seen = words_[0];
        if (seen != 0 && seen != last_seen_) {  // a new report
            if (on_since_ != 0 && probing_) linger_ = std::min<uint32_t>(linger_ * 2, 1u << 20);
            last_seen_ = seen;
            quiet_from_ = batch_no;
            if (on_since_ == 0) on_since_ = batch_no;
            probing_ = false;
        }
        if (on_since_ == 0) return false;
        if (batch_no - quiet_from_ > linger_) {  // quiet for long: let one batch through unsorted and see
            probing_ = true;
            quiet_from_ = batch_no;
            return false;
        }
        return true;
    }
    int ts_bits() const { return words_[1] != 0 ? 32 : 22; }

private:
    volatile uint32_t* words_ = nullptr;
    int mode_ = 1;
    uint32_t last_seen_ = 0, quiet_from_ = 0, on_since_ = 0, linger_ = 1024;
    bool probing_ = false, sticky_ = false;
};

}  // namespace gpu
}  // namespace wf
