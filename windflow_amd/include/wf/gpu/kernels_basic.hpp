// gfx950 kernels of the batched stateless operators: in-place / out-of-place map
// and the fused filter + stream compaction.
//
// What they replace in the reference:
//   K5 run_map_kernel_ip   (wf/map_gpu_node.hpp:56-63)
//   K6 run_map_kernel_nip  (wf/map_gpu_node.hpp:72-79)
//   K9 run_filter_kernel   (wf/filter_gpu_node.hpp:50-59)  + host-side re-emission
//   K11 compact (unused/buggy, wf/filter_gpu_node.hpp:81-98) and the Thrust copy_if
//   of the benchmarks (gpu_filter_stateless.cpp:481-492).
//
// Design for CDNA4: wave64 everywhere.  The filter evaluates the predicate,
// ranks survivors inside the wave with one __ballot + popcount of the lower
// lanes, combines the per-wave counts through LDS, obtains the tile's global
// offset with a single-pass decoupled look-back over 64-bit {status,value}
// descriptors (agent-scope relaxed atomics: value and flag share one word, so no
// fence is needed), and scatters the survivors in stable order.  Tiles draw a
// ticket from a global counter, so a tile only ever waits for tiles that have
// already started: forward progress does not depend on dispatch order.  The
// final count goes to the output batch header on the device and to a pinned host
// word, so neither a device->device successor nor the host needs a sync to size
// its next step.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "hip_utils.hpp"

namespace wf {
namespace gpu {

constexpr int WAVE = 64;
constexpr int BLOCK_THREADS = 256;  // 4 waves: one per SIMD of a CU
constexpr int FILTER_ITEMS = 4;     // tuples per thread in the compaction tile
constexpr int FILTER_TILE = BLOCK_THREADS * FILTER_ITEMS;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }
__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

// grid size for a memory-bound grid-stride kernel over n items (guide: cap at
// ~8 blocks per CU, 256 CUs)
inline unsigned grid_for(size_t n, unsigned block = BLOCK_THREADS, unsigned per_thread = 1) {
    size_t g = (n + (size_t)block * per_thread - 1) / ((size_t)block * per_thread);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (unsigned)g;
}

// ---- K5: in-place map -------------------------------------------------------
template <typename T, typename F>
__global__ void __launch_bounds__(BLOCK_THREADS) map_inplace_kernel(BatchHeader* hdr, T* data, F f) {
    const uint32_t n = hdr->count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        T t = data[i];
        f(t);
        data[i] = t;
    }
}

// ---- K6: out-of-place map ----------------------------------------------------
template <typename T, typename R, typename F>
__global__ void __launch_bounds__(BLOCK_THREADS)
    map_kernel(const BatchHeader* in_hdr, const T* in, BatchHeader* out_hdr, R* out, F f) {
    const uint32_t n = in_hdr->count;
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_hdr = *in_hdr;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const T t = in[i];
        R r{};
        f(t, r);
        out[i] = r;
    }
}

// A tuple held in registers as plain words between a global load and a later store.  A struct copied through a
// private temporary under divergent control flow is not always promoted to registers by the compiler and then
// lives in scratch memory (a memory round trip per tuple); an unrolled word array always is.
template <typename T>
struct TupleWords {
    static constexpr int N = sizeof(T) / 4;
    static constexpr int REM = sizeof(T) % 4;
    typedef uint32_t __attribute__((may_alias)) word_t;
    typedef uint8_t __attribute__((may_alias)) byte_t;
    uint32_t w[N > 0 ? N : 1];
    uint8_t tail[REM > 0 ? REM : 1];
    __device__ __forceinline__ void load(const T* p) {
#pragma unroll
        for (int k = 0; k < N; k++) w[k] = reinterpret_cast<const word_t*>(p)[k];
#pragma unroll
        for (int k = 0; k < REM; k++) tail[k] = reinterpret_cast<const byte_t*>(p)[4 * N + k];
    }
    __device__ __forceinline__ void store(T* p) const {
#pragma unroll
        for (int k = 0; k < N; k++) reinterpret_cast<word_t*>(p)[k] = w[k];
#pragma unroll
        for (int k = 0; k < REM; k++) reinterpret_cast<byte_t*>(p)[4 * N + k] = tail[k];
    }
};

// ---- K9: fused filter + stable stream compaction ----------------------------
// workspace layout: ws[0] = ticket counter, ws[1..] = tile descriptors.
// descriptor = (status << 32) | value; status 0 = empty, 1 = tile aggregate,
// 2 = inclusive prefix.  The workspace must be zeroed before every launch
// (hipMemsetAsync on the same stream).
constexpr unsigned long long DESC_AGG = 1ull << 32;
constexpr unsigned long long DESC_INC = 2ull << 32;

inline size_t filter_workspace_words(size_t n_upper) { return 1 + (n_upper + FILTER_TILE - 1) / FILTER_TILE + 1; }

template <typename T, typename P>
__global__ void __launch_bounds__(BLOCK_THREADS)
    filter_compact_kernel(const BatchHeader* in_hdr, const T* in, BatchHeader* out_hdr, T* out, volatile uint32_t* count_mirror,
                          unsigned long long* ws, P pred) {
    __shared__ uint32_t s_counts[FILTER_ITEMS * (BLOCK_THREADS / WAVE)];
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_base;
    const uint32_t n = in_hdr->count;
    const uint32_t n_tiles = (n + FILTER_TILE - 1) / FILTER_TILE;
    if (threadIdx.x == 0) s_tile = (uint32_t)atomicAdd(&ws[0], 1ull);
    __syncthreads();
    const uint32_t tile = s_tile;
    if (n_tiles == 0) {
        if (tile == 0 && threadIdx.x == 0) {
            *out_hdr = *in_hdr;
            out_hdr->count = 0;
            if (count_mirror) *count_mirror = 0;
        }
        return;
    }
    if (tile >= n_tiles) return;
    unsigned long long* desc = ws + 1;
    const int lane = lane_id();
    const int wave = wave_id();
    const uint32_t base = tile * FILTER_TILE;

    // 1. predicate + in-wave rank (striped: item k of thread t is element base + k*256 + t)
    TupleWords<T> vals[FILTER_ITEMS];
    bool keep[FILTER_ITEMS];
    uint32_t rank[FILTER_ITEMS];
#pragma unroll
    for (int k = 0; k < FILTER_ITEMS; k++) {
        const uint32_t idx = base + k * BLOCK_THREADS + threadIdx.x;
        keep[k] = false;
        if (idx < n) {
            vals[k].load(in + idx);
            keep[k] = pred(in[idx], idx);
        }
        const unsigned long long b = __ballot(keep[k]);
        rank[k] = (uint32_t)__popcll(b & lanemask_lt());
        if (lane == 0) s_counts[k * (BLOCK_THREADS / WAVE) + wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
    // 2. exclusive scan of the 16 (item, wave) counts: every thread does it redundantly from LDS broadcasts
    uint32_t my_off[FILTER_ITEMS];
    uint32_t running = 0;
#pragma unroll
    for (int k = 0; k < FILTER_ITEMS; k++) {
#pragma unroll
        for (int w = 0; w < BLOCK_THREADS / WAVE; w++) {
            const uint32_t c = s_counts[k * (BLOCK_THREADS / WAVE) + w];
            if (w == wave) my_off[k] = running;
            running += c;
        }
    }
    const uint32_t tile_total = running;
    // 3. decoupled look-back by wave 0
    if (wave == 0) {
        uint32_t exclusive = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&desc[0], DESC_INC | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&desc[tile], DESC_AGG | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int look = (int)tile - 1;  // highest predecessor not yet accounted for
            while (true) {
                const int p = look - lane;
                unsigned long long d = DESC_INC;  // lanes past tile 0 contribute an inclusive prefix of 0
                if (p >= 0) {
                    do {
                        d = __hip_atomic_load(&desc[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((d >> 32) == 0);
                }
                const unsigned long long inc_mask = __ballot((d >> 32) == 2);
                // lanes up to and including the first inclusive one contribute
                const int stop = inc_mask ? (int)__builtin_ctzll(inc_mask) : WAVE;
                uint32_t v = (lane <= stop) ? (uint32_t)(d & 0xffffffffu) : 0u;
#pragma unroll
                for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
                exclusive += v;
                if (inc_mask) break;
                look -= WAVE;
            }
            if (lane == 0)
                __hip_atomic_store(&desc[tile], DESC_INC | (unsigned long long)(exclusive + tile_total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_base = exclusive;
            if (tile == n_tiles - 1) {
                *out_hdr = *in_hdr;
                out_hdr->count = exclusive + tile_total;
                if (count_mirror) *count_mirror = exclusive + tile_total;
            }
        }
    }
    __syncthreads();
    // 4. stable scatter
    const uint32_t gbase = s_base;
#pragma unroll
    for (int k = 0; k < FILTER_ITEMS; k++) {
        if (keep[k]) vals[k].store(out + gbase + my_off[k] + rank[k]);
    }
}

// adapters so user functors can be written in either accepted form
//   Filter_GPU:  void(const T&, bool&)  (reference: wf/filter_gpu_node.hpp:102-105)  or  bool(const T&)
template <typename T, typename F>
struct PredicateAdapter {
    F f;
    // the compaction kernel passes the position of the tuple too (used by FlagPredicate)
    __host__ __device__ bool operator()(const T& t, uint32_t) const { return (*this)(t); }
    __host__ __device__ bool operator()(const T& t) const {
        if constexpr (std::is_same_v<decltype(call(f, t, 0)), bool>) {
            return call(f, t, 0);
        } else {
            bool m = false;
            f(t, m);
            return m;
        }
    }

private:
    template <typename G>
    __host__ __device__ static auto call(const G& g, const T& t, int) -> decltype(g(t)) {
        return g(t);
    }
    template <typename G>
    __host__ __device__ static void call(const G&, const T&, long) {}
};

// keep tuple i iff flags[i] != 0 (keyed filters evaluate the predicate in a separate,
// per-key ordered pass and compact afterwards)
template <typename T>
struct FlagPredicate {
    const uint8_t* flags;
    __host__ __device__ bool operator()(const T&, uint32_t i) const { return flags[i] != 0; }
};

}  // namespace gpu
}  // namespace wf
