// FlatFAT on the device.
//
// Two layers:
//  (1) device functions operating on one tree stored as a flat array of 2*cap
//      nodes (node 1 = root, leaves at [cap, 2cap)): fat_insert / fat_query /
//      fat_query_circular.  They are what a single thread uses to maintain the
//      tree of *its* key inside the keyed engine (key_ffat_gpu.hpp), where one
//      slab holds the trees of every key ("a slab of trees
//      indexed by dense key id" instead of one cudaMalloc per key).
//  (2) FlatFAT_GPU<result_t, comb_t>: one whole tree processed by many threads,
//      capability parity with /root/reference/wf/flatfat_gpu.hpp:49-455
//      (build / update / batched window results).  The reference launches one
//      kernel per tree level (K2/K3: InitTreeLevel / UpdateTreeLevel, :50-78,
//      log2(n) launches, update on the default stream); here a tile of 2*256
//      leaves is reduced through all its levels inside LDS by one workgroup and
//      a second launch finishes the top of the tree: 2 launches for any n.
//      K4 (ComputeResults, :87-130) becomes fat_results_kernel: one thread per
//      window, O(log n) ordered range query, windows addressed circularly.
// The combine function is treated as associative and NON-commutative: every
// query keeps left and right accumulators so operand order is arrival order.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <tuple>
#include <vector>

#include "hip_utils.hpp"
#include "kernels_basic.hpp"

namespace wf {
namespace gpu {

template <typename X>
__host__ __device__ __forceinline__ auto cf_key(const X& x) {
    return std::get<0>(x.getControlFields());
}
template <typename X>
__host__ __device__ __forceinline__ uint64_t cf_id(const X& x) {
    return std::get<1>(x.getControlFields());
}
template <typename X>
__host__ __device__ __forceinline__ uint64_t cf_ts(const X& x) {
    return std::get<2>(x.getControlFields());
}

// out = a (+) b with a older than b; control fields: key of a, id 0, latest ts
template <typename R, typename C>
__host__ __device__ __forceinline__ void fat_combine(const C& comb, const R& a, const R& b, R& out) {
    const uint64_t ta = cf_ts(a), tb = cf_ts(b);
    out.setControlFields(cf_key(a), 0, ta > tb ? ta : tb);
    comb(a, b, out);
}

// write leaf `pos` and recompute its ancestors
template <typename R, typename C>
__device__ __forceinline__ void fat_insert(R* tree, uint32_t cap, uint32_t pos, const R& v, const C& comb) {
    uint32_t node = cap + pos;
    tree[node] = v;
    R cur = v;
    while (node > 1) {
        const uint32_t sib = node ^ 1u;
        R parent;
        if (node & 1u)
            fat_combine(comb, tree[sib], cur, parent);
        else
            fat_combine(comb, cur, tree[sib], parent);
        node >>= 1;
        tree[node] = parent;
        cur = parent;
    }
}

// ordered aggregate of leaves [l, r), 0 <= l < r <= cap
template <typename R, typename C>
__device__ __forceinline__ void fat_query(const R* tree, uint32_t cap, uint32_t l, uint32_t r, R& out, const C& comb) {
    R left, right;
    bool hl = false, hr = false;
    for (l += cap, r += cap; l < r; l >>= 1, r >>= 1) {
        if (l & 1u) {
            if (hl) {
                R tmp;
                fat_combine(comb, left, tree[l], tmp);
                left = tmp;
            } else {
                left = tree[l];
                hl = true;
            }
            l++;
        }
        if (r & 1u) {
            --r;
            if (hr) {
                R tmp;
                fat_combine(comb, tree[r], right, tmp);
                right = tmp;
            } else {
                right = tree[r];
                hr = true;
            }
        }
    }
    if (hl && hr)
        fat_combine(comb, left, right, out);
    else
        out = hl ? left : right;
}

// ordered aggregate of `count` leaves starting at circular position `first`
template <typename R, typename C>
__device__ __forceinline__ void fat_query_circular(const R* tree, uint32_t cap, uint32_t first, uint32_t count, R& out, const C& comb) {
    first &= (cap - 1);
    if (first + count <= cap) {
        fat_query(tree, cap, first, first + count, out, comb);
    } else {
        R a, b;
        fat_query(tree, cap, first, cap, a, comb);
        fat_query(tree, cap, 0, first + count - cap, b, comb);
        fat_combine(comb, a, b, out);
    }
}

// ---- whole-tree kernels --------------------------------------------------------
constexpr int FAT_TILE_LEAVES = 2 * BLOCK_THREADS;  // 512 leaves reduced per workgroup

// K2': build every level of the sub-tree above one tile of leaves inside LDS.
// tree[cap + i] must already hold the leaves.  For cap < FAT_TILE_LEAVES one block
// builds the whole tree.
template <typename R, typename C>
__global__ void __launch_bounds__(BLOCK_THREADS) fat_build_tiles_kernel(R* tree, uint32_t cap, uint32_t first_tile, C comb) {
    __shared__ R s_nodes[FAT_TILE_LEAVES];  // level buffer, reduced in place
    const uint32_t tile_leaves = cap < (uint32_t)FAT_TILE_LEAVES ? cap : (uint32_t)FAT_TILE_LEAVES;
    const uint32_t first_leaf = (first_tile + blockIdx.x) * tile_leaves;
    for (uint32_t i = threadIdx.x; i < tile_leaves; i += BLOCK_THREADS) s_nodes[i] = tree[cap + first_leaf + i];
    __syncthreads();
    // level with `width` nodes of this tile lives at global index (cap >> lvl) + (first_leaf >> lvl)
    uint32_t width = tile_leaves >> 1;
    uint32_t lvl = 1;
    while (width >= 1) {
        R tmp;
        const bool act = threadIdx.x < width;
        if (act) fat_combine(comb, s_nodes[2 * threadIdx.x], s_nodes[2 * threadIdx.x + 1], tmp);
        __syncthreads();
        if (act) {
            s_nodes[threadIdx.x] = tmp;
            tree[(cap >> lvl) + (first_leaf >> lvl) + threadIdx.x] = tmp;
        }
        __syncthreads();
        if (width == 1) break;
        width >>= 1;
        lvl++;
    }
}

// top of the tree: nodes [1, top_nodes) where top_nodes = cap / FAT_TILE_LEAVES
// (their children at index >= top_nodes were written by fat_build_tiles_kernel)
template <typename R, typename C>
__global__ void __launch_bounds__(BLOCK_THREADS) fat_build_top_kernel(R* tree, uint32_t top_nodes, C comb) {
    for (uint32_t width = top_nodes >> 1; width >= 1; width >>= 1) {
        // nodes [width, 2*width)
        for (uint32_t i = threadIdx.x; i < width; i += BLOCK_THREADS) {
            const uint32_t node = width + i;
            R tmp;
            fat_combine(comb, tree[2 * node], tree[2 * node + 1], tmp);
            tree[node] = tmp;
        }
        __threadfence_block();
        __syncthreads();
        if (width == 1) break;
    }
}

// K4': window w of the batch = `win` leaves starting at circular leaf first + w*slide;
// results[w] keeps its pre-stamped control fields (key, gwid, ts come from the host)
template <typename R, typename C>
__global__ void __launch_bounds__(BLOCK_THREADS)
    fat_results_kernel(const R* tree, uint32_t cap, uint32_t first, uint32_t win, uint32_t slide, uint32_t n_windows, R* results, C comb) {
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_windows; w += gridDim.x * blockDim.x) {
        R agg;
        fat_query_circular(tree, cap, first + w * slide, win, agg, comb);
        const auto key = cf_key(results[w]);
        const uint64_t gwid = cf_id(results[w]);
        agg.setControlFields(key, gwid, cf_ts(agg));
        results[w] = agg;
    }
}

// One device-resident aggregator tree.  Leaves are a circular buffer of lifted
// values; build() loads a full batch of leaves, update() overwrites the oldest
// `n` leaves with new ones, results are computed for `n_windows` windows per call
// and copied back asynchronously.
template <typename result_t, typename comb_t>
class FlatFAT_GPU {
public:
    FlatFAT_GPU(comb_t comb, size_t capacity_leaves, size_t max_windows_per_batch, hipStream_t stream)
        : comb_(comb), stream_(stream), max_windows_(max_windows_per_batch) {
        cap_ = 1;
        while (cap_ < capacity_leaves) cap_ <<= 1;
        tree_.resize(2 * cap_);
        tree_.zero(stream_);
        d_results_.resize(max_windows_);
        h_leaves_.resize(cap_);
        h_results_.resize(max_windows_);
    }
    size_t capacity() const { return cap_; }
    size_t size() const { return count_; }

    // load `n` leaves from scratch (tree emptied first) - reference build()
    void build(const result_t* leaves, size_t n) {
        front_ = 0;
        count_ = 0;
        append(leaves, n);
    }
    // drop the `n_remove` oldest leaves and append `n` new ones - reference update()
    void update(const result_t* leaves, size_t n, size_t n_remove) {
        if (n_remove > count_) n_remove = count_;
        front_ = (front_ + n_remove) & (cap_ - 1);
        count_ -= n_remove;
        append(leaves, n);
    }
    // compute n_windows windows of `win` leaves sliding by `slide` from the oldest live
    // leaf; `stamped[w]` carries (key, gwid) of window w.  Asynchronous: waitResults().
    void computeResults(const result_t* stamped, size_t n_windows, size_t win, size_t slide) {
        if (n_windows > max_windows_) failwith("FlatFAT_GPU: too many windows in one batch");
        std::memcpy(h_results_.data(), stamped, n_windows * sizeof(result_t));
        WF_HIP_CHECK(hipMemcpyAsync(d_results_.data(), h_results_.data(), n_windows * sizeof(result_t), hipMemcpyHostToDevice, stream_));
        hipLaunchKernelGGL((fat_results_kernel<result_t, comb_t>), dim3(grid_for(n_windows)), dim3(BLOCK_THREADS), 0, stream_, tree_.data(),
                           (uint32_t)cap_, (uint32_t)front_, (uint32_t)win, (uint32_t)slide, (uint32_t)n_windows, d_results_.data(), comb_);
        WF_HIP_CHECK(hipMemcpyAsync(h_results_.data(), d_results_.data(), n_windows * sizeof(result_t), hipMemcpyDeviceToHost, stream_));
        n_pending_ = n_windows;
        kernels_ += 1;
        bytes_hd_ += n_windows * sizeof(result_t);
        bytes_dh_ += n_windows * sizeof(result_t);
    }
    // block until the last computeResults() finished; returns the host results
    const result_t* waitResults(size_t& n) {
        WF_HIP_CHECK(hipStreamSynchronize(stream_));
        n = n_pending_;
        n_pending_ = 0;
        return h_results_.data();
    }
    uint64_t num_kernels() const { return kernels_; }
    uint64_t bytes_hd() const { return bytes_hd_; }
    uint64_t bytes_dh() const { return bytes_dh_; }

private:
    void append(const result_t* leaves, size_t n) {
        if (count_ + n > cap_) failwith("FlatFAT_GPU: capacity exceeded");
        if (n == 0) return;
        // previous H2D from the pinned staging buffer must have completed
        WF_HIP_CHECK(hipStreamSynchronize(stream_));
        const size_t pos = (front_ + count_) & (cap_ - 1);
        const size_t first = std::min(n, cap_ - pos);
        std::memcpy(h_leaves_.data(), leaves, n * sizeof(result_t));
        WF_HIP_CHECK(hipMemcpyAsync(tree_.data() + cap_ + pos, h_leaves_.data(), first * sizeof(result_t), hipMemcpyHostToDevice, stream_));
        if (first < n)
            WF_HIP_CHECK(hipMemcpyAsync(tree_.data() + cap_, h_leaves_.data() + first, (n - first) * sizeof(result_t), hipMemcpyHostToDevice,
                                        stream_));
        bytes_hd_ += n * sizeof(result_t);
        count_ += n;
        rebuild(pos, n);
    }
    // recompute the ancestors of the circular leaf range [pos, pos+n)
    void rebuild(size_t pos, size_t n) {
        const size_t tile = std::min<size_t>(cap_, FAT_TILE_LEAVES);
        const size_t n_tiles = cap_ / tile;
        // dirty tiles (at most two contiguous runs because of the wrap-around)
        auto launch_tiles = [&](size_t first_tile, size_t cnt) {
            if (cnt == 0) return;
            hipLaunchKernelGGL((fat_build_tiles_kernel<result_t, comb_t>), dim3((unsigned)cnt), dim3(BLOCK_THREADS), 0, stream_, tree_.data(), (uint32_t)cap_,
                               (uint32_t)first_tile, comb_);
            kernels_++;
        };
        const size_t t0 = pos / tile;
        const size_t t1 = (pos + n - 1) / tile;  // may exceed n_tiles - 1 when wrapping
        if (n >= cap_ || t1 - t0 + 1 >= n_tiles) {
            launch_tiles(0, n_tiles);
        } else if (t1 < n_tiles) {
            launch_tiles(t0, t1 - t0 + 1);
        } else {
            launch_tiles(t0, n_tiles - t0);
            launch_tiles(0, t1 - n_tiles + 1);
        }
        if (n_tiles > 1) {
            hipLaunchKernelGGL((fat_build_top_kernel<result_t, comb_t>), dim3(1), dim3(BLOCK_THREADS), 0, stream_, tree_.data(),
                               (uint32_t)n_tiles, comb_);
            kernels_++;
        }
    }
    comb_t comb_;
    hipStream_t stream_;
    size_t cap_ = 1, front_ = 0, count_ = 0, max_windows_ = 0, n_pending_ = 0;
    DeviceBuffer<result_t> tree_;
    DeviceBuffer<result_t> d_results_;
    PinnedBuffer<result_t> h_leaves_;
    PinnedBuffer<result_t> h_results_;
    uint64_t kernels_ = 0, bytes_hd_ = 0, bytes_dh_ = 0;
};

}  // namespace gpu
}  // namespace wf
