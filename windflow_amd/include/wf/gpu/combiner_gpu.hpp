// Pane_Combiner_GPU: the combiner in front of a keyed exchange.
//
// A keyed aggregation behind a KeyBy across GPUs ships every tuple to the GPU that owns its key: with skewed keys (zipf:
// /root/reference/src/GPU_Tests/zipf.hpp:14-21) the link into the owner of the hot keys carries most of the stream, and one key
// can never use more than one GPU.  When the aggregation is lift + associative combine over time-based panes (Key_FFAT,
// /root/reference/wf/win_seqffat.hpp:422-513 folds the tuples of a quantum into one partial result before anything else
// happens to them), that first fold can run where the tuples ARE: every producer lane folds the tuples of a batch per
// (key, pane) on its own GPU and ships the partial results; the owner combines partials instead of lifting tuples.  It is
// MapReduce's combiner: a hot key costs its link one partial per pane per batch however many tuples it has, and its lifting
// work is spread over all the GPUs that see it.  The reference has no counterpart (single process, no exchange).
//
//   PaneCombinerGPU_Builder(lift, comb).withTBWindows(win, slide)      tuples  -> partials (result_t: key, pane, latest ts)
//   ExchangeGPU_Builder<result_t>(transport) ...                       partials travel
//   KeyFFATGPU_Builder(PartialLift<result_t>(), comb).withTBWindows(win, slide, delay) ...   windows over partials
//
// Per batch, on the device: the tuples are grouped by a hash of their key (stable radix sort: a key's tuples stay in arrival
// order, so its panes ascend when the lane is in timestamp order), a wave-parallel segmented scan folds runs of equal (key,
// pane) 64 tuples at a time, the surviving partials are compacted in order, and a second pass of the same kernel folds the
// partials of runs longer than a wave: a run of n tuples leaves as ceil(n / 4096) partials, in order.  Nothing needs to be
// exact for correctness - the owner combines whatever partials arrive, in arrival order - so hash collisions and runs cut
// at chunk borders only cost a few extra partials.
#pragma once
#include <numeric>

#include "batch_ops.hpp"
#include "flatfat_gpu.hpp"
#include "primitives.hpp"

namespace wf {
namespace gpu {

// the value of `v` held by lane (lane - d); any trivially copyable struct made of 32-bit words
template <typename X>
__device__ __forceinline__ X comb_shfl_up(const X& v, int d) {
    static_assert(sizeof(X) % 4 == 0, "wave shuffles move 32-bit words");
    int w[sizeof(X) / 4];
    __builtin_memcpy(w, &v, sizeof(X));
#pragma unroll
    for (size_t k = 0; k < sizeof(X) / 4; k++) w[k] = __shfl_up(w[k], d, WAVE);
    X o;
    __builtin_memcpy(&o, w, sizeof(X));
    return o;
}

template <typename T>
__global__ void __launch_bounds__(BLOCK_THREADS) comb_hash_kernel(const BatchHeader* hdr, const T* data, int bits, uint32_t* keys, uint32_t* idx) {
    const uint32_t n = hdr->count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        keys[i] = (uint32_t)(mix64((uint64_t)cf_key(data[i])) >> 11) & ((1u << bits) - 1u);
        idx[i] = i;
    }
}

// level 0 input: tuples, lifted; segment = (key, timestamp / pane_len)
template <typename T, typename R, typename L>
struct CombFromTuples {
    uint64_t pane_len;
    L lift;
    __device__ __forceinline__ uint64_t pane(const T& t) const { return cf_ts(t) / pane_len; }
    __device__ __forceinline__ R value(const T& t) const {
        R r{};
        r.setControlFields(cf_key(t), 0, cf_ts(t));
        lift(t, r);
        return r;
    }
};
// level 1 input: partials; their pane travels in the id field
template <typename R>
struct CombFromPartials {
    __device__ __forceinline__ uint64_t pane(const R& r) const { return cf_id(r); }
    __device__ __forceinline__ R value(const R& r) const { return r; }
};

// One wave per 64 consecutive elements (through `perm` when given): ordered fold of every run of equal (key, pane) inside the
// chunk; the last element of a run keeps the fold (tmp[i], flag[i] = 1), the others are dropped (flag[i] = 0).
template <typename In, typename R, typename From, typename C>
__global__ void __launch_bounds__(BLOCK_THREADS)
    comb_fold_kernel(const uint32_t* n_ptr, const In* in, const uint32_t* perm, From from, C comb, R* tmp, uint32_t* flag) {
    const uint32_t n = *n_ptr;
    const uint32_t chunks = (n + WAVE - 1) / WAVE;
    const uint32_t waves_total = gridDim.x * (BLOCK_THREADS / WAVE);
    const int lane = lane_id();
    for (uint32_t c = blockIdx.x * (BLOCK_THREADS / WAVE) + wave_id(); c < chunks; c += waves_total) {
        const uint32_t i = c * WAVE + (uint32_t)lane;
        const bool ok = i < n;
        R val{};
        uint64_t key = 0, pane = 0;
        if (ok) {
            const In x = in[perm ? perm[i] : i];
            key = (uint64_t)cf_key(x);
            pane = from.pane(x);
            val = from.value(x);
        }
        const uint64_t pkey = comb_shfl_up(key, 1), ppane = comb_shfl_up(pane, 1);
        const bool boundary = ok && (lane == 0 || pkey != key || ppane != pane);
        const unsigned long long bmask = __ballot(boundary);
        const unsigned long long okmask = __ballot(ok);
        const uint32_t cnt = (uint32_t)__popcll(okmask);
        const int seg = __popcll(bmask & (lanemask_lt() | (1ull << lane))) - 1;
        for (int d = 1; d < WAVE; d <<= 1) {
            const int oseg = __shfl_up(seg, d, WAVE);
            const bool take = ok && lane >= d && oseg == seg;
            if (__ballot(take) == 0) break;
            const R other = comb_shfl_up(val, d);
            if (take) {
                R t2;
                fat_combine(comb, other, val, t2);
                val = t2;
            }
        }
        const bool tail = ok && ((uint32_t)lane + 1 == cnt || ((bmask >> (lane + 1)) & 1ull));
        if (ok) flag[i] = tail ? 1u : 0u;
        if (tail) {
            val.setControlFields((decltype(cf_key(val)))key, pane, cf_ts(val));
            tmp[i] = val;
        }
    }
}

// stable compaction of the kept partials; the total goes to *n_out (and, for the last level, to the header / mirror of the batch)
template <typename R>
__global__ void __launch_bounds__(BLOCK_THREADS)
    comb_compact_kernel(const uint32_t* n_ptr, const R* tmp, const uint32_t* flag, const uint32_t* pos, const uint32_t* total, R* out, uint32_t* n_out,
                        BatchHeader* out_hdr, const BatchHeader* in_hdr, volatile uint32_t* mirror) {
    const uint32_t n = *n_ptr;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (flag[i]) out[pos[i]] = tmp[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t t = n ? *total : 0u;
        if (n_out) *n_out = t;
        if (out_hdr) {
            BatchHeader h = *in_hdr;
            h.count = t;
            *out_hdr = h;
            *mirror = t;
        }
    }
}

}  // namespace gpu

// lift of a Key_FFAT_GPU that consumes the partials of a Pane_Combiner_GPU: the partial IS the lifted value
template <typename R>
struct PartialLift {
    WF_HOST_DEVICE void operator()(const R& partial, R& out) const { out = partial; }
};

template <typename tuple_t, typename result_t, typename L, typename C>
class PaneCombinerGPU_Node : public GPU_Replica<tuple_t> {
    static_assert(std::is_trivially_copyable_v<result_t>, "partial results travel in raw batches: they must be trivially copyable");

public:
    PaneCombinerGPU_Node(L lift, C comb, uint64_t pane_len, GPUConfig cfg, edge_kind_t in_kind, std::string name, size_t par, size_t idx,
                         meta::closing_func_t closing, std::shared_ptr<StatsHub> hub, std::shared_ptr<std::atomic<uint64_t>> in_cnt,
                         std::shared_ptr<std::atomic<uint64_t>> out_cnt)
        : GPU_Replica<tuple_t>(cfg, in_kind, std::move(name), par, idx, std::move(closing), std::move(hub)), lift_(lift), comb_(comb), pane_len_(pane_len),
          in_cnt_(std::move(in_cnt)), out_cnt_(std::move(out_cnt)) {}

protected:
    void process_device(gpu::Batch* d) override {
        using namespace gpu;
        const size_t upper = d->count_on_host ? d->host_hdr.count : d->capacity;
        if (!out_pool_) {
            cap_ = d->capacity;
            out_pool_.reset(new BatchPool(MemKind::DEVICE, this->cfg_.gpu_id, sizeof(result_t), cap_, this->cfg_.pool_slabs));
            keys_.resize(cap_);
            idx_.resize(cap_);
            flag_.resize(cap_);
            pos_.resize(cap_);
            tmp_.resize(cap_);
            mid_.resize(cap_);
            ctl_.resize(4);
            scan_ws_.resize(scan_workspace_words(cap_));
            sorter_.reserve(cap_);
        }
        if (upper > cap_) failwith("Pane_Combiner_GPU: a batch larger than the first one received");
        hipStream_t s = this->stream();
        Batch* o = out_pool_->acquire(s);
        o->host_hdr = d->host_hdr;
        o->count_on_host = false;
        if (upper == 0) {
            o->host_hdr.count = 0;
            o->count_on_host = true;
            WF_HIP_CHECK(hipMemcpyAsync(o->slab, &o->host_hdr, sizeof(BatchHeader), hipMemcpyHostToDevice, s));
            this->release_input(d);
            this->emit_device(o);
            return;
        }
        const unsigned grid = grid_for(upper);
        const unsigned wgrid = (unsigned)std::min<size_t>(2048, std::max<size_t>(1, (upper / WAVE + BLOCK_THREADS / WAVE) / (BLOCK_THREADS / WAVE)));
        hipLaunchKernelGGL((comb_hash_kernel<tuple_t>), dim3(grid), dim3(BLOCK_THREADS), 0, s, d->header(), d->template data<tuple_t>(), HASH_BITS, keys_.data(),
                           idx_.data());
        const auto sorted = sorter_.sort(keys_.data(), idx_.data(), d->device_count_ptr(), upper, HASH_BITS, s);
        // level 0: tuples -> partials of at most 64 tuples
        hipLaunchKernelGGL((comb_fold_kernel<tuple_t, result_t, CombFromTuples<tuple_t, result_t, L>, C>), dim3(wgrid), dim3(BLOCK_THREADS), 0, s,
                           d->device_count_ptr(), d->template data<tuple_t>(), sorted.vals, CombFromTuples<tuple_t, result_t, L>{pane_len_, lift_}, comb_,
                           tmp_.data(), flag_.data());
        exclusive_scan_u32(flag_.data(), pos_.data(), upper, ctl_.data(), scan_ws_.data(), s, d->device_count_ptr());
        hipLaunchKernelGGL((comb_compact_kernel<result_t>), dim3(grid), dim3(BLOCK_THREADS), 0, s, d->device_count_ptr(), tmp_.data(), flag_.data(), pos_.data(),
                           ctl_.data(), mid_.data(), ctl_.data() + 1, (BatchHeader*)nullptr, (const BatchHeader*)nullptr, (volatile uint32_t*)nullptr);
        // level 1: partials of the same (key, pane) that were cut at chunk borders -> partials of at most 4096 tuples
        hipLaunchKernelGGL((comb_fold_kernel<result_t, result_t, CombFromPartials<result_t>, C>), dim3(wgrid), dim3(BLOCK_THREADS), 0, s, ctl_.data() + 1,
                           mid_.data(), (const uint32_t*)nullptr, CombFromPartials<result_t>{}, comb_, tmp_.data(), flag_.data());
        exclusive_scan_u32(flag_.data(), pos_.data(), upper, ctl_.data() + 2, scan_ws_.data(), s, ctl_.data() + 1);
        hipLaunchKernelGGL((comb_compact_kernel<result_t>), dim3(grid), dim3(BLOCK_THREADS), 0, s, ctl_.data() + 1, tmp_.data(), flag_.data(), pos_.data(),
                           ctl_.data() + 2, o->template data<result_t>(), (uint32_t*)nullptr, o->header(), d->header(), o->count_mirror);
        this->num_kernels_ += 8 + sorter_.kernels_per_sort(HASH_BITS);
        in_total_ += upper;
        this->release_input(d);
        this->emit_device(o);
    }
    void gpu_flush() override {
        if (in_cnt_) in_cnt_->fetch_add(in_total_);
    }

private:
    static constexpr int HASH_BITS = 22;  // two 11-bit sort passes
    L lift_;
    C comb_;
    uint64_t pane_len_;
    size_t cap_ = 0;
    uint64_t in_total_ = 0;
    std::shared_ptr<std::atomic<uint64_t>> in_cnt_, out_cnt_;
    std::unique_ptr<gpu::BatchPool> out_pool_;
    gpu::DeviceBuffer<uint32_t> keys_, idx_, flag_, pos_, ctl_;
    gpu::DeviceBuffer<result_t> tmp_, mid_;
    gpu::DeviceBuffer<unsigned long long> scan_ws_;
    gpu::RadixSorter sorter_;
};

// The operator: FORWARD, one replica per producer lane (chain it onto the producers), batches in, batches of partials out.
template <typename tuple_t, typename result_t>
class Pane_Combiner_GPU : public Simple_Operator<tuple_t, result_t> {
public:
    template <typename L, typename C>
    Pane_Combiner_GPU(L lift, C comb, uint64_t win_len, uint64_t slide_len, size_t parallelism, std::string name, GPUConfig cfg = GPUConfig(),
                      meta::closing_func_t closing = nullptr)
        : Simple_Operator<tuple_t, result_t>(name, parallelism, FORWARD, nullptr, nullptr), kind_(std::make_shared<edge_kind_t>(EDGE_TUPLE)), cfg_(cfg),
          tuples_in_(std::make_shared<std::atomic<uint64_t>>(0)), partials_out_(std::make_shared<std::atomic<uint64_t>>(0)) {
        if (win_len == 0 || slide_len == 0) fatal_error("Pane_Combiner_GPU: window length or slide cannot be zero");
        const uint64_t pane = std::gcd(win_len, slide_len);
        auto kind = kind_;
        auto in_cnt = tuples_in_;
        auto out_cnt = partials_out_;
        this->factory_ = [=](size_t i, size_t n, std::shared_ptr<StatsHub> hub) -> rt::Node* {
            return new PaneCombinerGPU_Node<tuple_t, result_t, L, C>(lift, comb, pane, cfg, *kind, name, n, i, closing, std::move(hub), in_cnt, out_cnt);
        };
    }
    bool isGPUOperator() const override { return true; }
    edge_kind_t output_kind() const override { return EDGE_DEVICE_BATCH; }
    bool accepts_input_kind(edge_kind_t) const override { return true; }
    void set_input_kind(edge_kind_t k) override { *kind_ = k; }
    std::function<rt::Node*(edge_kind_t)> output_adapter_factory() const override { return batch_adapter_factory<result_t>(cfg_.pool_slabs); }
    std::function<rt::Node*(size_t, size_t)> input_batcher_factory(Mode) const override { return batcher_factory<tuple_t>(cfg_); }
    // tuples folded so far (valid once the PipeGraph has ended)
    uint64_t tuples_in() const { return tuples_in_->load(); }

private:
    std::shared_ptr<edge_kind_t> kind_;
    GPUConfig cfg_;
    std::shared_ptr<std::atomic<uint64_t>> tuples_in_, partials_out_;
};

}  // namespace wf
