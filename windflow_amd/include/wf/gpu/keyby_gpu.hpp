// KeyBy on the device: partition a batch by destination without leaving the GPU.
//
// Capability parity with the fork's Standard_EmitterGPU
// (/root/reference/wf/standard_nodes_gpu.hpp:60-225: KEYBY of a device-resident
// batch into per-destination device sub-batches; its "linear" variant goes
// through the host, its "parallel" variant does one scan+scatter pass PER
// destination and is behind an #ifdef, getOutputQueue is unimplemented) and with
// the scattering studies (src/GPU_Tests/scattering/test_v1..v4).  Here: one
// destination computation, one stable radix pass on log2(P) bits, one offsets
// kernel and one scatter kernel, whatever P; the P sub-batches keep arrival order
// and carry their counts in device headers + pinned mirrors, so neither a GPU
// consumer nor the RCCL exchange (dist/exchange.hpp) needs the host to size them.
#pragma once
#include "../pipegraph.hpp"
#include "batch_ops.hpp"
#include "primitives.hpp"

namespace wf {
namespace gpu {

constexpr int MAX_PARTITIONS = 16;
constexpr int MAX_PACKED_PARTITIONS = 256;  // packed form (one output slab): destinations = ranks x lanes
constexpr uint32_t HASH_SHIFT_LANES = 24;
constexpr uint32_t HASH_SHIFT_RANKS = 48;

template <typename T>
struct DestTable {
    BatchHeader* hdr[MAX_PARTITIONS];
    T* data[MAX_PARTITIONS];
    volatile uint32_t* mirror[MAX_PARTITIONS];
};

// offsets[d] = first sorted position whose destination is >= d; offsets[n_dest] = n
__global__ void __launch_bounds__(BLOCK_THREADS)
    partition_offsets_kernel(const BatchHeader* hdr, const uint32_t* sorted_dest, uint32_t n_dest, uint32_t* offsets) {
    const uint32_t n = hdr->count;
    if (n == 0) {
        if (blockIdx.x == 0)
            for (uint32_t x = threadIdx.x; x <= n_dest; x += blockDim.x) offsets[x] = 0;
        return;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t d = sorted_dest[i];
        const uint32_t first = (i == 0) ? 0u : sorted_dest[i - 1] + 1u;
        for (uint32_t x = first; x <= d; x++) offsets[x] = i;
        if (i == n - 1)
            for (uint32_t x = d + 1; x <= n_dest; x++) offsets[x] = n;
    }
}

template <typename T>
__global__ void __launch_bounds__(BLOCK_THREADS)
    partition_scatter_kernel(const BatchHeader* in_hdr, const T* in, const uint32_t* sorted_dest, const uint32_t* perm, const uint32_t* offsets,
                             DestTable<T> tab, uint32_t n_dest) {
    const uint32_t n = in_hdr->count;
    if (blockIdx.x == 0 && threadIdx.x < n_dest) {
        BatchHeader h = *in_hdr;
        h.count = offsets[threadIdx.x + 1] - offsets[threadIdx.x];
        *tab.hdr[threadIdx.x] = h;
        *tab.mirror[threadIdx.x] = h.count;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t d = sorted_dest[i];
        tab.data[d][i - offsets[d]] = in[perm[i]];
    }
}

// two-level destination of the cross-GPU exchange: owner rank from bits 48-63 of the key hash, lane inside the rank from
// bits 24-47 (the device key tables use the low bits): dest = rank * n_lanes + lane
template <typename T, typename KeyFn>
__global__ void __launch_bounds__(BLOCK_THREADS)
    dest_keys2_kernel(const BatchHeader* hdr, const T* data, uint32_t n_ranks, uint32_t n_lanes, uint32_t* dest, uint32_t* idx, KeyFn keyfn) {
    const uint32_t n = hdr->count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t h = mix64(keyfn(data[i]));
        const uint32_t rank = (uint32_t)(((h >> HASH_SHIFT_RANKS) & 0xffffull) % n_ranks);
        const uint32_t lane = (uint32_t)(((h >> HASH_SHIFT_LANES) & 0xffffffull) % n_lanes);
        dest[i] = rank * n_lanes + lane;
        idx[i] = i;
    }
}

// destination from a user functor (device split: MultiPipe::split_gpu)
template <typename T, typename DestFn>
__global__ void __launch_bounds__(BLOCK_THREADS)
    dest_fn_kernel(const BatchHeader* hdr, const T* data, uint32_t n_dest, uint32_t* dest, uint32_t* idx, DestFn fn) {
    const uint32_t n = hdr->count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        dest[i] = (uint32_t)((uint64_t)fn(data[i]) % n_dest);
        idx[i] = i;
    }
}

// packed form of the scatter: ONE output slab holding the tuples grouped by destination (stable), plus the n_dest + 1
// segment offsets mirrored into pinned memory - the send buffer and the send counts of an all-to-all-v
template <typename T>
__global__ void __launch_bounds__(BLOCK_THREADS)
    partition_pack_kernel(const BatchHeader* in_hdr, const T* in, const uint32_t* perm, const uint32_t* offsets, BatchHeader* out_hdr, T* out,
                          volatile uint32_t* offsets_mirror, uint32_t n_dest) {
    const uint32_t n = in_hdr->count;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) *out_hdr = *in_hdr;
        for (uint32_t d = threadIdx.x; d <= n_dest; d += blockDim.x) offsets_mirror[d] = offsets[d];
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[perm[i]];
}

// Partitions device batches of T into n_dest device batches (stable within a destination).
template <typename T>
class BatchPartitioner {
public:
    // hash_shift selects which 24 bits of mix64(key) pick the destination.  Every level that divides the key space
    // must use bits of its own or the inner level sees a single residue class: the device key tables use the low
    // bits, lanes inside a GPU use bits 24-47 (HASH_SHIFT_LANES), ranks use bits 48-63 (HASH_SHIFT_RANKS).
    BatchPartitioner(size_t n_dest, int device, size_t pool_slabs, uint32_t hash_shift = HASH_SHIFT_LANES)
        : n_dest_(n_dest), device_(device), pool_slabs_(pool_slabs), hash_shift_(hash_shift) {
        if (n_dest_ == 0 || n_dest_ > (size_t)MAX_PARTITIONS) failwith("BatchPartitioner: unsupported number of destinations");
        bits_ = 1;
        while ((1u << bits_) < n_dest_) bits_++;
    }
    // enqueue on `s`; `outs` receives n_dest batches whose counts live on the device (+ pinned mirrors)
    void partition(Batch* in, hipStream_t s, std::vector<Batch*>& outs, uint64_t& kernels) { partition_with(in, s, outs, kernels, HashDest{hash_shift_}); }
    // the same with the destination chosen by `fn(tuple)` modulo n_dest (any functor callable on the device)
    template <typename DestFn>
    void partition_with(Batch* in, hipStream_t s, std::vector<Batch*>& outs, uint64_t& kernels, DestFn fn) {
        const size_t upper = in->count_on_host ? in->host_hdr.count : in->capacity;
        if (pool_ && upper > pool_->capacity()) failwith("BatchPartitioner: a batch larger than the first one received");
        if (!pool_) {
            pool_.reset(new BatchPool(MemKind::DEVICE, device_, sizeof(T), in->capacity, n_dest_ * pool_slabs_));
            dest_.resize(in->capacity);
            idx_.resize(in->capacity);
            offsets_.resize(MAX_PARTITIONS + 1);
            sorter_.reserve(in->capacity);
        }
        outs.clear();
        DestTable<T> tab{};
        for (size_t d = 0; d < n_dest_; d++) {
            Batch* o = pool_->acquire(s);
            o->host_hdr = in->host_hdr;
            o->count_on_host = false;
            tab.hdr[d] = o->header();
            tab.data[d] = o->template data<T>();
            tab.mirror[d] = o->count_mirror;
            outs.push_back(o);
        }
        const unsigned grid = grid_for(std::max<size_t>(upper, 1));
        hipLaunchKernelGGL((dest_fn_kernel<T, DestFn>), dim3(grid), dim3(BLOCK_THREADS), 0, s, in->header(), in->template data<T>(),
                           (uint32_t)n_dest_, dest_.data(), idx_.data(), fn);
        const auto sorted = sorter_.sort(dest_.data(), idx_.data(), in->device_count_ptr(), std::max<size_t>(upper, 1), bits_, s);
        hipLaunchKernelGGL(partition_offsets_kernel, dim3(grid), dim3(BLOCK_THREADS), 0, s, in->header(), sorted.keys, (uint32_t)n_dest_,
                           offsets_.data());
        hipLaunchKernelGGL((partition_scatter_kernel<T>), dim3(grid), dim3(BLOCK_THREADS), 0, s, in->header(), in->template data<T>(),
                           sorted.keys, sorted.vals, offsets_.data(), tab, (uint32_t)n_dest_);
        kernels += 3 + sorter_.kernels_per_sort(bits_);
    }
    // A handful of destinations (the device split of a PipeGraph): one stable compaction pass per destination - the fused
    // predicate + compaction kernel of Filter_GPU with the predicate "fn(tuple) % n_dest == d" - instead of the sort: n_dest
    // launches that stream the batch, no index arrays.
    template <typename DestFn>
    struct DestIs {
        DestFn fn;
        uint32_t n_dest, d;
        __host__ __device__ bool operator()(const T& t, uint32_t) const { return (uint32_t)((uint64_t)fn(t) % n_dest) == d; }
    };
    template <typename DestFn>
    void partition_few(Batch* in, hipStream_t s, std::vector<Batch*>& outs, uint64_t& kernels, DestFn fn) {
        const size_t upper = in->count_on_host ? in->host_hdr.count : in->capacity;
        if (pool_ && upper > pool_->capacity()) failwith("BatchPartitioner: a batch larger than the first one received");
        if (!pool_) {
            pool_.reset(new BatchPool(MemKind::DEVICE, device_, sizeof(T), in->capacity, n_dest_ * pool_slabs_));
            ws_words_ = filter_workspace_words(in->capacity);
            ws_.resize(ws_words_ * n_dest_);
        }
        outs.clear();
        const size_t words = filter_workspace_words(std::max<size_t>(upper, 1));
        WF_HIP_CHECK(hipMemsetAsync(ws_.data(), 0, ws_words_ * n_dest_ * sizeof(unsigned long long), s));
        const unsigned cgrid = (unsigned)std::max<size_t>(1, (upper + FILTER_TILE - 1) / FILTER_TILE);
        (void)words;
        for (size_t d = 0; d < n_dest_; d++) {
            Batch* o = pool_->acquire(s);
            o->host_hdr = in->host_hdr;
            o->count_on_host = false;
            hipLaunchKernelGGL((filter_compact_kernel<T, DestIs<DestFn>>), dim3(cgrid), dim3(BLOCK_THREADS), 0, s, in->header(), in->template data<T>(),
                               o->header(), o->template data<T>(), o->count_mirror, ws_.data() + d * ws_words_,
                               DestIs<DestFn>{fn, (uint32_t)n_dest_, (uint32_t)d});
            outs.push_back(o);
        }
        kernels += n_dest_;
    }
    static constexpr size_t FEW_DESTINATIONS = 4;
    // Packed form for the cross-GPU exchange: destinations are (rank, lane) pairs, the output is one slab ordered by
    // destination; mirror_for(slab) names the pinned array (n_ranks * n_lanes + 1 words) that receives the segment offsets
    // of that slab.
    template <typename MirrorFn>
    Batch* partition_packed(Batch* in, size_t n_ranks, size_t n_lanes, MirrorFn mirror_for, hipStream_t s, uint64_t& kernels) {
        const size_t D = n_ranks * n_lanes;
        if (D == 0 || D > (size_t)MAX_PACKED_PARTITIONS) failwith("BatchPartitioner: unsupported number of (rank, lane) destinations");
        const size_t upper = in->count_on_host ? in->host_hdr.count : in->capacity;
        if (pool_ && upper > pool_->capacity()) failwith("BatchPartitioner: a batch larger than the first one received");
        if (!pool_) {
            pool_.reset(new BatchPool(MemKind::DEVICE, device_, sizeof(T), in->capacity, pool_slabs_));
            dest_.resize(in->capacity);
            idx_.resize(in->capacity);
            offsets_.resize(MAX_PACKED_PARTITIONS + 1);
            sorter_.reserve(in->capacity);
        }
        int bits = 1;
        while ((1u << bits) < D) bits++;
        Batch* o = pool_->acquire(s);
        o->host_hdr = in->host_hdr;
        o->count_on_host = in->count_on_host;
        volatile uint32_t* offsets_mirror = mirror_for(o);
        const unsigned grid = grid_for(std::max<size_t>(upper, 1));
        hipLaunchKernelGGL((dest_keys2_kernel<T, KeyOfAny>), dim3(grid), dim3(BLOCK_THREADS), 0, s, in->header(), in->template data<T>(), (uint32_t)n_ranks,
                           (uint32_t)n_lanes, dest_.data(), idx_.data(), KeyOfAny{});
        const auto sorted = sorter_.sort(dest_.data(), idx_.data(), in->device_count_ptr(), std::max<size_t>(upper, 1), bits, s);
        hipLaunchKernelGGL(partition_offsets_kernel, dim3(grid), dim3(BLOCK_THREADS), 0, s, in->header(), sorted.keys, (uint32_t)D, offsets_.data());
        hipLaunchKernelGGL((partition_pack_kernel<T>), dim3(grid), dim3(BLOCK_THREADS), 0, s, in->header(), in->template data<T>(), sorted.vals,
                           offsets_.data(), o->header(), o->template data<T>(), offsets_mirror, (uint32_t)D);
        kernels += 3 + sorter_.kernels_per_sort(bits);
        return o;
    }
    size_t n_dest() const { return n_dest_; }

private:
    struct KeyOfAny {
        __host__ __device__ uint64_t operator()(const T& t) const { return key_hash64(std::get<0>(t.getControlFields())); }
    };
    // KeyBy: 24 bits of the key hash picked by hash_shift
    struct HashDest {
        uint32_t shift;
        __host__ __device__ uint64_t operator()(const T& t) const {
            return (mix64(key_hash64(std::get<0>(t.getControlFields()))) >> shift) & 0xffffffull;
        }
    };
    size_t n_dest_;
    int device_;
    size_t pool_slabs_;
    uint32_t hash_shift_;
    int bits_ = 1;
    std::unique_ptr<BatchPool> pool_;
    DeviceBuffer<uint32_t> dest_, idx_, offsets_;
    DeviceBuffer<unsigned long long> ws_;
    size_t ws_words_ = 0;
    RadixSorter sorter_;
};

// destination of a key under the device KeyBy (host code can call it too)
__host__ __device__ inline uint32_t keyby_destination(uint64_t key, uint32_t n_dest, uint32_t hash_shift = HASH_SHIFT_LANES) {
    return (uint32_t)(((mix64(key) >> hash_shift) & 0xffffffull) % n_dest);
}

}  // namespace gpu

// Emitter node fused on a producer of batches: ships host batches to the device if needed, partitions them on the
// device and sends sub-batch d to destination d.  DestFn = void: by key hash (KeyBy); otherwise by `fn(tuple) % n_dest`
// (the device form of MultiPipe::split).
template <typename T, typename DestFn = void>
class Partition_EmitterGPU : public rt::Node {
    struct NoFn {};
    using fn_t = std::conditional_t<std::is_void_v<DestFn>, NoFn, DestFn>;

public:
    Partition_EmitterGPU(size_t n_dest, edge_kind_t in_kind, int device, size_t pool_slabs, fn_t fn = fn_t{})
        : n_dest_(n_dest), in_kind_(in_kind), device_(device), slabs_(pool_slabs), part_(n_dest, device, pool_slabs), fn_(fn) {
        for (size_t i = 0; i < n_dest_; i++) add_out_port();
        gpu::preload_module(device_);
    }
    void svc_init() override {
        WF_HIP_CHECK(hipSetDevice(device_));
        stream_.reset(new gpu::Stream());
    }
    void svc(void* msg, int) override {
        gpu::Batch* b = static_cast<gpu::Batch*>(msg);
        gpu::Batch* d = b;
        hipStream_t s = stream_->get();
        if (in_kind_ == EDGE_HOST_BATCH) {
            if (!in_pool_) in_pool_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, device_, sizeof(T), b->capacity, slabs_));
            if (b->capacity > in_pool_->capacity()) gpu::failwith("Partition_EmitterGPU: a batch larger than the first one received");
            d = in_pool_->acquire(s);
            *b->header() = b->host_hdr;
            WF_HIP_CHECK(hipMemcpyAsync(d->slab, b->slab, b->bytes_used(), hipMemcpyHostToDevice, s));
            d->host_hdr = b->host_hdr;
            d->count_on_host = true;
            gpu::release_after(b, s, device_, foreign_);  // the pinned slab may belong to another GPU's context
        } else {
            if (d->device != device_)
                gpu::failwith("Partition_EmitterGPU: device batch produced on another GPU; place the producer and the operator behind the partitioning on the same GPU");
            d->ready.edge_wait(s);
        }
        if (n_dest_ == 1) {
            d->ready.record(s);
            emit_to(d, 0);
            return;
        }
        std::vector<gpu::Batch*> outs;
        if constexpr (std::is_void_v<DestFn>)
            part_.partition(d, s, outs, kernels_);
        else if (n_dest_ <= gpu::BatchPartitioner<T>::FEW_DESTINATIONS)
            part_.partition_few(d, s, outs, kernels_, fn_);
        else
            part_.partition_with(d, s, outs, kernels_, fn_);
        WF_HIP_CHECK(hipGetLastError());
        d->release(s);
        reaper_.tick(s);
        for (size_t p = 0; p < n_dest_; p++) {
            outs[p]->ready.record(s);
            emit_to(outs[p], p);
        }
    }
    void flush() override { foreign_.drain(); }
    void svc_end() override {
        if (stream_) stream_->sync();
    }

private:
    gpu::ForeignReleaser foreign_;
    size_t n_dest_;
    edge_kind_t in_kind_;
    int device_;
    size_t slabs_;
    gpu::BatchPartitioner<T> part_;
    fn_t fn_;
    std::unique_ptr<gpu::Stream> stream_;
    gpu::StreamReaper reaper_;
    std::unique_ptr<gpu::BatchPool> in_pool_;
    uint64_t kernels_ = 0;
};
template <typename T>
using KeyBy_EmitterGPU = Partition_EmitterGPU<T, void>;

// Device form of MultiPipe::split (reference: MultiPipe::split, wf/pipegraph.hpp:1893-1958, per-tuple on the CPU): the
// branch of a tuple is fn(tuple) % cardinality, evaluated on the GPU; every batch becomes one device sub-batch per
// branch and the branches continue on device-batch edges.  fn: any functor integral(const tuple_t&) callable on the
// device.  Branches of device batches can be merged again (MultiPipe::merge keeps the batches).
template <typename F>
inline MultiPipe& split_gpu(MultiPipe& mp, F fn, size_t cardinality, GPUConfig cfg = GPUConfig()) {
    using tuple_t = typename meta::split_sig<F>::tuple_t;
    static_assert(!meta::split_sig<F>::multi, "split_gpu: the splitting function returns one branch index (no multicast on batch edges)");
    if (cardinality > (size_t)gpu::MAX_PARTITIONS) fatal_error("split_gpu: at most 16 branches");
    const int device = cfg.device_for(0);
    const size_t slabs = cfg.pool_slabs;
    return mp.split_custom(
        cardinality, typeid(tuple_t), [](edge_kind_t k) { return k == EDGE_HOST_BATCH || k == EDGE_DEVICE_BATCH; },
        [=](edge_kind_t kind) -> rt::Node* { return new Partition_EmitterGPU<tuple_t, F>(cardinality, kind, device, slabs, fn); }, EDGE_DEVICE_BATCH,
        batch_adapter_factory<tuple_t>(slabs));
}

}  // namespace wf
