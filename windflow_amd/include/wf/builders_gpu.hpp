// Fluent builders of the GPU operators.  Option names follow
// /root/reference/wf/builders_gpu.hpp (withBatch :120, withGPUConfiguration :132,
// withScratchpad :156, MapGPU_Builder :1433 with withTupleBufferCapacity :1533,
// withGPUThreadsPerBlock :1544, withScratchpadSize :1549); FilterGPU_Builder,
// BatchSource_Builder and BatchSink_Builder are additions (the reference has no
// FilterGPU builder and ships batches only from benchmark code).
#pragma once
#include "builders.hpp"
#include "gpu/batch_ops.hpp"
#include "gpu/filter_gpu.hpp"
#include "gpu/map_gpu.hpp"

namespace wf {

// options shared by every GPU builder
template <typename Derived>
class GPU_Builder_Base : public Builder_Base<Derived> {
public:
    // number of tuples (or windows, for window operators) per device batch
    Derived& withBatch(size_t batch_len) {
        cfg_.batch_len = batch_len;
        return this->self();
    }
    Derived& withTupleBufferCapacity(size_t n) { return withBatch(n); }
    Derived& withGPUConfiguration(int gpu_id = 0, size_t n_thread_block = DEFAULT_GPU_THREADS_PER_BLOCK) {
        cfg_.gpu_id = gpu_id;
        cfg_.n_thread_block = n_thread_block;
        return this->self();
    }
    Derived& withGPUThreadsPerBlock(size_t n) {
        cfg_.n_thread_block = n;
        return this->self();
    }
    Derived& withScratchpad(size_t bytes) {
        cfg_.scratchpad_size = bytes;
        return this->self();
    }
    Derived& withScratchpadSize(size_t bytes) { return withScratchpad(bytes); }
    // batches in flight per replica (depth of the shipping pipeline)
    Derived& withInFlightBatches(size_t n) {
        cfg_.pool_slabs = n < 2 ? 2 : n;
        return this->self();
    }

protected:
    using Builder_Base<Derived>::Builder_Base;
    GPUConfig cfg_;
};

template <typename F>
class MapGPU_Builder : public GPU_Builder_Base<MapGPU_Builder<F>> {
public:
    using tuple_t = meta::arg_bare_t<F, 0>;
    static constexpr bool in_place = meta::arity_v<F> == 1;
    static_assert(meta::arity_v<F> == 1 || meta::arity_v<F> == 2,
                  "WindFlow Compilation Error - unknown signature passed to the MapGPU_Builder:\n"
                  "  Candidate 1 : __host__ __device__ void(tuple_t &)\n"
                  "  Candidate 2 : __host__ __device__ void(const tuple_t &, result_t &)\n"
                  "  (keyed forms with a scratchpad: see KeyedMapGPU_Builder)");
    template <bool B, typename X>
    struct res {
        using type = tuple_t;
    };
    template <typename X>
    struct res<false, X> {
        using type = meta::arg_bare_t<X, 1>;
    };
    using result_t = typename res<in_place, F>::type;
    using map_t = Map_GPU<tuple_t, result_t>;
    explicit MapGPU_Builder(F func) : GPU_Builder_Base<MapGPU_Builder<F>>("map_gpu"), func_(std::move(func)) {}
    map_t build() { return map_t(func_, this->parallelism_, this->name_, this->cfg_, this->closing_); }
    map_t* build_ptr() { return new map_t(func_, this->parallelism_, this->name_, this->cfg_, this->closing_); }
    std::unique_ptr<map_t> build_unique() { return std::unique_ptr<map_t>(build_ptr()); }

private:
    F func_;
};

template <typename F>
class FilterGPU_Builder : public GPU_Builder_Base<FilterGPU_Builder<F>> {
public:
    using tuple_t = meta::arg_bare_t<F, 0>;
    static_assert(meta::arity_v<F> == 1 || meta::arity_v<F> == 2,
                  "WindFlow Compilation Error - unknown signature passed to the FilterGPU_Builder:\n"
                  "  Candidate 1 : __host__ __device__ void(const tuple_t &, bool &)\n"
                  "  Candidate 2 : __host__ __device__ bool(const tuple_t &)");
    using filter_t = Filter_GPU<tuple_t>;
    explicit FilterGPU_Builder(F func) : GPU_Builder_Base<FilterGPU_Builder<F>>("filter_gpu"), func_(std::move(func)) {}
    filter_t build() { return filter_t(func_, this->parallelism_, this->name_, this->cfg_, this->closing_); }
    filter_t* build_ptr() { return new filter_t(func_, this->parallelism_, this->name_, this->cfg_, this->closing_); }
    std::unique_ptr<filter_t> build_unique() { return std::unique_ptr<filter_t>(build_ptr()); }

private:
    F func_;
};

template <typename F>
class BatchSource_Builder : public GPU_Builder_Base<BatchSource_Builder<F>> {
public:
    template <bool Bulk, typename X>
    struct tup {
        using type = meta::arg_bare_t<X, 0>;
    };
    template <typename X>
    struct tup<true, X> {
        using type = std::remove_pointer_t<meta::arg_bare_t<X, 0>>;
    };
    using tuple_t = typename tup<meta::base_arity_v<F> == 3, F>::type;
    using source_t = Batch_Source<tuple_t>;
    explicit BatchSource_Builder(F func) : GPU_Builder_Base<BatchSource_Builder<F>>("batch_source"), func_(std::move(func)) {}
    source_t build() { return source_t(func_, this->parallelism_, this->name_, this->cfg_.batch_len, this->cfg_.pool_slabs, this->cfg_.gpu_id, this->closing_); }
    source_t* build_ptr() {
        return new source_t(func_, this->parallelism_, this->name_, this->cfg_.batch_len, this->cfg_.pool_slabs, this->cfg_.gpu_id, this->closing_);
    }
    std::unique_ptr<source_t> build_unique() { return std::unique_ptr<source_t>(build_ptr()); }

private:
    F func_;
};

template <typename F>
class BatchSink_Builder : public Builder_Base<BatchSink_Builder<F>> {
public:
    using tuple_t = std::remove_cv_t<std::remove_pointer_t<meta::arg_bare_t<F, 0>>>;
    using sink_t = Batch_Sink<tuple_t>;
    explicit BatchSink_Builder(F func) : Builder_Base<BatchSink_Builder<F>>("batch_sink"), func_(std::move(func)) {}
    sink_t build() { return sink_t(func_, this->parallelism_, this->name_, this->closing_); }
    sink_t* build_ptr() { return new sink_t(func_, this->parallelism_, this->name_, this->closing_); }
    std::unique_ptr<sink_t> build_unique() { return std::unique_ptr<sink_t>(build_ptr()); }

private:
    F func_;
};

}  // namespace wf
