// Fluent builders of the GPU operators.  Option names follow
// /root/reference/wf/builders_gpu.hpp (withBatch :120, withGPUConfiguration :132,
// withScratchpad :156, MapGPU_Builder :1433 with withTupleBufferCapacity :1533,
// withGPUThreadsPerBlock :1544, withScratchpadSize :1549); FilterGPU_Builder,
// BatchSource_Builder and BatchSink_Builder are additions (the reference has no
// FilterGPU builder and ships batches only from benchmark code).
#pragma once
#include "builders.hpp"
#include "gpu/batch_ops.hpp"
#include "gpu/combiner_gpu.hpp"
#include "gpu/filter_gpu.hpp"
#include "gpu/map_gpu.hpp"
#include "gpu/key_ffat_gpu.hpp"
#include "gpu/keyed_gpu.hpp"
#include "gpu/keyby_gpu.hpp"
#include "gpu/win_seq_gpu.hpp"

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
    // replica i of the operator runs on gpu_ids[i % gpu_ids.size()] (one process driving several GPUs)
    Derived& withGPUPlacement(std::vector<int> gpu_ids) {
        cfg_.gpu_ids = std::move(gpu_ids);
        if (!cfg_.gpu_ids.empty()) cfg_.gpu_id = cfg_.gpu_ids[0];
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
    // source compatibility with the fork's MapGPU_Builder (wf/builders_gpu.hpp:1512-1520): whether an operator takes
    // or leaves its batches on the device is negotiated by the MultiPipe, so these are accepted and ignored
    Derived& withGPUInput() { return this->self(); }
    Derived& withGPUOutput() { return this->self(); }
    // batches in flight per replica (depth of the shipping pipeline)
    Derived& withInFlightBatches(size_t n) {
        cfg_.pool_slabs = n < 2 ? 2 : n;
        return this->self();
    }

protected:
    using Builder_Base<Derived>::Builder_Base;
    GPUConfig cfg_;
};

// Map_GPU: the functor signature selects the variant (reference: wf/map_gpu.hpp:103-117)
//   void(T&)                                   stateless, in place
//   void(const T&, R&)                         stateless, out of place
//   void(T&, char*, size_t)                    keyed, in place, per-key scratchpad
//   void(const T&, R&, char*, size_t)          keyed, out of place, per-key scratchpad
template <typename F>
class MapGPU_Builder : public GPU_Builder_Base<MapGPU_Builder<F>> {
public:
    static constexpr size_t arity = meta::arity_v<F>;
    static_assert(arity >= 1 && arity <= 4,
                  "WindFlow Compilation Error - unknown signature passed to the MapGPU_Builder:\n"
                  "  Candidate 1 : __host__ __device__ void(tuple_t &)\n"
                  "  Candidate 2 : __host__ __device__ void(const tuple_t &, result_t &)\n"
                  "  Candidate 3 : __host__ __device__ void(tuple_t &, char *, size_t)\n"
                  "  Candidate 4 : __host__ __device__ void(const tuple_t &, result_t &, char *, size_t)");
    using tuple_t = meta::arg_bare_t<F, 0>;
    static constexpr bool keyed = arity >= 3;
    static constexpr bool in_place = (arity == 1 || arity == 3);
    template <bool B, typename X>
    struct res {
        using type = tuple_t;
    };
    template <typename X>
    struct res<false, X> {
        using type = meta::arg_bare_t<X, 1>;
    };
    using result_t = typename res<in_place, F>::type;
    using map_t = std::conditional_t<keyed, Keyed_GPU_Operator<in_place ? 0 : 1, tuple_t, result_t>, Map_GPU<tuple_t, result_t>>;
    explicit MapGPU_Builder(F func) : GPU_Builder_Base<MapGPU_Builder<F>>("map_gpu"), func_(std::move(func)) {}
    // keyed variants: distinct keys per replica / custom host routing / key-disjoint batch lanes
    MapGPU_Builder& withMaxKeys(size_t n) {
        kcfg_.max_keys = n;
        return *this;
    }
    MapGPU_Builder& enable_KeyBy(routing_func_t r = default_routing) {
        routing_ = std::move(r);
        return *this;
    }
    // fork spelling (wf/builders_gpu.hpp:1507, :1522): keyedness comes from the functor signature here
    MapGPU_Builder& withKeyBy(bool) { return *this; }
    MapGPU_Builder& withRoutingFunction(routing_func_t r) { return enable_KeyBy(std::move(r)); }
    MapGPU_Builder& withKeyDisjointLanes(bool b = true) {
        kcfg_.lanes_are_key_disjoint = b;
        return *this;
    }
    map_t* build_ptr() {
        if constexpr (keyed) {
            kcfg_.state_size = this->cfg_.scratchpad_size ? this->cfg_.scratchpad_size : 64;
            return new map_t(func_, this->parallelism_, this->name_, kcfg_, this->cfg_, this->closing_, routing_);
        } else {
            return new map_t(func_, this->parallelism_, this->name_, this->cfg_, this->closing_);
        }
    }
    map_t build() { return *std::unique_ptr<map_t>(build_ptr()); }
    std::unique_ptr<map_t> build_unique() { return std::unique_ptr<map_t>(build_ptr()); }

private:
    F func_;
    KeyedGPUConfig kcfg_;
    routing_func_t routing_ = default_routing;
};

// keyed_scan(lift, combine, apply): the associative form of a keyed in-place Map_GPU (gpu/keyed_gpu.hpp): hand the result
// to ScanMapGPU_Builder.
template <typename L, typename C, typename A>
inline gpu::KeyedScan<L, C, A> keyed_scan(L lift, C combine, A apply) {
    return gpu::KeyedScan<L, C, A>{lift, combine, apply};
}
template <typename F>
class ScanMapGPU_Builder : public GPU_Builder_Base<ScanMapGPU_Builder<F>> {
    static_assert(gpu::is_keyed_scan<F>::value, "ScanMapGPU_Builder takes keyed_scan(lift, combine, apply)");

public:
    using tuple_t = typename F::tuple_t;
    using map_t = Keyed_GPU_Operator<3, tuple_t, tuple_t>;
    explicit ScanMapGPU_Builder(F func) : GPU_Builder_Base<ScanMapGPU_Builder<F>>("scan_map_gpu"), func_(std::move(func)) {}
    ScanMapGPU_Builder& withMaxKeys(size_t n) {
        kcfg_.max_keys = n;
        return *this;
    }
    ScanMapGPU_Builder& withRoutingFunction(routing_func_t r) {
        routing_ = std::move(r);
        return *this;
    }
    ScanMapGPU_Builder& withKeyDisjointLanes(bool b = true) {
        kcfg_.lanes_are_key_disjoint = b;
        return *this;
    }
    map_t* build_ptr() {
        kcfg_.state_size = F::record_size;
        return new map_t(func_, this->parallelism_, this->name_, kcfg_, this->cfg_, this->closing_, routing_);
    }
    map_t build() { return *std::unique_ptr<map_t>(build_ptr()); }
    std::unique_ptr<map_t> build_unique() { return std::unique_ptr<map_t>(build_ptr()); }

private:
    F func_;
    KeyedGPUConfig kcfg_;
    routing_func_t routing_ = default_routing;
};

// Filter_GPU:  void(const T&, bool&) | bool(const T&)   stateless
//              void(const T&, bool&, char*, size_t)      keyed, per-key scratchpad
template <typename F>
class FilterGPU_Builder : public GPU_Builder_Base<FilterGPU_Builder<F>> {
public:
    static constexpr size_t arity = meta::arity_v<F>;
    static_assert(arity == 1 || arity == 2 || arity == 4,
                  "WindFlow Compilation Error - unknown signature passed to the FilterGPU_Builder:\n"
                  "  Candidate 1 : __host__ __device__ void(const tuple_t &, bool &)\n"
                  "  Candidate 2 : __host__ __device__ bool(const tuple_t &)\n"
                  "  Candidate 3 : __host__ __device__ void(const tuple_t &, bool &, char *, size_t)");
    using tuple_t = meta::arg_bare_t<F, 0>;
    static constexpr bool keyed = arity == 4;
    using filter_t = std::conditional_t<keyed, Keyed_GPU_Operator<2, tuple_t, tuple_t>, Filter_GPU<tuple_t>>;
    explicit FilterGPU_Builder(F func) : GPU_Builder_Base<FilterGPU_Builder<F>>("filter_gpu"), func_(std::move(func)) {}
    FilterGPU_Builder& withMaxKeys(size_t n) {
        kcfg_.max_keys = n;
        return *this;
    }
    FilterGPU_Builder& enable_KeyBy(routing_func_t r = default_routing) {
        routing_ = std::move(r);
        return *this;
    }
    FilterGPU_Builder& withKeyBy(bool) { return *this; }
    FilterGPU_Builder& withRoutingFunction(routing_func_t r) { return enable_KeyBy(std::move(r)); }
    FilterGPU_Builder& withKeyDisjointLanes(bool b = true) {
        kcfg_.lanes_are_key_disjoint = b;
        return *this;
    }
    filter_t* build_ptr() {
        if constexpr (keyed) {
            kcfg_.state_size = this->cfg_.scratchpad_size ? this->cfg_.scratchpad_size : 64;
            return new filter_t(func_, this->parallelism_, this->name_, kcfg_, this->cfg_, this->closing_, routing_);
        } else {
            return new filter_t(func_, this->parallelism_, this->name_, this->cfg_, this->closing_);
        }
    }
    filter_t build() { return *std::unique_ptr<filter_t>(build_ptr()); }
    std::unique_ptr<filter_t> build_unique() { return std::unique_ptr<filter_t>(build_ptr()); }

private:
    F func_;
    KeyedGPUConfig kcfg_;
    routing_func_t routing_ = default_routing;
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

// Device_Source: generator __host__ __device__ void(uint64_t seq, uint32_t replica, tuple_t&)
template <typename G>
class DeviceSource_Builder : public GPU_Builder_Base<DeviceSource_Builder<G>> {
public:
    using tuple_t = meta::arg_bare_t<G, 2>;
    using source_t = Device_Source<tuple_t>;
    explicit DeviceSource_Builder(G gen) : GPU_Builder_Base<DeviceSource_Builder<G>>("device_source"), gen_(gen) {}
    // stop after this many tuples per replica (rounded up to whole batches)
    DeviceSource_Builder& withLimit(uint64_t tuples_per_replica) {
        more_ = [tuples_per_replica](uint64_t seq) { return seq < tuples_per_replica; };
        return *this;
    }
    // asked before every batch with the number of tuples generated so far by the replica
    DeviceSource_Builder& withStopCondition(std::function<bool(uint64_t)> more) {
        more_ = std::move(more);
        return *this;
    }
    // the host sizes (and may punctuate) every batch: bool(seq, replica, capacity, n&, punctuation&), false = end of stream
    DeviceSource_Builder& withControl(device_source_control_t control) {
        control_ = std::move(control);
        return *this;
    }
    source_t* build_ptr() {
        if (control_) return new source_t(gen_, control_, this->parallelism_, this->name_, this->cfg_.batch_len, this->cfg_.pool_slabs, this->cfg_, this->closing_);
        if (!more_) fatal_error("DeviceSource_Builder: withLimit(), withStopCondition() or withControl() is required");
        return new source_t(gen_, more_, this->parallelism_, this->name_, this->cfg_.batch_len, this->cfg_.pool_slabs, this->cfg_, this->closing_);
    }
    std::unique_ptr<source_t> build_unique() { return std::unique_ptr<source_t>(build_ptr()); }

private:
    G gen_;
    std::function<bool(uint64_t)> more_;
    device_source_control_t control_;
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


// ---- GPU window builders -------------------------------------------------------
// options shared by the GPU window builders: windows + GPU configuration
template <typename Derived>
class GPU_Window_Builder_Base : public GPU_Builder_Base<Derived> {
public:
    Derived& withCBWindows(uint64_t win_len, uint64_t slide_len) {
        win_ = win_len;
        slide_ = slide_len;
        type_ = CB;
        delay_ = 0;
        return this->self();
    }
    Derived& withTBWindows(std::chrono::microseconds win_len, std::chrono::microseconds slide_len,
                           std::chrono::microseconds triggering_delay = std::chrono::microseconds::zero()) {
        win_ = (uint64_t)win_len.count();
        slide_ = (uint64_t)slide_len.count();
        delay_ = (uint64_t)triggering_delay.count();
        type_ = TB;
        return this->self();
    }

    // distinct keys per replica the device-resident state is sized for (each key owns an archive area in HBM)
    Derived& withMaxKeys(size_t n) {
        wopt_.max_keys = n;
        return this->self();
    }
    // tuples a key may keep on the device for its open windows between two batches (default: the window length for
    // count-based windows, 1024 for time-based ones, where it depends on the rate of the stream)
    Derived& withArchiveCapacity(size_t n) {
        wopt_.carry_capacity = n;
        return this->self();
    }
    // results per output batch (default: twice the input batch length, at least 8192)
    Derived& withOutputCapacity(size_t n) {
        wopt_.out_capacity = n;
        return this->self();
    }
    // batch input: lane i already carries exactly the keys of replica i (no device KeyBy in front of the replicas)
    Derived& withKeyDisjointLanes(bool b = true) {
        wopt_.lanes_are_key_disjoint = b;
        return this->self();
    }
    // count-based windows: number the tuples of a key by arrival on the device and ignore their ids (always on for batch input)
    Derived& withRenumbering(bool b = true) {
        wopt_.renumber = b;
        return this->self();
    }

protected:
    // withBatch(n): tuples per device batch when the operator is fed by tuples and builds its batches itself (the reference
    // counts windows per batch, wf/builders_gpu.hpp:120; here a batch is a slice of the input stream, whatever it fires)
    explicit GPU_Window_Builder_Base(std::string n) : GPU_Builder_Base<Derived>(std::move(n)) { this->cfg_.batch_len = 4096; }
    uint64_t win_ = 1, slide_ = 1, delay_ = 0;
    win_type_t type_ = CB;
    gpu::WinGPUOptions wopt_;
};

namespace detail {
// tuple / result types of a device window function: the 6-argument functor or wave_reduce(lift, combine)
template <typename F, bool Reduce>
struct gpu_win_types {
    using tuple_t = typename meta::win_gpu_sig<F>::tuple_t;
    using result_t = typename meta::win_gpu_sig<F>::result_t;
};
template <typename F>
struct gpu_win_types<F, true> {
    using tuple_t = typename meta::lift_sig<decltype(F::lift)>::tuple_t;
    using result_t = typename meta::lift_sig<decltype(F::lift)>::result_t;
};
template <typename F>
using gpu_win_types_t = gpu_win_types<F, gpu::is_wave_reduce_t<F>::value>;
}  // namespace detail

template <typename F>
class WinSeqGPU_Builder : public GPU_Window_Builder_Base<WinSeqGPU_Builder<F>> {
public:
    using tuple_t = typename detail::gpu_win_types_t<F>::tuple_t;
    using result_t = typename detail::gpu_win_types_t<F>::result_t;
    using winseq_t = Win_Seq_GPU<tuple_t, result_t>;
    explicit WinSeqGPU_Builder(F func) : GPU_Window_Builder_Base<WinSeqGPU_Builder<F>>("win_seq_gpu"), func_(std::move(func)) {}
    winseq_t* build_ptr() {
        return new winseq_t(func_, this->win_, this->slide_, this->delay_, this->type_, this->name_, this->cfg_, this->closing_, this->wopt_);
    }
    winseq_t build() { return *std::unique_ptr<winseq_t>(build_ptr()); }
    std::unique_ptr<winseq_t> build_unique() { return std::unique_ptr<winseq_t>(build_ptr()); }

private:
    F func_;
};

namespace detail {
template <typename T, bool IsOp>
struct gpu_farm_input_types {
    using tuple_t = typename gpu_win_types_t<T>::tuple_t;
    using result_t = typename gpu_win_types_t<T>::result_t;
};
template <typename A, typename B>
struct gpu_farm_input_types<Pane_Farm_GPU<A, B>, true> {
    using tuple_t = A;
    using result_t = B;
};
template <typename A, typename B>
struct gpu_farm_input_types<Win_MapReduce_GPU<A, B>, true> {
    using tuple_t = A;
    using result_t = B;
};
}  // namespace detail

template <typename T>
class KeyFarmGPU_Builder : public GPU_Window_Builder_Base<KeyFarmGPU_Builder<T>> {
public:
    static constexpr bool nested = std::is_base_of_v<Basic_Operator, T>;
    using tuple_t = typename detail::gpu_farm_input_types<T, nested>::tuple_t;
    using result_t = typename detail::gpu_farm_input_types<T, nested>::result_t;
    using keyfarm_t = Key_Farm_GPU<tuple_t, result_t>;
    template <typename U>
    explicit KeyFarmGPU_Builder(U&& input) : GPU_Window_Builder_Base<KeyFarmGPU_Builder<T>>("key_farm_gpu"), input_(store(std::forward<U>(input))) {
        if constexpr (nested) {
            this->win_ = input.getWinLen();
            this->slide_ = input.getSlideLen();
            this->delay_ = input.getTriggeringDelay();
            this->type_ = input.getWinType();
        }
    }
    KeyFarmGPU_Builder& set_KeyBy(routing_func_t r) {
        routing_ = std::move(r);
        return *this;
    }
    KeyFarmGPU_Builder& withOptLevel(opt_level_t o) {
        opt_ = o;
        return *this;
    }
    keyfarm_t* build_ptr() {
        if constexpr (nested)
            return new keyfarm_t(*input_, this->win_, this->slide_, this->delay_, this->type_, this->parallelism_, this->name_, this->closing_, routing_, opt_);
        else
            return new keyfarm_t(input_, this->win_, this->slide_, this->delay_, this->type_, this->parallelism_, this->name_, this->cfg_,
                                 this->closing_, routing_, this->wopt_);
    }
    keyfarm_t build() { return *std::unique_ptr<keyfarm_t>(build_ptr()); }
    std::unique_ptr<keyfarm_t> build_unique() { return std::unique_ptr<keyfarm_t>(build_ptr()); }

private:
    using stored_t = std::conditional_t<nested, T*, T>;
    template <typename U>
    static stored_t store(U&& u) {
        if constexpr (nested)
            return &u;
        else
            return stored_t(std::forward<U>(u));
    }
    stored_t input_;
    routing_func_t routing_ = default_routing;
    opt_level_t opt_ = LEVEL2;
};
template <typename U>
KeyFarmGPU_Builder(U&&) -> KeyFarmGPU_Builder<std::decay_t<U>>;

template <typename T>
class WinFarmGPU_Builder : public GPU_Window_Builder_Base<WinFarmGPU_Builder<T>> {
public:
    static constexpr bool nested = std::is_base_of_v<Basic_Operator, T>;
    using tuple_t = typename detail::gpu_farm_input_types<T, nested>::tuple_t;
    using result_t = typename detail::gpu_farm_input_types<T, nested>::result_t;
    using winfarm_t = Win_Farm_GPU<tuple_t, result_t>;
    template <typename U>
    explicit WinFarmGPU_Builder(U&& input) : GPU_Window_Builder_Base<WinFarmGPU_Builder<T>>("win_farm_gpu"), input_(store(std::forward<U>(input))) {
        if constexpr (nested) {
            this->win_ = input.getWinLen();
            this->slide_ = input.getSlideLen();
            this->delay_ = input.getTriggeringDelay();
            this->type_ = input.getWinType();
        }
    }
    WinFarmGPU_Builder& withOrdering(bool o) {
        ordered_ = o;
        return *this;
    }
    WinFarmGPU_Builder& withOptLevel(opt_level_t o) {
        opt_ = o;
        return *this;
    }
    winfarm_t* build_ptr() {
        if constexpr (nested)
            return new winfarm_t(*input_, this->win_, this->slide_, this->delay_, this->type_, this->parallelism_, this->name_, this->closing_, ordered_, opt_);
        else
            return new winfarm_t(input_, this->win_, this->slide_, this->delay_, this->type_, this->parallelism_, this->name_, this->cfg_,
                                 this->closing_, ordered_, this->wopt_);
    }
    winfarm_t build() { return *std::unique_ptr<winfarm_t>(build_ptr()); }
    std::unique_ptr<winfarm_t> build_unique() { return std::unique_ptr<winfarm_t>(build_ptr()); }

private:
    using stored_t = std::conditional_t<nested, T*, T>;
    template <typename U>
    static stored_t store(U&& u) {
        if constexpr (nested)
            return &u;
        else
            return stored_t(std::forward<U>(u));
    }
    stored_t input_;
    bool ordered_ = true;
    opt_level_t opt_ = LEVEL2;
};
template <typename U>
WinFarmGPU_Builder(U&&) -> WinFarmGPU_Builder<std::decay_t<U>>;

namespace detail {
// tuple / result types of a two-stage GPU operator from whichever stage is a host function
template <typename F, typename G, bool FirstIsGPU>
struct two_stage_gpu_types {
    using tuple_t = typename gpu_win_types_t<F>::tuple_t;
    using result_t = typename gpu_win_types_t<F>::result_t;
};
template <typename F, typename G>
struct two_stage_gpu_types<F, G, false> {
    using tuple_t = typename meta::win_sig<F>::tuple_t;
    using result_t = typename meta::win_sig<F>::result_t;
};
}  // namespace detail

template <typename Derived>
class GPU_TwoStage_Builder_Base : public GPU_Window_Builder_Base<Derived> {
public:
    Derived& withParallelism(size_t first, size_t second) {
        par1_ = first;
        par2_ = second;
        return this->self();
    }
    Derived& withOrdering(bool ordered) {
        ordered_ = ordered;
        return this->self();
    }
    Derived& withOptLevel(opt_level_t o) {
        opt_ = o;
        return this->self();
    }
    Derived& prepare4Nesting() {
        opt_ = LEVEL2;
        nesting_ = true;
        return this->self();
    }

protected:
    using GPU_Window_Builder_Base<Derived>::GPU_Window_Builder_Base;
    size_t par1_ = 1, par2_ = 1;
    bool ordered_ = true;
    opt_level_t opt_ = LEVEL0;
    bool nesting_ = false;
};

template <typename F, typename G>
class PaneFarmGPU_Builder : public GPU_TwoStage_Builder_Base<PaneFarmGPU_Builder<F, G>> {
public:
    using types = detail::two_stage_gpu_types<F, G, meta::is_gpu_stage_func<F>()>;
    using tuple_t = typename types::tuple_t;
    using result_t = typename types::result_t;
    using panefarm_t = Pane_Farm_GPU<tuple_t, result_t>;
    PaneFarmGPU_Builder(F plq, G wlq) : GPU_TwoStage_Builder_Base<PaneFarmGPU_Builder<F, G>>("pane_farm_gpu"), plq_(std::move(plq)), wlq_(std::move(wlq)) {}
    panefarm_t* build_ptr() {
        auto* p = new panefarm_t(plq_, wlq_, this->win_, this->slide_, this->delay_, this->type_, this->par1_, this->par2_, this->name_, this->cfg_,
                                 this->closing_, this->ordered_, this->opt_, this->wopt_);
        if (this->nesting_) p->setPrepared4Nesting();
        return p;
    }
    panefarm_t build() { return *std::unique_ptr<panefarm_t>(build_ptr()); }
    std::unique_ptr<panefarm_t> build_unique() { return std::unique_ptr<panefarm_t>(build_ptr()); }

private:
    F plq_;
    G wlq_;
};

template <typename F, typename G>
class WinMapReduceGPU_Builder : public GPU_TwoStage_Builder_Base<WinMapReduceGPU_Builder<F, G>> {
public:
    using types = detail::two_stage_gpu_types<F, G, meta::is_gpu_stage_func<F>()>;
    using tuple_t = typename types::tuple_t;
    using result_t = typename types::result_t;
    using wmr_t = Win_MapReduce_GPU<tuple_t, result_t>;
    WinMapReduceGPU_Builder(F map, G reduce) : GPU_TwoStage_Builder_Base<WinMapReduceGPU_Builder<F, G>>("win_mapreduce_gpu"), map_(std::move(map)), reduce_(std::move(reduce)) {
        this->par1_ = 2;
    }
    wmr_t* build_ptr() {
        auto* p = new wmr_t(map_, reduce_, this->win_, this->slide_, this->delay_, this->type_, this->par1_, this->par2_, this->name_, this->cfg_,
                            this->closing_, this->ordered_, this->opt_, this->wopt_);
        if (this->nesting_) p->setPrepared4Nesting();
        return p;
    }
    wmr_t build() { return *std::unique_ptr<wmr_t>(build_ptr()); }
    std::unique_ptr<wmr_t> build_unique() { return std::unique_ptr<wmr_t>(build_ptr()); }

private:
    F map_;
    G reduce_;
};

// Key_FFAT_GPU / Win_SeqFFAT_GPU: lift and combine, both device-callable
template <typename L, typename C>
class KeyFFATGPU_Builder : public GPU_Window_Builder_Base<KeyFFATGPU_Builder<L, C>> {
public:
    using tuple_t = typename meta::lift_sig<L>::tuple_t;
    using result_t = typename meta::lift_sig<L>::result_t;
    using keyffat_t = Key_FFAT_GPU<tuple_t, result_t>;
    KeyFFATGPU_Builder(L lift, C comb) : GPU_Window_Builder_Base<KeyFFATGPU_Builder<L, C>>("key_ffat_gpu"), lift_(std::move(lift)), comb_(std::move(comb)) {
        this->cfg_.batch_len = 65536;
    }
    // distinct keys per replica the device-resident state is sized for
    KeyFFATGPU_Builder& withMaxKeys(size_t n) {
        max_keys_ = n;
        return *this;
    }
    // results per output batch (default: twice the input batch length)
    KeyFFATGPU_Builder& withOutputCapacity(size_t n) {
        out_capacity_ = n;
        return *this;
    }
    KeyFFATGPU_Builder& withKeyDisjointLanes(bool b = true) {
        disjoint_ = b;
        return *this;
    }
    KeyFFATGPU_Builder& skipEmptyGaps(bool b = true) {
        skip_gaps_ = b;
        return *this;
    }
    // kept for API parity (reference builders_gpu.hpp:955): the tree of a key is always updated incrementally
    KeyFFATGPU_Builder& withRebuild(bool) { return *this; }
    KeyFFATGPU_Builder& set_KeyBy(routing_func_t r) {
        routing_ = std::move(r);
        return *this;
    }
    keyffat_t* build_ptr() {
        KeyFFATGPUConfig w;
        w.win_len = this->win_;
        w.slide_len = this->slide_;
        w.delay = this->delay_;
        w.type = this->type_;
        w.max_keys = max_keys_;
        w.out_capacity = out_capacity_;
        w.skip_empty_gaps = skip_gaps_;
        w.lanes_are_key_disjoint = disjoint_;
        return new keyffat_t(lift_, comb_, w, this->parallelism_, this->name_, this->cfg_, this->closing_, routing_);
    }
    keyffat_t build() { return *std::unique_ptr<keyffat_t>(build_ptr()); }
    std::unique_ptr<keyffat_t> build_unique() { return std::unique_ptr<keyffat_t>(build_ptr()); }

private:
    L lift_;
    C comb_;
    size_t max_keys_ = 1 << 16;
    size_t out_capacity_ = 0;
    bool skip_gaps_ = false;
    bool disjoint_ = false;
    routing_func_t routing_ = default_routing;
};

// Pane_Combiner_GPU (gpu/combiner_gpu.hpp): lift + combine of a time-based keyed aggregation, applied per (key, pane) to every
// batch where the tuples are produced; its partial results feed an Exchange_GPU and a Key_FFAT_GPU built with
// KeyFFATGPU_Builder(PartialLift<result_t>(), comb)
template <typename L, typename C>
class PaneCombinerGPU_Builder : public GPU_Window_Builder_Base<PaneCombinerGPU_Builder<L, C>> {
public:
    using tuple_t = typename meta::lift_sig<L>::tuple_t;
    using result_t = typename meta::lift_sig<L>::result_t;
    using combiner_t = Pane_Combiner_GPU<tuple_t, result_t>;
    PaneCombinerGPU_Builder(L lift, C comb) : GPU_Window_Builder_Base<PaneCombinerGPU_Builder<L, C>>("pane_combiner_gpu"), lift_(std::move(lift)), comb_(std::move(comb)) {
        this->cfg_.batch_len = 65536;
    }
    combiner_t* build_ptr() {
        if (this->type_ != TB) fatal_error("Pane_Combiner_GPU: only time-based windows have panes that several producers can fill independently");
        return new combiner_t(lift_, comb_, this->win_, this->slide_, this->parallelism_, this->name_, this->cfg_, this->closing_);
    }
    std::unique_ptr<combiner_t> build_unique() { return std::unique_ptr<combiner_t>(build_ptr()); }

private:
    L lift_;
    C comb_;
};

// sequential flavour: one replica (reference WinSeqFFATGPU_Builder, builders_gpu.hpp:195)
template <typename L, typename C>
class WinSeqFFATGPU_Builder : public KeyFFATGPU_Builder<L, C> {
public:
    WinSeqFFATGPU_Builder(L lift, C comb) : KeyFFATGPU_Builder<L, C>(std::move(lift), std::move(comb)) {
        this->name_ = "win_seqffat_gpu";
        this->parallelism_ = 1;
    }
};

}  // namespace wf
