// Map_GPU: batched map on the device.
//
// Capability parity with the fork's MapGPU / MapGPU_Node
// (/root/reference/wf/map_gpu.hpp:78-291, wf/map_gpu_node.hpp:127-720): in-place
// `void(T&)` and out-of-place `void(const T&, R&)` device functors; tuples from a
// CPU predecessor are staged in pinned memory and shipped asynchronously, a
// device batch from a GPU predecessor is adopted without copies, the partial
// batch at end-of-stream goes through the same kernel.  Unlike the fork, the
// operator can be added to a MultiPipe (`MultiPipe::add/chain`), and whether its
// input / output stays on the device is negotiated by the MultiPipe instead of
// being set by hand (`withGPUInput/Output`, wf/builders_gpu.hpp:1512-1520).
// The keyed (stateful) forms live in keyed_gpu.hpp.
#pragma once
#include "batch_ops.hpp"
#include "kernels_basic.hpp"

namespace wf {

// A device functor that reads side tables in HBM (the lookup table of the reference's stateless benchmark,
// gpu_map_stateless.cpp:369-386; the campaign table of the Yahoo benchmark) may define a host-only
//     void bind_device(int gpu_id)
// which every replica calls on ITS copy of the functor before the first batch, on the thread and device it runs on: the place
// to upload the table to that GPU and store the device pointers in the functor (replicas of one operator can sit on different
// GPUs: withGPUPlacement).
template <typename F, typename = void>
struct has_bind_device : std::false_type {};
template <typename F>
struct has_bind_device<F, std::void_t<decltype(std::declval<F&>().bind_device(0))>> : std::true_type {};

template <typename tuple_t, typename result_t, typename F>
class MapGPU_Node : public GPU_Replica<tuple_t> {
public:
    static constexpr bool in_place = meta::arity_v<F> == 1;
    MapGPU_Node(F func, GPUConfig cfg, edge_kind_t in_kind, std::string name, size_t par, size_t idx, meta::closing_func_t closing,
                std::shared_ptr<StatsHub> hub)
        : GPU_Replica<tuple_t>(cfg, in_kind, std::move(name), par, idx, std::move(closing), std::move(hub)), func_(std::move(func)) {}

protected:
    void gpu_init() override {
        if constexpr (has_bind_device<F>::value) func_.bind_device(this->cfg_.gpu_id);
    }
    void process_device(gpu::Batch* d) override {
        const size_t upper = d->count_on_host ? d->host_hdr.count : d->capacity;
        const unsigned grid = gpu::grid_for(upper);
        if constexpr (in_place) {
            hipLaunchKernelGGL((gpu::map_inplace_kernel<tuple_t, F>), dim3(grid), dim3(gpu::BLOCK_THREADS), 0, this->stream(), d->header(),
                               d->template data<tuple_t>(), func_);
            this->num_kernels_++;
            this->emit_device(d);
        } else {
            if (!out_pool_)
                out_pool_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, this->cfg_.gpu_id, sizeof(result_t), d->capacity, this->cfg_.pool_slabs));
            gpu::Batch* o = out_pool_->acquire(this->stream());
            hipLaunchKernelGGL((gpu::map_kernel<tuple_t, result_t, F>), dim3(grid), dim3(gpu::BLOCK_THREADS), 0, this->stream(), d->header(),
                               d->template data<tuple_t>(), o->header(), o->template data<result_t>(), func_);
            this->num_kernels_++;
            o->host_hdr = d->host_hdr;
            o->count_on_host = d->count_on_host;
            if (!d->count_on_host) {
                // the count is only known on the device: mirror it for host consumers
                WF_HIP_CHECK(hipMemcpyAsync((void*)o->count_mirror, d->device_count_ptr(), sizeof(uint32_t), hipMemcpyDeviceToHost, this->stream()));
            }
            d->release(this->stream());
            this->emit_device(o);
        }
    }

private:
    F func_;
    std::unique_ptr<gpu::BatchPool> out_pool_;
};

template <typename tuple_t, typename result_t>
class Map_GPU : public Simple_Operator<tuple_t, result_t> {
public:
    template <typename F>
    Map_GPU(F func, size_t parallelism, std::string name, GPUConfig cfg = GPUConfig(), meta::closing_func_t closing = nullptr)
        : Simple_Operator<tuple_t, result_t>(name, parallelism, FORWARD, nullptr, nullptr), kind_(std::make_shared<edge_kind_t>(EDGE_TUPLE)), slabs_(cfg.pool_slabs), cfg_(cfg) {
        auto kind = kind_;
        this->factory_ = [func, cfg, kind, name, closing](size_t i, size_t n, std::shared_ptr<StatsHub> hub) -> rt::Node* {
            return new MapGPU_Node<tuple_t, result_t, F>(func, cfg, *kind, name, n, i, closing, std::move(hub));
        };
    }
    bool isGPUOperator() const override { return true; }
    edge_kind_t output_kind() const override { return EDGE_DEVICE_BATCH; }
    bool accepts_input_kind(edge_kind_t) const override { return true; }
    void set_input_kind(edge_kind_t k) override { *kind_ = k; }
    std::function<rt::Node*(edge_kind_t)> output_adapter_factory() const override { return batch_adapter_factory<result_t>(slabs_); }
    std::function<rt::Node*(size_t, size_t)> input_batcher_factory(Mode) const override { return batcher_factory<tuple_t>(cfg_); }

private:
    std::shared_ptr<edge_kind_t> kind_;
    size_t slabs_;
    GPUConfig cfg_;
};

}  // namespace wf
