// Filter_GPU: batched filter with on-device stream compaction.
//
// Capability parity with the fork's FilterGPU / FilterGPU_Node
// (/root/reference/wf/filter_gpu.hpp:61-217, wf/filter_gpu_node.hpp:100-414, which
// does not compile as shipped - SURVEY.md §2.2) and with the stateless filter of
// the benchmarks (Thrust copy_if, gpu_filter_stateless.cpp:481-492).  Device
// functor forms: void(const T&, bool&) as in the reference, or bool(const T&).
// The survivors are compacted on the device in stable order by one fused kernel
// (kernels_basic.hpp), the count reaches the host through a pinned word, and the
// compacted batch can feed another GPU operator without touching the host.
// A FilterGPU_Builder exists (the reference has none).
#pragma once
#include "batch_ops.hpp"
#include "kernels_basic.hpp"

namespace wf {

template <typename tuple_t, typename F>
class FilterGPU_Node : public GPU_Replica<tuple_t> {
public:
    FilterGPU_Node(F func, GPUConfig cfg, edge_kind_t in_kind, std::string name, size_t par, size_t idx, meta::closing_func_t closing,
                   std::shared_ptr<StatsHub> hub)
        : GPU_Replica<tuple_t>(cfg, in_kind, std::move(name), par, idx, std::move(closing), std::move(hub)), pred_{std::move(func)} {}

protected:
    void process_device(gpu::Batch* d) override {
        if (!out_pool_)
            out_pool_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, this->cfg_.gpu_id, sizeof(tuple_t), d->capacity, this->cfg_.pool_slabs));
        const size_t upper = d->count_on_host ? d->host_hdr.count : d->capacity;
        const size_t words = gpu::filter_workspace_words(std::max<size_t>(upper, 1));
        // one workspace per in-flight batch (round-robin over pool_slabs)
        if (ws_.size() < this->cfg_.pool_slabs) ws_.resize(this->cfg_.pool_slabs);
        gpu::DeviceBuffer<unsigned long long>& ws = ws_[ws_next_];
        ws_next_ = (ws_next_ + 1) % ws_.size();
        ws.resize(gpu::filter_workspace_words(d->capacity));
        WF_HIP_CHECK(hipMemsetAsync(ws.data(), 0, words * sizeof(unsigned long long), this->stream()));
        gpu::Batch* o = out_pool_->acquire(this->stream());
        const unsigned grid = (unsigned)std::max<size_t>(1, (upper + gpu::FILTER_TILE - 1) / gpu::FILTER_TILE);
        hipLaunchKernelGGL((gpu::filter_compact_kernel<tuple_t, gpu::PredicateAdapter<tuple_t, F>>), dim3(grid), dim3(gpu::BLOCK_THREADS), 0,
                           this->stream(), d->header(), d->template data<tuple_t>(), o->header(), o->template data<tuple_t>(),
                           o->count_mirror, ws.data(), pred_);
        this->num_kernels_++;
        o->host_hdr = d->host_hdr;
        o->count_on_host = false;  // data dependent: device header + pinned mirror know it
        d->release(this->stream());
        this->emit_device(o);
    }

private:
    gpu::PredicateAdapter<tuple_t, F> pred_;
    std::unique_ptr<gpu::BatchPool> out_pool_;
    std::vector<gpu::DeviceBuffer<unsigned long long>> ws_;
    size_t ws_next_ = 0;
};

template <typename tuple_t>
class Filter_GPU : public Simple_Operator<tuple_t, tuple_t> {
public:
    template <typename F>
    Filter_GPU(F func, size_t parallelism, std::string name, GPUConfig cfg = GPUConfig(), meta::closing_func_t closing = nullptr)
        : Simple_Operator<tuple_t, tuple_t>(name, parallelism, FORWARD, nullptr, nullptr), kind_(std::make_shared<edge_kind_t>(EDGE_TUPLE)), slabs_(cfg.pool_slabs), cfg_(cfg) {
        auto kind = kind_;
        this->factory_ = [func, cfg, kind, name, closing](size_t i, size_t n, std::shared_ptr<StatsHub> hub) -> rt::Node* {
            return new FilterGPU_Node<tuple_t, F>(func, cfg, *kind, name, n, i, closing, std::move(hub));
        };
    }
    bool isGPUOperator() const override { return true; }
    edge_kind_t output_kind() const override { return EDGE_DEVICE_BATCH; }
    bool accepts_input_kind(edge_kind_t) const override { return true; }
    void set_input_kind(edge_kind_t k) override { *kind_ = k; }
    std::function<rt::Node*(edge_kind_t)> output_adapter_factory() const override { return batch_adapter_factory<tuple_t>(slabs_); }
    std::function<rt::Node*(size_t, size_t)> input_batcher_factory(Mode) const override { return batcher_factory<tuple_t>(cfg_); }

private:
    std::shared_ptr<edge_kind_t> kind_;
    size_t slabs_;
    GPUConfig cfg_;
};

}  // namespace wf
