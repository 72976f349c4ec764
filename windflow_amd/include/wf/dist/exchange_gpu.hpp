// Exchange_GPU: KeyBy across GPUs - the operator and its builder (the nodes are in exchange_nodes.hpp, which also explains
// the protocol).  Every rank (process or thread owning one GPU) places one Exchange_GPU in its MultiPipe; it delivers
// `withOutputLanes(L)` key-disjoint streams: give the keyed operator behind it L replicas and withKeyDisjointLanes().
#pragma once
#include "../builders_gpu.hpp"
#include "exchange_nodes.hpp"
#include "transport.hpp"

namespace wf {

// The operator: every producer endpoint gets its own packing node (in the producer's thread: per-producer order is what
// keeps the tuples of a key in order), all feeding the rank's single endpoint, which delivers `out_lanes` key-disjoint
// streams.  (`parallelism` is accepted for symmetry with the other builders; the number of packing nodes is the number
// of producers.)
template <typename tuple_t>
class Exchange_GPU : public Basic_Operator {
public:
    Exchange_GPU(std::shared_ptr<dist::Transport> transport, size_t parallelism = 1, std::string name = "exchange_gpu", GPUConfig cfg = GPUConfig(),
                 ExchangeConfig xcfg = ExchangeConfig(), meta::closing_func_t closing = nullptr)
        : tr_(std::move(transport)), par_(parallelism), name_(std::move(name)), cfg_(cfg), xcfg_(xcfg), closing_(std::move(closing)),
          xstats_(std::make_shared<ExchangeStats>()) {
        if (par_ == 0) fatal_error("operator " + name_ + " has parallelism zero");
        if (!tr_) fatal_error("operator " + name_ + " needs a transport");
        if ((size_t)tr_->world() * xcfg_.out_lanes > (size_t)gpu::MAX_PACKED_PARTITIONS) fatal_error("operator " + name_ + ": too many ranks x lanes");
        // a rank is one GPU: the lanes, the endpoint and their slabs all live on the first device of the configuration
        cfg_.gpu_id = cfg_.device_for(0);
        cfg_.gpu_ids.clear();
    }
    std::string getName() const override { return name_; }
    size_t getParallelism() const override { return par_; }
    routing_modes_t getRoutingMode() const override { return FORWARD; }
    bool isGPUOperator() const override { return true; }
    const std::type_info& input_type() const override { return typeid(tuple_t); }
    const std::type_info& output_type() const override { return typeid(tuple_t); }
    edge_kind_t output_kind() const override { return EDGE_DEVICE_BATCH; }
    bool accepts_input_kind(edge_kind_t) const override { return true; }
    void set_input_kind(edge_kind_t k) override { kind_ = k; }
    std::function<rt::Node*(edge_kind_t)> output_adapter_factory() const override { return batch_adapter_factory<tuple_t>(cfg_.pool_slabs); }
    // valid once the PipeGraph has ended
    ExchangeStats stats() const {
        ExchangeStats x = *xstats_;
        if (x.elapsed_us == 0 && x.start_us != 0) x.elapsed_us = current_time_usecs() - x.start_us;  // still running
        return x;
    }
    size_t getOutputLanes() const { return xcfg_.out_lanes; }

    // DETERMINISTIC graphs: the exchange is an ordering point - every output lane carries its tuples in timestamp order, released
    // under the low-watermark rule over every producer lane of every rank (exchange_nodes.hpp)
    bool merges_batch_lanes_in_order() const override { return true; }
    std::vector<StageSpec> build_stages(rt::Graph& g, Mode m) override {
        ExchangeConfig x = xcfg_;
        if (m == Mode::DETERMINISTIC) x.deterministic = true;
        return exchange_stages<tuple_t>(g, tr_, name_, cfg_, x, kind_, hub_, xstats_, closing_);
    }

private:
    std::shared_ptr<dist::Transport> tr_;
    size_t par_;
    std::string name_;
    GPUConfig cfg_;
    ExchangeConfig xcfg_;
    meta::closing_func_t closing_;
    std::shared_ptr<ExchangeStats> xstats_;
    edge_kind_t kind_ = EDGE_TUPLE;
};

template <typename tuple_t>
class ExchangeGPU_Builder : public GPU_Builder_Base<ExchangeGPU_Builder<tuple_t>> {
public:
    explicit ExchangeGPU_Builder(std::shared_ptr<dist::Transport> transport)
        : GPU_Builder_Base<ExchangeGPU_Builder<tuple_t>>("exchange_gpu"), transport_(std::move(transport)) {}
    // number of local producers whose step punctuations are merged (see Batch_Source)
    ExchangeGPU_Builder& withLocalLanes(size_t n) {
        xcfg_.local_lanes = n;
        return *this;
    }
    // key-disjoint streams delivered downstream: give the keyed operator behind this many replicas and withKeyDisjointLanes()
    ExchangeGPU_Builder& withOutputLanes(size_t n) {
        xcfg_.out_lanes = n;
        return *this;
    }
    // input batches a round may carry (default: one per local lane); 1 = every batch travels alone
    ExchangeGPU_Builder& withMaxBatchesPerRound(size_t n) {
        xcfg_.max_batches_per_round = n;
        return *this;
    }
    // how long a batch may wait for company before its round leaves anyway (see ExchangeConfig::linger_us)
    ExchangeGPU_Builder& withLinger(uint64_t usec) {
        xcfg_.linger_us = usec;
        return *this;
    }
    ExchangeGPU_Builder& withIdleRoundInterval(uint64_t usec) {
        xcfg_.idle_round_us = usec;
        return *this;
    }
    Exchange_GPU<tuple_t> build() { return Exchange_GPU<tuple_t>(transport_, this->parallelism_, this->name_, this->cfg_, xcfg_, this->closing_); }
    std::unique_ptr<Exchange_GPU<tuple_t>> build_unique() {
        return std::make_unique<Exchange_GPU<tuple_t>>(transport_, this->parallelism_, this->name_, this->cfg_, xcfg_, this->closing_);
    }

private:
    std::shared_ptr<dist::Transport> transport_;
    ExchangeConfig xcfg_;
};

}  // namespace wf
