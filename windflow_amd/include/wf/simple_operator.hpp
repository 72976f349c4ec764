// Shared machinery of the single-stage operators (Source, Filter, Map, FlatMap,
// Accumulator, Sink and the batched GPU Map/Filter): a replica base class that
// carries RuntimeContext, closing function and statistics, and an operator base
// class that describes itself as one StageSpec.
//
// In the reference each of these operators repeats ~200 lines of farm
// construction (e.g. /root/reference/wf/map.hpp:243-383); the behaviour encoded
// here is the common part: ctor I = FORWARD, ctor II = KEYBY with a routing
// function, replicas created up front, Ordering_Node(TS) in DETERMINISTIC mode
// (wf/pipegraph.hpp:1581, :1660, :1739, :1817, :2929).
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "basic_operator.hpp"
#include "context.hpp"
#include "emitters.hpp"
#include "meta.hpp"
#include "ordering_node.hpp"

namespace wf {

// base of every replica node: context + closing function + statistics
class Replica_Base : public rt::Node {
public:
    Replica_Base(std::string op_name, size_t parallelism, size_t index, meta::closing_func_t closing,
                 std::shared_ptr<StatsHub> hub, bool is_win = false, bool is_gpu = false)
        : context(parallelism, index), closing_(std::move(closing)), hub_(std::move(hub)),
          stats_(std::move(op_name), "replica_" + std::to_string(index), is_win, is_gpu) {
        set_node_name(stats_.nameOP + "_" + std::to_string(index));
    }

    void svc_init() override { stats_.start_time_us = current_time_usecs(); }

    void svc_end() override {
        if (closing_) closing_(context);
        stats_.end_time_us = current_time_usecs();
        stats_.inputs_received = counters().inputs;
        stats_.outputs_sent = counters().outputs;
        stats_.bytes_received = counters().inputs * in_bytes_;
        stats_.bytes_sent = counters().outputs * out_bytes_;
        const double run_us = (double)(stats_.end_time_us - stats_.start_time_us);
        // per-input service time as the thread loop measured it (reference: wf/win_seq.hpp:306-313, :453-463 measure every
        // svc() call; here one pair of clock reads covers a bulk of up to 64 messages).  A replica fused behind another node
        // has no measurement of its own: its time is part of the head's.
        stats_.busy_time_us = counters().busy_ns / 1000;
        stats_.parks = counters().parks;
        stats_.service_time_measured = counters().busy_ns > 0;
        if (stats_.service_time_measured) {
            const uint64_t units = stats_.inputs_received > 0 ? stats_.inputs_received : stats_.outputs_sent;  // sources: per output
            if (units > 0) stats_.service_time_us = (double)counters().busy_ns / 1000.0 / (double)units;
        }
        if (stats_.outputs_sent > 0) stats_.eff_service_time_us = run_us / (double)stats_.outputs_sent;
        finalize_stats(stats_);
        if (hub_) hub_->add(stats_);
        if (runtime_config().trace) stats_.dump_toFile(runtime_config().log_dir);
    }

protected:
    virtual void finalize_stats(Stats_Record&) {}
    void set_io_bytes(size_t in_b, size_t out_b) {
        in_bytes_ = in_b;
        out_bytes_ = out_b;
    }
    RuntimeContext context;

private:
    meta::closing_func_t closing_;
    std::shared_ptr<StatsHub> hub_;
    Stats_Record stats_;
    size_t in_bytes_ = 0, out_bytes_ = 0;
};

// Operator with exactly one stage.
template <typename in_t, typename out_t>
class Simple_Operator : public Basic_Operator {
public:
    using replica_factory_t = std::function<rt::Node*(size_t index, size_t parallelism, std::shared_ptr<StatsHub>)>;

    Simple_Operator(std::string name, size_t parallelism, routing_modes_t routing, routing_func_t routing_func,
                    replica_factory_t factory)
        : name_(std::move(name)), parallelism_(parallelism), routing_(routing), routing_func_(std::move(routing_func)),
          factory_(std::move(factory)) {
        if (parallelism_ == 0) fatal_error("operator " + name_ + " has parallelism zero");
    }

    std::string getName() const override { return name_; }
    size_t getParallelism() const override { return parallelism_; }
    routing_modes_t getRoutingMode() const override { return routing_; }
    const std::type_info& input_type() const override { return typeid(in_t); }
    const std::type_info& output_type() const override { return typeid(out_t); }

    std::vector<StageSpec> build_stages(rt::Graph& g, Mode) override {
        StageSpec s;
        s.name = name_;
        s.routing = routing_;
        for (size_t i = 0; i < parallelism_; i++) s.replicas.push_back(g.adopt(factory_(i, parallelism_, hub_)));
        if constexpr (!std::is_void_v<in_t>) {
            const size_t n = parallelism_;
            const bool keyed = (routing_ == KEYBY);
            routing_func_t rf = routing_func_;
            s.make_emitter = [n, keyed, rf](rt::Graph& gr, size_t) {
                auto* e = keyed ? new Standard_Emitter<in_t>(rf, n) : new Standard_Emitter<in_t>(n);
                return single_emitter(gr, e, n);
            };
            s.make_ordering = [](Mode m) -> rt::Node* {
                return m == Mode::DETERMINISTIC ? new Ordering_Node<in_t>(TS) : nullptr;
            };
        }
        return {s};
    }

protected:
    std::string name_;
    size_t parallelism_;
    routing_modes_t routing_;
    routing_func_t routing_func_;
    replica_factory_t factory_;
};

}  // namespace wf
