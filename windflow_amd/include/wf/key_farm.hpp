// Key_Farm: key parallelism.  Every key is owned by one replica (hash routing);
// each replica is a complete window engine.  Semantics: /root/reference/wf/
// key_farm.hpp:127-220 (ctor I), :251-457 (nesting with Pane_Farm /
// Win_MapReduce: each replica is a whole two-stage operator), wf/kf_nodes.hpp.
#pragma once
#include "pane_farm.hpp"
#include "two_stage.hpp"
#include "win_mapreduce.hpp"

namespace wf {

template <typename tuple_t, typename result_t>
class Key_Farm : public Basic_Operator {
public:
    template <typename F, typename = std::enable_if_t<!std::is_base_of_v<Basic_Operator, F>>>
    Key_Farm(F func, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type, size_t parallelism, std::string name,
             meta::closing_func_t closing = nullptr, routing_func_t routing = default_routing, opt_level_t opt_level = LEVEL0)
        : Key_Farm(engines_tag{}, nullptr, win_len, slide_len, triggering_delay, type, parallelism, std::move(name), std::move(routing), opt_level) {
        set_engine(cpu_win_engine<tuple_t, result_t>(func, closing, hub_));
    }
    // generic form: replicas built by an arbitrary engine maker (used by Key_Farm_GPU)
    struct engines_tag {};
    Key_Farm(engines_tag, win_engine_maker_t maker, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type,
             size_t parallelism, std::string name, routing_func_t routing = default_routing, opt_level_t opt_level = LEVEL0)
        : name_(std::move(name)), win_(win_len), slide_(slide_len), delay_(triggering_delay), type_(type), par_(parallelism),
          routing_(std::move(routing)), opt_(opt_level) {
        check_params();
        if (maker) set_engine(std::move(maker));
    }

protected:
    void set_engine(win_engine_maker_t maker) {
        const win_type_t type = type_;
        const std::string n = name_;
        const uint64_t win = win_, slide = slide_, delay = delay_;
        const size_t N = par_;
        flat_factory_ = [=](rt::Graph& g) {
            std::vector<WinEngineNodes> reps;
            for (size_t i = 0; i < N; i++) {
                WinParams p;
                p.win_len = win;
                p.slide_len = slide;
                p.triggering_delay = delay;
                p.winType = type;
                p.config = WinOperatorConfig(0, 1, slide, 0, 1, slide);
                p.role = SEQ;
                reps.push_back(maker(g, p, n, N, i));
            }
            return reps;
        };
    }

public:
    Key_Farm(TwoStage_Operator<tuple_t, result_t>& inner, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type,
             size_t num_replicas, std::string name, meta::closing_func_t closing = nullptr, routing_func_t routing = default_routing,
             opt_level_t opt_level = LEVEL2)
        : name_(std::move(name)), win_(win_len), slide_(slide_len), delay_(triggering_delay), type_(type), par_(num_replicas),
          routing_(std::move(routing)), opt_(opt_level), inner_(&inner) {
        (void)closing;
        check_params();
        if (inner.isUsed4Nesting() || inner.isUsed()) fatal_error("inner operator has already been used (nested or in a MultiPipe)");
        inner.markUsed4Nesting();
        if (inner.getWinLen() != win_ || inner.getSlideLen() != slide_ || inner.getTriggeringDelay() != delay_ || inner.getWinType() != type_)
            fatal_error("incompatible windowing parameters between Key_Farm and its inner operator");
        if (inner.getOptLevel() != LEVEL2 || opt_ != LEVEL2)
            fatal_error("nesting requires both operators built at LEVEL2 (prepare4Nesting on the inner one)");
        inner_par_ = inner.getInnerParallelisms();
    }

    std::string getName() const override { return name_; }
    size_t getParallelism() const override { return inner_ ? par_ * (inner_par_.first + inner_par_.second) : par_; }
    routing_modes_t getRoutingMode() const override { return COMPLEX; }
    bool isWindowed() const override { return true; }
    const std::type_info& input_type() const override { return typeid(tuple_t); }
    const std::type_info& output_type() const override { return typeid(result_t); }
    bool forces_shuffle_after() const override { return inner_ != nullptr; }
    size_t getNumComplexReplicas() const { return inner_ ? par_ : 0; }
    std::pair<size_t, size_t> getInnerParallelisms() const { return inner_par_; }

    std::vector<StageSpec> build_stages(rt::Graph& g, Mode) override {
        const size_t N = par_;
        routing_func_t rf = routing_;
        if (inner_ != nullptr) {
            std::vector<TwoStageGroup> groups;
            for (size_t i = 0; i < N; i++) {
                NestCtx ctx;  // every inner replica sees the complete sub-stream of its keys
                ctx.slide_outer = slide_;
                groups.push_back(inner_->make_group(g, ctx, name_ + "_inner_" + std::to_string(i)));
            }
            auto root = [rf](size_t G) -> rt::Node* { return new KF_Emitter<tuple_t>(rf, G); };
            // results of a key come from one inner replica only, already ordered when it has one
            // second-stage replica; with more the collector restores the per-key order
            return assemble_two_stage<tuple_t, result_t>(g, groups, type_, name_, root, inner_->isOrdered());
        }
        StageSpec s;
        s.name = name_;
        s.routing = COMPLEX;
        for (const WinEngineNodes& e : flat_factory_(g)) {
            s.replicas.push_back(e.entry);
            s.outputs.push_back(e.output);
        }
        s.make_emitter = [rf, N](rt::Graph& gr, size_t) { return single_emitter(gr, new KF_Emitter<tuple_t>(rf, N), N); };
        s.make_ordering = window_ordering<tuple_t>(type_, name_);
        return {s};
    }

private:
    void check_params() const {
        if (win_ == 0 || slide_ == 0) fatal_error("window length or slide in Key_Farm cannot be zero");
        if (par_ == 0) fatal_error("Key_Farm has parallelism zero");
    }
    std::string name_;
    uint64_t win_, slide_, delay_;
    win_type_t type_;
    size_t par_;
    routing_func_t routing_;
    opt_level_t opt_;
    TwoStage_Operator<tuple_t, result_t>* inner_ = nullptr;
    std::pair<size_t, size_t> inner_par_{0, 0};
    std::function<std::vector<WinEngineNodes>(rt::Graph&)> flat_factory_;
};

}  // namespace wf
