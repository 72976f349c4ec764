// Key_FFAT: key parallelism over incremental FlatFAT engines (lift + combine).
// Semantics: /root/reference/wf/key_ffat.hpp:105-154 (N Win_SeqFFAT + KF_Emitter,
// sliding windows only).
#pragma once
#include "win_seqffat.hpp"

namespace wf {

template <typename tuple_t, typename result_t>
class Key_FFAT : public Basic_Operator {
public:
    template <typename lift_t, typename comb_t>
    Key_FFAT(lift_t lift, comb_t comb, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type, size_t parallelism,
             std::string name, meta::closing_func_t closing = nullptr, routing_func_t routing = default_routing)
        : name_(std::move(name)), type_(type), par_(parallelism), routing_(std::move(routing)) {
        if (win_len == 0 || slide_len == 0) fatal_error("window length or slide in Key_FFAT cannot be zero");
        if (par_ == 0) fatal_error("Key_FFAT has parallelism zero");
        if (slide_len >= win_len) fatal_error("Key_FFAT can be used with sliding windows only (s<w)");
        auto hub = hub_;
        const std::string n = name_;
        const size_t N = par_;
        WinParams p;
        p.win_len = win_len;
        p.slide_len = slide_len;
        p.triggering_delay = triggering_delay;
        p.winType = type;
        p.config = WinOperatorConfig(0, 1, slide_len, 0, 1, slide_len);
        factory_ = [=](rt::Graph& g) {
            std::vector<rt::Node*> reps;
            for (size_t i = 0; i < N; i++)
                reps.push_back(g.adopt(new Win_SeqFFAT_Node<tuple_t, result_t, lift_t, comb_t>(lift, comb, p, n, N, i, closing, hub)));
            return reps;
        };
    }
    std::string getName() const override { return name_; }
    size_t getParallelism() const override { return par_; }
    routing_modes_t getRoutingMode() const override { return COMPLEX; }
    bool isWindowed() const override { return true; }
    const std::type_info& input_type() const override { return typeid(tuple_t); }
    const std::type_info& output_type() const override { return typeid(result_t); }
    std::vector<StageSpec> build_stages(rt::Graph& g, Mode) override {
        StageSpec s;
        s.name = name_;
        s.routing = COMPLEX;
        s.replicas = factory_(g);
        const size_t N = par_;
        routing_func_t rf = routing_;
        s.make_emitter = [rf, N](rt::Graph& gr, size_t) { return single_emitter(gr, new KF_Emitter<tuple_t>(rf, N), N); };
        s.make_ordering = window_ordering<tuple_t>(type_, name_);
        return {s};
    }

private:
    std::string name_;
    win_type_t type_;
    size_t par_;
    routing_func_t routing_;
    std::function<std::vector<rt::Node*>(rt::Graph&)> factory_;
};

}  // namespace wf
