// Pane_Farm: pane parallelism.  A window of length w sliding by s is cut into
// panes of gcd(w, s); the PLQ stage computes one partial result per pane
// (tumbling windows of pane length, role PLQ), the WLQ stage computes each window
// from w/pane consecutive pane results (count-based windows over pane ids, role
// WLQ).  Semantics: /root/reference/wf/pane_farm.hpp:124-218, SURVEY.md A.8.
// Each of the two functions may be non-incremental or incremental, plain or rich.
#pragma once
#include <numeric>

#include "two_stage.hpp"

namespace wf {

template <typename tuple_t, typename result_t>
class Pane_Farm : public TwoStage_Operator<tuple_t, result_t> {
public:
    template <typename plq_t, typename wlq_t>
    Pane_Farm(plq_t plq_func, wlq_t wlq_func, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type,
              size_t plq_parallelism, size_t wlq_parallelism, std::string name, meta::closing_func_t closing = nullptr, bool ordered = true,
              opt_level_t opt_level = LEVEL0)
        : Pane_Farm(engines_tag{}, nullptr, nullptr, win_len, slide_len, triggering_delay, type, plq_parallelism, wlq_parallelism, std::move(name),
                    ordered, opt_level) {
        set_engines(cpu_win_engine<tuple_t, result_t>(plq_func, closing, this->hub_), cpu_win_engine<result_t, result_t>(wlq_func, closing, this->hub_));
    }

    // generic form: the two stages are built by arbitrary engine makers (used by Pane_Farm_GPU)
    struct engines_tag {};
    Pane_Farm(engines_tag, win_engine_maker_t plq_maker, win_engine_maker_t wlq_maker, uint64_t win_len, uint64_t slide_len,
              uint64_t triggering_delay, win_type_t type, size_t plq_parallelism, size_t wlq_parallelism, std::string name, bool ordered = true,
              opt_level_t opt_level = LEVEL0)
        : name_(std::move(name)), win_(win_len), slide_(slide_len), delay_(triggering_delay), type_(type), plq_par_(plq_parallelism),
          wlq_par_(wlq_parallelism), ordered_(ordered), opt_(opt_level) {
        if (win_ == 0 || slide_ == 0) fatal_error("window length or slide in Pane_Farm cannot be zero");
        if (plq_par_ == 0 || wlq_par_ == 0) fatal_error("Pane_Farm has parallelism (PLQ or WLQ) equal to zero");
        if (win_ <= slide_) fatal_error("Pane_Farm can be used with sliding windows only (s<w)");
        if (plq_maker && wlq_maker) set_engines(std::move(plq_maker), std::move(wlq_maker));
    }

protected:
    void set_engines(win_engine_maker_t plq_maker, win_engine_maker_t wlq_maker) {
        const uint64_t win = win_, slide = slide_, delay = delay_;
        const size_t P = plq_par_, W = wlq_par_;
        const win_type_t type = type_;
        group_factory_ = [=](rt::Graph& g, const NestCtx& ctx, const std::string& prefix, bool with_plq) {
            TwoStageGroup grp;
            const uint64_t my_slide = slide * ctx.slide_mult;
            const uint64_t pane = std::gcd(win, my_slide);
            const uint64_t win_w = win / pane, slide_w = my_slide / pane;
            // ---- PLQ: tumbling windows of one pane ---------------------------
            for (size_t i = 0; with_plq && i < P; i++) {
                WinParams p;
                p.win_len = pane;
                p.slide_len = pane * P;
                p.triggering_delay = delay;
                p.winType = type;
                p.config = WinOperatorConfig(ctx.id, ctx.n, ctx.slide_outer, i, P, pane);
                p.role = PLQ;
                const WinEngineNodes n = plq_maker(g, p, prefix + "_plq", P, i);
                grp.first_entries.push_back(n.entry);
                grp.first_outputs.push_back(n.output);
            }
            grp.first_emitter_tb = [=](rt::Graph& gr, size_t off) {
                if (P == 1) return single_emitter(gr, new Standard_Emitter<tuple_t>(1), 1, off);
                return single_emitter(gr, new WF_Emitter<tuple_t>(type, pane, pane, P, ctx.id, ctx.n, ctx.slide_outer, PLQ), P, off);
            };
            grp.first_emitter_cb = [=](rt::Graph& gr, size_t off) {
                if (P == 1) return single_emitter(gr, new Standard_Emitter<tuple_t>(1), 1, off);
                return single_emitter(gr, new Broadcast_Emitter<tuple_t>(P), P, off);
            };
            // ---- WLQ: count-based windows over pane ids -----------------------
            for (size_t k = 0; k < W; k++) {
                WinParams p;
                p.win_len = win_w;
                p.slide_len = slide_w * W;
                p.triggering_delay = 0;
                p.winType = CB;
                p.config = WinOperatorConfig(ctx.id, ctx.n, ctx.slide_outer, k, W, slide_w);
                p.role = WLQ;
                const WinEngineNodes n = wlq_maker(g, p, prefix + "_wlq", W, k);
                grp.second.push_back(n.entry);
                grp.second_outputs.push_back(n.output);
            }
            grp.second_emitter = [=](rt::Graph& gr, size_t off) {
                if (W == 1) return single_emitter(gr, new Standard_Emitter<result_t>(1), 1, off);
                return single_emitter(gr, new WF_Emitter<result_t>(CB, win_w, slide_w, W, ctx.id, ctx.n, ctx.slide_outer, WLQ), W, off);
            };
            return grp;
        };
    }

public:
    // ---- Basic_Operator -------------------------------------------------------
    std::string getName() const override { return name_; }
    size_t getParallelism() const override { return plq_par_ + wlq_par_; }
    routing_modes_t getRoutingMode() const override { return COMPLEX; }
    bool isWindowed() const override { return true; }
    const std::type_info& input_type() const override { return typeid(tuple_t); }
    const std::type_info& output_type() const override { return typeid(result_t); }
    std::vector<StageSpec> build_stages(rt::Graph& g, Mode) override {
        if (opt_ == LEVEL2 && prepared4nesting_)
            fatal_error("Pane_Farm [" + name_ + "] has been prepared for nesting, it cannot be added directly to a MultiPipe");
        NestCtx ctx;
        ctx.slide_outer = slide_;
        std::vector<TwoStageGroup> groups{group_factory_(g, ctx, name_, true)};
        return assemble_two_stage<tuple_t, result_t>(g, groups, type_, name_, nullptr, ordered_);
    }
    // ---- TwoStage_Operator ----------------------------------------------------
    TwoStageGroup make_group(rt::Graph& g, const NestCtx& ctx, const std::string& prefix) override { return group_factory_(g, ctx, prefix, true); }
    uint64_t getWinLen() const override { return win_; }
    uint64_t getSlideLen() const override { return slide_; }
    uint64_t getTriggeringDelay() const override { return delay_; }
    win_type_t getWinType() const override { return type_; }
    std::pair<size_t, size_t> getInnerParallelisms() const override { return {plq_par_, wlq_par_}; }
    opt_level_t getOptLevel() const override { return opt_; }
    bool isOrdered() const override { return ordered_; }
    size_t getPLQParallelism() const { return plq_par_; }
    size_t getWLQParallelism() const { return wlq_par_; }
    void setPrepared4Nesting() { prepared4nesting_ = true; }

protected:
    // the WLQ stage alone (its replicas and the emitter that feeds them): for operators that compute the panes by other means
    // (Pane_Farm_GPU fed by batches: one keyed device stage instead of the pane-parallel PLQ replicas)
    TwoStageGroup make_wlq_group(rt::Graph& g, const std::string& prefix) {
        NestCtx ctx;
        ctx.slide_outer = slide_;
        return group_factory_(g, ctx, prefix, false);
    }
    bool prepared4nesting() const { return prepared4nesting_; }

private:
    std::string name_;
    uint64_t win_, slide_, delay_;
    win_type_t type_;
    size_t plq_par_, wlq_par_;
    bool ordered_;
    opt_level_t opt_;
    bool prepared4nesting_ = false;
    std::function<TwoStageGroup(rt::Graph&, const NestCtx&, const std::string&, bool with_plq)> group_factory_;
};

}  // namespace wf
