// Win_MapReduce: intra-window parallelism.  The tuples of each key are dealt
// round-robin to M >= 2 MAP replicas, each computing a partial result of every
// window over its stripe (role MAP: replica i emits ids i, i+M, i+2M, ...); the
// REDUCE stage merges the M partials of a window with count-based windows of
// length = slide = M over the partial ids (role REDUCE).
// Semantics: /root/reference/wf/win_mapreduce.hpp:127-230, wf/wm_nodes.hpp,
// SURVEY.md A.9.  With count-based windows the MAP stage is fed by a broadcast of
// the renumbered stream and each replica keeps its stripe through a
// WinMap_Dropper (reference pipegraph.hpp:2762-2779).
#pragma once
#include "two_stage.hpp"

namespace wf {

template <typename tuple_t, typename result_t>
class Win_MapReduce : public TwoStage_Operator<tuple_t, result_t> {
public:
    template <typename map_t, typename reduce_t>
    Win_MapReduce(map_t map_func, reduce_t reduce_func, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type,
                  size_t map_parallelism, size_t reduce_parallelism, std::string name, meta::closing_func_t closing = nullptr,
                  bool ordered = true, opt_level_t opt_level = LEVEL0)
        : Win_MapReduce(engines_tag{}, nullptr, nullptr, win_len, slide_len, triggering_delay, type, map_parallelism, reduce_parallelism,
                        std::move(name), ordered, opt_level) {
        set_engines(cpu_win_engine<tuple_t, result_t>(map_func, closing, this->hub_),
                    cpu_win_engine<result_t, result_t>(reduce_func, closing, this->hub_));
    }

    // generic form: the two stages are built by arbitrary engine makers (used by Win_MapReduce_GPU)
    struct engines_tag {};
    Win_MapReduce(engines_tag, win_engine_maker_t map_maker, win_engine_maker_t reduce_maker, uint64_t win_len, uint64_t slide_len,
                  uint64_t triggering_delay, win_type_t type, size_t map_parallelism, size_t reduce_parallelism, std::string name,
                  bool ordered = true, opt_level_t opt_level = LEVEL0)
        : name_(std::move(name)), win_(win_len), slide_(slide_len), delay_(triggering_delay), type_(type), map_par_(map_parallelism),
          reduce_par_(reduce_parallelism), ordered_(ordered), opt_(opt_level) {
        if (win_ == 0 || slide_ == 0) fatal_error("window length or slide in Win_MapReduce cannot be zero");
        if (map_par_ < 2) fatal_error("Win_MapReduce must have a parallel MAP stage");
        if (reduce_par_ == 0) fatal_error("parallelism of the REDUCE stage cannot be zero");
        if (map_maker && reduce_maker) set_engines(std::move(map_maker), std::move(reduce_maker));
    }

protected:
    void set_engines(win_engine_maker_t map_maker, win_engine_maker_t reduce_maker) {
        const uint64_t win = win_, slide = slide_, delay = delay_;
        const size_t M = map_par_, R = reduce_par_;
        const win_type_t type = type_;
        group_factory_ = [=](rt::Graph& g, const NestCtx& ctx, const std::string& prefix, bool with_map) {
            TwoStageGroup grp;
            const uint64_t my_slide = slide * ctx.slide_mult;
            // ---- MAP ------------------------------------------------------------
            for (size_t i = 0; with_map && i < M; i++) {
                WinParams p;
                p.win_len = win;
                p.slide_len = my_slide;
                p.triggering_delay = delay;
                p.winType = type;
                p.config = WinOperatorConfig(ctx.id, ctx.n, ctx.slide_outer, 0, 1, my_slide);
                p.role = MAP;
                p.map_first = i;
                p.map_stride = M;
                const WinEngineNodes seq = map_maker(g, p, prefix + "_map", M, i);
                if (type == CB) {
                    rt::Node* drop = g.adopt(new WinMap_Dropper<tuple_t>(i, M));
                    g.connect_fused(drop, drop->add_out_port(), seq.entry);
                    grp.first_entries.push_back(drop);
                } else {
                    grp.first_entries.push_back(seq.entry);
                }
                grp.first_outputs.push_back(seq.output);
            }
            grp.first_emitter_tb = [=](rt::Graph& gr, size_t off) { return single_emitter(gr, new WinMap_Emitter<tuple_t>(type, M), M, off); };
            grp.first_emitter_cb = [=](rt::Graph& gr, size_t off) { return single_emitter(gr, new Broadcast_Emitter<tuple_t>(M), M, off); };
            // ---- REDUCE ------------------------------------------------------------
            for (size_t k = 0; k < R; k++) {
                WinParams p;
                p.win_len = M;
                p.slide_len = M * R;
                p.triggering_delay = 0;
                p.winType = CB;
                p.config = WinOperatorConfig(ctx.id, ctx.n, ctx.slide_outer, k, R, M);
                p.role = REDUCE;
                const WinEngineNodes n = reduce_maker(g, p, prefix + "_reduce", R, k);
                grp.second.push_back(n.entry);
                grp.second_outputs.push_back(n.output);
            }
            grp.second_emitter = [=](rt::Graph& gr, size_t off) {
                if (R == 1) return single_emitter(gr, new Standard_Emitter<result_t>(1), 1, off);
                return single_emitter(gr, new WF_Emitter<result_t>(CB, M, M, R, ctx.id, ctx.n, ctx.slide_outer, REDUCE), R, off);
            };
            return grp;
        };
    }

public:
    std::string getName() const override { return name_; }
    size_t getParallelism() const override { return map_par_ + reduce_par_; }
    routing_modes_t getRoutingMode() const override { return COMPLEX; }
    bool isWindowed() const override { return true; }
    const std::type_info& input_type() const override { return typeid(tuple_t); }
    const std::type_info& output_type() const override { return typeid(result_t); }
    std::vector<StageSpec> build_stages(rt::Graph& g, Mode) override {
        if (opt_ == LEVEL2 && prepared4nesting_)
            fatal_error("Win_MapReduce [" + name_ + "] has been prepared for nesting, it cannot be added directly to a MultiPipe");
        NestCtx ctx;
        ctx.slide_outer = slide_;
        std::vector<TwoStageGroup> groups{group_factory_(g, ctx, name_, true)};
        return assemble_two_stage<tuple_t, result_t>(g, groups, type_, name_, nullptr, ordered_);
    }
    TwoStageGroup make_group(rt::Graph& g, const NestCtx& ctx, const std::string& prefix) override { return group_factory_(g, ctx, prefix, true); }
    uint64_t getWinLen() const override { return win_; }
    uint64_t getSlideLen() const override { return slide_; }
    uint64_t getTriggeringDelay() const override { return delay_; }
    win_type_t getWinType() const override { return type_; }
    std::pair<size_t, size_t> getInnerParallelisms() const override { return {map_par_, reduce_par_}; }
    opt_level_t getOptLevel() const override { return opt_; }
    bool isOrdered() const override { return ordered_; }
    size_t getMAPParallelism() const { return map_par_; }
    size_t getREDUCEParallelism() const { return reduce_par_; }
    void setPrepared4Nesting() { prepared4nesting_ = true; }

protected:
    // the REDUCE stage alone (its replicas and the emitter that feeds them): for operators that compute the partial results by
    // other means (Win_MapReduce_GPU fed by batches: one keyed device stage instead of the M MAP replicas)
    TwoStageGroup make_reduce_group(rt::Graph& g, const std::string& prefix) {
        NestCtx ctx;
        ctx.slide_outer = slide_;
        return group_factory_(g, ctx, prefix, false);
    }
    bool prepared4nesting() const { return prepared4nesting_; }

private:
    std::string name_;
    uint64_t win_, slide_, delay_;
    win_type_t type_;
    size_t map_par_, reduce_par_;
    bool ordered_;
    opt_level_t opt_;
    bool prepared4nesting_ = false;
    std::function<TwoStageGroup(rt::Graph&, const NestCtx&, const std::string&, bool with_map)> group_factory_;
};

}  // namespace wf
