// Shared machinery of the two-stage window operators (Pane_Farm = PLQ+WLQ,
// Win_MapReduce = MAP+REDUCE) and of their nesting inside Win_Farm / Key_Farm.
//
// The reference nests by replicating whole FastFlow pipelines inside a farm and
// then performs graph surgery according to the optimisation level
// (/root/reference/wf/win_farm.hpp:266-484, wf/key_farm.hpp:251-457,
// wf/pane_farm.hpp:221-264, wf/transformations.hpp).  Here a two-stage operator
// knows how to build one *group* = {first-stage replicas, second-stage replicas,
// emitter into the first stage, emitter from a first-stage replica into the
// second stage} for a given position inside an outer operator; flat operators
// are one group, nested ones are G groups under a root emitter.  Stage wiring is
// always "emitter fused on the producer, ordering node fused on the consumer",
// which is what the reference reaches at LEVEL2 (combine_farms + Ordering_Node).
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "basic_operator.hpp"
#include "emitters.hpp"
#include "ordering_node.hpp"
#include "win_seq.hpp"
#include "window_nodes.hpp"

namespace wf {

// position of an inner operator inside an outer window-parallel operator
struct NestCtx {
    size_t id = 0;             // index of this inner replica
    size_t n = 1;              // number of inner replicas (window-parallel outer), 1 otherwise
    uint64_t slide_outer = 0;  // slide of the outer operator
    uint64_t slide_mult = 1;   // the inner operator slides by slide * slide_mult
};

struct TwoStageGroup {
    std::vector<rt::Node*> first_entries;  // node receiving the input of each first-stage replica
    std::vector<rt::Node*> first_outputs;  // node emitting the output of each first-stage replica
    std::vector<rt::Node*> second;         // second-stage replicas (entry nodes)
    std::vector<rt::Node*> second_outputs; // node emitting the output of each second-stage replica (empty: same as `second`)
    // emitter into the first stage for time-based / count-based windows; leaves are
    // numbered dest_offset + local index
    std::function<EmitterAssembly(rt::Graph&, size_t dest_offset)> first_emitter_tb;
    std::function<EmitterAssembly(rt::Graph&, size_t dest_offset)> first_emitter_cb;
    // emitter fused on every first-stage replica, routing into the second stage
    std::function<EmitterAssembly(rt::Graph&, size_t dest_offset)> second_emitter;
};

// Interface implemented by Pane_Farm and Win_MapReduce so that Win_Farm and
// Key_Farm can replicate them.
template <typename tuple_t, typename result_t>
class TwoStage_Operator : public Basic_Operator {
public:
    virtual TwoStageGroup make_group(rt::Graph& g, const NestCtx& ctx, const std::string& name_prefix) = 0;
    virtual uint64_t getWinLen() const = 0;
    virtual uint64_t getSlideLen() const = 0;
    virtual uint64_t getTriggeringDelay() const = 0;
    virtual win_type_t getWinType() const = 0;
    virtual std::pair<size_t, size_t> getInnerParallelisms() const = 0;
    virtual opt_level_t getOptLevel() const = 0;
    virtual bool isOrdered() const = 0;
    bool isUsed4Nesting() const { return used4nesting_; }
    void markUsed4Nesting() { used4nesting_ = true; }

protected:
    bool used4nesting_ = false;
};

// turn G groups into stage specs.  `root_emitter(g, G)` builds the node that picks
// a group (nullptr when G == 1); `ordered` appends a per-key ordered collector.
template <typename tuple_t, typename result_t>
inline std::vector<StageSpec> assemble_two_stage(rt::Graph& g, std::vector<TwoStageGroup>& groups, win_type_t type, const std::string& name,
                                                 std::function<rt::Node*(size_t n_groups)> root_emitter, bool ordered) {
    const size_t G = groups.size();
    auto shared = std::make_shared<std::vector<TwoStageGroup>>(groups);
    std::vector<size_t> off1(G + 1, 0), off2(G + 1, 0);
    for (size_t j = 0; j < G; j++) {
        off1[j + 1] = off1[j] + groups[j].first_entries.size();
        off2[j + 1] = off2[j] + groups[j].second.size();
    }
    StageSpec a;
    a.name = name + "_stage1";
    a.routing = COMPLEX;
    for (auto& gr : groups) {
        a.replicas.insert(a.replicas.end(), gr.first_entries.begin(), gr.first_entries.end());
        a.outputs.insert(a.outputs.end(), gr.first_outputs.begin(), gr.first_outputs.end());
    }
    a.make_emitter = [shared, off1, type, root_emitter, G](rt::Graph& gr, size_t) {
        std::vector<EmitterAssembly> children;
        for (size_t j = 0; j < G; j++) {
            auto& grp = (*shared)[j];
            children.push_back(type == CB ? grp.first_emitter_cb(gr, off1[j]) : grp.first_emitter_tb(gr, off1[j]));
        }
        if (G == 1) return children[0];
        rt::Node* root = gr.adopt(root_emitter(G));
        return tree_emitter(gr, root, children);
    };
    a.make_ordering = window_ordering<tuple_t>(type, name);

    StageSpec b;
    b.name = name + "_stage2";
    b.routing = COMPLEX;
    b.internal = true;
    for (auto& gr : groups) {
        b.replicas.insert(b.replicas.end(), gr.second.begin(), gr.second.end());
        const auto& outs = gr.second_outputs.empty() ? gr.second : gr.second_outputs;
        b.outputs.insert(b.outputs.end(), outs.begin(), outs.end());
    }
    b.make_emitter = [shared, off1, off2, G](rt::Graph& gr, size_t src) {
        size_t j = 0;
        while (j + 1 < G && src >= off1[j + 1]) j++;
        return (*shared)[j].second_emitter(gr, off2[j]);
    };
    // the second stage always orders by id, whatever the mode
    b.make_ordering = [](Mode) -> rt::Node* { return new Ordering_Node<result_t>(ID); };

    std::vector<StageSpec> out{a, b};
    if (ordered && b.replicas.size() > 1) {
        StageSpec c;
        c.name = name + "_collector";
        c.routing = FORWARD;
        c.internal = true;
        c.replicas.push_back(g.adopt(new Ordered_Collector<result_t>()));
        c.make_emitter = [](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<result_t>(1), 1); };
        out.push_back(c);
    }
    return out;
}

// A window engine maker builds one replica of a window stage for given parameters.
// Operators are written against makers so that the same composition logic serves
// the CPU engines (Win_Seq_Node) and the GPU ones (Win_Seq_GPU_Node).
// A replica may be a short chain of fused nodes (a GPU engine followed by its device->host and un-batching adapters):
// `entry` receives the tuples, `output` emits the results.
struct WinEngineNodes {
    rt::Node* entry = nullptr;
    rt::Node* output = nullptr;
};
using win_engine_maker_t = std::function<WinEngineNodes(rt::Graph&, const WinParams&, const std::string& name, size_t par, size_t idx)>;

template <typename in_t, typename out_t, typename F>
inline win_engine_maker_t cpu_win_engine(F func, meta::closing_func_t closing, std::shared_ptr<StatsHub> hub) {
    return [func, closing, hub](rt::Graph& g, const WinParams& p, const std::string& name, size_t par, size_t idx) -> WinEngineNodes {
        rt::Node* n = g.adopt(new Win_Seq_Node<in_t, out_t, F>(func, p, name, par, idx, closing, hub));
        return {n, n};
    };
}

}  // namespace wf
