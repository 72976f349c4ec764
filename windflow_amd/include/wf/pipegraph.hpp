// PipeGraph + MultiPipe: the topology layer.
//
// Public API parity with /root/reference/wf/pipegraph.hpp:104-244 (PipeGraph) and
// :255-571 (MultiPipe): add_source / add / chain / add_sink / chain_sink / merge /
// split / select / run / start / wait_end / getNumThreads / get_ListOperators /
// dump_DOTGraph, and the legality rules of SURVEY.md A.12-A.13.
//
// Design (ours): the reference stores a MultiPipe as nested FastFlow all-to-all
// "matrioskas" and needs one add()/chain() overload per operator type plus graph
// surgery (normalize, combine_with_laststage...) for split and merge.  Here a
// MultiPipe is just a list of *open endpoints* - (node, unconnected output port)
// pairs - and there is a single generic wiring routine driven by the StageSpec
// list an operator produces:
//   * direct connection  (same parallelism, FORWARD, no forced shuffle):
//     endpoint i --ring--> replica i                    (reference "Case 2")
//   * shuffle: an emitter assembly is fused onto every endpoint and its leaves
//     are connected by rings to [ordering node ->] replica   (reference "Case 3")
//   * chain: the replica is fused onto the endpoint, same thread
//   * split: a Splitting_Emitter is fused onto every endpoint; branch b of the
//     split owns port b of each of them        (reference "Case 1"/Tree_Emitter)
//   * merge: the merged MultiPipe owns the union of the operands' endpoints.
// Because endpoints are first-class, any set of open leaf MultiPipes can be
// merged; the three cases the reference supports (merge-ind, merge-full,
// merge-partial, pipegraph.hpp:808-971) are classified for diagnostics only.
#pragma once
#include <algorithm>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <typeinfo>
#include <vector>

#include "basic.hpp"
#include "basic_operator.hpp"
#include "emitters.hpp"
#include "meta.hpp"
#include "rt/graph.hpp"
#include "source.hpp"

namespace wf {

class PipeGraph;

class MultiPipe {
public:
    MultiPipe(const MultiPipe&) = delete;
    MultiPipe& operator=(const MultiPipe&) = delete;

    // add an operator with a new set of threads (shuffle or direct connection)
    MultiPipe& add(Basic_Operator& op) {
        attach(op, false);
        return *this;
    }
    // try to fuse the operator's replicas in the threads of the previous operator
    MultiPipe& chain(Basic_Operator& op) {
        attach(op, true);
        return *this;
    }
    MultiPipe& add_sink(Basic_Operator& sink) {
        attach(sink, false);
        return *this;
    }
    MultiPipe& chain_sink(Basic_Operator& sink) {
        attach(sink, true);
        return *this;
    }

    // merge this MultiPipe with others; returns the new MultiPipe
    template <typename... MPS>
    MultiPipe& merge(MPS&... others);

    // split into `cardinality` branches with a user function returning a branch
    // index or a vector of branch indices
    template <typename F>
    MultiPipe& split(F func, size_t cardinality);

    // Generic form of split (the device split of gpu/split_gpu.hpp builds on it): the edges are first adapted until
    // `accepts(kind)` holds, then `make(kind)` builds the splitting node fused onto every endpoint - it owns one output
    // port per branch - and the branches start on edges of kind `branch_kind` with `branch_adapter` as their converter.
    MultiPipe& split_custom(size_t cardinality, const std::type_info& tuple_type, const std::function<bool(edge_kind_t)>& accepts,
                            const std::function<rt::Node*(edge_kind_t)>& make, edge_kind_t branch_kind,
                            std::function<rt::Node*(edge_kind_t)> branch_adapter);

    // branch idx of a split MultiPipe
    MultiPipe& select(size_t idx) const {
        if (!is_split_) fatal_error("MultiPipe has not been split, select() cannot be used");
        if (idx >= children_.size()) fatal_error("index of select() is out of range");
        return *children_[idx];
    }

    bool isRunnable() const { return has_source_ && (has_sink_ || is_merged_ || is_split_); }
    bool hasSink() const { return has_sink_; }
    bool hasSource() const { return has_source_; }
    size_t getNumEndpoints() const { return tails_.size(); }

private:
    friend class PipeGraph;
    struct Endpoint {
        rt::Node* node;
        size_t port;
    };

    explicit MultiPipe(PipeGraph* g) : graph_(g) {}

    void attach(Basic_Operator& op, bool try_chain);
    // insert adapter nodes until the edge kind satisfies `ok`
    template <typename Pred>
    void adapt_tails(Pred ok);
    void check_editable(const char* what) const {
        if (!has_source_) fatal_error(std::string("MultiPipe does not have a Source, ") + what);
        if (has_sink_) fatal_error(std::string("MultiPipe is terminated by a Sink, ") + what);
        if (is_merged_) fatal_error(std::string("MultiPipe has been merged, ") + what);
        if (is_split_) fatal_error(std::string("MultiPipe has been split, ") + what);
    }

    PipeGraph* graph_;
    std::vector<Endpoint> tails_;
    const std::type_info* out_type_ = nullptr;
    bool has_source_ = false;
    bool has_sink_ = false;
    bool is_merged_ = false;
    bool is_split_ = false;
    bool fresh_from_split_ = false;  // no operator added yet after a split
    bool force_shuffle_ = false;     // after merges and nested window operators
    std::vector<MultiPipe*> children_;
    edge_kind_t tail_kind_ = EDGE_TUPLE;                     // what the endpoints emit
    std::function<rt::Node*(edge_kind_t)> tail_adapter_;     // from the producing operator
    std::vector<size_t> dot_tail_;  // DOT ids of the operators feeding the endpoints
    std::string split_label_;
};

// ---------------------------------------------------------------------------
class PipeGraph {
public:
    explicit PipeGraph(std::string name, Mode mode = Mode::DEFAULT)
        : name_(std::move(name)), mode_(mode), root_(new AppNode{nullptr, nullptr, {}}) {}
    ~PipeGraph() {
        if (started_ && !ended_) wait_end();
    }
    PipeGraph(const PipeGraph&) = delete;
    PipeGraph& operator=(const PipeGraph&) = delete;

    // any source-like operator (Source<T>, Batch_Source<T>, ...): no input, one stage
    MultiPipe& add_source(Basic_Operator& source) {
        if (source.input_type() != typeid(void)) fatal_error("add_source() needs a Source operator");
        if (source.isUsed()) fatal_error("Source operator has already been used in a MultiPipe");
        MultiPipe* mp = new_mp();
        auto stages = source.build_stages(graph_, mode_);
        source.mark_used();
        for (rt::Node* n : stages[0].replicas) mp->tails_.push_back({n, n->add_out_port()});
        mp->has_source_ = true;
        mp->out_type_ = &source.output_type();
        mp->tail_kind_ = source.output_kind();
        mp->tail_adapter_ = source.output_adapter_factory();
        mp->dot_tail_ = {dot_add_node(source.getName(), source.getParallelism(), "Source", false)};
        root_->children.push_back(new_app_node(mp, root_.get()));
        operators_.push_back(&source);
        return *mp;
    }

    // start + wait_end
    int run() {
        start();
        return wait_end();
    }

    int start() {
        if (started_) fatal_error("PipeGraph [" + name_ + "] has already been started");
        for (auto& mp : mps_)
            if (!mp->isRunnable()) fatal_error("some MultiPipe is not runnable, check Sinks and Sources presence");
        started_ = true;
        if (!quiet()) {
            std::cout << WF_GREEN << "WindFlow-AMD" << WF_DEFAULT_COLOR << " PipeGraph [" << name_ << "] is running with "
                      << getNumThreads() << " threads" << std::endl
                      << "  |- execution mode: " << (mode_ == Mode::DETERMINISTIC ? "DETERMINISTIC" : "DEFAULT") << std::endl
                      << "  |- backpressure: ON (bounded rings, capacity " << graph_.config().queue_capacity << ")" << std::endl
                      << "  |- waiting: " << (graph_.config().blocking ? "blocking" : "spin/yield/sleep back-off") << std::endl
                      << "  |- thread pinning: " << (graph_.config().pin_threads ? "ON" : "OFF") << std::endl
                      << "  |- statistics dump: " << (graph_.config().trace ? "ON" : "OFF") << std::endl;
        }
        start_us_ = current_time_usecs();
        graph_.run();
        return 0;
    }

    int wait_end() {
        if (!started_) fatal_error("PipeGraph [" + name_ + "] is not started yet");
        graph_.wait();
        ended_ = true;
        elapsed_us_ = current_time_usecs() - start_us_;
        if (graph_.config().trace) dump_trace();
        if (!quiet()) std::cout << WF_GREEN << "WindFlow-AMD" << WF_DEFAULT_COLOR << " PipeGraph [" << name_ << "] executed successfully" << std::endl;
        return 0;
    }

    bool isStarted() const { return started_; }
    bool isEnded() const { return ended_; }
    Mode getMode() const { return mode_; }
    const std::string& getName() const { return name_; }
    size_t getNumThreads() const { return graph_.num_threads(); }
    uint64_t getElapsedTimeUsec() const { return elapsed_us_; }
    std::vector<Basic_Operator*> get_ListOperators() const { return operators_; }
    rt::Graph& runtime() { return graph_; }
    rt::RuntimeConfig& config() { return graph_.config(); }

    // Graphviz description of the logical graph (no graphviz link dependency:
    // the reference needs -DGRAPHVIZ_WINDFLOW + libgvc, pipegraph.hpp:1450-1518)
    std::string get_DOTGraph() const {
        std::ostringstream o;
        o << "digraph \"" << name_ << "\" {\n  rankdir=LR;\n  node [shape=box, style=rounded];\n";
        for (size_t i = 0; i < dot_nodes_.size(); i++) {
            const auto& n = dot_nodes_[i];
            o << "  n" << i << " [label=\"" << n.label << " (" << n.parallelism << ")\\n" << n.kind << "\"" << (n.gpu ? ", color=red" : "")
              << "];\n";
        }
        for (const auto& e : dot_edges_) {
            o << "  n" << e.from << " -> n" << e.to;
            if (!e.label.empty()) o << " [label=\"" << e.label << "\"]";
            o << ";\n";
        }
        o << "}\n";
        return o.str();
    }
    void dump_DOTGraph(const std::string& path = "") const {
        std::ofstream f(path.empty() ? name_ + ".gv" : path);
        f << get_DOTGraph();
    }

private:
    friend class MultiPipe;
    struct AppNode {
        MultiPipe* mp;
        AppNode* parent;
        std::vector<AppNode*> children;
    };
    struct DotNode {
        std::string label;
        size_t parallelism;
        std::string kind;
        bool gpu;
    };
    struct DotEdge {
        size_t from, to;
        std::string label;
    };

    static bool quiet() {
        static const bool q = std::getenv("WF_QUIET") != nullptr && std::atoi(std::getenv("WF_QUIET")) != 0;
        return q;
    }

    MultiPipe* new_mp() {
        mps_.emplace_back(new MultiPipe(this));
        return mps_.back().get();
    }
    AppNode* new_app_node(MultiPipe* mp, AppNode* parent) {
        app_nodes_.emplace_back(new AppNode{mp, parent, {}});
        return app_nodes_.back().get();
    }
    AppNode* find_app_node(MultiPipe* mp) const {
        for (auto& n : app_nodes_)
            if (n->mp == mp) return n.get();
        return nullptr;
    }
    static void leaves_of(AppNode* n, std::vector<AppNode*>& out) {
        if (n->children.empty())
            out.push_back(n);
        else
            for (AppNode* c : n->children) leaves_of(c, out);
    }
    static void detach(AppNode* n) {
        AppNode* p = n->parent;
        if (p == nullptr) return;
        p->children.erase(std::remove(p->children.begin(), p->children.end(), n), p->children.end());
        n->parent = nullptr;
    }

    size_t dot_add_node(const std::string& label, size_t par, const std::string& kind, bool gpu) {
        dot_nodes_.push_back({label, par, kind, gpu});
        return dot_nodes_.size() - 1;
    }
    void dot_add_edge(size_t from, size_t to, const std::string& label = "") { dot_edges_.push_back({from, to, label}); }

    // classify a merge the way the reference does (pipegraph.hpp:808-971) and
    // update the Application Tree
    std::string register_merge(const std::vector<MultiPipe*>& ops, MultiPipe* merged) {
        std::vector<AppNode*> nodes;
        for (MultiPipe* mp : ops) {
            AppNode* n = find_app_node(mp);
            if (n == nullptr) fatal_error("MultiPipe to be merged does not belong to this PipeGraph");
            if (!n->children.empty()) fatal_error("MultiPipe to be merged has been split");
            nodes.push_back(n);
        }
        // lowest common ancestor
        auto path_to_root = [](AppNode* n) {
            std::vector<AppNode*> p;
            for (; n != nullptr; n = n->parent) p.push_back(n);
            std::reverse(p.begin(), p.end());
            return p;
        };
        std::vector<AppNode*> lca_path = path_to_root(nodes[0]);
        for (size_t i = 1; i < nodes.size(); i++) {
            auto p = path_to_root(nodes[i]);
            size_t k = 0;
            while (k < lca_path.size() && k < p.size() && lca_path[k] == p[k]) k++;
            lca_path.resize(k);
        }
        AppNode* lca = lca_path.back();
        std::string kind;
        std::vector<AppNode*> lca_leaves;
        leaves_of(lca, lca_leaves);
        const bool covers_all = lca_leaves.size() == nodes.size();
        if (lca == root_.get()) {
            bool all_root_children = true;
            for (AppNode* n : nodes) all_root_children &= (n->parent == root_.get());
            kind = all_root_children ? "merge-ind" : "merge-general";
        } else if (covers_all) {
            kind = "merge-full";
        } else {
            bool siblings = true;
            for (AppNode* n : nodes) siblings &= (n->parent == lca);
            kind = siblings ? "merge-partial" : "merge-general";
        }
        // tree update: operands disappear, the merged MultiPipe hangs under the LCA
        // (or replaces it when the whole sub-tree was merged)
        AppNode* attach_to = lca;
        if (covers_all && lca != root_.get()) {
            attach_to = lca->parent;
            detach(lca);
        } else {
            for (AppNode* n : nodes) {
                AppNode* p = n->parent;
                detach(n);
                // prune interior nodes that lost all their children
                while (p != nullptr && p != lca && p->children.empty()) {
                    AppNode* pp = p->parent;
                    detach(p);
                    p = pp;
                }
            }
        }
        AppNode* mn = new_app_node(merged, attach_to);
        attach_to->children.push_back(mn);
        return kind;
    }

    void dump_trace() {
        const std::string dir = graph_.config().log_dir;
        ::mkdir(dir.c_str(), 0777);
        std::ofstream f(dir + "/graph_" + name_ + "_" + std::to_string(::getpid()) + ".json");
        f << "{\"graph\":\"" << name_ << "\",\"threads\":" << getNumThreads() << ",\"elapsed_us\":" << elapsed_us_ << ",\"operators\":[";
        bool first = true;
        for (Basic_Operator* op : operators_) {
            for (const auto& r : op->get_StatsRecords()) {
                f << (first ? "" : ",") << r.to_json();
                first = false;
            }
        }
        f << "]}\n";
    }

    std::string name_;
    Mode mode_;
    rt::Graph graph_;
    std::vector<std::unique_ptr<MultiPipe>> mps_;
    std::vector<std::unique_ptr<AppNode>> app_nodes_;
    std::unique_ptr<AppNode> root_;
    std::vector<Basic_Operator*> operators_;
    std::vector<DotNode> dot_nodes_;
    std::vector<DotEdge> dot_edges_;
    bool started_ = false;
    bool ended_ = false;
    uint64_t start_us_ = 0, elapsed_us_ = 0;
};

// ---------------------------------------------------------------------------
inline void MultiPipe::attach(Basic_Operator& op, bool try_chain) {
    check_editable("operator cannot be added");
    if (graph_->started_) fatal_error("PipeGraph already started, operator cannot be added");
    if (op.isUsed()) fatal_error("operator " + op.getName() + " has already been used in a MultiPipe");
    if (out_type_ != nullptr && *out_type_ != op.input_type())
        fatal_error("output type of the last operator in the MultiPipe does not match the input type of operator " + op.getName());
    rt::Graph& g = graph_->graph_;
    const Mode mode = graph_->mode_;
    // DETERMINISTIC mode: batches of several lanes cannot simply be shuffled (lanes overlap in time).  Keyed GPU operators order
    // them themselves; in front of anything else that is not fed lane by lane the batches become tuples again and the
    // ordering nodes do their work.
    if (mode == Mode::DETERMINISTIC && tail_kind_ != EDGE_TUPLE && !op.merges_batch_lanes_in_order() &&
        (tails_.size() != op.getParallelism() || force_shuffle_ ||
         !(op.getRoutingMode() == FORWARD || op.takes_batch_lanes_one_to_one())))  // (a branch of a device split keeps its lanes: port b of splitter i)
        adapt_tails([](edge_kind_t k) { return k == EDGE_TUPLE; });
    // upward adaptation: a batch operator behind a producer of tuples gets a Batcher fused onto every producer (see
    // Basic_Operator::input_batcher_factory) - unless the mode needs an order that only the tuple world's ordering nodes can
    // give (DETERMINISTIC, lanes shuffled into an operator that does not merge its batch lanes by timestamp itself)
    if (tail_kind_ == EDGE_TUPLE && !tails_.empty() && op.accepts_input_kind(EDGE_HOST_BATCH) &&
        (mode != Mode::DETERMINISTIC || op.merges_batch_lanes_in_order() ||
         (tails_.size() == op.getParallelism() && !force_shuffle_ && !fresh_from_split_ && op.getRoutingMode() == FORWARD))) {
        if (auto make = op.input_batcher_factory(mode)) {
            for (size_t i = 0; i < tails_.size(); i++) {
                rt::Node* b = g.adopt(make(i, tails_.size()));
                g.connect_fused(tails_[i].node, tails_[i].port, b);
                tails_[i] = {b, b->add_out_port()};
            }
            tail_kind_ = EDGE_HOST_BATCH;
            tail_adapter_ = nullptr;  // (the operator accepts host batches: nothing converts them back)
            fresh_from_split_ = false;  // the batchers are the first nodes of a split branch: it now looks like any batch producer
        }
    }
    // edge negotiation: hand batches over when the operator can take them
    adapt_tails([&op](edge_kind_t k) { return op.accepts_input_kind(k); });
    op.set_input_kind(tail_kind_);
    std::vector<StageSpec> stages = op.build_stages(g, mode);
    op.mark_used();
    const bool is_sink = (op.output_type() == typeid(void));
    bool chained = false;
    std::vector<Endpoint> prev = tails_;
    for (size_t si = 0; si < stages.size(); si++) {
        StageSpec& s = stages[si];
        if (s.per_producer) {
            // one node per upstream endpoint, in the producer's thread
            for (size_t i = 0; i < prev.size(); i++) {
                rt::Node* node = g.adopt(s.per_producer(i, prev.size()));
                g.connect_fused(prev[i].node, prev[i].port, node);
                s.replicas.push_back(node);
            }
            force_shuffle_ = false;
            fresh_from_split_ = false;
            prev.clear();
            for (rt::Node* r : s.replicas) prev.push_back({r, r->add_out_port()});
            continue;
        }
        const size_t n = s.replicas.size();
        const bool last_stage = (si + 1 == stages.size());
        // (the first operator of a split branch cannot be chained on tuple edges - reference: MultiPipe::chain after split - but on
        // batch edges a branch is just one more output port of the device splitter: its work can stay in the splitter's thread)
        const bool can_chain = try_chain && stages.size() == 1 && s.routing == FORWARD && (!fresh_from_split_ || tail_kind_ != EDGE_TUPLE) &&
                               !force_shuffle_ && prev.size() == n;
        const bool direct = !s.internal && s.routing == FORWARD && (!fresh_from_split_ || tail_kind_ != EDGE_TUPLE) && !force_shuffle_ && prev.size() == n;
        if (can_chain) {
            // same thread as the producer (reference: chain_operator, pipegraph.hpp:1273-1318)
            for (size_t i = 0; i < n; i++) g.connect_fused(prev[i].node, prev[i].port, s.replicas[i]);
            chained = true;
        } else if (direct) {
            for (size_t i = 0; i < n; i++) g.connect_queue(prev[i].node, prev[i].port, s.replicas[i]);
        } else {
            // heads of the new threads: [ordering ->] replica.  Ordering nodes merge individual tuples (batch edges reach this
            // point in DETERMINISTIC mode only when the operator orders its lanes itself, see the top of this function)
            const bool batch_edge = !s.internal && tail_kind_ != EDGE_TUPLE;
            std::vector<rt::Node*> heads(n);
            for (size_t i = 0; i < n; i++) {
                rt::Node* ord = (s.make_ordering && !batch_edge) ? s.make_ordering(mode) : nullptr;
                if (ord != nullptr) {
                    g.adopt(ord);
                    g.connect_fused(ord, ord->add_out_port(), s.replicas[i]);
                    heads[i] = ord;
                } else {
                    heads[i] = s.replicas[i];
                }
            }
            for (size_t j = 0; j < prev.size(); j++) {
                EmitterAssembly a = s.make_emitter(g, j);
                g.connect_fused(prev[j].node, prev[j].port, a.root);
                for (const EmitterLeaf& l : a.leaves) g.connect_queue(l.node, l.port, heads.at(l.dest));
            }
            force_shuffle_ = false;
        }
        fresh_from_split_ = false;
        prev.clear();
        if (!(is_sink && last_stage))
            for (rt::Node* r : (s.outputs.empty() ? s.replicas : s.outputs)) prev.push_back({r, r->add_out_port()});
    }
    tails_ = prev;
    if (is_sink) {
        has_sink_ = true;
        out_type_ = nullptr;
    } else {
        out_type_ = &op.output_type();
    }
    if (op.forces_shuffle_after()) force_shuffle_ = true;
    tail_kind_ = op.output_kind();
    tail_adapter_ = op.output_adapter_factory();
    // logical graph
    const size_t id = graph_->dot_add_node(op.getName(), op.getParallelism(), is_sink ? "Sink" : (op.isWindowed() ? "Window" : "Operator"),
                                           op.isGPUOperator());
    for (size_t from : dot_tail_) graph_->dot_add_edge(from, id, chained ? "chained" : split_label_);
    split_label_.clear();
    dot_tail_ = {id};
    graph_->operators_.push_back(&op);
}

template <typename Pred>
inline void MultiPipe::adapt_tails(Pred ok) {
    while (!ok(tail_kind_)) {
        if (tail_kind_ == EDGE_TUPLE || !tail_adapter_) fatal_error("operator cannot consume the output of the previous operator");
        rt::Graph& g = graph_->graph_;
        for (Endpoint& e : tails_) {
            rt::Node* a = g.adopt(tail_adapter_(tail_kind_));
            g.connect_fused(e.node, e.port, a);
            e = {a, a->add_out_port()};
        }
        tail_kind_ = (tail_kind_ == EDGE_DEVICE_BATCH) ? EDGE_HOST_BATCH : EDGE_TUPLE;
    }
}

template <typename... MPS>
inline MultiPipe& MultiPipe::merge(MPS&... others) {
    std::vector<MultiPipe*> ops = {this, (&others)...};
    if (ops.size() < 2) fatal_error("merge() needs at least one other MultiPipe");
    for (size_t i = 0; i < ops.size(); i++) {
        MultiPipe* mp = ops[i];
        if (mp->graph_ != graph_) fatal_error("MultiPipe to be merged does not belong to this PipeGraph");
        for (size_t j = 0; j < i; j++)
            if (ops[j] == mp) fatal_error("a MultiPipe cannot be merged with itself");
        if (!mp->has_source_) fatal_error("MultiPipe does not have a Source, it cannot be merged");
        if (mp->has_sink_) fatal_error("MultiPipe is terminated by a Sink, it cannot be merged");
        if (mp->is_merged_) fatal_error("MultiPipe has already been merged");
        if (mp->is_split_) fatal_error("MultiPipe has been split, merge its branches instead");
        if (mp->out_type_ == nullptr || *mp->out_type_ != *out_type_) fatal_error("MultiPipes to be merged must have the same output type");
    }
    // operands that carry batches keep them: every operand is brought down to the lowest kind among them (device
    // batches -> host batches -> tuples), so a merge of device-batch branches stays on the device
    edge_kind_t common = tail_kind_;
    for (MultiPipe* mp : ops) common = std::min(common, mp->tail_kind_);
    for (MultiPipe* mp : ops) mp->adapt_tails([common](edge_kind_t k) { return k <= common; });
    MultiPipe* merged = graph_->new_mp();
    merged->has_source_ = true;
    merged->out_type_ = out_type_;
    merged->tail_kind_ = common;
    merged->tail_adapter_ = tail_adapter_;
    merged->force_shuffle_ = true;  // reference: MultiPipe ctor II, pipegraph.hpp:298-329
    for (MultiPipe* mp : ops) {
        merged->tails_.insert(merged->tails_.end(), mp->tails_.begin(), mp->tails_.end());
        merged->dot_tail_.insert(merged->dot_tail_.end(), mp->dot_tail_.begin(), mp->dot_tail_.end());
        mp->tails_.clear();
        mp->is_merged_ = true;
    }
    graph_->register_merge(ops, merged);
    return *merged;
}

template <typename F>
inline MultiPipe& MultiPipe::split(F func, size_t cardinality) {
    using tuple_t = typename meta::split_sig<F>::tuple_t;
    return split_custom(
        cardinality, typeid(tuple_t), [](edge_kind_t k) { return k == EDGE_TUPLE; },
        [func, cardinality](edge_kind_t) -> rt::Node* { return new Splitting_Emitter<tuple_t, F>(func, cardinality); }, EDGE_TUPLE, nullptr);
}

inline MultiPipe& MultiPipe::split_custom(size_t cardinality, const std::type_info& tuple_type, const std::function<bool(edge_kind_t)>& accepts,
                                          const std::function<rt::Node*(edge_kind_t)>& make, edge_kind_t branch_kind,
                                          std::function<rt::Node*(edge_kind_t)> branch_adapter) {
    check_editable("it cannot be split");
    if (tails_.empty() || fresh_from_split_) fatal_error("MultiPipe is empty, it cannot be split");
    if (cardinality < 2) fatal_error("split() needs at least two branches");
    if (*out_type_ != tuple_type) fatal_error("input type of the splitting function does not match the output type of the MultiPipe");
    adapt_tails(accepts);
    rt::Graph& g = graph_->graph_;
    std::vector<rt::Node*> splitters;
    for (Endpoint& e : tails_) {
        rt::Node* s = g.adopt(make(tail_kind_));
        g.connect_fused(e.node, e.port, s);
        splitters.push_back(s);
    }
    PipeGraph::AppNode* me = graph_->find_app_node(this);
    for (size_t b = 0; b < cardinality; b++) {
        MultiPipe* c = graph_->new_mp();
        c->has_source_ = true;
        c->out_type_ = out_type_;
        c->fresh_from_split_ = true;
        c->tail_kind_ = branch_kind;
        c->tail_adapter_ = branch_adapter;
        c->dot_tail_ = dot_tail_;
        c->split_label_ = "branch " + std::to_string(b);
        for (rt::Node* s : splitters) c->tails_.push_back({s, b});
        children_.push_back(c);
        me->children.push_back(graph_->new_app_node(c, me));
    }
    tails_.clear();
    is_split_ = true;
    return *this;
}

}  // namespace wf
