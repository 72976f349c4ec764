// Basic_Operator + StageSpec: the contract between operators and the MultiPipe.
//
// In the reference every operator *is* a FastFlow farm/pipeline/a2a and the
// MultiPipe has one hand-written add()/chain() overload per operator type with
// the emitter/ordering/collector combination spelled out
// (/root/reference/wf/pipegraph.hpp:1565-2990, wf/basic_operator.hpp:47-79).
// Here an operator only *describes* itself as an ordered list of stages; a stage
// says: these are my replica nodes, this is how to build an emitter that routes
// into them, and this is the ordering node I need in front of each replica for a
// given execution mode.  The MultiPipe has exactly one generic wiring routine.
#pragma once
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <typeinfo>
#include <vector>

#include "basic.hpp"
#include "rt/graph.hpp"
#include "stats_record.hpp"

namespace wf {

// (node, port) of an emitter that must be connected to destination replica `dest`
struct EmitterLeaf {
    size_t dest;
    rt::Node* node;
    size_t port;
};
// An emitter is a small tree of nodes fused in the producer's thread.  `root`
// receives the producer's output; `leaves` are the open ports, flattened
// (this is what the reference's Tree_Emitter does by summing getNDestinations(),
// wf/tree_emitter.hpp:119-144).
struct EmitterAssembly {
    rt::Node* root = nullptr;
    std::vector<EmitterLeaf> leaves;
};

struct StageSpec {
    std::string name;
    routing_modes_t routing = FORWARD;
    std::vector<rt::Node*> replicas;  // entry node of each replica, already adopted by the graph
    // node whose output feeds the next stage, when a replica is a fused chain of
    // several nodes (empty: same as `replicas`)
    std::vector<rt::Node*> outputs;
    // build (and adopt into the graph) the emitter used by upstream producer
    // `src_idx` (index of the replica of the previous stage when `internal`, else
    // the index of the upstream endpoint)
    std::function<EmitterAssembly(rt::Graph&, size_t src_idx)> make_emitter;
    // ordering node to put in front of each replica for this mode (nullptr: none)
    std::function<rt::Node*(Mode)> make_ordering;
    bool internal = false;  // fed by the previous stage of the same operator
    // Instead of `replicas`: the stage has exactly one node per upstream endpoint, built (not adopted) by this factory
    // and FUSED onto its endpoint (node i of n) - work that belongs to the producer's thread, such as packing a batch
    // by destination before it is handed to an exchange.
    std::function<rt::Node*(size_t i, size_t n)> per_producer;
};

// What flows on the edge leaving an operator: individual heap tuples, batches in
// pinned host memory, or batches resident in device memory.  The MultiPipe
// negotiates the kind between consecutive operators and inserts adapter nodes
// (device->host copy, un-batching) where the consumer cannot take batches.  This
// generalises the reference's manual have_gpu_input / have_gpu_output flags
// (/root/reference/wf/map_gpu.hpp:148-149, wf/builders_gpu.hpp:1512-1520).
enum edge_kind_t { EDGE_TUPLE = 0, EDGE_HOST_BATCH = 1, EDGE_DEVICE_BATCH = 2 };

// where replicas deposit their statistics at svc_end (shared with the operator)
struct StatsHub {
    std::mutex m;
    std::vector<Stats_Record> recs;
    void add(const Stats_Record& r) {
        std::lock_guard<std::mutex> l(m);
        recs.push_back(r);
    }
};

class Basic_Operator {
public:
    virtual ~Basic_Operator() = default;
    virtual std::string getName() const = 0;
    virtual size_t getParallelism() const = 0;
    virtual routing_modes_t getRoutingMode() const = 0;
    virtual bool isUsed() const { return used_; }
    virtual bool isGPUOperator() const { return false; }
    virtual bool isWindowed() const { return false; }
    // type names used for the inter-operator type check (the reference compares
    // typeid(T).name() strings, pipegraph.hpp:1573-1578)
    virtual const std::type_info& input_type() const = 0;
    virtual const std::type_info& output_type() const = 0;
    // materialise the operator into stages owned by `g`
    virtual std::vector<StageSpec> build_stages(rt::Graph& g, Mode mode) = 0;
    // edge negotiation (see edge_kind_t)
    virtual edge_kind_t output_kind() const { return EDGE_TUPLE; }
    virtual bool accepts_input_kind(edge_kind_t k) const { return k == EDGE_TUPLE; }
    virtual void set_input_kind(edge_kind_t) {}
    // node converting this operator's output one step towards tuples
    // (DEVICE_BATCH -> HOST_BATCH -> TUPLE); nullptr if the operator emits tuples
    virtual std::function<rt::Node*(edge_kind_t from)> output_adapter_factory() const { return nullptr; }
    // Upward adaptation (tuples -> batches): an operator that works on batches and is fed by a producer of individual
    // tuples may name a node - built per producer, FUSED onto it - that writes the producer's tuples straight into pinned
    // host batches (gpu/batch_ops.hpp Batcher).  The operator is then wired as if a batch source fed it: no heap tuple,
    // no queue hop and no ordering node per tuple between the CPU stage and the GPU operator.  nullptr: feed me tuples.
    virtual std::function<rt::Node*(size_t producer, size_t n_producers)> input_batcher_factory(Mode) const { return nullptr; }
    // after a nested window operator the next operator must shuffle (A.12)
    virtual bool forces_shuffle_after() const { return false; }
    // DETERMINISTIC graphs: true if the operator itself merges the batch lanes that feed it in timestamp order (keyed GPU
    // operators do, through the device KeyBy endpoint); otherwise batches coming from a different number of lanes are
    // turned back into tuples so that the ordering nodes of the tuple world apply
    virtual bool merges_batch_lanes_in_order() const { return false; }
    // true for a keyed operator that was promised key-disjoint batch lanes (withKeyDisjointLanes): lane i feeds replica i
    // directly, whatever routing the operator would use on tuples
    virtual bool takes_batch_lanes_one_to_one() const { return false; }
    // statistics of every replica (filled at the end of the run)
    virtual std::vector<Stats_Record> get_StatsRecords() const {
        std::lock_guard<std::mutex> l(hub_->m);
        return hub_->recs;
    }

    // tuples discarded by the replicas (late tuples of window operators, reference: getNumDroppedTuples,
    // wf/win_farm.hpp:592-617; tuples rejected by a Filter); valid once the PipeGraph has ended
    size_t getNumDroppedTuples() const {
        size_t n = 0;
        for (const Stats_Record& r : get_StatsRecords()) n += r.inputs_ignored;
        return n;
    }

    void mark_used() { used_ = true; }
    std::shared_ptr<StatsHub> stats_hub() const { return hub_; }

protected:
    bool used_ = false;
    std::shared_ptr<StatsHub> hub_ = std::make_shared<StatsHub>();
};

}  // namespace wf
