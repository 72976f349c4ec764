// The non-incremental GPU window operators: Win_Seq_GPU, Key_Farm_GPU, Win_Farm_GPU,
// Pane_Farm_GPU, Win_MapReduce_GPU.  Windows are evaluated on the device by an
// arbitrary user functor
//     __host__ __device__ void(uint64_t gwid, const tuple_t* data, size_t n,
//                              result_t* res, char* scratch, size_t scratch_size)
// (/root/reference/API:102-152) or by wave_reduce(lift, combine).
//
// Capability parity with /root/reference/wf/win_seq_gpu.hpp:83-669 and
// wf/{win_farm,key_farm,pane_farm,win_mapreduce}_gpu.hpp.  Every replica is a
// WinBatchGPU_Node (win_batch_gpu.hpp): keyed state, window bookkeeping and
// evaluation all live on the device; the host threads only move batches.
//  * fed by batches (Batch_Source, Device_Source, another GPU operator, an
//    Exchange_GPU) a flat operator keeps the stream on the device end to end; replica
//    i takes lane i when the lanes carry disjoint key sets, otherwise the batches go
//    through the device KeyBy.  Win_Farm_GPU takes the same path: on a GPU the
//    windows of a key are evaluated in parallel by the waves of one kernel, so
//    "consecutive windows on different replicas" is what the hardware does inside a
//    replica, and replicas only divide the keys;
//  * fed by tuples (CPU producers) the replicas stage them into pinned batches; the
//    composite operators (window farms, pane farms, map-reduce, nesting) are the CPU
//    composition classes instantiated with a GPU engine maker, the engine followed
//    by its device->host and un-batching adapters in the same thread.
#pragma once
#include "../key_farm.hpp"
#include "../pane_farm.hpp"
#include "../win_farm.hpp"
#include "../win_mapreduce.hpp"
#include "../win_seq.hpp"
#include "batch_ops.hpp"
#include "../dist/exchange_nodes.hpp"
#include "keyby_gpu.hpp"
#include "win_batch_gpu.hpp"

namespace wf {

// engine maker for the composition classes (two_stage.hpp): a GPU replica fed by tuples whose results leave as tuples
template <typename in_t, typename out_t, typename F>
inline win_engine_maker_t gpu_win_engine(F func, GPUConfig cfg, gpu::WinGPUOptions opt, meta::closing_func_t closing, std::shared_ptr<StatsHub> hub) {
    return [func, cfg, opt, closing, hub](rt::Graph& g, const WinParams& p, const std::string& name, size_t par, size_t idx) -> WinEngineNodes {
        rt::Node* eng = g.adopt(new WinBatchGPU_Node<in_t, out_t, F>(func, p, opt, cfg, EDGE_TUPLE, name, par, idx, closing, hub));
        auto adapters = batch_adapter_factory<out_t>(cfg.pool_slabs);
        rt::Node* d2h = g.adopt(adapters(EDGE_DEVICE_BATCH));
        rt::Node* unb = g.adopt(adapters(EDGE_HOST_BATCH));
        g.connect_fused(eng, eng->add_out_port(), d2h);
        g.connect_fused(d2h, d2h->add_out_port(), unb);
        return {eng, unb};
    };
}

// Stage construction of a flat (single-stage) GPU window operator whose replicas own disjoint key sets.
template <typename tuple_t, typename result_t>
class GPU_Keyed_Window_Stage {
public:
    using node_factory_t = std::function<rt::Node*(const WinParams&, const gpu::WinGPUOptions&, edge_kind_t, size_t par, size_t idx)>;
    GPU_Keyed_Window_Stage(node_factory_t factory, uint64_t win, uint64_t slide, uint64_t delay, win_type_t type, size_t par, std::string name,
                           GPUConfig cfg, gpu::WinGPUOptions opt, routing_func_t routing)
        : factory_(std::move(factory)), win_(win), slide_(slide), delay_(delay), type_(type), par_(par), name_(std::move(name)), cfg_(cfg), opt_(opt),
          routing_(std::move(routing)) {}

    edge_kind_t kind = EDGE_TUPLE;
    size_t slabs() const { return cfg_.pool_slabs; }
    bool key_disjoint() const { return opt_.lanes_are_key_disjoint; }
    // tuple producers fill batches themselves (count-based windows keep the reference's rule: DETERMINISTIC graphs only, which
    // the tuple path enforces)
    std::function<rt::Node*(size_t, size_t)> batcher(Mode m) const {
        if (opt_.lanes_are_key_disjoint || (type_ == CB && m != Mode::DETERMINISTIC)) return nullptr;
        return batcher_factory<tuple_t>(cfg_);
    }

    std::vector<StageSpec> build(rt::Graph& g, std::shared_ptr<StatsHub> hub_ = nullptr, bool deterministic = false) {
        StageSpec s;
        s.name = name_;
        const size_t N = par_;
        WinParams p;
        p.win_len = win_;
        p.slide_len = slide_;
        p.triggering_delay = delay_;
        p.winType = type_;
        p.config = WinOperatorConfig(0, 1, slide_, 0, 1, slide_);
        p.role = SEQ;
        gpu::WinGPUOptions opt = opt_;
        // batches carry no per-key renumbering stage in front of them: count-based windows count arrivals on the device
        if (kind != EDGE_TUPLE && type_ == CB) opt.renumber = true;
        // time-based windows behind several unsynchronised producers (DEFAULT graph): every replica sees a merge of sorted pieces
        if (type_ == TB && !deterministic && !opt.lanes_are_key_disjoint) opt.expect_disorder = true;
        const bool keyby = kind != EDGE_TUPLE && !opt.lanes_are_key_disjoint && (N > 1 || deterministic);
        const edge_kind_t node_kind = keyby ? EDGE_DEVICE_BATCH : kind;
        for (size_t i = 0; i < N; i++) s.replicas.push_back(g.adopt(factory_(p, opt, node_kind, N, i)));
        if (kind == EDGE_TUPLE) {
            s.routing = COMPLEX;
            routing_func_t rf = routing_;
            s.make_emitter = [rf, N](rt::Graph& gr, size_t) { return single_emitter(gr, new KF_Emitter<tuple_t>(rf, N), N); };
            s.make_ordering = window_ordering<tuple_t>(type_, name_);
        } else if (keyby) {
            // device KeyBy (dist/exchange_nodes.hpp): producers pack their batches by replica, one endpoint merges them per replica
            std::vector<StageSpec> stages = device_keyby_stages<tuple_t>(g, name_, cfg_, N, kind, hub_, deterministic);
            s.routing = FORWARD;
            s.make_emitter = [N](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<tuple_t>(N), N); };
            stages.push_back(s);
            return stages;
        } else {
            s.routing = FORWARD;  // lane i -> replica i
            s.make_emitter = [N](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<tuple_t>(N), N); };
        }
        return {s};
    }

private:
    node_factory_t factory_;
    uint64_t win_, slide_, delay_;
    win_type_t type_;
    size_t par_;
    std::string name_;
    GPUConfig cfg_;
    gpu::WinGPUOptions opt_;
    routing_func_t routing_;
};

template <typename tuple_t, typename result_t, typename F>
inline typename GPU_Keyed_Window_Stage<tuple_t, result_t>::node_factory_t gpu_window_node_factory(
    F func, GPUConfig cfg, std::string name, meta::closing_func_t closing, std::shared_ptr<StatsHub> hub,
    std::shared_ptr<std::vector<gpu::WinCounters>> counters) {
    return [=](const WinParams& p, const gpu::WinGPUOptions& opt, edge_kind_t kind, size_t par, size_t idx) -> rt::Node* {
        return new WinBatchGPU_Node<tuple_t, result_t, F>(func, p, opt, cfg, kind, name, par, idx, closing, hub, counters);
    };
}

#define WF_COMMA ,
// which of the two stage functions are device functors (6 arguments, or wave_reduce)
namespace meta {
template <typename F>
constexpr bool is_gpu_stage_func() {
    if constexpr (gpu::is_wave_reduce_t<F>::value)
        return true;
    else
        return is_gpu_win_func<F>();
}
}  // namespace meta

// ---- both stages of a pane farm on the device (batch path) ----------------------------------------------------------
// One replica = two engines on one stream: the PLQ engine turns a batch of tuples into the batch of the panes it completes
// (tumbling windows of gcd(win, slide), role PLQ: results numbered per key), the WLQ engine consumes that device batch and
// fires the windows (count-based windows of win / pane pane results sliding by slide / pane, role WLQ).  Pane results never
// leave HBM.  (The reference needs exactly one of the two functions on the host: wf/pane_farm_gpu.hpp:124-548.)
template <typename tuple_t, typename result_t, typename F1, typename F2>
class PaneFarmBatchGPU_Node : public GPU_Replica<tuple_t> {
public:
    PaneFarmBatchGPU_Node(F1 plq, F2 wlq, WinParams p, gpu::WinGPUOptions opt, GPUConfig cfg, edge_kind_t in_kind, std::string name, size_t par, size_t idx,
                          meta::closing_func_t closing, std::shared_ptr<StatsHub> hub, std::shared_ptr<std::vector<gpu::WinCounters>> counters)
        : GPU_Replica<tuple_t>(cfg, in_kind, std::move(name), par, idx, std::move(closing), std::move(hub), true), plq_f_(plq), wlq_f_(wlq), p_(p), opt_(opt),
          counters_(std::move(counters)) {
        if (p_.win_len == 0 || p_.slide_len == 0) fatal_error("window length or slide cannot be zero");
    }

protected:
    void process_device(gpu::Batch* d) override {
        ensure_engines(d->capacity);
        hipStream_t s = this->stream();
        gpu::Batch* panes = plq_->process(d, s, this->num_kernels_);
        this->release_input(d);
        gpu::Batch* out = wlq_->process(panes, s, this->num_kernels_);
        panes->release(s);
        this->emit_device(out);
        if (++since_check_ >= 256) {
            since_check_ = 0;
            check(plq_->read_counters(s), "PLQ");
            check(wlq_->read_counters(s), "WLQ");
        }
    }
    void gpu_flush() override {
        if (!plq_) return;
        hipStream_t s = this->stream();
        size_t cursor = 0;
        gpu::Batch* panes = nullptr;
        while (plq_->flush_range(cursor, s, this->num_kernels_, panes)) {
            gpu::Batch* out = wlq_->process(panes, s, this->num_kernels_);
            panes->release(s);
            this->emit_device(out);
        }
        cursor = 0;
        gpu::Batch* out = nullptr;
        while (wlq_->flush_range(cursor, s, this->num_kernels_, out)) this->emit_device(out);
        const gpu::WinCounters c1 = plq_->read_counters(s), c2 = wlq_->read_counters(s);
        if (counters_) (*counters_)[this->context.getReplicaIndex()] = c1;
        dropped_ = c1.dropped;
        check(c1, "PLQ");
        check(c2, "WLQ");
    }
    void finalize_stats(Stats_Record& st) override {
        GPU_Replica<tuple_t>::finalize_stats(st);
        st.inputs_ignored = dropped_;
    }

private:
    void ensure_engines(size_t batch_capacity) {
        if (plq_) return;
        const uint64_t pane = std::gcd(p_.win_len, p_.slide_len);
        WinParams p1 = p_;
        p1.win_len = pane;
        p1.slide_len = pane;
        p1.config = WinOperatorConfig(0, 1, pane, 0, 1, pane);
        p1.role = PLQ;
        gpu::WinGPUOptions o1 = opt_;
        o1.out_capacity = 0;  // panes per batch: the engine's default (twice the input batch), the rest is deferred
        if (p_.winType == TB && o1.carry_capacity == 0) o1.carry_capacity = 1024;
        plq_.reset(new gpu::WinBatchEngine<tuple_t, result_t, F1>(plq_f_, p1, o1, batch_capacity, this->cfg_.gpu_id, this->cfg_.pool_slabs, this->cfg_.scratchpad_size));
        WinParams p2;
        p2.win_len = p_.win_len / pane;
        p2.slide_len = p_.slide_len / pane;
        p2.triggering_delay = 0;
        p2.winType = CB;
        p2.config = WinOperatorConfig(0, 1, p2.slide_len, 0, 1, p2.slide_len);
        p2.role = WLQ;
        gpu::WinGPUOptions o2 = opt_;
        o2.renumber = false;     // the ids of the pane results ARE the pane numbers
        o2.carry_capacity = 0;   // count-based: the window length
        o2.expect_disorder = false;
        wlq_.reset(new gpu::WinBatchEngine<result_t, result_t, F2>(wlq_f_, p2, o2, plq_->geometry().out_cap, this->cfg_.gpu_id, this->cfg_.pool_slabs,
                                                                  this->cfg_.scratchpad_size));
    }
    void check(const gpu::WinCounters& c, const char* stage) {
        if (c.table_full > 0)
            fatal_error(std::string("Pane_Farm_GPU (") + stage + "): tuples without a state slot - more keys than withMaxKeys(), or the reserved key 2^64-1");
        if (c.carry_overflow > 0)
            fatal_error(std::string("Pane_Farm_GPU (") + stage + "): " + std::to_string(c.carry_overflow) +
                        " tuples needed by open windows did not fit their key's archive; raise withArchiveCapacity()");
        if (c.work_overflow > 0) fatal_error(std::string("Pane_Farm_GPU (") + stage + "): work buffer overflow (internal sizing error)");
    }
    F1 plq_f_;
    F2 wlq_f_;
    WinParams p_;
    gpu::WinGPUOptions opt_;
    std::shared_ptr<std::vector<gpu::WinCounters>> counters_;
    std::unique_ptr<gpu::WinBatchEngine<tuple_t, result_t, F1>> plq_;
    std::unique_ptr<gpu::WinBatchEngine<result_t, result_t, F2>> wlq_;
    uint64_t dropped_ = 0;
    uint32_t since_check_ = 0;
};

// Base of the two-stage GPU operators: what Key_Farm_GPU / Win_Farm_GPU need to replicate their batch path
template <typename tuple_t, typename result_t>
class TwoStage_GPU_Mixin {
public:
    virtual ~TwoStage_GPU_Mixin() = default;
    // both stage functions run on the device: the operator has a batch path (replicas divide the keys, each runs both stages)
    bool has_batch_path() const { return (bool)batch_factory_; }
    std::unique_ptr<GPU_Keyed_Window_Stage<tuple_t, result_t>> make_flat_stage(size_t replicas, const std::string& name, routing_func_t routing) const {
        return std::unique_ptr<GPU_Keyed_Window_Stage<tuple_t, result_t>>(new GPU_Keyed_Window_Stage<tuple_t, result_t>(
            batch_factory_, bwin_, bslide_, bdelay_, btype_, replicas, name, bcfg_, bopt_, std::move(routing)));
    }
    std::shared_ptr<std::vector<gpu::WinCounters>> batch_counters(size_t replicas) {
        bcounters_->resize(std::max(bcounters_->size(), replicas));
        return bcounters_;
    }

protected:
    typename GPU_Keyed_Window_Stage<tuple_t, result_t>::node_factory_t batch_factory_;
    uint64_t bwin_ = 0, bslide_ = 0, bdelay_ = 0;
    win_type_t btype_ = CB;
    GPUConfig bcfg_;
    gpu::WinGPUOptions bopt_;
    std::shared_ptr<std::vector<gpu::WinCounters>> bcounters_ = std::make_shared<std::vector<gpu::WinCounters>>();
};

// Device first stage (keyed, fed by batches) -> device-to-host + un-batching adapters in the replicas' threads -> host second stage.
// The reference's own form of the two-stage GPU operators (one stage on the device, the other on the host:
// wf/pane_farm_gpu.hpp:124-337, wf/win_mapreduce_gpu.hpp:127-360) on the batch path: the first stage's replicas divide the KEYS
// (on a GPU the panes / stripes of a key are evaluated in parallel inside a replica), its results - numbered per key - come back
// as tuples for the second-stage replicas on the host.
template <typename tuple_t, typename result_t>
inline std::vector<StageSpec> hybrid_two_stage(rt::Graph& g, GPU_Keyed_Window_Stage<tuple_t, result_t>& first, std::shared_ptr<StatsHub> hub, Mode m,
                                               TwoStageGroup second, const std::string& name, bool ordered) {
    std::vector<StageSpec> stages = first.build(g, hub, m == Mode::DETERMINISTIC);
    StageSpec& dev = stages.back();
    auto adapters = batch_adapter_factory<result_t>(first.slabs());
    for (rt::Node* r : dev.replicas) {
        rt::Node* d2h = g.adopt(adapters(EDGE_DEVICE_BATCH));
        rt::Node* unb = g.adopt(adapters(EDGE_HOST_BATCH));
        g.connect_fused(r, r->add_out_port(), d2h);
        g.connect_fused(d2h, d2h->add_out_port(), unb);
        dev.outputs.push_back(unb);
    }
    auto grp = std::make_shared<TwoStageGroup>(std::move(second));
    StageSpec host;
    host.name = name + "_stage2";
    host.routing = COMPLEX;
    host.internal = true;
    host.replicas = grp->second;
    host.outputs = grp->second_outputs.empty() ? grp->second : grp->second_outputs;
    host.make_emitter = [grp](rt::Graph& gr, size_t) { return grp->second_emitter(gr, 0); };
    // no ordering node: the first-stage results of a key come from ONE device replica, in order (an ordering by id over the
    // channels would wait for ids from replicas that never see the key)
    const size_t second_par = host.replicas.size();
    stages.push_back(host);
    if (ordered && second_par > 1) {
        StageSpec c;
        c.name = name + "_collector";
        c.routing = FORWARD;
        c.internal = true;
        c.replicas.push_back(g.adopt(new Ordered_Collector<result_t>()));
        c.make_emitter = [](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<result_t>(1), 1); };
        stages.push_back(c);
    }
    return stages;
}

// members shared by the two flat two-stage GPU operators (edge negotiation of the batch paths).  flat_: both stages on the
// device, one keyed stage whose replicas run both; hybrid_: first stage on the device, second on the host (SECOND_GROUP builds
// the host replicas and the emitter that feeds them)
#define WF_TWO_STAGE_GPU_BATCH_PATH(BASE, SECOND_GROUP)                                                                                         \
    bool isGPUOperator() const override { return true; }                                                                                        \
    bool merges_batch_lanes_in_order() const override { return in_stage() && !in_stage()->key_disjoint(); }                                     \
    bool takes_batch_lanes_one_to_one() const override { return in_stage() && in_stage()->key_disjoint(); }                                     \
    std::function<rt::Node*(size_t, size_t)> input_batcher_factory(Mode m) const override {                                                     \
        return in_stage() && !this->prepared4nesting() ? in_stage()->batcher(m) : nullptr;                                                      \
    }                                                                                                                                           \
    edge_kind_t output_kind() const override { return batch_path() ? EDGE_DEVICE_BATCH : EDGE_TUPLE; }                                          \
    bool accepts_input_kind(edge_kind_t k) const override { return in_stage() ? true : k == EDGE_TUPLE; }                                       \
    void set_input_kind(edge_kind_t k) override {                                                                                               \
        if (in_stage()) in_stage()->kind = k;                                                                                                   \
    }                                                                                                                                           \
    std::function<rt::Node*(edge_kind_t)> output_adapter_factory() const override {                                                             \
        return batch_path() ? batch_adapter_factory<result_t>(flat_->slabs()) : nullptr;                                                        \
    }                                                                                                                                           \
    std::vector<StageSpec> build_stages(rt::Graph& g, Mode m) override {                                                                        \
        if (batch_path()) return flat_->build(g, this->hub_, m == Mode::DETERMINISTIC);                                                         \
        if (hybrid_ && hybrid_->kind != EDGE_TUPLE)                                                                                             \
            return hybrid_two_stage<tuple_t, result_t>(g, *hybrid_, this->hub_, m, SECOND_GROUP, this->getName(), hybrid_ordered_);             \
        return BASE::build_stages(g, m);                                                                                                        \
    }                                                                                                                                           \
    std::vector<gpu::WinCounters> getDeviceCounters() const { return *this->bcounters_; }                                                       \
                                                                                                                                                \
private:                                                                                                                                        \
    bool batch_path() const { return flat_ && flat_->kind != EDGE_TUPLE; }                                                                      \
    GPU_Keyed_Window_Stage<tuple_t, result_t>* in_stage() const { return flat_ ? flat_.get() : hybrid_.get(); }                                 \
    std::unique_ptr<GPU_Keyed_Window_Stage<tuple_t, result_t>> flat_, hybrid_;                                                                  \
    bool hybrid_ordered_ = true;

template <typename tuple_t, typename result_t>
class Pane_Farm_GPU : public Pane_Farm<tuple_t, result_t>, public TwoStage_GPU_Mixin<tuple_t, result_t> {
public:
    template <typename plq_t, typename wlq_t>
    Pane_Farm_GPU(plq_t plq, wlq_t wlq, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type, size_t plq_parallelism,
                  size_t wlq_parallelism, std::string name, GPUConfig cfg, meta::closing_func_t closing = nullptr, bool ordered = true,
                  opt_level_t opt_level = LEVEL0, gpu::WinGPUOptions opt = gpu::WinGPUOptions())
        : Pane_Farm<tuple_t, result_t>(typename Pane_Farm<tuple_t, result_t>::engines_tag{}, nullptr, nullptr, win_len, slide_len,
                                       triggering_delay, type, plq_parallelism, wlq_parallelism, name, ordered, opt_level) {
        constexpr bool gpu_plq = meta::is_gpu_stage_func<plq_t>();
        constexpr bool gpu_wlq = meta::is_gpu_stage_func<wlq_t>();
        static_assert(gpu_plq || gpu_wlq, "WindFlow Compilation Error - Pane_Farm_GPU: at least one of PLQ and WLQ must be a device functor");
        win_engine_maker_t m1, m2;
        if constexpr (gpu_plq)
            m1 = gpu_win_engine<tuple_t, result_t>(plq, cfg, opt, closing, this->hub_);
        else
            m1 = cpu_win_engine<tuple_t, result_t>(plq, closing, this->hub_);
        if constexpr (gpu_wlq)
            m2 = gpu_win_engine<result_t, result_t>(wlq, cfg, opt, closing, this->hub_);
        else
            m2 = cpu_win_engine<result_t, result_t>(wlq, closing, this->hub_);
        this->set_engines(std::move(m1), std::move(m2));
        if constexpr (gpu_plq && gpu_wlq) {
            // both functions on the device: fed by batches (or by tuple producers through their batchers) the replicas divide the
            // keys and each runs PLQ and WLQ back to back on its stream
            this->bwin_ = win_len;
            this->bslide_ = slide_len;
            this->bdelay_ = triggering_delay;
            this->btype_ = type;
            this->bcfg_ = cfg;
            this->bopt_ = opt;
            auto hub = this->hub_;
            auto counters = this->bcounters_;
            this->batch_factory_ = [=](const WinParams& p, const gpu::WinGPUOptions& o, edge_kind_t kind, size_t par, size_t idx) -> rt::Node* {
                if (counters->size() < par) counters->resize(par);
                return new PaneFarmBatchGPU_Node<tuple_t, result_t, plq_t, wlq_t>(plq, wlq, p, o, cfg, kind, name, par, idx, closing, hub, counters);
            };
            flat_ = this->make_flat_stage(std::max(plq_parallelism, wlq_parallelism), name, default_routing);
        } else if constexpr (gpu_plq) {
            // The reference's own form - PLQ on the device, WLQ on the host (wf/pane_farm_gpu.hpp:124-337) - also gets a batch path:
            // the panes are the tumbling windows of a KEYED device stage (replicas divide the keys instead of the panes: on a GPU the
            // panes of a key are evaluated in parallel inside a replica), fed by the producers' batchers and the device KeyBy; its
            // pane results - one per pane and key, numbered per key - come back as tuples for the WLQ replicas on the host.
            const uint64_t pane = std::gcd(win_len, slide_len);
            auto hub = this->hub_;
            auto counters = this->bcounters_;
            auto factory = [=](const WinParams& p0, const gpu::WinGPUOptions& o, edge_kind_t kind, size_t par, size_t idx) -> rt::Node* {
                if (counters->size() < par) counters->resize(par);
                WinParams p = p0;
                p.role = PLQ;  // pane results are numbered per key: the ids the WLQ windows count
                return new WinBatchGPU_Node<tuple_t, result_t, plq_t>(plq, p, o, cfg, kind, name + "_plq", par, idx, closing, hub, counters);
            };
            gpu::WinGPUOptions o1 = opt;
            if (type == TB && o1.carry_capacity == 0) o1.carry_capacity = 1024;
            hybrid_.reset(new GPU_Keyed_Window_Stage<tuple_t, result_t>(factory, pane, pane, triggering_delay, type, plq_parallelism, name + "_plq", cfg, o1,
                                                                       default_routing));
            hybrid_ordered_ = ordered;
        }
    }
    WF_TWO_STAGE_GPU_BATCH_PATH(Pane_Farm<tuple_t WF_COMMA result_t>, this->make_wlq_group(g, this->getName()))
};

template <typename tuple_t, typename result_t>
class Win_MapReduce_GPU : public Win_MapReduce<tuple_t, result_t>, public TwoStage_GPU_Mixin<tuple_t, result_t> {
public:
    template <typename map_t, typename reduce_t>
    Win_MapReduce_GPU(map_t map, reduce_t reduce, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type,
                      size_t map_parallelism, size_t reduce_parallelism, std::string name, GPUConfig cfg, meta::closing_func_t closing = nullptr,
                      bool ordered = true, opt_level_t opt_level = LEVEL0, gpu::WinGPUOptions opt = gpu::WinGPUOptions())
        : Win_MapReduce<tuple_t, result_t>(typename Win_MapReduce<tuple_t, result_t>::engines_tag{}, nullptr, nullptr, win_len, slide_len,
                                           triggering_delay, type, map_parallelism, reduce_parallelism, name, ordered, opt_level) {
        constexpr bool gpu_map = meta::is_gpu_stage_func<map_t>();
        constexpr bool gpu_red = meta::is_gpu_stage_func<reduce_t>();
        static_assert(gpu_map || gpu_red, "WindFlow Compilation Error - Win_MapReduce_GPU: at least one of MAP and REDUCE must be a device functor");
        win_engine_maker_t m1, m2;
        if constexpr (gpu_map)
            m1 = gpu_win_engine<tuple_t, result_t>(map, cfg, opt, closing, this->hub_);
        else
            m1 = cpu_win_engine<tuple_t, result_t>(map, closing, this->hub_);
        if constexpr (gpu_red)
            m2 = gpu_win_engine<result_t, result_t>(reduce, cfg, opt, closing, this->hub_);
        else
            m2 = cpu_win_engine<result_t, result_t>(reduce, closing, this->hub_);
        this->set_engines(std::move(m1), std::move(m2));
        if constexpr (gpu_map && gpu_red && !gpu::is_wave_reduce_t<map_t>::value && !gpu::is_wave_reduce_t<reduce_t>::value) {
            // both functions on the device: every window is cut into map_parallelism stripes inside the engine (gpu::map_reduce)
            this->bwin_ = win_len;
            this->bslide_ = slide_len;
            this->bdelay_ = triggering_delay;
            this->btype_ = type;
            this->bcfg_ = cfg;
            this->bopt_ = opt;
            auto hub = this->hub_;
            auto counters = this->bcounters_;
            const auto mr = gpu::map_reduce(map, reduce, map_parallelism);
            using mr_t = std::decay_t<decltype(mr)>;
            this->batch_factory_ = [=](const WinParams& p, const gpu::WinGPUOptions& o, edge_kind_t kind, size_t par, size_t idx) -> rt::Node* {
                if (counters->size() < par) counters->resize(par);
                return new WinBatchGPU_Node<tuple_t, result_t, mr_t>(mr, p, o, cfg, kind, name, par, idx, closing, hub, counters);
            };
            flat_ = this->make_flat_stage(reduce_parallelism, name, default_routing);
        } else if constexpr (gpu_map && !gpu::is_wave_reduce_t<map_t>::value) {
            // The reference's own form - MAP on the device, REDUCE on the host (wf/win_mapreduce_gpu.hpp:127-360) - on the batch
            // path: a keyed device stage emits the map_parallelism partial results of every window (gpu::map_only; numbered
            // M * k + m per key, as the M MAP replicas would), the REDUCE replicas on the host combine them
            auto hub = this->hub_;
            auto counters = this->bcounters_;
            const auto mo = gpu::map_only(map, map_parallelism);
            using mo_t = std::decay_t<decltype(mo)>;
            const size_t M = map_parallelism;
            auto factory = [=](const WinParams& p0, const gpu::WinGPUOptions& o, edge_kind_t kind, size_t par, size_t idx) -> rt::Node* {
                if (counters->size() < par) counters->resize(par);
                WinParams p = p0;
                p.role = MAP;
                p.map_first = 0;
                p.map_stride = M;
                return new WinBatchGPU_Node<tuple_t, result_t, mo_t>(mo, p, o, cfg, kind, name + "_map", par, idx, closing, hub, counters);
            };
            hybrid_.reset(new GPU_Keyed_Window_Stage<tuple_t, result_t>(factory, win_len, slide_len, triggering_delay, type, reduce_parallelism,
                                                                       name + "_map", cfg, opt, default_routing));
            hybrid_ordered_ = ordered;
        }
    }
    WF_TWO_STAGE_GPU_BATCH_PATH(Win_MapReduce<tuple_t WF_COMMA result_t>, this->make_reduce_group(g, this->getName()))
};

// ---- operators -------------------------------------------------------------------
template <typename tuple_t, typename result_t>
class Key_Farm_GPU : public Key_Farm<tuple_t, result_t> {
    using base_t = Key_Farm<tuple_t, result_t>;

public:
    template <typename F, typename = std::enable_if_t<!std::is_base_of_v<Basic_Operator, F>>>
    Key_Farm_GPU(F func, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type, size_t parallelism, std::string name,
                 GPUConfig cfg, meta::closing_func_t closing = nullptr, routing_func_t routing = default_routing,
                 gpu::WinGPUOptions opt = gpu::WinGPUOptions())
        : base_t(typename base_t::engines_tag{}, nullptr, win_len, slide_len, triggering_delay, type, parallelism, name),
          counters_(std::make_shared<std::vector<gpu::WinCounters>>(parallelism)) {
        flat_.reset(new GPU_Keyed_Window_Stage<tuple_t, result_t>(gpu_window_node_factory<tuple_t, result_t>(func, cfg, name, closing, this->hub_, counters_),
                                                                 win_len, slide_len, triggering_delay, type, parallelism, name, cfg, opt,
                                                                 std::move(routing)));
    }
    // nesting: replicate a Pane_Farm_GPU / Win_MapReduce_GPU
    Key_Farm_GPU(TwoStage_Operator<tuple_t, result_t>& inner, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type,
                 size_t num_replicas, std::string name, meta::closing_func_t closing = nullptr, routing_func_t routing = default_routing,
                 opt_level_t opt_level = LEVEL2)
        : base_t(inner, win_len, slide_len, triggering_delay, type, num_replicas, name, closing, routing, opt_level) {
        // an inner operator whose two stages both run on the device has a batch path: the outer replicas then divide the keys and
        // each runs the inner operator's stages on its stream
        if (auto* two = dynamic_cast<TwoStage_GPU_Mixin<tuple_t, result_t>*>(&inner); two != nullptr && two->has_batch_path()) {
            counters_ = two->batch_counters(num_replicas);
            flat_ = two->make_flat_stage(num_replicas, name, routing);
        }
    }
    bool isGPUOperator() const override { return true; }
    bool merges_batch_lanes_in_order() const override { return flat_ && !flat_->key_disjoint(); }
    bool takes_batch_lanes_one_to_one() const override { return flat_ && flat_->key_disjoint(); }
    std::function<rt::Node*(size_t, size_t)> input_batcher_factory(Mode m) const override { return flat_ ? flat_->batcher(m) : nullptr; }
    edge_kind_t output_kind() const override { return flat_ ? EDGE_DEVICE_BATCH : EDGE_TUPLE; }
    bool accepts_input_kind(edge_kind_t k) const override { return flat_ ? true : k == EDGE_TUPLE; }
    void set_input_kind(edge_kind_t k) override {
        if (flat_) flat_->kind = k;
    }
    std::function<rt::Node*(edge_kind_t)> output_adapter_factory() const override {
        return flat_ ? batch_adapter_factory<result_t>(flat_->slabs()) : nullptr;
    }
    std::vector<StageSpec> build_stages(rt::Graph& g, Mode m) override { return flat_ ? flat_->build(g, this->hub_, m == Mode::DETERMINISTIC) : base_t::build_stages(g, m); }
    // per-replica device counters, valid after the end of the run
    std::vector<gpu::WinCounters> getDeviceCounters() const { return counters_ ? *counters_ : std::vector<gpu::WinCounters>(); }

private:
    std::shared_ptr<std::vector<gpu::WinCounters>> counters_;
    std::unique_ptr<GPU_Keyed_Window_Stage<tuple_t, result_t>> flat_;
};

// one replica (reference: WinSeqGPU_Builder, builders_gpu.hpp:51)
template <typename tuple_t, typename result_t>
class Win_Seq_GPU : public Key_Farm_GPU<tuple_t, result_t> {
public:
    template <typename F>
    Win_Seq_GPU(F func, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type, std::string name, GPUConfig cfg,
                meta::closing_func_t closing = nullptr, gpu::WinGPUOptions opt = gpu::WinGPUOptions())
        : Key_Farm_GPU<tuple_t, result_t>(func, win_len, slide_len, triggering_delay, type, 1, std::move(name), cfg, closing, default_routing, opt) {}
};

template <typename tuple_t, typename result_t>
class Win_Farm_GPU : public Win_Farm<tuple_t, result_t> {
    using base_t = Win_Farm<tuple_t, result_t>;

public:
    template <typename F, typename = std::enable_if_t<!std::is_base_of_v<Basic_Operator, F>>>
    Win_Farm_GPU(F func, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type, size_t parallelism, std::string name,
                 GPUConfig cfg, meta::closing_func_t closing = nullptr, bool ordered = true, gpu::WinGPUOptions opt = gpu::WinGPUOptions())
        : base_t(typename base_t::engines_tag{}, nullptr, win_len, slide_len, triggering_delay, type, parallelism, name, ordered),
          counters_(std::make_shared<std::vector<gpu::WinCounters>>(parallelism)) {
        // tuples: window-parallel replicas, as on the CPU
        this->set_engine(gpu_win_engine<tuple_t, result_t>(func, cfg, opt, closing, this->hub_));
        // batches: replicas divide the keys, the windows of a key are evaluated in parallel inside the replica
        flat_.reset(new GPU_Keyed_Window_Stage<tuple_t, result_t>(gpu_window_node_factory<tuple_t, result_t>(func, cfg, name, closing, this->hub_, counters_),
                                                                 win_len, slide_len, triggering_delay, type, parallelism, name, cfg, opt,
                                                                 default_routing));
    }
    Win_Farm_GPU(TwoStage_Operator<tuple_t, result_t>& inner, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type,
                 size_t num_replicas, std::string name, meta::closing_func_t closing = nullptr, bool ordered = true, opt_level_t opt_level = LEVEL2)
        : base_t(inner, win_len, slide_len, triggering_delay, type, num_replicas, name, closing, ordered, opt_level) {
        if (auto* two = dynamic_cast<TwoStage_GPU_Mixin<tuple_t, result_t>*>(&inner); two != nullptr && two->has_batch_path()) {
            counters_ = two->batch_counters(num_replicas);
            flat_ = two->make_flat_stage(num_replicas, name, default_routing);
        }
    }
    bool isGPUOperator() const override { return true; }
    bool merges_batch_lanes_in_order() const override { return flat_ && !flat_->key_disjoint(); }
    bool takes_batch_lanes_one_to_one() const override { return flat_ && flat_->key_disjoint(); }
    std::function<rt::Node*(size_t, size_t)> input_batcher_factory(Mode m) const override { return flat_ ? flat_->batcher(m) : nullptr; }
    edge_kind_t output_kind() const override { return batch_path() ? EDGE_DEVICE_BATCH : EDGE_TUPLE; }
    bool accepts_input_kind(edge_kind_t k) const override { return flat_ ? true : k == EDGE_TUPLE; }
    void set_input_kind(edge_kind_t k) override {
        if (flat_) flat_->kind = k;
    }
    std::function<rt::Node*(edge_kind_t)> output_adapter_factory() const override {
        return batch_path() ? batch_adapter_factory<result_t>(flat_->slabs()) : nullptr;
    }
    std::vector<StageSpec> build_stages(rt::Graph& g, Mode m) override { return batch_path() ? flat_->build(g, this->hub_, m == Mode::DETERMINISTIC) : base_t::build_stages(g, m); }
    std::vector<gpu::WinCounters> getDeviceCounters() const { return counters_ ? *counters_ : std::vector<gpu::WinCounters>(); }

private:
    bool batch_path() const { return flat_ && flat_->kind != EDGE_TUPLE; }
    std::shared_ptr<std::vector<gpu::WinCounters>> counters_;
    std::unique_ptr<GPU_Keyed_Window_Stage<tuple_t, result_t>> flat_;
};

}  // namespace wf
