// Exchange_GPU: KeyBy across GPUs.  Every rank (process or thread owning one GPU)
// places one Exchange_GPU in its MultiPipe; batches entering it are partitioned by
// key on the device and re-distributed with one all-to-all per round, so that a key
// is only ever seen by the rank that owns it (owner = bits 48-63 of mix64(key) mod
// world).  Downstream keyed operators (Key_FFAT_GPU, keyed Map_GPU ...) then run
// unchanged on purely local state.
//
// No counterpart in the reference (single process, no inter-GPU traffic, SURVEY.md
// §2.6); this is the piece that turns "N replicas on N GPUs" into a scalable
// one-process-per-GPU deployment over xGMI.
//
// Protocol (all ranks run the same sequence of collectives, in lock step):
//   lanes:  partition every input batch into `world` device sub-batches (stage 1, one thread per lane)
//   round:  1. the endpoint (stage 2, one thread per rank) takes the next partitioned batch, possibly none
//           2. all-gather {counts[world], done, step punctuation, watermark}
//           3. all-to-all-v with the exact sizes learned in 2; slices land
//              rank-major in one output batch, so per-key order is (round, source
//              rank, position): FIFO per producer is preserved
//           4. emit the output batch (count known on the host, no further sync)
//   a rank without input contributes an empty round (from on_idle, rate limited);
//   at end of stream a rank keeps contributing empty rounds with done = 1 until
//   every rank is done.  Step punctuations (header.reserved[0], see Batch_Source)
//   are delivered downstream once every rank has reported them.
#pragma once
#include <cstddef>
#include <map>

#include "../builders_gpu.hpp"
#include "../gpu/keyby_gpu.hpp"
#include "transport.hpp"

namespace wf {

struct ExchangeConfig {
    size_t local_lanes = 1;         // producers feeding this exchange on this rank (punctuations to wait for per step)
    uint64_t idle_round_us = 200;   // with no input, contribute an empty round this often
};

// What a partition lane hands to the rank's endpoint: the `world` device sub-batches of one input
// batch with their (host-known) counts.  parts is empty when world == 1 (pass-through of `whole`).
struct ExchangeBundle {
    std::vector<gpu::Batch*> parts;
    std::vector<uint32_t> counts;
    gpu::Batch* whole = nullptr;
    gpu::BatchHeader hdr{};
    size_t capacity = 0;
};

// ---- stage 1: partition lanes (parallel) ------------------------------------------------------
// Ships the batch to the device if needed and splits it by owner rank.  The lane waits for its own
// stream so that the counts travel with the bundle: the endpoint never waits for a lane's kernels.
template <typename tuple_t>
class ExchangePart_Node : public GPU_Replica<tuple_t> {
public:
    ExchangePart_Node(int world, GPUConfig cfg, edge_kind_t in_kind, std::string name, size_t par, size_t idx, meta::closing_func_t closing,
                      std::shared_ptr<StatsHub> hub)
        : GPU_Replica<tuple_t>(cfg, in_kind, std::move(name), par, idx, std::move(closing), std::move(hub)), world_(world) {}

protected:
    void gpu_init() override {
        if (world_ > 1)
            part_.reset(new gpu::BatchPartitioner<tuple_t>((size_t)world_, this->cfg_.gpu_id, std::max<size_t>(2, this->cfg_.pool_slabs / 2),
                                                           gpu::HASH_SHIFT_RANKS));
    }
    void process_device(gpu::Batch* d) override {
        gpu::TraceRange trace("wf:exchange_partition");
        auto* b = new ExchangeBundle();
        b->hdr = d->host_hdr;
        b->capacity = d->capacity;
        hipStream_t s = this->stream();
        if (world_ == 1) {
            d->ready.record(s);
            b->whole = d;
        } else {
            part_->partition(d, s, b->parts, this->num_kernels_);
            d->release(s);
            WF_HIP_CHECK(hipStreamSynchronize(s));  // the sub-batch counts are needed on the host to size the exchange
            b->counts.resize((size_t)world_);
            for (int p = 0; p < world_; p++) b->counts[p] = *b->parts[p]->count_mirror;
            for (gpu::Batch* p : b->parts) p->ready.record(s);
        }
        this->emit(b);
    }

private:
    int world_;
    std::unique_ptr<gpu::BatchPartitioner<tuple_t>> part_;
};

// ---- stage 2: the rank's endpoint on the fabric (one thread) -------------------------------------
template <typename tuple_t>
class ExchangeEndpoint_Node : public Replica_Base {
public:
    ExchangeEndpoint_Node(std::shared_ptr<dist::Transport> tr, ExchangeConfig xcfg, GPUConfig cfg, std::string name, meta::closing_func_t closing,
                          std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), 1, 0, std::move(closing), std::move(hub), false, true), tr_(std::move(tr)), xcfg_(xcfg), cfg_(cfg) {
        if (!tr_) gpu::failwith("Exchange_GPU: no transport");
        if (tr_->world() > gpu::MAX_PARTITIONS) gpu::failwith("Exchange_GPU: world size above MAX_PARTITIONS");
    }
    void svc_init() override {
        Replica_Base::svc_init();
        WF_HIP_CHECK(hipSetDevice(cfg_.gpu_id));
        stream_.reset(new gpu::Stream());
        last_round_us_ = current_time_usecs();
    }
    void svc(void* msg, int) override {
        std::unique_ptr<ExchangeBundle> b(static_cast<ExchangeBundle*>(msg));
        round(b.get(), false);
    }
    bool wants_idle() const override { return tr_->world() > 1; }
    void on_idle() override {
        if (finished_) return;
        if (current_time_usecs() - last_round_us_ < xcfg_.idle_round_us) return;
        round(nullptr, false);
    }
    void flush() override {
        // end of my stream: keep serving rounds until everybody is done
        while (!finished_) round(nullptr, true);
    }
    void svc_end() override {
        if (stream_) stream_->sync();
        Replica_Base::svc_end();
    }
    uint64_t rounds() const { return rounds_; }
    uint64_t bytes_received() const { return bytes_exchanged_; }
    uint64_t max_link_bytes_per_round() const { return max_link_bytes_; }

protected:
    void finalize_stats(Stats_Record& s) override { s.num_kernels = rounds_; }

private:
    static constexpr size_t META = gpu::MAX_PARTITIONS + 5;  // counts[], done, step, watermark, created_ns, batch capacity

    void round(ExchangeBundle* b, bool done) {
        gpu::TraceRange trace("wf:exchange_round");
        const int world = tr_->world(), me = tr_->rank();
        hipStream_t s = stream_->get();
        uint64_t mine[META] = {};
        if (b != nullptr) {
            my_cap_ = std::max(my_cap_, b->capacity);
            if (b->hdr.reserved[0] != 0) note_local_punctuation(b->hdr.reserved[0]);
            if (world == 1) {  // nothing to exchange: pass through (punctuations are still merged over the local lanes)
                gpu::Batch* d = b->whole;
                d->ready.wait_on(s);
                const uint64_t step = take_global_step(local_step_done_);
                if (step != d->host_hdr.reserved[0]) {
                    d->host_hdr.reserved[0] = step;
                    WF_HIP_CHECK(hipMemcpyAsync(d->slab + offsetof(gpu::BatchHeader, reserved), &d->host_hdr.reserved[0], sizeof(uint64_t),
                                                hipMemcpyHostToDevice, s));
                }
                d->ready.record(s);
                emit(d);
                return;
            }
            for (int p = 0; p < world; p++) {
                mine[p] = b->counts[p];
                b->parts[p]->ready.wait_on(s);
            }
            mine[gpu::MAX_PARTITIONS + 2] = b->hdr.watermark;
            mine[gpu::MAX_PARTITIONS + 3] = b->hdr.created_ns;
        } else if (world == 1) {
            finished_ = done;
            return;
        }
        mine[gpu::MAX_PARTITIONS + 0] = done ? 1 : 0;
        mine[gpu::MAX_PARTITIONS + 1] = local_step_done_;
        mine[gpu::MAX_PARTITIONS + 4] = my_cap_;
        uint64_t all[META * dist::MAX_WORLD];
        tr_->all_gather_host(mine, all, META, s);
        // what I receive, rank-major
        size_t total = 0;
        bool all_done = true, any_traffic = false;
        uint64_t min_step = UINT64_MAX, created = 0;
        size_t recv_cnt[dist::MAX_WORLD];
        for (int r = 0; r < world; r++) {
            const uint64_t* row = all + (size_t)r * META;
            recv_cnt[r] = (size_t)row[me];
            total += recv_cnt[r];
            for (int p = 0; p < world; p++) any_traffic = any_traffic || row[p] != 0;
            all_done = all_done && row[gpu::MAX_PARTITIONS + 0] != 0;
            min_step = std::min(min_step, row[gpu::MAX_PARTITIONS + 1]);
            // per-rank high marks; what I forward is the LOW watermark: the minimum over the ranks still producing
            rank_wm_[r] = std::max(rank_wm_[r], row[gpu::MAX_PARTITIONS + 2]);
            rank_done_[r] = row[gpu::MAX_PARTITIONS + 0] != 0;
            cap_ = std::max<size_t>(cap_, (size_t)row[gpu::MAX_PARTITIONS + 4]);
            if (row[gpu::MAX_PARTITIONS + 3] != 0 && recv_cnt[r] > 0)
                created = created == 0 ? row[gpu::MAX_PARTITIONS + 3] : std::min(created, row[gpu::MAX_PARTITIONS + 3]);
        }
        uint64_t wm = UINT64_MAX;
        for (int r = 0; r < world; r++)
            if (!rank_done_[r]) wm = std::min(wm, rank_wm_[r]);
        if (wm == UINT64_MAX)
            for (int r = 0; r < world; r++) wm = (r == 0) ? rank_wm_[r] : std::max(wm, rank_wm_[r]);
        const uint64_t step_out = take_global_step(min_step);
        gpu::Batch* out = nullptr;
        if (any_traffic) {
            if (total > 0 || step_out != 0) out = acquire_out(total);
            dist::PeerSlices sl;
            size_t off = 0;
            for (int p = 0; p < world; p++) {
                if (b != nullptr) {
                    sl.send[p] = b->parts[p]->template data<tuple_t>();
                    sl.send_bytes[p] = (size_t)mine[p] * sizeof(tuple_t);
                }
                if (recv_cnt[p] > 0) {
                    sl.recv[p] = out->template data<tuple_t>() + off;
                    sl.recv_bytes[p] = recv_cnt[p] * sizeof(tuple_t);
                    off += recv_cnt[p];
                }
            }
            tr_->all_to_all_v(sl, s);
            bytes_exchanged_ += off * sizeof(tuple_t);
            // xGMI is point to point: the exchange is bound by the busiest link, not by the total
            for (int p = 0; p < world; p++)
                if (p != me) max_link_bytes_ = std::max<uint64_t>(max_link_bytes_, std::max(sl.send_bytes[p], sl.recv_bytes[p]));
        } else if (step_out != 0) {
            out = acquire_out(0);
        }
        if (b != nullptr)
            for (gpu::Batch* p : b->parts) p->release(s);
        if (out != nullptr) {
            out->host_hdr.count = (uint32_t)total;
            out->host_hdr.source_id = (uint32_t)me;
            out->host_hdr.watermark = wm;
            out->host_hdr.created_ns = created;
            out->host_hdr.first_seq = seq_;
            out->host_hdr.reserved[0] = step_out;
            out->count_on_host = true;
            seq_ += total;
            WF_HIP_CHECK(hipMemcpyAsync(out->slab, &out->host_hdr, sizeof(gpu::BatchHeader), hipMemcpyHostToDevice, s));
            out->ready.record(s);
            emit(out);
        }
        rounds_++;
        last_round_us_ = current_time_usecs();
        if (all_done && !any_traffic) finished_ = true;
    }

    gpu::Batch* acquire_out(size_t total) {
        // worst case every rank sends me a full batch (cap_ = largest batch capacity announced by any rank)
        const size_t need = std::max<size_t>(std::max(cap_ * (size_t)tr_->world(), total), 1);
        if (!out_pool_ || out_cap_ < need) {
            // batches of the old pool may still be in flight downstream: keep it alive until the end
            if (out_pool_) retired_pools_.push_back(std::move(out_pool_));
            out_cap_ = need;
            out_pool_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, cfg_.gpu_id, sizeof(tuple_t), out_cap_, cfg_.pool_slabs));
        }
        return out_pool_->acquire(stream_->get());
    }

    // a step is locally complete when every local lane has punctuated it
    void note_local_punctuation(uint64_t step) {
        if (++punct_seen_[step] >= xcfg_.local_lanes) {
            local_step_done_ = std::max(local_step_done_, step);
            punct_seen_.erase(step);
        }
    }
    // returns the step id to announce downstream (0: none) given the minimum step all ranks completed
    uint64_t take_global_step(uint64_t min_step) {
        if (min_step != UINT64_MAX && min_step > announced_step_) {
            announced_step_ = min_step;
            return min_step;
        }
        return 0;
    }

    std::shared_ptr<dist::Transport> tr_;
    ExchangeConfig xcfg_;
    GPUConfig cfg_;
    std::unique_ptr<gpu::Stream> stream_;
    std::unique_ptr<gpu::BatchPool> out_pool_;
    std::vector<std::unique_ptr<gpu::BatchPool>> retired_pools_;
    size_t my_cap_ = 0, cap_ = 0, out_cap_ = 0;
    uint64_t seq_ = 0, rounds_ = 0, bytes_exchanged_ = 0, last_round_us_ = 0;
    uint64_t local_step_done_ = 0, announced_step_ = 0, max_link_bytes_ = 0;
    uint64_t rank_wm_[dist::MAX_WORLD] = {};
    bool rank_done_[dist::MAX_WORLD] = {};
    std::map<uint64_t, size_t> punct_seen_;
    bool finished_ = false;
};

// The operator: `parallelism` partition lanes (give it the parallelism of the producer so that every
// lane ships and splits its own batches) feeding the rank's single endpoint.
template <typename tuple_t>
class Exchange_GPU : public Basic_Operator {
public:
    Exchange_GPU(std::shared_ptr<dist::Transport> transport, size_t parallelism = 1, std::string name = "exchange_gpu", GPUConfig cfg = GPUConfig(),
                 ExchangeConfig xcfg = ExchangeConfig(), meta::closing_func_t closing = nullptr)
        : tr_(std::move(transport)), par_(parallelism), name_(std::move(name)), cfg_(cfg), xcfg_(xcfg), closing_(std::move(closing)) {
        if (par_ == 0) fatal_error("operator " + name_ + " has parallelism zero");
        if (!tr_) fatal_error("operator " + name_ + " needs a transport");
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

    std::vector<StageSpec> build_stages(rt::Graph& g, Mode) override {
        const size_t N = par_;
        StageSpec a;
        a.name = name_ + "_partition";
        a.routing = FORWARD;
        for (size_t i = 0; i < N; i++)
            a.replicas.push_back(g.adopt(new ExchangePart_Node<tuple_t>(tr_->world(), cfg_, kind_, name_ + "_partition", N, i, nullptr, hub_)));
        a.make_emitter = [N](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<tuple_t>(N), N); };
        StageSpec b;
        b.name = name_;
        b.routing = FORWARD;
        b.internal = true;
        b.replicas.push_back(g.adopt(new ExchangeEndpoint_Node<tuple_t>(tr_, xcfg_, cfg_, name_, closing_, hub_)));
        b.make_emitter = [](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<tuple_t>(1), 1); };
        return {a, b};
    }

private:
    std::shared_ptr<dist::Transport> tr_;
    size_t par_;
    std::string name_;
    GPUConfig cfg_;
    ExchangeConfig xcfg_;
    meta::closing_func_t closing_;
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
