// Win_Seq_GPU and the window operators built on it (Win_Farm_GPU, Key_Farm_GPU,
// Pane_Farm_GPU, Win_MapReduce_GPU): windows are evaluated on the device by an
// arbitrary user functor
//     __host__ __device__ void(uint64_t gwid, const tuple_t* data, size_t n,
//                              result_t* res, char* scratch, size_t scratch_size)
// (/root/reference/API:102-152).
//
// Capability parity with /root/reference/wf/win_seq_gpu.hpp:83-669 (window
// bookkeeping on the host, fired windows batched, one batch in flight per replica
// while the host prepares the next one, optional per-thread scratchpad) and with
// wf/{win_farm,key_farm,pane_farm,win_mapreduce}_gpu.hpp.  Differences by design:
//  * batches are formed ACROSS keys: a batch is "the next batch_len fired windows
//    of this replica", whatever their keys, so launches stay large when there are
//    many keys (the reference batches per key, win_seq_gpu.hpp:466);
//  * the tuples a batch needs are copied once per key into a pinned staging buffer
//    (the reference memcpy's from a pageable std::vector);
//  * the partial batch at end-of-stream runs through the same kernel (the
//    reference computes leftovers on the CPU, :562-620);
//  * the farm / pane / map-reduce GPU operators are the CPU composition classes
//    instantiated with a GPU engine maker instead of four more copies of the code.
// The functor is opaque and sequential over its window, hence one thread per
// window; associative queries should use Key_FFAT_GPU, which parallelises inside
// windows.
#pragma once
#include <deque>
#include <unordered_map>

#include "../key_farm.hpp"
#include "../pane_farm.hpp"
#include "../win_farm.hpp"
#include "../win_mapreduce.hpp"
#include "../win_seq.hpp"
#include "batch_ops.hpp"
#include "kernels_basic.hpp"

namespace wf {
namespace gpu {

struct WinDesc {
    uint64_t gwid;
    uint32_t off;  // first tuple of the window in the batch buffer
    uint32_t len;  // number of tuples
};

// K1: one thread per window of the batch
template <typename T, typename R, typename F>
__global__ void __launch_bounds__(BLOCK_THREADS)
    win_batch_kernel(const T* tuples, const WinDesc* desc, R* results, uint32_t n_windows, char* scratch, size_t scratch_size, F f) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    char* my_scratch = scratch_size ? scratch + (size_t)tid * scratch_size : nullptr;
    for (uint32_t w = tid; w < n_windows; w += gridDim.x * blockDim.x) {
        const WinDesc d = desc[w];
        f(d.gwid, tuples + d.off, (size_t)d.len, &results[w], my_scratch, scratch_size);
    }
}

}  // namespace gpu

template <typename tuple_t, typename result_t, typename F>
class Win_Seq_GPU_Node : public Replica_Base {
public:
    using key_t = key_of_t<tuple_t>;
    Win_Seq_GPU_Node(F func, WinParams p, GPUConfig cfg, std::string name, size_t par, size_t idx, meta::closing_func_t closing,
                     std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub), true, true), func_(func), p_(p), cfg_g_(cfg) {
        if (p_.win_len == 0 || p_.slide_len == 0) fatal_error("window length or slide cannot be zero");
        if (cfg_g_.batch_len == 0) fatal_error("batch length cannot be zero");
        cfg_g_.gpu_id = cfg.device_for(idx);
        gpu::preload_module(cfg_g_.gpu_id);
        set_io_bytes(sizeof(tuple_t), sizeof(result_t));
    }

    void svc_init() override {
        Replica_Base::svc_init();
        WF_HIP_CHECK(hipSetDevice(cfg_g_.gpu_id));
        stream_.reset(new gpu::Stream());
        const size_t B = cfg_g_.batch_len;
        h_desc_.resize(B);
        h_res_.resize(B);
        d_desc_.resize(B);
        d_res_.resize(B);
        grid_ = gpu::grid_for(B, (unsigned)cfg_g_.n_thread_block);
        if (cfg_g_.scratchpad_size > 0) d_scratch_.resize((size_t)grid_ * cfg_g_.n_thread_block * cfg_g_.scratchpad_size);
        pending_.reserve(B);
    }

    void svc(void* msg, int) override {
        tuple_t* t = untag<tuple_t>(msg);
        const bool marker = is_marker(msg);
        const key_t key = key_of(*t);
        const size_t hashcode = std::hash<key_t>()(key);
        const uint64_t ord = (p_.winType == CB) ? id_of(*t) : ts_of(*t);
        auto it = keys_.find(key);
        if (it == keys_.end()) it = keys_.emplace(key, KeyD(p_)).first;
        KeyD& kd = it->second;
        const WinAssign wa(p_.config, hashcode, p_.role);
        const uint64_t min_boundary = (kd.last_lwid >= 0) ? p_.win_len + (uint64_t)kd.last_lwid * p_.slide_len : 0;
        if (ord < wa.initial_id + min_boundary) {
            if (kd.last_lwid >= 0 && !marker) dropped_++;
            tuple_delete(t);
            return;
        }
        long last_w;
        const uint64_t rel = ord - wa.initial_id;
        if (p_.win_len >= p_.slide_len) {
            last_w = (long)((rel + 1 + p_.slide_len - 1) / p_.slide_len) - 1;
        } else {
            const uint64_t n = rel / p_.slide_len;
            last_w = (long)n;
            if (rel >= n * p_.slide_len + p_.win_len && !marker) {
                tuple_delete(t);
                return;
            }
        }
        if (!marker) kd.archive.insert(*t);
        for (long lwid = (long)kd.next_lwid; lwid <= last_w; lwid++) {
            const uint64_t gwid = wa.first_gwid_key + (uint64_t)lwid * wa.gwid_stride;
            kd.wins.emplace_back(key, (uint64_t)lwid, gwid, wa.initial_id + (uint64_t)lwid * p_.slide_len, p_.win_len, p_.winType);
            kd.next_lwid++;
        }
        const uint64_t fire_slack = (p_.winType == TB) ? p_.triggering_delay : 0;
        while (!kd.wins.empty() && ord >= kd.wins.front().end + fire_slack) {
            schedule(key, hashcode, kd, kd.wins.front());
            kd.wins.pop_front();
        }
        if (!marker && !kd.wins.empty() && p_.winType == CB) {
            // count-based results carry the timestamp of the last tuple that fell in the window
            for (auto& w : kd.wins)
                if (ord >= w.start && ord < w.end) w.result.setControlFields(key, w.gwid, ts_of(*t));
        }
        tuple_delete(t);
    }

    void flush() override {
        for (auto& kv : keys_) {
            const size_t hashcode = std::hash<key_t>()(kv.first);
            for (auto& w : kv.second.wins) schedule(kv.first, hashcode, kv.second, w);
            kv.second.wins.clear();
        }
        if (!pending_.empty()) launch_batch();
        drain();
    }

    void svc_end() override {
        if (stream_) stream_->sync();
        Replica_Base::svc_end();
    }
    uint64_t getNumDroppedTuples() const { return dropped_; }

protected:
    void finalize_stats(Stats_Record& s) override {
        s.inputs_ignored = dropped_;
        s.num_kernels = kernels_;
        s.bytes_copied_hd = bytes_hd_;
        s.bytes_copied_dh = bytes_dh_;
    }

private:
    struct OrdFn {
        win_type_t type;
        uint64_t operator()(const tuple_t& t) const { return type == CB ? id_of(t) : ts_of(t); }
    };
    struct KeyD {
        explicit KeyD(const WinParams& p) : archive(OrdFn{p.winType}), emit_counter(p.role == MAP ? p.map_first : 0) {}
        StreamArchive<tuple_t, OrdFn> archive;
        std::deque<Window<result_t, key_t>> wins;
        uint64_t emit_counter;
        uint64_t next_lwid = 0;
        long last_lwid = -1;
        // windows of this key scheduled in the batch under construction
        bool in_batch = false;
        uint64_t batch_lo = 0, batch_hi = 0, purge_to = 0;
    };
    struct Pending {
        KeyD* kd;
        uint64_t start, end;
        uint64_t gwid;
        result_t result;
    };

    // a fired window joins the batch under construction
    void schedule(const key_t& key, size_t hashcode, KeyD& kd, Window<result_t, key_t>& w) {
        kd.last_lwid++;
        result_t res = w.result;
        if (p_.role == MAP) {
            res.setControlFields(key, kd.emit_counter, ts_of(res));
            kd.emit_counter += p_.map_stride;
        } else if (p_.role == PLQ) {
            const auto& c = p_.config;
            const uint64_t new_id = ((c.id_inner + c.n_inner - (hashcode % c.n_inner)) % c.n_inner) + kd.emit_counter * c.n_inner;
            res.setControlFields(key, new_id, ts_of(res));
            kd.emit_counter++;
        }
        if (!kd.in_batch) {
            kd.in_batch = true;
            kd.batch_lo = w.start;
            batch_keys_.push_back(&kd);
        }
        kd.batch_hi = w.end;
        kd.purge_to = w.start + p_.slide_len;
        pending_.push_back({&kd, w.start, w.end, w.gwid, res});
        if (pending_.size() == cfg_g_.batch_len) launch_batch();
    }

    // wait for the batch in flight (if any) and emit its results in order
    void drain() {
        if (in_flight_ == 0) return;
        stream_->sync();
        for (size_t i = 0; i < in_flight_; i++) emit(tuple_new<result_t>(h_res_[i]));
        in_flight_ = 0;
    }

    void launch_batch() {
        drain();  // one batch in flight: its pinned buffers are reused below
        const size_t n = pending_.size();
        // copy, once per key, the span of tuples its scheduled windows cover
        size_t total = 0;
        for (KeyD* kd : batch_keys_) total += kd->archive.getDistance(kd->batch_lo, kd->batch_hi);
        if (h_tuples_.size() < total) h_tuples_.resize(std::max(total, 2 * h_tuples_.size()));
        if (d_tuples_.size() < total) d_tuples_.resize(std::max(total, 2 * d_tuples_.size()));
        size_t off = 0;
        std::unordered_map<KeyD*, std::pair<size_t, typename StreamArchive<tuple_t, OrdFn>::const_iterator>> base;
        for (KeyD* kd : batch_keys_) {
            auto r = kd->archive.getWinRange(kd->batch_lo, kd->batch_hi);
            base[kd] = {off, r.first};
            for (auto itx = r.first; itx != r.second; ++itx) h_tuples_[off++] = *itx;
        }
        for (size_t i = 0; i < n; i++) {
            Pending& pw = pending_[i];
            auto& b = base[pw.kd];
            auto r = pw.kd->archive.getWinRange(pw.start, pw.end);
            h_desc_[i].gwid = pw.gwid;
            h_desc_[i].off = (uint32_t)(b.first + (size_t)std::distance(b.second, r.first));
            h_desc_[i].len = (uint32_t)std::distance(r.first, r.second);
            h_res_[i] = pw.result;
        }
        hipStream_t s = stream_->get();
        if (total > 0) WF_HIP_CHECK(hipMemcpyAsync(d_tuples_.data(), h_tuples_.data(), total * sizeof(tuple_t), hipMemcpyHostToDevice, s));
        WF_HIP_CHECK(hipMemcpyAsync(d_desc_.data(), h_desc_.data(), n * sizeof(gpu::WinDesc), hipMemcpyHostToDevice, s));
        WF_HIP_CHECK(hipMemcpyAsync(d_res_.data(), h_res_.data(), n * sizeof(result_t), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL((gpu::win_batch_kernel<tuple_t, result_t, F>), dim3(gpu::grid_for(n, (unsigned)cfg_g_.n_thread_block)),
                           dim3((unsigned)cfg_g_.n_thread_block), 0, s, d_tuples_.data(), d_desc_.data(), d_res_.data(), (uint32_t)n,
                           d_scratch_.data(), cfg_g_.scratchpad_size, func_);
        WF_HIP_CHECK(hipMemcpyAsync(h_res_.data(), d_res_.data(), n * sizeof(result_t), hipMemcpyDeviceToHost, s));
        kernels_++;
        bytes_hd_ += total * sizeof(tuple_t) + n * (sizeof(gpu::WinDesc) + sizeof(result_t));
        bytes_dh_ += n * sizeof(result_t);
        in_flight_ = n;
        // the archives can now forget what no future window needs
        for (KeyD* kd : batch_keys_) {
            kd->archive.purge(kd->purge_to);
            kd->in_batch = false;
        }
        batch_keys_.clear();
        pending_.clear();
    }

    F func_;
    WinParams p_;
    GPUConfig cfg_g_;
    std::unordered_map<key_t, KeyD> keys_;
    std::vector<Pending> pending_;
    std::vector<KeyD*> batch_keys_;
    std::unique_ptr<gpu::Stream> stream_;
    gpu::PinnedBuffer<tuple_t> h_tuples_;
    gpu::PinnedBuffer<gpu::WinDesc> h_desc_;
    gpu::PinnedBuffer<result_t> h_res_;
    gpu::DeviceBuffer<tuple_t> d_tuples_;
    gpu::DeviceBuffer<gpu::WinDesc> d_desc_;
    gpu::DeviceBuffer<result_t> d_res_;
    gpu::DeviceBuffer<char> d_scratch_;
    unsigned grid_ = 1;
    size_t in_flight_ = 0;
    uint64_t dropped_ = 0, kernels_ = 0, bytes_hd_ = 0, bytes_dh_ = 0;
};

// engine maker for the composition classes (two_stage.hpp)
template <typename in_t, typename out_t, typename F>
inline win_engine_maker_t gpu_win_engine(F func, GPUConfig cfg, meta::closing_func_t closing, std::shared_ptr<StatsHub> hub) {
    return [func, cfg, closing, hub](rt::Graph& g, const WinParams& p, const std::string& name, size_t par, size_t idx) -> rt::Node* {
        return g.adopt(new Win_Seq_GPU_Node<in_t, out_t, F>(func, p, cfg, name, par, idx, closing, hub));
    };
}

// ---- operators -------------------------------------------------------------------
template <typename tuple_t, typename result_t>
class Win_Seq_GPU : public Key_Farm<tuple_t, result_t> {
public:
    template <typename F>
    Win_Seq_GPU(F func, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type, std::string name, GPUConfig cfg,
                meta::closing_func_t closing = nullptr)
        : Key_Farm<tuple_t, result_t>(typename Key_Farm<tuple_t, result_t>::engines_tag{}, nullptr, win_len, slide_len, triggering_delay, type, 1,
                                      std::move(name)) {
        this->set_engine(gpu_win_engine<tuple_t, result_t>(func, cfg, closing, this->hub_));
    }
    bool isGPUOperator() const override { return true; }
};

template <typename tuple_t, typename result_t>
class Key_Farm_GPU : public Key_Farm<tuple_t, result_t> {
public:
    template <typename F, typename = std::enable_if_t<!std::is_base_of_v<Basic_Operator, F>>>
    Key_Farm_GPU(F func, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type, size_t parallelism, std::string name,
                 GPUConfig cfg, meta::closing_func_t closing = nullptr, routing_func_t routing = default_routing)
        : Key_Farm<tuple_t, result_t>(typename Key_Farm<tuple_t, result_t>::engines_tag{}, nullptr, win_len, slide_len, triggering_delay, type,
                                      parallelism, std::move(name), std::move(routing)) {
        this->set_engine(gpu_win_engine<tuple_t, result_t>(func, cfg, closing, this->hub_));
    }
    // nesting: replicate a Pane_Farm_GPU / Win_MapReduce_GPU
    Key_Farm_GPU(TwoStage_Operator<tuple_t, result_t>& inner, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type,
                 size_t num_replicas, std::string name, meta::closing_func_t closing = nullptr, routing_func_t routing = default_routing,
                 opt_level_t opt_level = LEVEL2)
        : Key_Farm<tuple_t, result_t>(inner, win_len, slide_len, triggering_delay, type, num_replicas, std::move(name), closing,
                                      std::move(routing), opt_level) {}
    bool isGPUOperator() const override { return true; }
};

template <typename tuple_t, typename result_t>
class Win_Farm_GPU : public Win_Farm<tuple_t, result_t> {
public:
    template <typename F, typename = std::enable_if_t<!std::is_base_of_v<Basic_Operator, F>>>
    Win_Farm_GPU(F func, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type, size_t parallelism, std::string name,
                 GPUConfig cfg, meta::closing_func_t closing = nullptr, bool ordered = true)
        : Win_Farm<tuple_t, result_t>(typename Win_Farm<tuple_t, result_t>::engines_tag{}, nullptr, win_len, slide_len, triggering_delay, type,
                                      parallelism, std::move(name), ordered) {
        this->set_engine(gpu_win_engine<tuple_t, result_t>(func, cfg, closing, this->hub_));
    }
    Win_Farm_GPU(TwoStage_Operator<tuple_t, result_t>& inner, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type,
                 size_t num_replicas, std::string name, meta::closing_func_t closing = nullptr, bool ordered = true, opt_level_t opt_level = LEVEL2)
        : Win_Farm<tuple_t, result_t>(inner, win_len, slide_len, triggering_delay, type, num_replicas, std::move(name), closing, ordered,
                                      opt_level) {}
    bool isGPUOperator() const override { return true; }
};

// exactly one of the two stage functions is a device functor (6 arguments)
template <typename tuple_t, typename result_t>
class Pane_Farm_GPU : public Pane_Farm<tuple_t, result_t> {
public:
    template <typename plq_t, typename wlq_t>
    Pane_Farm_GPU(plq_t plq, wlq_t wlq, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type, size_t plq_parallelism,
                  size_t wlq_parallelism, std::string name, GPUConfig cfg, meta::closing_func_t closing = nullptr, bool ordered = true,
                  opt_level_t opt_level = LEVEL0)
        : Pane_Farm<tuple_t, result_t>(typename Pane_Farm<tuple_t, result_t>::engines_tag{}, nullptr, nullptr, win_len, slide_len,
                                       triggering_delay, type, plq_parallelism, wlq_parallelism, std::move(name), ordered, opt_level) {
        constexpr bool gpu_plq = meta::is_gpu_win_func<plq_t>();
        constexpr bool gpu_wlq = meta::is_gpu_win_func<wlq_t>();
        static_assert(gpu_plq != gpu_wlq, "WindFlow Compilation Error - Pane_Farm_GPU: exactly one of PLQ and WLQ must be a device functor");
        win_engine_maker_t m1, m2;
        if constexpr (gpu_plq) {
            m1 = gpu_win_engine<tuple_t, result_t>(plq, cfg, closing, this->hub_);
            m2 = cpu_win_engine<result_t, result_t>(wlq, closing, this->hub_);
        } else {
            m1 = cpu_win_engine<tuple_t, result_t>(plq, closing, this->hub_);
            m2 = gpu_win_engine<result_t, result_t>(wlq, cfg, closing, this->hub_);
        }
        this->set_engines(std::move(m1), std::move(m2));
    }
    bool isGPUOperator() const override { return true; }
};

template <typename tuple_t, typename result_t>
class Win_MapReduce_GPU : public Win_MapReduce<tuple_t, result_t> {
public:
    template <typename map_t, typename reduce_t>
    Win_MapReduce_GPU(map_t map, reduce_t reduce, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type,
                      size_t map_parallelism, size_t reduce_parallelism, std::string name, GPUConfig cfg, meta::closing_func_t closing = nullptr,
                      bool ordered = true, opt_level_t opt_level = LEVEL0)
        : Win_MapReduce<tuple_t, result_t>(typename Win_MapReduce<tuple_t, result_t>::engines_tag{}, nullptr, nullptr, win_len, slide_len,
                                           triggering_delay, type, map_parallelism, reduce_parallelism, std::move(name), ordered, opt_level) {
        constexpr bool gpu_map = meta::is_gpu_win_func<map_t>();
        constexpr bool gpu_red = meta::is_gpu_win_func<reduce_t>();
        static_assert(gpu_map != gpu_red, "WindFlow Compilation Error - Win_MapReduce_GPU: exactly one of MAP and REDUCE must be a device functor");
        win_engine_maker_t m1, m2;
        if constexpr (gpu_map) {
            m1 = gpu_win_engine<tuple_t, result_t>(map, cfg, closing, this->hub_);
            m2 = cpu_win_engine<result_t, result_t>(reduce, closing, this->hub_);
        } else {
            m1 = cpu_win_engine<tuple_t, result_t>(map, closing, this->hub_);
            m2 = gpu_win_engine<result_t, result_t>(reduce, cfg, closing, this->hub_);
        }
        this->set_engines(std::move(m1), std::move(m2));
    }
    bool isGPUOperator() const override { return true; }
};

}  // namespace wf
