// Batch transport between host operators and GPU operators:
//   Batch_Source   source that writes tuples straight into pinned slabs
//   Batch_Sink     sink that consumes whole host batches (+ latency histogram)
//   DeviceToHost   adapter node: device batch -> pinned host batch (async D2H)
//   Unbatcher      adapter node: host batch -> individual heap tuples
//   GPU_Replica    base class of the batched GPU replicas (Map_GPU, Filter_GPU...)
//
// The reference ships batches from its benchmark Sources with two pinned buffers
// and two streams (gpu_map_stateful.cpp:222-226, :330-349) and re-emits results
// one tuple at a time from the GPU node (wf/map_gpu_node.hpp:476-486).  Here the
// transport is part of the library: slabs come from bounded stream-ordered pools
// (hip_utils.hpp), a GPU replica never blocks on the device in steady state, and
// the device->host direction is a pipelined adapter that the MultiPipe inserts
// only when the consumer is a CPU operator.
#pragma once
#include <algorithm>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>

#include "../simple_operator.hpp"
#include "hip_utils.hpp"
#include "trace.hpp"

namespace wf {

struct GPUConfig {
    int gpu_id = 0;
    // placement per replica: replica i runs on gpu_ids[i % size]; empty = every replica on gpu_id.
    // A device batch produced on another GPU is pulled over with a peer copy (xGMI) by the consuming replica.
    std::vector<int> gpu_ids;
    int device_for(size_t replica) const { return gpu_ids.empty() ? gpu_id : gpu_ids[replica % gpu_ids.size()]; }
    size_t batch_len = 65536;       // tuples per batch when the operator builds batches itself
    size_t n_thread_block = DEFAULT_GPU_THREADS_PER_BLOCK;
    size_t pool_slabs = 6;          // slabs per pool = batches in flight (back-pressure)
    size_t scratchpad_size = 0;     // per-key (keyed map/filter) or per-thread (windows) bytes
    // tuple producers in front of the operator fill pinned batches of batch_len tuples (Batcher); a batch that has been open
    // for this long is shipped as it is (checked when a tuple arrives: a fused node has no timer of its own)
    uint64_t batch_linger_us = 5000;
};

// ---- latency accounting (BASELINE.json asks for p99; the reference only has a
// mean, src/yahoo_test_cpu/ysb_nodes.hpp:194-218) --------------------------------
// Fixed memory: a log-scale histogram (64 buckets per power of two: percentiles are exact to 1.1 %), so a pipeline
// that runs for days records every batch without growing.
class LatencyRecorder {
public:
    void add(uint64_t ns, uint64_t weight = 1) {
        std::lock_guard<std::mutex> l(m_);
        hist_[bucket_of(ns)] += weight;
        sum_ += (long double)ns * weight;
        weight_ += weight;
        samples_++;
    }
    // weighted percentile in microseconds (p in [0,100]): upper bound of the bucket that holds it
    double percentile_us(double p) const {
        std::lock_guard<std::mutex> l(m_);
        if (weight_ == 0) return 0.0;
        const double target = p / 100.0 * (double)weight_;
        uint64_t acc = 0;
        for (size_t b = 0; b < hist_.size(); b++) {
            acc += hist_[b];
            if (hist_[b] != 0 && (double)acc >= target) return (double)upper_of(b) / 1000.0;
        }
        return 0.0;
    }
    double mean_us() const {
        std::lock_guard<std::mutex> l(m_);
        return weight_ > 0 ? (double)(sum_ / (long double)weight_ / 1000.0) : 0.0;
    }
    size_t count() const {
        std::lock_guard<std::mutex> l(m_);
        return samples_;
    }
    void clear() {
        std::lock_guard<std::mutex> l(m_);
        std::fill(hist_.begin(), hist_.end(), 0);
        sum_ = 0;
        weight_ = 0;
        samples_ = 0;
    }

private:
    static constexpr int SUB_BITS = 6;  // 64 buckets per octave
    static constexpr int OCTAVES = 44;  // up to 2^44 ns = 4.9 hours
    static size_t bucket_of(uint64_t ns) {
        if (ns < (1ull << SUB_BITS)) return (size_t)ns;
        int msb = 63 - __builtin_clzll(ns);
        if (msb >= OCTAVES) {
            msb = OCTAVES - 1;
            ns = (1ull << OCTAVES) - 1;
        }
        const uint64_t sub = (ns >> (msb - SUB_BITS)) & ((1ull << SUB_BITS) - 1);
        return (size_t)(((uint64_t)(msb - SUB_BITS + 1) << SUB_BITS) + sub);
    }
    static uint64_t upper_of(size_t b) {
        if (b < (1ull << SUB_BITS)) return b;
        const uint64_t oct = (b >> SUB_BITS) - 1, sub = b & ((1ull << SUB_BITS) - 1);
        return (((1ull << SUB_BITS) | sub) + 1) << oct;
    }
    mutable std::mutex m_;
    std::vector<uint64_t> hist_ = std::vector<uint64_t>((size_t)(OCTAVES - SUB_BITS + 1) << SUB_BITS, 0);
    long double sum_ = 0;
    uint64_t weight_ = 0;
    size_t samples_ = 0;
};

// ---- adapters ---------------------------------------------------------------
template <typename T>
class Unbatcher : public rt::Node {
public:
    void svc(void* msg, int) override {
        gpu::Batch* b = static_cast<gpu::Batch*>(msg);
        const T* d = b->template data<T>();
        const size_t n = b->host_hdr.count;
        for (size_t i = 0; i < n; i++) emit(tuple_new<T>(d[i]));
        b->release_host();
    }
};

template <typename T>
class DeviceToHost : public rt::Node {
public:
    explicit DeviceToHost(size_t depth = 4) : depth_(depth) {}
    void svc(void* msg, int) override {
        gpu::Batch* d = static_cast<gpu::Batch*>(msg);
        if (!stream_) {
            WF_HIP_CHECK(hipSetDevice(d->device));
            if (gpu::d2h_shared_stream()) {
                stream_ = gpu::shared_copy_stream(d->device);
            } else {
                own_stream_.reset(new gpu::Stream());
                stream_ = own_stream_->get();
            }
            pool_.reset(new gpu::BatchPool(gpu::MemKind::PINNED_HOST, d->device, sizeof(T), d->capacity, depth_ + 2));
        }
        fifo_.push_back({d, nullptr});
        pump(false);
        while (fifo_.size() > depth_) pump(true);
    }
    void flush() override {
        while (!fifo_.empty()) pump(true);
        if (stream_) WF_HIP_CHECK(hipStreamSynchronize(stream_));
    }
    bool wants_idle() const override { return true; }
    void on_idle() override {
        if (!fifo_.empty()) pump(false);
    }

private:
    struct InFlight {
        gpu::Batch* dev;
        gpu::Batch* host;  // nullptr until the copy has been enqueued
    };
    // advance the head of the FIFO; `block` waits for the head's current step
    void pump(bool block) {
        while (!fifo_.empty()) {
            InFlight& f = fifo_.front();
            if (f.host == nullptr) {
                // step 1: wait on the HOST for the producer (the count must be known before sizing the copy, and a
                // device-side wait would park a barrier packet in this stream's hardware queue, which it shares
                // with other lanes' compute streams: their kernels would queue up behind this lane's)
                if (block)
                    f.dev->ready.host_sync();
                else if (!f.dev->ready.done())
                    return;
                if (!f.dev->count_on_host) {
                    f.dev->host_hdr.count = *f.dev->count_mirror;
                    f.dev->count_on_host = true;
                }
                gpu::TraceRange tr("wf:drain_batch");
                if (f.dev->host_hdr.count > pool_->capacity()) gpu::failwith("DeviceToHost: a batch larger than the first one received");
                gpu::Batch* h = pool_->acquire_host();
                h->host_hdr = f.dev->host_hdr;
                const size_t bytes = (size_t)h->host_hdr.count * sizeof(T);
                if (bytes > 0) WF_HIP_CHECK(hipMemcpyAsync(h->template data<T>(), f.dev->template data<T>(), bytes, hipMemcpyDeviceToHost, stream_));
                h->ready.record(stream_);
                f.dev->release(stream_);
                reaper_.tick(stream_);
                f.host = h;
            }
            // step 2: copy completion
            if (block)
                f.host->ready.host_sync();
            else if (!f.host->ready.done())
                return;
            emit(f.host);
            fifo_.pop_front();
            block = false;
        }
    }
    size_t depth_;
    hipStream_t stream_ = nullptr;  // own_stream_, or the device's shared copy stream (WF_D2H_STREAMS=shared)
    std::unique_ptr<gpu::Stream> own_stream_;
    gpu::StreamReaper reaper_;
    std::unique_ptr<gpu::BatchPool> pool_;
    std::deque<InFlight> fifo_;
};

// Batcher: adapter node in the other direction, tuples -> pinned host batches.  It is fused onto a CPU producer (Source, Map,
// Filter, FlatMap ... replica) whose consumer is a GPU operator: the producer's tuples are written straight into the slab
// that the GPU operator will upload - by value when the producer supports it (rt::Node::emit_value: no heap tuple at all),
// otherwise the pooled tuple is copied and freed in the thread that created it.  What leaves the producer's thread is one
// message per batch instead of one per tuple, and the device KeyBy / deterministic merge of the GPU operator replaces the
// per-tuple emitters and ordering nodes of the tuple world.  (Reference: every GPU operator re-stages the tuples it receives
// one by one, wf/win_seq_gpu.hpp:352-539, wf/map_gpu_node.hpp:249-291.)
template <typename T>
class Batcher : public rt::Node {
    static_assert(std::is_trivially_copyable_v<T>, "tuples processed on the GPU travel in raw batches: they must be trivially copyable");

public:
    Batcher(size_t batch_len, size_t slabs, int device, uint64_t linger_us, size_t producer)
        : batch_len_(std::max<size_t>(batch_len, 1)), slabs_(std::max<size_t>(slabs, 2)), device_(device), linger_ns_(linger_us * 1000),
          producer_(producer) {
        gpu::preload_module(device_);
    }
    void svc_init() override {
        WF_HIP_CHECK(hipSetDevice(device_));
        pool_.reset(new gpu::BatchPool(gpu::MemKind::PINNED_HOST, device_, sizeof(T), batch_len_, slabs_));
    }
    bool accepts_values() const override { return true; }
    void svc_value(const void* v, int) override { append(*static_cast<const T*>(v)); }
    void svc(void* msg, int) override {
        T* t = static_cast<T*>(msg);
        append(*t);
        tuple_delete(t);
    }
    void flush() override { ship(); }

private:
    inline void append(const T& t) {
        if (cur_ == nullptr) open();
        data_[n_++] = t;
        const uint64_t ts = ts_of(t);
        if (ts > wm_) wm_ = ts;
        if (n_ == batch_len_)
            ship();
        else if (linger_ns_ != 0 && (n_ & 63) == 0 && current_time_nsecs() - cur_->host_hdr.created_ns > linger_ns_)
            ship();
    }
    void open() {
        cur_ = pool_->acquire_host();
        cur_->host_hdr.created_ns = current_time_nsecs();
        data_ = cur_->template data<T>();
        n_ = 0;
    }
    void ship() {
        if (cur_ == nullptr) return;
        if (n_ == 0) {
            cur_->release_host();
        } else {
            cur_->host_hdr.count = (uint32_t)n_;
            cur_->host_hdr.source_id = (uint32_t)producer_;
            cur_->host_hdr.first_seq = seq_;
            cur_->host_hdr.watermark = wm_;
            seq_ += n_;
            emit(cur_);
        }
        cur_ = nullptr;
        n_ = 0;
    }
    size_t batch_len_, slabs_;
    int device_;
    uint64_t linger_ns_;
    size_t producer_;
    std::unique_ptr<gpu::BatchPool> pool_;
    gpu::Batch* cur_ = nullptr;
    T* data_ = nullptr;
    size_t n_ = 0;
    uint64_t wm_ = 0, seq_ = 0;
};

// WF_TUPLE_PATH=1: GPU operators behind CPU producers are fed tuple by tuple, as in the reference (A/B switch)
inline bool upward_batching_enabled() {
    static const bool on = [] {
        const char* e = std::getenv("WF_TUPLE_PATH");
        return !(e && std::atoi(e) != 0);
    }();
    return on;
}
template <typename T>
inline std::function<rt::Node*(size_t, size_t)> batcher_factory(const GPUConfig& cfg) {
    if (!upward_batching_enabled()) return nullptr;
    const size_t len = cfg.batch_len, slabs = cfg.pool_slabs;
    const uint64_t linger = cfg.batch_linger_us;
    const GPUConfig c = cfg;
    return [len, slabs, linger, c](size_t i, size_t) -> rt::Node* { return new Batcher<T>(len, slabs, c.device_for(0), linger, i); };
}

// `producer_slabs` = size of the pool the device batches come from: the adapter
// never holds more than producer_slabs - 1 of them, so the producer (which runs in
// the same thread when fused) can always acquire one more
template <typename T>
inline std::function<rt::Node*(edge_kind_t)> batch_adapter_factory(size_t producer_slabs = 6) {
    const size_t depth = producer_slabs >= 3 ? std::min<size_t>(4, producer_slabs - 2) : 0;
    return [depth](edge_kind_t from) -> rt::Node* {
        if (from == EDGE_DEVICE_BATCH) return new DeviceToHost<T>(depth);
        return new Unbatcher<T>();
    };
}

// ---- Batch_Source -----------------------------------------------------------
// Functor forms:  bool(tuple_t&)                          one tuple per call, written in place
//                 bool(tuple_t* buf, size_t cap, size_t& n) fill up to cap tuples, set n
// (+ RuntimeContext& as last argument).  false = end of stream.
template <typename tuple_t, typename F>
class Batch_Source_Node : public Replica_Base {
    static_assert(std::is_trivially_copyable_v<tuple_t>, "batched tuples are shipped as raw bytes: they must be trivially copyable");

public:
    Batch_Source_Node(F func, size_t batch_len, size_t slabs, int device, std::string name, size_t par, size_t idx,
                      meta::closing_func_t closing, std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub)), func_(std::move(func)), batch_len_(batch_len),
          slabs_(slabs), device_(device) {
        set_io_bytes(0, sizeof(tuple_t) * batch_len);
        gpu::preload_module(device_);
    }
    bool is_source() const override { return true; }
    void svc_init() override {
        Replica_Base::svc_init();
        WF_HIP_CHECK(hipSetDevice(device_));
        pool_.reset(new gpu::BatchPool(gpu::MemKind::PINNED_HOST, device_, sizeof(tuple_t), batch_len_, slabs_));
    }
    bool generate() override {
        gpu::Batch* b = pool_->acquire_host();
        b->host_hdr.created_ns = current_time_nsecs();
        b->host_hdr.source_id = (uint32_t)context.getReplicaIndex();
        b->host_hdr.first_seq = seq_;
        tuple_t* buf = b->template data<tuple_t>();
        size_t n = 0;
        bool more = true;
        if constexpr (meta::base_arity_v<F> == 3) {
            more = meta::invoke_rich(func_, context, buf, batch_len_, n);
        } else {
            while (n < batch_len_ && more) {
                new (&buf[n]) tuple_t();
                more = meta::invoke_rich(func_, context, buf[n]);
                n++;
            }
        }
        b->host_hdr.count = (uint32_t)n;
        if (n > 0) b->host_hdr.watermark = ts_of(buf[n - 1]);
        seq_ += n;
        if constexpr (has_punctuation<F>::value) b->host_hdr.reserved[0] = func_.punctuation();
        if constexpr (has_opened_ns<F>::value) b->host_hdr.created_ns = func_.opened_ns();
        if (n > 0 || b->host_hdr.reserved[0] != 0)
            emit(b);
        else
            b->release_host();
        return more;
    }

private:
    // optional hook: uint64_t F::punctuation() - a non-zero value marks the batch as
    // the last one of a driver-defined step (header.reserved[0]); every batch
    // operator forwards it, so a sink can tell when a step has fully drained
    template <typename G, typename = void>
    struct has_punctuation : std::false_type {};
    template <typename G>
    struct has_punctuation<G, std::void_t<decltype(std::declval<G&>().punctuation())>> : std::true_type {};
    // optional hook: uint64_t F::opened_ns() - host clock when the first tuple of the
    // batch was produced (default: when generate() was entered)
    template <typename G, typename = void>
    struct has_opened_ns : std::false_type {};
    template <typename G>
    struct has_opened_ns<G, std::void_t<decltype(std::declval<G&>().opened_ns())>> : std::true_type {};
    F func_;
    size_t batch_len_, slabs_;
    int device_;
    uint64_t seq_ = 0;
    std::unique_ptr<gpu::BatchPool> pool_;
};

template <typename tuple_t>
class Batch_Source : public Simple_Operator<void, tuple_t> {
public:
    template <typename F>
    Batch_Source(F func, size_t parallelism, std::string name, size_t batch_len, size_t slabs = 6, int device = 0,
                 meta::closing_func_t closing = nullptr)
        : Simple_Operator<void, tuple_t>(name, parallelism, NONE, nullptr,
                                         [=](size_t i, size_t n, std::shared_ptr<StatsHub> hub) -> rt::Node* {
                                             return new Batch_Source_Node<tuple_t, F>(func, batch_len, slabs, device, name, n, i, closing,
                                                                                     std::move(hub));
                                         }) {}
    edge_kind_t output_kind() const override { return EDGE_HOST_BATCH; }
    std::function<rt::Node*(edge_kind_t)> output_adapter_factory() const override { return batch_adapter_factory<tuple_t>(); }
};

// ---- Device_Source ------------------------------------------------------------
// A source whose tuples are born in device memory (synthetic load, replay of device-resident data, a NIC that
// writes into HBM): no host->device copy, the stream enters the pipeline as device batches.  The generator is a
// pure function of the replica and of the replica-local sequence number,
//     __host__ __device__ void(uint64_t seq, uint32_t replica, tuple_t& t)
// evaluated by one thread per tuple; the host decides when to stop: `more(seq)` is asked before every batch.
// The reference has no counterpart (its GPU operators are always fed from host memory).
namespace gpu {
template <typename T, typename G>
__global__ void __launch_bounds__(BLOCK_THREADS)
    device_source_kernel(BatchHeader* hdr, T* data, uint64_t first_seq, uint32_t n, uint32_t replica, BatchHeader host_view, G gen) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        T t{};
        gen(first_seq + i, replica, t);
        data[i] = t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *hdr = host_view;
}
}  // namespace gpu

// What the host decides for a device source, batch by batch: how many tuples the next batch holds (0 < n <= cap), whether it
// closes a driver-defined step (punctuation, see Batch_Source) - or that the stream is over (return false).  May block
// (a benchmark driver releasing a quota per step).
using device_source_control_t = std::function<bool(uint64_t seq, uint32_t replica, size_t cap, size_t& n, uint64_t& punctuation)>;

template <typename tuple_t, typename G>
class Device_Source_Node : public Replica_Base {
    static_assert(std::is_trivially_copyable_v<tuple_t>, "batched tuples are raw bytes: they must be trivially copyable");

public:
    Device_Source_Node(G gen, device_source_control_t control, size_t batch_len, size_t slabs, int device, std::string name, size_t par,
                       size_t idx, meta::closing_func_t closing, std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub), false, true), gen_(gen), control_(std::move(control)),
          batch_len_(batch_len), slabs_(slabs), device_(device) {
        set_io_bytes(0, sizeof(tuple_t) * batch_len);
        gpu::preload_module(device_);
    }
    bool is_source() const override { return true; }
    void svc_init() override {
        Replica_Base::svc_init();
        WF_HIP_CHECK(hipSetDevice(device_));
        stream_.reset(new gpu::Stream());
        pool_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, device_, sizeof(tuple_t), batch_len_, slabs_));
    }
    bool generate() override {
        const uint32_t replica = (uint32_t)context.getReplicaIndex();
        size_t n = 0;
        uint64_t punct = 0;
        if (!control_(seq_, replica, batch_len_, n, punct)) return false;
        if (n > batch_len_) n = batch_len_;
        if (n == 0 && punct == 0) return true;
        gpu::Batch* d = pool_->acquire(stream_->get());
        d->host_hdr.count = (uint32_t)n;
        d->host_hdr.source_id = replica;
        d->host_hdr.first_seq = seq_;
        d->host_hdr.created_ns = current_time_nsecs();
        d->host_hdr.reserved[0] = punct;
        if (n > 0) {
            tuple_t last{};
            gen_(seq_ + n - 1, replica, last);  // the generator runs on the host too: the watermark is known here
            d->host_hdr.watermark = ts_of(last);
        } else {
            d->host_hdr.watermark = last_wm_;
        }
        last_wm_ = d->host_hdr.watermark;
        d->count_on_host = true;
        hipLaunchKernelGGL((gpu::device_source_kernel<tuple_t, G>), dim3(gpu::grid_for(std::max<size_t>(n, 1))), dim3(gpu::BLOCK_THREADS), 0,
                           stream_->get(), d->header(), d->template data<tuple_t>(), seq_, (uint32_t)n, replica, d->host_hdr, gen_);
        WF_HIP_CHECK(hipGetLastError());
        d->ready.record(stream_->get());
        reaper_.tick(stream_->get());
        kernels_++;
        seq_ += n;
        emit(d);
        return true;
    }
    void svc_end() override {
        if (stream_) stream_->sync();
        Replica_Base::svc_end();
    }

protected:
    void finalize_stats(Stats_Record& s) override { s.num_kernels = kernels_; }

private:
    G gen_;
    device_source_control_t control_;
    size_t batch_len_, slabs_;
    int device_;
    uint64_t seq_ = 0, kernels_ = 0, last_wm_ = 0;
    std::unique_ptr<gpu::Stream> stream_;
    std::unique_ptr<gpu::BatchPool> pool_;
    gpu::StreamReaper reaper_;
};

template <typename tuple_t>
class Device_Source : public Simple_Operator<void, tuple_t> {
public:
    // `more(seq)`: asked before every (full) batch with the number of tuples the replica has generated so far
    template <typename G>
    Device_Source(G gen, std::function<bool(uint64_t)> more, size_t parallelism, std::string name, size_t batch_len, size_t slabs = 6,
                  GPUConfig cfg = GPUConfig(), meta::closing_func_t closing = nullptr)
        : Device_Source(gen, control_of(std::move(more)), parallelism, std::move(name), batch_len, slabs, cfg, std::move(closing)) {}
    // the general form: the host sizes every batch and may punctuate it (device_source_control_t)
    template <typename G>
    Device_Source(G gen, device_source_control_t control, size_t parallelism, std::string name, size_t batch_len, size_t slabs = 6,
                  GPUConfig cfg = GPUConfig(), meta::closing_func_t closing = nullptr)
        : Simple_Operator<void, tuple_t>(name, parallelism, NONE, nullptr,
                                         [=](size_t i, size_t n, std::shared_ptr<StatsHub> hub) -> rt::Node* {
                                             return new Device_Source_Node<tuple_t, G>(gen, control, batch_len, slabs, cfg.device_for(i), name, n, i,
                                                                                      closing, std::move(hub));
                                         }) {}
    edge_kind_t output_kind() const override { return EDGE_DEVICE_BATCH; }
    std::function<rt::Node*(edge_kind_t)> output_adapter_factory() const override { return batch_adapter_factory<tuple_t>(); }

private:
    static device_source_control_t control_of(std::function<bool(uint64_t)> more) {
        return [more](uint64_t seq, uint32_t, size_t cap, size_t& n, uint64_t& punct) {
            punct = 0;
            n = cap;
            return more(seq);
        };
    }
};

// ---- Batch_Sink --------------------------------------------------------------
// Functor: void(const tuple_t* data, size_t n) or
//          void(const tuple_t* data, size_t n, const gpu::BatchHeader&)  [+ RuntimeContext&];
// called with (nullptr, 0) at end of stream.
template <typename tuple_t, typename F>
class Batch_Sink_Node : public Replica_Base {
public:
    Batch_Sink_Node(F func, std::shared_ptr<LatencyRecorder> lat, std::string name, size_t par, size_t idx, meta::closing_func_t closing,
                    std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub)), func_(std::move(func)), lat_(std::move(lat)) {}
    void svc(void* msg, int) override {
        gpu::Batch* b = static_cast<gpu::Batch*>(msg);
        if constexpr (meta::base_arity_v<F> == 3)
            meta::invoke_rich(func_, context, static_cast<const tuple_t*>(b->template data<tuple_t>()), (size_t)b->host_hdr.count,
                              static_cast<const gpu::BatchHeader&>(b->host_hdr));
        else
            meta::invoke_rich(func_, context, static_cast<const tuple_t*>(b->template data<tuple_t>()), (size_t)b->host_hdr.count);
        if (lat_ && b->host_hdr.created_ns != 0) lat_->add(current_time_nsecs() - b->host_hdr.created_ns, b->host_hdr.count ? b->host_hdr.count : 1);
        b->release_host();
    }
    void flush() override {
        if constexpr (meta::base_arity_v<F> == 3) {
            gpu::BatchHeader h{};
            meta::invoke_rich(func_, context, static_cast<const tuple_t*>(nullptr), (size_t)0, static_cast<const gpu::BatchHeader&>(h));
        } else {
            meta::invoke_rich(func_, context, static_cast<const tuple_t*>(nullptr), (size_t)0);
        }
    }

private:
    F func_;
    std::shared_ptr<LatencyRecorder> lat_;
};

template <typename tuple_t>
class Batch_Sink : public Simple_Operator<tuple_t, void> {
public:
    template <typename F>
    Batch_Sink(F func, size_t parallelism, std::string name, meta::closing_func_t closing = nullptr)
        : Simple_Operator<tuple_t, void>(name, parallelism, FORWARD, nullptr, nullptr), lat_(std::make_shared<LatencyRecorder>()) {
        auto lat = lat_;
        this->factory_ = [func, lat, name, closing](size_t i, size_t n, std::shared_ptr<StatsHub> hub) -> rt::Node* {
            return new Batch_Sink_Node<tuple_t, F>(func, lat, name, n, i, closing, std::move(hub));
        };
    }
    bool accepts_input_kind(edge_kind_t k) const override { return k == EDGE_HOST_BATCH; }
    std::shared_ptr<LatencyRecorder> latency() const { return lat_; }

private:
    std::shared_ptr<LatencyRecorder> lat_;
};

// ---- base of the batched GPU replicas ----------------------------------------
template <typename in_t>
class GPU_Replica : public Replica_Base {
    static_assert(std::is_trivially_copyable_v<in_t>, "tuples processed on the GPU travel in raw batches: they must be trivially copyable");

public:
    GPU_Replica(GPUConfig cfg, edge_kind_t in_kind, std::string name, size_t par, size_t idx, meta::closing_func_t closing,
                std::shared_ptr<StatsHub> hub, bool is_win = false)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub), is_win, true), cfg_(cfg), in_kind_(in_kind) {
        cfg_.gpu_id = cfg.device_for(idx);
        gpu::preload_module(cfg_.gpu_id);
    }

    void svc_init() override {
        Replica_Base::svc_init();
        WF_HIP_CHECK(hipSetDevice(cfg_.gpu_id));
        stream_.reset(new gpu::Stream());
        gpu_init();
    }

    void svc(void* msg, int) override {
        switch (in_kind_) {
            case EDGE_TUPLE: {
                in_t* t = static_cast<in_t*>(msg);
                if (staging_ == nullptr) open_staging();
                staging_->template data<in_t>()[staging_->host_hdr.count++] = *t;
                staging_->host_hdr.watermark = std::max<uint64_t>(staging_->host_hdr.watermark, ts_of(*t));
                tuple_delete(t);
                if (staging_->host_hdr.count == cfg_.batch_len) {
                    ship(staging_);
                    staging_ = nullptr;
                }
                break;
            }
            case EDGE_HOST_BATCH:
                ship(static_cast<gpu::Batch*>(msg));
                break;
            case EDGE_DEVICE_BATCH: {
                gpu::Batch* d = static_cast<gpu::Batch*>(msg);
                if (d->device != cfg_.gpu_id) d = pull_from_peer(d);
                // Fused onto its only producer, the replica continues on the PRODUCER'S stream: the two never run
                // concurrently anyway, stream order replaces the event wait between them, and the chain of a lane
                // occupies one stream instead of one per stage (streams share a few hardware queues: fewer of them, and
                // no barrier packets between them, is what keeps the queues moving - profiles/r1_sync_policy).
                if (inherit_stream_ && this->fed_in_thread_by_one() && d->device == cfg_.gpu_id && d->ready.recorded_on() != nullptr &&
                    d->ready.recorded_on() != cur_stream_) {
                    // what this replica enqueued so far (table set-up in gpu_init, earlier batches) stays ahead of what follows
                    if (!handoff_) handoff_.reset(new gpu::Event());
                    handoff_->record(stream());
                    cur_stream_ = d->ready.recorded_on();
                    handoff_->wait_on(cur_stream_);
                }
                d->ready.edge_wait(stream());
                reaper_.tick(stream());
                gpu::TraceRange tr("wf:enqueue_kernels");
                process_device(d);
                WF_HIP_CHECK(hipGetLastError());  // a refused launch (bad grid, too much LDS) must not pass silently
                break;
            }
        }
    }

    // Several host batches queued on one channel: the replica has fallen behind its producer.  Small ones are uploaded into
    // ONE device batch (in arrival order) and go through the kernels together - a batch costs ~10 launches whatever its size
    // (flagship at 16 Ki batches: the replica thread was 97 % busy issuing them, profiles/r3_small_batch).  A batch that
    // carries a token in its header (a marker batch, the benchmark's step barrier) ends its group: its header fields go to
    // the merged batch, nothing is reordered.
    void svc_bulk(void** msgs, size_t n, int ch) override {
        if (in_kind_ != EDGE_HOST_BATCH || n < 2 || gpu::coalesce_tuples() == 0) {
            for (size_t i = 0; i < n; i++) svc(msgs[i], ch);
            return;
        }
        size_t i = 0;
        while (i < n) {
            gpu::Batch* first = static_cast<gpu::Batch*>(msgs[i]);
            const size_t room = group_room(first);
            size_t k = 1, tuples = first->host_hdr.count;
            while (i + k < n && !has_token(static_cast<gpu::Batch*>(msgs[i + k - 1]))) {
                gpu::Batch* next = static_cast<gpu::Batch*>(msgs[i + k]);
                if (tuples + next->host_hdr.count > room || next->capacity != first->capacity) break;
                tuples += next->host_hdr.count;
                k++;
            }
            if (k == 1)
                ship(first);
            else
                ship(reinterpret_cast<gpu::Batch**>(msgs + i), k);
            i += k;
        }
    }

    // end of stream: the partial batch goes through the same kernels (the
    // reference falls back to the CPU here, wf/map_gpu_node.hpp:502-604)
    void flush() override {
        if (staging_ != nullptr) {
            if (staging_->host_hdr.count > 0)
                ship(staging_);
            else
                staging_->release_host();
            staging_ = nullptr;
        }
        while (!uploading_.empty()) pump_uploads(true);
        gpu_flush();
        foreign_.drain();
    }

    // uploads complete while the input queue is empty: hand them to the kernels without waiting for the next batch
    bool wants_idle() const override { return in_kind_ != EDGE_DEVICE_BATCH; }
    void on_idle() override {
        if (!uploading_.empty()) pump_uploads(false);
    }

    void svc_end() override {
        if (cur_stream_) (void)hipStreamSynchronize(cur_stream_);
        if (stream_) stream_->sync();
        Replica_Base::svc_end();
    }

protected:
    virtual void gpu_init() {}
    virtual void gpu_flush() {}
    // enqueue the operator's kernels for device batch `d` on stream(); takes ownership of d
    virtual void process_device(gpu::Batch* d) = 0;

    void finalize_stats(Stats_Record& s) override {
        s.num_kernels = num_kernels_;
        s.bytes_copied_hd = bytes_hd_;
        s.bytes_copied_dh = bytes_dh_;
    }
    hipStream_t stream() const { return cur_stream_ ? cur_stream_ : stream_->get(); }
    // how many of its producers' host batches one device input batch may hold (see svc_bulk): what an operator sizes per
    // input batch on the user's word (an output capacity) grows by this factor
    size_t input_scale() const { return input_scale_; }
    // A user's output capacity is meant per batch of the operator's configured length (withBatch): input batches that hold
    // several of those - small host batches coalesced here or upstream - get that many times the room, up to twice the
    // batch (the automatic size).  0 stays 0 (automatic).
    size_t scaled_out_capacity(size_t user, size_t device_capacity) const {
        if (user == 0) return 0;
        const size_t unit = std::max<size_t>(cfg_.batch_len, 1);
        const size_t scale = std::max<size_t>(input_scale_, (device_capacity + unit - 1) / unit);
        return scale <= 1 ? user : std::min<size_t>(std::max(2 * device_capacity, user), user * scale);
    }
    // hand an input slab back after the work enqueued so far (whichever GPU's context owns the slab)
    void release_input(gpu::Batch* b) { gpu::release_after(b, stream(), cfg_.gpu_id, foreign_); }
    // record completion on the batch and pass it downstream
    void emit_device(gpu::Batch* d) {
        d->ready.record(stream());
        emit(d);
    }

    GPUConfig cfg_;
    edge_kind_t in_kind_;
    uint64_t num_kernels_ = 0, bytes_hd_ = 0, bytes_dh_ = 0;

private:
    void open_staging() {
        if (!staging_pool_)
            staging_pool_.reset(new gpu::BatchPool(gpu::MemKind::PINNED_HOST, cfg_.gpu_id, sizeof(in_t), cfg_.batch_len, cfg_.pool_slabs));
        staging_ = staging_pool_->acquire_host();
        staging_->host_hdr.created_ns = current_time_nsecs();
    }
    // a device batch that lives on another GPU: copy it into a slab of this replica's device (peer copy over xGMI)
    gpu::Batch* pull_from_peer(gpu::Batch* d) {
        gpu::TraceRange tr("wf:peer_copy");
        if (!d->count_on_host) {  // the copy is sized on the host
            d->ready.host_sync();
            d->host_hdr.count = *d->count_mirror;
            d->count_on_host = true;
        }
        if (!dev_in_pool_)
            dev_in_pool_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, cfg_.gpu_id, sizeof(in_t), d->capacity, cfg_.pool_slabs));
        if (d->capacity > dev_in_pool_->capacity()) gpu::failwith("GPU operator: a batch larger than the first one received (producers with different batch lengths?)");
        gpu::Batch* mine = dev_in_pool_->acquire(stream());
        // the producer's event belongs to the other GPU: the copy must not start before it, and a cross-device
        // hipStreamWaitEvent is legal, but the wait happens on the host like every other inter-stream wait here
        d->ready.host_sync();
        mine->host_hdr = d->host_hdr;
        mine->count_on_host = true;
        WF_HIP_CHECK(hipMemcpyAsync(mine->slab, &mine->host_hdr, sizeof(gpu::BatchHeader), hipMemcpyHostToDevice, stream()));
        const size_t bytes = (size_t)d->host_hdr.count * sizeof(in_t);
        if (bytes > 0)
            WF_HIP_CHECK(hipMemcpyPeerAsync(mine->template data<in_t>(), cfg_.gpu_id, d->template data<in_t>(), d->device, bytes, stream()));
        foreign_.release(d, stream());  // d's own event cannot be recorded on a stream of this GPU
        mine->ready.record(stream());
        return mine;
    }
    // host batch -> device batch (async), then the operator's kernels: on the replica's compute stream, or with
    // WF_UPLOAD_STREAMS=n on a shared upload stream, the kernels following once the copy is known to be complete
    // tuples a device input slab holds: the producer's batch capacity, or a multiple of it up to WF_COALESCE_TUPLES when the
    // replica has an input ring to find several batches in (see svc_bulk)
    // (a replica fed by several producers has that many times the tuples to get through - one replica aggregating a single
    // key that 8 sources emit: 0.96 -> 1.20 G tuples/s at the same latency - so its limit is up to four times the knob's)
    size_t coalesce_capacity(size_t host_capacity) const {
        const size_t lim = gpu::coalesce_tuples() * std::min<size_t>(4, std::max<size_t>(1, this->num_in()));
        if (!this->has_input_rings() || host_capacity == 0 || lim < 2 * host_capacity) return host_capacity;
        return (lim / host_capacity) * host_capacity;
    }
    // what one group of svc_bulk may hold: the device slab, and with a single producer no more than its share of the slabs
    // that producer keeps in flight (gpu::coalesce_share) - the producer needs the others to go on filling
    size_t group_room(const gpu::Batch* first) const {
        size_t room = coalesce_capacity(first->capacity);
        if (this->num_in() <= 1 && first->pool != nullptr && gpu::coalesce_share() > 1) {
            const size_t share = (first->pool->max_slabs() + gpu::coalesce_share() - 1) / gpu::coalesce_share() * first->capacity;
            room = std::min(room, std::max(share, gpu::COALESCE_SMALL_GROUP));
        }
        return room;
    }
    static bool has_token(const gpu::Batch* b) {
        const uint64_t* r = b->host_hdr.reserved;
        return (r[0] | r[1] | r[2] | r[3]) != 0;
    }
    void ship(gpu::Batch* h) { ship(&h, 1); }
    void ship(gpu::Batch** hs, size_t k) {
        gpu::TraceRange tr("wf:ship_batch");
        gpu::Batch* h = hs[0];
        if (!dev_in_pool_) {
            dev_in_pool_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, cfg_.gpu_id, sizeof(in_t), coalesce_capacity(h->capacity), cfg_.pool_slabs));
            input_scale_ = std::max<size_t>(1, coalesce_capacity(h->capacity) / std::max<size_t>(1, h->capacity));
            // (a replica fused onto its producer is never polled between inputs: its uploads stay in its compute stream)
            const size_t ups = this->has_input_rings() ? gpu::upload_streams(this->context.getParallelism()) : 0;
            upload_stream_ = ups == 0 ? stream() : gpu::shared_copy_stream(cfg_.gpu_id, 1 + (int)(this->context.getReplicaIndex() % ups));
        }
        gpu::BatchHeader merged = h->host_hdr;
        for (size_t i = 1; i < k; i++) {
            const gpu::BatchHeader& o = hs[i]->host_hdr;
            merged.count += o.count;
            merged.watermark = std::max(merged.watermark, o.watermark);
            if (o.created_ns != 0 && (merged.created_ns == 0 || o.created_ns < merged.created_ns)) merged.created_ns = o.created_ns;
            for (int r = 0; r < 4; r++)
                if (o.reserved[r] != 0) merged.reserved[r] = o.reserved[r];  // (only the last batch of a group carries tokens)
        }
        if (merged.count > dev_in_pool_->capacity()) gpu::failwith("GPU operator: a batch larger than the first one received (producers with different batch lengths?)");
        gpu::Batch* d = dev_in_pool_->acquire(upload_stream_);
        *h->header() = merged;  // the header travels in the slab
        size_t off = 0;
        for (size_t i = 0; i < k; i++) {
            const size_t data_bytes = (size_t)hs[i]->host_hdr.count * sizeof(in_t);
            if (i == 0) {
                WF_HIP_CHECK(hipMemcpyAsync(d->slab, h->slab, sizeof(gpu::BatchHeader) + data_bytes, hipMemcpyHostToDevice, upload_stream_));
                bytes_hd_ += sizeof(gpu::BatchHeader);
            } else if (data_bytes > 0) {
                WF_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char*>(d->template data<in_t>()) + off, hs[i]->template data<in_t>(), data_bytes,
                                            hipMemcpyHostToDevice, upload_stream_));
            }
            off += data_bytes;
            bytes_hd_ += data_bytes;
        }
        d->host_hdr = merged;
        d->count_on_host = true;
        for (size_t i = 0; i < k; i++) gpu::release_after(hs[i], upload_stream_, cfg_.gpu_id, foreign_);  // the pinned slab may belong to another GPU's context
        reaper_.tick(stream());
        // (the copy stream of a lone replica needs the same care: nobody else ever synchronises it, and HIP's host-side
        // bookkeeping of its copies and event records grew by ~12 MB/s - apps/bench/r3_check33.sh)
        if (upload_stream_ != stream()) upload_reaper_.tick(upload_stream_);
        if (upload_stream_ == stream()) {  // uploads in the compute stream: stream order is all the kernels need
            process_device(d);
            WF_HIP_CHECK(hipGetLastError());
            return;
        }
        d->ready.record(upload_stream_);
        uploading_.push_back(d);
        pump_uploads(false);
        // double buffering: at most one upload in flight behind the batch being handed to the kernels
        while (uploading_.size() > 1) pump_uploads(true);
    }
    void pump_uploads(bool block) {
        while (!uploading_.empty()) {
            gpu::Batch* d = uploading_.front();
            if (block)
                d->ready.host_sync();
            else if (!d->ready.done())
                return;
            uploading_.pop_front();
            gpu::TraceRange tr("wf:enqueue_kernels");
            process_device(d);
            WF_HIP_CHECK(hipGetLastError());  // a refused launch (bad grid, too much LDS) must not pass silently
            block = false;
        }
    }

    size_t input_scale_ = 1;
    hipStream_t upload_stream_ = nullptr;
    hipStream_t cur_stream_ = nullptr;  // the producer's stream, once inherited (see svc)
    bool inherit_stream_ = gpu::inherit_streams();
    std::unique_ptr<gpu::Event> handoff_;
    gpu::ForeignReleaser foreign_;
    std::deque<gpu::Batch*> uploading_;
    gpu::StreamReaper reaper_, upload_reaper_;
    std::unique_ptr<gpu::Stream> stream_;
    std::unique_ptr<gpu::BatchPool> staging_pool_;
    std::unique_ptr<gpu::BatchPool> dev_in_pool_;
    gpu::Batch* staging_ = nullptr;
};

}  // namespace wf
