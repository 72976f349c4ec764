// Python extension (pybind11, built with hipcc for gfx950): exposes
//   * kernel-level entry points used by the numerics tests (device pointers come
//     from torch tensors, results are compared against plain PyTorch references)
//   * persistent benchmark pipelines built from the C++ library (PipeGraph,
//     Batch_Source, Map_GPU, Filter_GPU, ...) that bench.py drives step by step.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <wf/windflow.hpp>
#include <wf/windflow_gpu.hpp>

#include "bench_common.hpp"
#include "ysb_gpu_common.hpp"  // apps/ysb

namespace py = pybind11;
using namespace wf;
using wfbench::result_t;
using wfbench::tuple_t;

// ---------------------------------------------------------------------------
// kernel-level entry points (pointers are device addresses, stream 0)
// ---------------------------------------------------------------------------
struct ScaleValue {
    double factor;
    __host__ __device__ void operator()(tuple_t& t) const { t.value = t.value * factor + t.aux; }
};
struct ValueBelow {
    double threshold;
    __host__ __device__ bool operator()(const tuple_t& t) const { return t.value < threshold; }
};
struct EvenKeyMask {
    __host__ __device__ void operator()(const tuple_t& t, bool& keep) const { keep = (t.key & 1) == 0; }
};

static void check_ptr(uintptr_t p, const char* what) {
    if (p == 0) throw std::invalid_argument(std::string("null device pointer: ") + what);
}

// in-place map over n tuples: value = value*factor + aux
static void map_scale_inplace(uintptr_t tuples, size_t n, double factor) {
    check_ptr(tuples, "tuples");
    gpu::DeviceBuffer<gpu::BatchHeader> hdr(1);
    gpu::BatchHeader h{};
    h.count = (uint32_t)n;
    WF_HIP_CHECK(hipMemcpy(hdr.data(), &h, sizeof(h), hipMemcpyHostToDevice));
    hipLaunchKernelGGL((gpu::map_inplace_kernel<tuple_t, ScaleValue>), dim3(gpu::grid_for(n)), dim3(gpu::BLOCK_THREADS), 0, 0, hdr.data(),
                       reinterpret_cast<tuple_t*>(tuples), ScaleValue{factor});
    WF_HIP_CHECK(hipGetLastError());
    WF_HIP_CHECK(hipDeviceSynchronize());
}

// out-of-place map (K6, gpu::map_kernel) over n tuples: result = (key, id, ts, value*factor + aux); also copies the header
struct ScaleToResult {
    double factor;
    __host__ __device__ void operator()(const tuple_t& t, result_t& r) const {
        r.setControlFields(t.key, t.id, t.ts);
        r.value = t.value * factor + t.aux;
    }
};
static size_t map_scale_out_of_place(uintptr_t tuples, size_t n, uintptr_t results, double factor) {
    check_ptr(tuples, "tuples");
    check_ptr(results, "results");
    gpu::DeviceBuffer<gpu::BatchHeader> hdr(2);
    gpu::BatchHeader h{};
    h.count = (uint32_t)n;
    h.watermark = 77;
    WF_HIP_CHECK(hipMemcpy(hdr.data(), &h, sizeof(h), hipMemcpyHostToDevice));
    hipLaunchKernelGGL((gpu::map_kernel<tuple_t, result_t, ScaleToResult>), dim3(gpu::grid_for(std::max<size_t>(n, 1))), dim3(gpu::BLOCK_THREADS), 0, 0,
                       hdr.data(), reinterpret_cast<const tuple_t*>(tuples), hdr.data() + 1, reinterpret_cast<result_t*>(results), ScaleToResult{factor});
    WF_HIP_CHECK(hipGetLastError());
    WF_HIP_CHECK(hipDeviceSynchronize());
    gpu::BatchHeader out{};
    WF_HIP_CHECK(hipMemcpy(&out, hdr.data() + 1, sizeof(out), hipMemcpyDeviceToHost));
    if (out.watermark != 77) throw std::runtime_error("map_kernel did not copy the batch header");
    return out.count;
}

// stable compaction of the tuples with value < threshold (mode 0) or even key (mode 1);
// returns the number of survivors
static size_t filter_compact(uintptr_t in, size_t n, uintptr_t out, int mode, double threshold) {
    check_ptr(in, "in");
    check_ptr(out, "out");
    gpu::DeviceBuffer<gpu::BatchHeader> hdr(2);
    gpu::BatchHeader h{};
    h.count = (uint32_t)n;
    WF_HIP_CHECK(hipMemcpy(hdr.data(), &h, sizeof(h), hipMemcpyHostToDevice));
    const size_t words = gpu::filter_workspace_words(std::max<size_t>(n, 1));
    gpu::DeviceBuffer<unsigned long long> ws(words);
    WF_HIP_CHECK(hipMemset(ws.data(), 0, words * sizeof(unsigned long long)));
    gpu::PinnedBuffer<uint32_t> mirror(16);
    mirror[0] = 0xffffffffu;
    const unsigned grid = (unsigned)std::max<size_t>(1, (n + gpu::FILTER_TILE - 1) / gpu::FILTER_TILE);
    if (mode == 0) {
        using P = gpu::PredicateAdapter<tuple_t, ValueBelow>;
        hipLaunchKernelGGL((gpu::filter_compact_kernel<tuple_t, P>), dim3(grid), dim3(gpu::BLOCK_THREADS), 0, 0, hdr.data(),
                           reinterpret_cast<const tuple_t*>(in), hdr.data() + 1, reinterpret_cast<tuple_t*>(out), mirror.data(), ws.data(),
                           P{ValueBelow{threshold}});
    } else {
        using P = gpu::PredicateAdapter<tuple_t, EvenKeyMask>;
        hipLaunchKernelGGL((gpu::filter_compact_kernel<tuple_t, P>), dim3(grid), dim3(gpu::BLOCK_THREADS), 0, 0, hdr.data(),
                           reinterpret_cast<const tuple_t*>(in), hdr.data() + 1, reinterpret_cast<tuple_t*>(out), mirror.data(), ws.data(),
                           P{EvenKeyMask{}});
    }
    WF_HIP_CHECK(hipGetLastError());
    WF_HIP_CHECK(hipDeviceSynchronize());
    gpu::BatchHeader oh{};
    WF_HIP_CHECK(hipMemcpy(&oh, hdr.data() + 1, sizeof(oh), hipMemcpyDeviceToHost));
    if (oh.count != mirror[0]) throw std::runtime_error("filter_compact: device header and pinned mirror disagree");
    return oh.count;
}

// ---------------------------------------------------------------------------
// benchmark pipelines
// ---------------------------------------------------------------------------
struct SinkTotals {
    std::mutex m;
    uint64_t tuples = 0;
    double value_sum = 0.0;
    uint64_t batches = 0;
};

// S gated batch sources -> Map_GPU (in place) -> Filter_GPU (compaction) -> D2H -> batch sink
// (BASELINE.json config 2; reference workload: gpu_map_stateless.cpp + gpu_filter_stateless.cpp)
class MapFilterPipeline {
public:
    MapFilterPipeline(int gpu_id, size_t n_sources, size_t batch_len, size_t n_keys, size_t inflight)
        : n_sources_(n_sources), gates_(new wfbench::StepGate[n_sources]), graph_("map_filter_gpu", Mode::DEFAULT) {
        auto totals = totals_ = std::make_shared<SinkTotals>();
        auto barrier = barrier_ = std::make_shared<wfbench::StepBarrier>();
        wfbench::GatedGenerator gen(gates_.get(), n_keys, 1, n_sources);
        source_ = BatchSource_Builder(gen).withName("source").withParallelism(n_sources).withBatch(batch_len).withInFlightBatches(inflight)
                      .withGPUConfiguration(gpu_id).build_unique();
        MultiPipe& mp = graph_.add_source(*source_);
        map_ = MapGPU_Builder(ScaleValue{2.0}).withName("map_gpu").withParallelism(n_sources).withInFlightBatches(inflight)
                   .withGPUConfiguration(gpu_id).build_unique();
        mp.add(*map_);
        filter_ = FilterGPU_Builder(ValueBelow{1024.0}).withName("filter_gpu").withParallelism(n_sources).withInFlightBatches(inflight)
                      .withGPUConfiguration(gpu_id).build_unique();
        mp.chain(*filter_);
        auto sink_f = [totals, barrier](const tuple_t* data, size_t n, const gpu::BatchHeader& h) {
            double s = 0.0;
            for (size_t i = 0; i < n; i++) s += data[i].value;
            {
                std::lock_guard<std::mutex> l(totals->m);
                totals->tuples += n;
                totals->value_sum += s;
                totals->batches += (data != nullptr);
            }
            if (h.reserved[0] != 0) barrier->arrive(h.reserved[0]);
        };
        sink_ = BatchSink_Builder(sink_f).withName("sink").withParallelism(n_sources).build_unique();
        mp.chain_sink(*sink_);
        latency_ = sink_->latency();
        graph_.start();
    }
    ~MapFilterPipeline() { close(); }

    // feed `tuples_per_source` tuples from every source and wait until every result reached the sinks
    void step(int64_t tuples_per_source) {
        py::gil_scoped_release rel;
        step_id_++;
        for (size_t i = 0; i < n_sources_; i++) gates_[i].release(step_id_, tuples_per_source);
        barrier_->wait(step_id_, (int)n_sources_);
    }
    // k steps streamed back to back (no drain between them): returns when the last one has reached the sinks
    void steps(int64_t k, int64_t tuples_per_source) {
        py::gil_scoped_release rel;
        step_id_ += (uint64_t)k;
        for (size_t i = 0; i < n_sources_; i++) gates_[i].release(step_id_, k * tuples_per_source);
        barrier_->wait(step_id_, (int)n_sources_);
    }
    void close() {
        if (closed_) return;
        closed_ = true;
        py::gil_scoped_release rel;
        for (size_t i = 0; i < n_sources_; i++) gates_[i].shutdown();
        graph_.wait_end();
    }
    py::dict stats() {
        py::dict d;
        std::lock_guard<std::mutex> l(totals_->m);
        d["sink_tuples"] = totals_->tuples;
        d["sink_value_sum"] = totals_->value_sum;
        d["sink_batches"] = totals_->batches;
        d["latency_p50_us"] = latency_->percentile_us(50);
        d["latency_p99_us"] = latency_->percentile_us(99);
        d["latency_mean_us"] = latency_->mean_us();
        d["threads"] = graph_.getNumThreads();
        return d;
    }
    void reset_latency() { latency_->clear(); }
    std::string dot() { return graph_.get_DOTGraph(); }

private:
    size_t n_sources_;
    std::unique_ptr<wfbench::StepGate[]> gates_;
    PipeGraph graph_;
    std::shared_ptr<SinkTotals> totals_;
    std::shared_ptr<wfbench::StepBarrier> barrier_;
    std::shared_ptr<LatencyRecorder> latency_;
    std::unique_ptr<Batch_Source<tuple_t>> source_;
    std::unique_ptr<Map_GPU<tuple_t, tuple_t>> map_;
    std::unique_ptr<Filter_GPU<tuple_t>> filter_;
    std::unique_ptr<Batch_Sink<tuple_t>> sink_;
    uint64_t step_id_ = 0;
    bool closed_ = false;
};

// ---------------------------------------------------------------------------
// keyed FlatFAT engine driven directly (numerics tests): tuples is an [n,5] int64
// array of 40-byte tuples, processed in batches of batch_len; returns an [m,4]
// int64 array of 32-byte results (key, gwid, ts, value-bits) + the device counters
// ---------------------------------------------------------------------------
// combine used by the engine tests: 0 = sum, 1 = newest value wins, 2 = oldest value wins (the last two are
// associative but NOT commutative: any reordering inside the engine shows up in the results)
struct CombByMode {
    int mode;
    __host__ __device__ void operator()(const result_t& a, const result_t& b, result_t& out) const {
        out.value = mode == 0 ? a.value + b.value : (mode == 1 ? b.value : a.value);
    }
};

// FlatFAT_GPU driven from Python: one device-resident aggregator tree whose leaves are a circular buffer
// (reference: wf/flatfat_gpu.hpp:251-434 build / update / getResults)
class FlatFatSession {
public:
    FlatFatSession(size_t capacity, size_t max_windows, int comb_mode) : fat_(CombByMode{comb_mode}, capacity, max_windows, stream_.get()) {}
    size_t capacity() const { return fat_.capacity(); }
    size_t size() const { return fat_.size(); }
    void build(const std::vector<double>& values) { fat_.build(leaves(values).data(), values.size()); }
    void update(const std::vector<double>& values, size_t n_remove) { fat_.update(leaves(values).data(), values.size(), n_remove); }
    // n_windows windows of `win` leaves sliding by `slide` from the oldest live leaf: (value, ts) per window
    std::vector<std::pair<double, uint64_t>> results(size_t n_windows, size_t win, size_t slide) {
        std::vector<result_t> stamped(n_windows);
        for (size_t w = 0; w < n_windows; w++) stamped[w].setControlFields(7, w, 0);
        fat_.computeResults(stamped.data(), n_windows, win, slide);
        size_t n = 0;
        const result_t* r = fat_.waitResults(n);
        std::vector<std::pair<double, uint64_t>> out;
        for (size_t w = 0; w < n; w++) {
            if (r[w].key != 7 || r[w].id != w) throw std::runtime_error("FlatFAT_GPU lost the stamped control fields");
            out.push_back({r[w].value, r[w].ts});
        }
        return out;
    }
    uint64_t kernels() const { return fat_.num_kernels(); }

private:
    std::vector<result_t> leaves(const std::vector<double>& values) {
        std::vector<result_t> l(values.size());
        for (size_t i = 0; i < values.size(); i++) {
            l[i].setControlFields(7, 0, next_ts_++);
            l[i].value = values[i];
        }
        return l;
    }
    gpu::Stream stream_;
    gpu::FlatFAT_GPU<result_t, CombByMode> fat_;
    uint64_t next_ts_ = 1;
};

static py::tuple key_ffat_run(py::array_t<int64_t, py::array::c_style | py::array::forcecast> tuples, uint64_t win, uint64_t slide,
                              uint64_t delay, bool time_based, size_t max_keys, size_t batch_len, size_t out_capacity, int comb_mode,
                              bool skip_gaps) {
    auto buf = tuples.request();
    if (buf.ndim != 2 || buf.shape[1] != 5) throw std::invalid_argument("tuples must be [n,5] int64");
    const size_t n = (size_t)buf.shape[0];
    const tuple_t* host = static_cast<const tuple_t*>(buf.ptr);
    WF_HIP_CHECK(hipSetDevice(0));
    gpu::Stream stream;
    if (out_capacity == 0) out_capacity = batch_len;
    gpu::KeyedFFATEngine<tuple_t, result_t, wfbench::LiftValue, CombByMode> engine(wfbench::LiftValue{}, CombByMode{comb_mode}, win, slide, delay,
                                                                                    time_based ? TB : CB, max_keys, batch_len, out_capacity, 0, 3, skip_gaps);
    gpu::BatchPool in_pool(gpu::MemKind::DEVICE, 0, sizeof(tuple_t), batch_len, 2);
    std::vector<result_t> results;
    uint64_t kernels = 0;
    auto drain = [&](gpu::Batch* out) {
        out->ready.record(stream);
        stream.sync();
        const uint32_t cnt = *out->count_mirror;
        const size_t old = results.size();
        results.resize(old + cnt);
        if (cnt) WF_HIP_CHECK(hipMemcpy(results.data() + old, out->data<result_t>(), cnt * sizeof(result_t), hipMemcpyDeviceToHost));
        out->release(stream);
    };
    for (size_t off = 0; off < n; off += batch_len) {
        const size_t cnt = std::min(batch_len, n - off);
        gpu::Batch* d = in_pool.acquire(stream);
        d->host_hdr.count = (uint32_t)cnt;
        WF_HIP_CHECK(hipMemcpyAsync(d->header(), &d->host_hdr, sizeof(gpu::BatchHeader), hipMemcpyHostToDevice, stream));
        WF_HIP_CHECK(hipMemcpyAsync(d->data<tuple_t>(), host + off, cnt * sizeof(tuple_t), hipMemcpyHostToDevice, stream));
        gpu::Batch* out = engine.process(d, stream, kernels);
        d->release(stream);
        drain(out);
    }
    {
        size_t cursor = 0;
        gpu::Batch* out = nullptr;
        while (engine.flush_range(cursor, stream, kernels, out)) drain(out);
    }
    gpu::FFATCounters c = engine.read_counters(stream);
    py::array_t<int64_t> arr({(py::ssize_t)results.size(), (py::ssize_t)4});
    if (!results.empty()) std::memcpy(arr.mutable_data(), results.data(), results.size() * sizeof(result_t));
    py::dict counters;
    counters["dropped"] = c.dropped;
    counters["overflow"] = c.overflow;
    counters["skipped_windows"] = c.skipped_windows;
    counters["huge_runs"] = c.huge_runs;
    counters["table_full"] = c.table_full;
    counters["kernels"] = kernels;
    counters["device_bytes"] = engine.device_bytes();
    return py::make_tuple(arr, counters);
}

// stable radix sort of (key, value) pairs on the device (numerics test of the primitive)
// mode 0: the multi-launch radix sort; 1: behind an OPEN gate (batches of at most 16 Ki pairs: the one-workgroup bitonic network);
// 2: behind an open gate in rescue mode (the one-workgroup radix sort); 3: behind a CLOSED gate (nothing may move)
static void radix_sort_pairs(uintptr_t keys, uintptr_t vals, size_t n, int bits, int mode) {
    check_ptr(keys, "keys");
    check_ptr(vals, "vals");
    gpu::DeviceBuffer<uint32_t> n_dev(2);
    const uint32_t words[2] = {(uint32_t)n, 7u};
    WF_HIP_CHECK(hipMemcpy(n_dev.data(), words, 8, hipMemcpyHostToDevice));
    gpu::SortGate gate{};
    if (mode != 0) gate = gpu::SortGate{n_dev.data() + 1, mode == 3 ? 8u : 7u, mode == 2 ? 1u : 0u};
    gpu::RadixSorter sorter;
    gpu::Stream st;
    hipStream_t s0 = st.get();
    auto sorted = sorter.sort(reinterpret_cast<uint32_t*>(keys), reinterpret_cast<uint32_t*>(vals), n_dev.data(), n, bits, s0, gate);
    if (mode == 3) {  // closed gate: the consumers read the unsorted arrays
        WF_HIP_CHECK(hipStreamSynchronize(s0));
        return;
    }
    if (sorted.keys != reinterpret_cast<uint32_t*>(keys)) {
        WF_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<void*>(keys), sorted.keys, n * 4, hipMemcpyDeviceToDevice, s0));
        WF_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<void*>(vals), sorted.vals, n * 4, hipMemcpyDeviceToDevice, s0));
    }
    WF_HIP_CHECK(hipGetLastError());
    WF_HIP_CHECK(hipStreamSynchronize(s0));
}

static uint32_t exclusive_scan(uintptr_t in, uintptr_t out, size_t n) {
    check_ptr(in, "in");
    check_ptr(out, "out");
    gpu::DeviceBuffer<unsigned long long> ws(gpu::scan_workspace_words(n));
    gpu::DeviceBuffer<uint32_t> total(1);
    gpu::exclusive_scan_u32(reinterpret_cast<const uint32_t*>(in), reinterpret_cast<uint32_t*>(out), n, total.data(), ws.data(), 0);
    WF_HIP_CHECK(hipGetLastError());
    WF_HIP_CHECK(hipDeviceSynchronize());
    uint32_t t = 0;
    WF_HIP_CHECK(hipMemcpy(&t, total.data(), 4, hipMemcpyDeviceToHost));
    return t;
}

static void exchange_stats(py::dict& d, const ExchangeStats& x, size_t lanes) {
    d["exchange_rounds"] = x.rounds;
    d["exchange_empty_rounds"] = x.empty_rounds;
    d["exchange_batches_in"] = x.bundles;
    d["exchange_batches_out"] = x.lane_batches;
    d["exchange_tuples_sent"] = x.tuples_sent;
    d["exchange_tuples_received"] = x.tuples_received;
    d["exchange_max_link_bytes_per_round"] = x.max_link_bytes_per_round;
    uint64_t busiest = 0, all_links = 0;
    for (uint64_t b : x.link_bytes_sent) {
        busiest = std::max(busiest, b);
        all_links += b;
    }
    d["exchange_busiest_link_bytes"] = busiest;  // bytes this rank put on its busiest outgoing link over the whole run
    d["exchange_link_bytes_total"] = all_links;
    d["exchange_rounds_per_s"] = x.elapsed_us ? (double)x.rounds * 1e6 / (double)x.elapsed_us : 0.0;
    d["exchange_lanes"] = lanes;
}

// S gated batch sources -> Key_FFAT_GPU (device-resident keyed sliding-window sum) -> D2H -> batch sinks
// (BASELINE.json config 4, the flagship "sliding-window aggregation" pipeline)
// sources that emit the same keys advance together: within the process, and across the ranks of a transport through the
// progress counters of its control board (bench_common.hpp SkewBound)
static std::shared_ptr<wfbench::SkewBound> make_skew_bound(const std::shared_ptr<dist::Transport>& tr, size_t n_sources) {
    if (!tr || tr->world() == 1 || n_sources > (size_t)dist::ControlBoard::PROGRESS_LANES) return std::make_shared<wfbench::SkewBound>(n_sources, 4);
    std::vector<std::atomic<uint64_t>*> rows;
    for (int r = 0; r < tr->world(); r++) rows.push_back(tr->board().progress(r));
    return std::make_shared<wfbench::SkewBound>(std::move(rows), (size_t)tr->rank(), n_sources, 8);
}

class WinAggPipeline {
public:
    // transport == nullptr: every source lane feeds its own Key_FFAT_GPU replica (keys are partitioned at the
    // sources).  With a transport the tuples are re-partitioned by key across the ranks first (Exchange_GPU:
    // RCCL all-to-all over xGMI) and then across the local replicas (device KeyBy).
    // key_dist / shared_keys: how the sources draw keys (bench_common.hpp KeySpace).  With shared keys every source emits every
    // key, so the batches go through the device KeyBy in front of the aggregation replicas.
    WinAggPipeline(int gpu_id, size_t n_sources, size_t batch_len, size_t n_keys, uint64_t win, uint64_t slide, bool time_based, uint64_t delay,
                   size_t inflight, std::shared_ptr<dist::Transport> transport = nullptr, size_t agg_replicas = 0, int key_dist = 0,
                   double zipf_alpha = 1.3, bool shared_keys = false, bool deterministic = false, bool device_source = false, bool combiner = false)
        : n_sources_(n_sources), n_sinks_(n_sources), gates_(new wfbench::StepGate[n_sources]),
          graph_("win_agg_gpu", deterministic ? Mode::DETERMINISTIC : Mode::DEFAULT) {
        auto totals = totals_ = std::make_shared<SinkTotals>();
        auto barrier = barrier_ = std::make_shared<wfbench::StepBarrier>();
        const size_t rank = transport ? (size_t)transport->rank() : 0, world = transport ? (size_t)transport->world() : 1;
        wfbench::KeySpace keyspace = wfbench::KeySpace::make(n_keys, key_dist, zipf_alpha, shared_keys);
        if (device_source) {
            // the tuples are born in HBM (Device_Source): same stream, no host generator and no PCIe upload in the way
            wfbench::DeviceGenerator dgen;
            dgen.n_keys = keyspace.n_keys;
            dgen.kmul = world * n_sources;
            dgen.src_offset = rank * n_sources;
            dgen.dist = key_dist;
            dgen.shared = shared_keys ? 1u : 0u;
            if (keyspace.quantiles) {
                WF_HIP_CHECK(hipSetDevice(gpu_id));
                zipf_dev_.resize(keyspace.quantiles->size());
                WF_HIP_CHECK(hipMemcpy(zipf_dev_.data(), keyspace.quantiles->data(), keyspace.quantiles->size() * sizeof(wfbench::KeySpace::Bucket),
                                       hipMemcpyHostToDevice));
                zipf_host_ = keyspace.quantiles;
                dgen.q_dev = zipf_dev_.data();
                dgen.q_host = zipf_host_->data();
            }
            wfbench::GatedControl ctl;
            ctl.gates = gates_.get();
            if (shared_keys) ctl.skew = make_skew_bound(transport, n_sources);
            source_ = DeviceSource_Builder(dgen).withControl(ctl).withName("device_source").withParallelism(n_sources).withBatch(batch_len)
                          .withInFlightBatches(inflight).withGPUConfiguration(gpu_id).build_unique();
        } else {
            wfbench::GatedGenerator gen(gates_.get(), keyspace, 1, n_sources, rank * n_sources, world * n_sources);
            // sources that emit the same keys read one feed: at most 4 batches apart (bench_common.hpp SkewBound)
            if (shared_keys) gen.bound_skew(make_skew_bound(transport, n_sources));
            source_ = BatchSource_Builder(gen).withName("source").withParallelism(n_sources).withBatch(batch_len).withInFlightBatches(inflight)
                          .withGPUConfiguration(gpu_id).build_unique();
        }
        MultiPipe& mp = graph_.add_source(*source_);
        KeyFFATGPUConfig wcfg;
        wcfg.win_len = win;
        wcfg.slide_len = slide;
        wcfg.delay = delay;
        wcfg.type = time_based ? TB : CB;
        wcfg.max_keys = n_keys;
        wcfg.out_capacity = batch_len;
        wcfg.lanes_are_key_disjoint = true;  // the generator gives every source its own keys
        GPUConfig cfg;
        cfg.gpu_id = gpu_id;
        cfg.batch_len = batch_len;
        cfg.pool_slabs = inflight;
        const bool use_combiner = combiner && transport && time_based;
        if (combiner && !use_combiner) throw std::invalid_argument("WinAggPipeline: the combiner needs a transport and time-based windows");
        if (transport) {
            // the exchange delivers `lanes` key-disjoint streams (owner rank, then lane inside the rank, both from the key hash):
            // one aggregation replica per lane, no second partitioning pass behind the all-to-all
            n_sinks_ = agg_replicas ? agg_replicas : std::min<size_t>(n_sources, 4);
            if (use_combiner) {
                // fold every batch per (key, pane) where it is produced; the partial results travel instead of the tuples
                combiner_ = PaneCombinerGPU_Builder(wfbench::LiftValue{}, wfbench::CombSum{}).withTBWindows(std::chrono::microseconds(win), std::chrono::microseconds(slide))
                                .withName("pane_combiner").withParallelism(n_sources).withGPUConfiguration(gpu_id).withInFlightBatches(inflight).build_unique();
                mp.chain(*combiner_);
                auto ex = ExchangeGPU_Builder<result_t>(transport).withName("exchange").withParallelism(n_sources).withLocalLanes(n_sources)
                              .withOutputLanes(n_sinks_).withGPUConfiguration(gpu_id).withInFlightBatches(inflight).build_unique();
                auto* raw = ex.get();
                exchange_stats_ = [raw] { return raw->stats(); };
                mp.add(*ex);
                exchange_ = std::move(ex);
            } else {
                auto ex = ExchangeGPU_Builder<tuple_t>(transport).withName("exchange").withParallelism(n_sources).withLocalLanes(n_sources)
                              .withOutputLanes(n_sinks_).withGPUConfiguration(gpu_id).withInFlightBatches(inflight).build_unique();
                auto* raw = ex.get();
                exchange_stats_ = [raw] { return raw->stats(); };
                mp.add(*ex);
                exchange_ = std::move(ex);
            }
            wcfg.lanes_are_key_disjoint = true;
            // hash partitioning is only balanced on average: leave headroom in the per-replica key tables and in the
            // result batches (a lane receives what `world` ranks send it in a round)
            // (shared keys: n_keys is the key space of the whole job, spread by hash over world x lanes replicas)
            wcfg.max_keys = shared_keys ? (n_keys + world * n_sinks_ - 1) / (world * n_sinks_) : n_keys * n_sources / n_sinks_;
            wcfg.max_keys += wcfg.max_keys / 4 + 1024;
            // a round carries up to one batch per local lane and rank: the lane batches hold up to 2 * n_sources / lanes input batches
            wcfg.out_capacity = std::max<size_t>(2 * n_sources * batch_len / n_sinks_, batch_len);
        } else if (shared_keys) {
            // every lane carries every key: device KeyBy in front of the replicas (the batches of all sources are packed by replica
            // and merged: a replica receives batches of up to 2 * n_sources / replicas input batches)
            wcfg.lanes_are_key_disjoint = false;
            // Few keys mean runs of thousands of tuples per key and batch, which the engine spreads over the whole chip key by key:
            // several replicas doing that at once only get in each other's way (64 keys, DEFAULT graph: one replica 0.98-1.09 G
            // tuples/s with p99 7 ms in every run; four replicas 0.97 G / 13 ms or 0.5-0.65 G / 23 ms depending on how their
            // kernels happened to interleave; 512 keys: 1.01-1.07 G against 0.78-1.09 G - profiles/r3_shared_keys/replicas.md).  From about
            // a thousand keys on, 3-4 replicas win (1.24-1.33 G against 1.09-1.17 G).
            n_sinks_ = agg_replicas ? agg_replicas : (n_keys <= 512 && !deterministic) ? 1 : std::min<size_t>(n_sources, 4);
            // (a DETERMINISTIC graph keeps four: the lanes are merged in timestamp order in front of the replicas either way, and
            // one replica behind that merge measured 0.53 G / p99 32 ms against 0.89-1.0 G / 22 ms)
            wcfg.max_keys = n_keys / n_sinks_;  // shared: n_keys is the whole key space
            wcfg.max_keys += wcfg.max_keys / 4 + 1024;
            wcfg.out_capacity = std::max<size_t>(2 * n_sources * batch_len / n_sinks_, 2 * batch_len);
            // no KeyBy in front of a single replica: it sees every lane's punctuation - unless the graph is DETERMINISTIC, where the
            // merge in front of it hands on one punctuation per step
            if (n_sinks_ == 1 && !deterministic) punct_expected_ = n_sources;
        }
        if (punct_expected_ == 0) punct_expected_ = n_sinks_;
        if (use_combiner) {
            auto k = std::make_unique<Key_FFAT_GPU<result_t, result_t>>(PartialLift<result_t>{}, wfbench::CombSum{}, wcfg, n_sinks_, "key_ffat_gpu", cfg);
            auto* raw = k.get();
            counters_ = [raw] { return raw->getDeviceCounters(); };
            mp.add(*k);
            kff_ = std::move(k);
        } else {
            auto k = std::make_unique<Key_FFAT_GPU<tuple_t, result_t>>(wfbench::LiftValue{}, wfbench::CombSum{}, wcfg, n_sinks_, "key_ffat_gpu", cfg);
            auto* raw = k.get();
            counters_ = [raw] { return raw->getDeviceCounters(); };
            mp.add(*k);
            kff_ = std::move(k);
        }
        auto sink_f = [totals, barrier](const result_t* data, size_t n, const gpu::BatchHeader& h) {
            double s = 0.0;
            for (size_t i = 0; i < n; i++) s += data[i].value;
            {
                std::lock_guard<std::mutex> l(totals->m);
                totals->tuples += n;
                totals->value_sum += s;
                totals->batches += (data != nullptr);
            }
            if (h.reserved[0] != 0) barrier->arrive(h.reserved[0]);
        };
        sink_ = BatchSink_Builder(sink_f).withName("sink").withParallelism(n_sinks_).build_unique();
        mp.chain_sink(*sink_);
        latency_ = sink_->latency();
        graph_.start();
    }
    ~WinAggPipeline() { close(); }
    void step(int64_t tuples_per_source) {
        py::gil_scoped_release rel;
        step_id_++;
        for (size_t i = 0; i < n_sources_; i++) gates_[i].release(step_id_, tuples_per_source);
        barrier_->wait(step_id_, (int)punct_expected_);
    }
    // k steps streamed back to back (no drain between them): returns when the last one has reached the sinks
    void steps(int64_t k, int64_t tuples_per_source) {
        py::gil_scoped_release rel;
        step_id_ += (uint64_t)k;
        for (size_t i = 0; i < n_sources_; i++) gates_[i].release(step_id_, k * tuples_per_source);
        barrier_->wait(step_id_, (int)punct_expected_);
    }
    void close() {
        if (closed_) return;
        closed_ = true;
        py::gil_scoped_release rel;
        for (size_t i = 0; i < n_sources_; i++) gates_[i].shutdown();
        graph_.wait_end();
    }
    py::dict stats() {
        py::dict d;
        {
            std::lock_guard<std::mutex> l(totals_->m);
            d["sink_tuples"] = totals_->tuples;
            d["sink_value_sum"] = totals_->value_sum;
            d["sink_batches"] = totals_->batches;
        }
        d["latency_p50_us"] = latency_->percentile_us(50);
        d["latency_p99_us"] = latency_->percentile_us(99);
        d["latency_mean_us"] = latency_->mean_us();
        d["latency_samples"] = latency_->count();
        d["threads"] = graph_.getNumThreads();
        uint64_t dropped = 0, overflow = 0, skipped = 0, full = 0, huge = 0;
        for (auto& c : counters_()) {
            huge += c.huge_runs;
            dropped += c.dropped;
            overflow += c.overflow;
            skipped += c.skipped_windows;
            full += c.table_full;
        }
        d["dropped"] = dropped;
        d["overflow"] = overflow;
        d["skipped_windows"] = skipped;
        d["table_full"] = full;
        d["huge_runs"] = huge;
        if (exchange_) exchange_stats(d, exchange_stats_(), n_sinks_);
        if (combiner_) d["combiner_tuples_in"] = combiner_->tuples_in();
        return d;
    }
    void reset_latency() { latency_->clear(); }
    std::string dot() { return graph_.get_DOTGraph(); }

private:
    size_t n_sources_, n_sinks_, punct_expected_ = 0;
    std::unique_ptr<wfbench::StepGate[]> gates_;
    PipeGraph graph_;
    std::shared_ptr<SinkTotals> totals_;
    std::shared_ptr<wfbench::StepBarrier> barrier_;
    std::shared_ptr<LatencyRecorder> latency_;
    gpu::DeviceBuffer<wfbench::KeySpace::Bucket> zipf_dev_;
    std::shared_ptr<std::vector<wfbench::KeySpace::Bucket>> zipf_host_;
    std::unique_ptr<Basic_Operator> source_, exchange_, kff_;
    std::unique_ptr<Pane_Combiner_GPU<tuple_t, result_t>> combiner_;
    std::function<ExchangeStats()> exchange_stats_;
    std::function<std::vector<gpu::FFATCounters>()> counters_;
    std::unique_ptr<Batch_Sink<result_t>> sink_;
    uint64_t step_id_ = 0;
    bool closed_ = false;
};

// ---------------------------------------------------------------------------
// non-incremental GPU windows (BASELINE.json config 3): S gated batch sources -> Key_Farm_GPU -> D2H -> batch sinks
// ---------------------------------------------------------------------------
// the "non-incremental user kernel": an opaque sequential window function (sum of the values)
struct WinSumGPU {
    __host__ __device__ void operator()(uint64_t, const tuple_t* data, size_t n, result_t* res, char*, size_t) const {
        double s = 0.0;
        for (size_t i = 0; i < n; i++) s += data[i].value;
        res->value = s;
    }
};
// window function of the engine tests: 0 = sum, 1 = newest value, 2 = oldest value, 3 = number of tuples
struct WinFuncByMode {
    int mode;
    __host__ __device__ void operator()(uint64_t, const tuple_t* data, size_t n, result_t* res, char*, size_t) const {
        if (mode == 0) {
            double s = 0.0;
            for (size_t i = 0; i < n; i++) s += data[i].value;
            res->value = s;
        } else if (mode == 1) {
            res->value = n ? data[n - 1].value : -1.0;
        } else if (mode == 2) {
            res->value = n ? data[0].value : -1.0;
        } else {
            res->value = (double)n;
        }
    }
};

class WinSeqPipeline {
public:
    // replicas = 0: one per source (lanes key-disjoint unless shared_keys); wave_reduce: the associative form of the same sum
    WinSeqPipeline(int gpu_id, size_t n_sources, size_t batch_len, size_t n_keys, uint64_t win, uint64_t slide, bool time_based, size_t inflight,
                   size_t replicas, bool wave_reduce, int key_dist, double zipf_alpha, bool shared_keys)
        : n_sources_(n_sources), n_sinks_(replicas ? replicas : n_sources), gates_(new wfbench::StepGate[n_sources]),
          graph_("win_seq_gpu", Mode::DEFAULT) {
        auto totals = totals_ = std::make_shared<SinkTotals>();
        auto barrier = barrier_ = std::make_shared<wfbench::StepBarrier>();
        wfbench::GatedGenerator gen(gates_.get(), wfbench::KeySpace::make(n_keys, key_dist, zipf_alpha, shared_keys), 1,
                                    n_sources);
        source_ = BatchSource_Builder(gen).withName("source").withParallelism(n_sources).withBatch(batch_len).withInFlightBatches(inflight)
                      .withGPUConfiguration(gpu_id).build_unique();
        MultiPipe& mp = graph_.add_source(*source_);
        const bool disjoint = !shared_keys && n_sinks_ == n_sources;
        keyby_free_ = disjoint;
        const size_t keys_per_replica = (shared_keys ? n_keys : n_keys * n_sources) / n_sinks_;  // shared: n_keys is the whole key space
        const size_t max_keys = keys_per_replica + keys_per_replica / 4 + 16;
        auto configure = [&](auto builder) {
            if (time_based)
                builder.withTBWindows(std::chrono::microseconds(win), std::chrono::microseconds(slide));
            else
                builder.withCBWindows(win, slide);
            builder.withName("key_farm_gpu").withParallelism(n_sinks_).withInFlightBatches(inflight).withGPUConfiguration(gpu_id).withMaxKeys(max_keys)
                .withOutputCapacity(std::max<size_t>(batch_len / std::max<uint64_t>(1, time_based ? 1 : slide) + max_keys, 8192))
                .withKeyDisjointLanes(disjoint);
            if (time_based) builder.withArchiveCapacity((size_t)std::min<uint64_t>(2 * win + 64, 1 << 22));
            auto op = builder.build_unique();
            auto* raw = op.get();
            counters_ = [raw] { return raw->getDeviceCounters(); };
            return op;
        };
        if (wave_reduce)
            op_ = configure(KeyFarmGPU_Builder(gpu::wave_reduce(wfbench::LiftValue{}, wfbench::CombSum{})));
        else
            op_ = configure(KeyFarmGPU_Builder(WinSumGPU{}));
        mp.add(*op_);
        auto sink_f = [totals, barrier](const result_t* data, size_t n, const gpu::BatchHeader& h) {
            double s = 0.0;
            for (size_t i = 0; i < n; i++) s += data[i].value;
            {
                std::lock_guard<std::mutex> l(totals->m);
                totals->tuples += n;
                totals->value_sum += s;
                totals->batches += (data != nullptr);
            }
            if (h.reserved[0] != 0) barrier->arrive(h.reserved[0]);
        };
        sink_ = BatchSink_Builder(sink_f).withName("sink").withParallelism(n_sinks_).build_unique();
        mp.chain_sink(*sink_);
        latency_ = sink_->latency();
        graph_.start();
    }
    ~WinSeqPipeline() { close(); }
    void step(int64_t tuples_per_source) { steps(1, tuples_per_source); }
    void steps(int64_t k, int64_t tuples_per_source) {
        py::gil_scoped_release rel;
        step_id_ += (uint64_t)k;
        for (size_t i = 0; i < n_sources_; i++) gates_[i].release(step_id_, k * tuples_per_source);
        barrier_->wait(step_id_, (int)punct_expected());
    }
    void close() {
        if (closed_) return;
        closed_ = true;
        py::gil_scoped_release rel;
        for (size_t i = 0; i < n_sources_; i++) gates_[i].shutdown();
        graph_.wait_end();
    }
    py::dict stats() {
        py::dict d;
        {
            std::lock_guard<std::mutex> l(totals_->m);
            d["sink_tuples"] = totals_->tuples;
            d["sink_value_sum"] = totals_->value_sum;
            d["sink_batches"] = totals_->batches;
        }
        d["latency_p50_us"] = latency_->percentile_us(50);
        d["latency_p99_us"] = latency_->percentile_us(99);
        d["latency_mean_us"] = latency_->mean_us();
        d["latency_samples"] = latency_->count();
        d["threads"] = graph_.getNumThreads();
        uint64_t dropped = 0, deferred = 0, full = 0, carry = 0;
        for (auto& c : counters_()) {
            dropped += c.dropped;
            deferred += c.deferred;
            full += c.table_full;
            carry += c.carry_overflow;
        }
        d["dropped"] = dropped;
        d["deferred_windows"] = deferred;
        d["table_full"] = full;
        d["carry_overflow"] = carry;
        d["overflow"] = carry;
        return d;
    }
    void reset_latency() { latency_->clear(); }
    std::string dot() { return graph_.get_DOTGraph(); }

private:
    // a step's punctuation reaches every sink once: lanes map 1:1, or the device KeyBy merges the punctuations of the sources
    size_t punct_expected() const { return (n_sinks_ == 1 && n_sources_ > 1) ? n_sources_ : n_sinks_; }
    size_t n_sources_, n_sinks_;
    bool keyby_free_ = true;
    std::unique_ptr<wfbench::StepGate[]> gates_;
    PipeGraph graph_;
    std::shared_ptr<SinkTotals> totals_;
    std::shared_ptr<wfbench::StepBarrier> barrier_;
    std::shared_ptr<LatencyRecorder> latency_;
    std::unique_ptr<Batch_Source<tuple_t>> source_;
    std::unique_ptr<Basic_Operator> op_;
    std::function<std::vector<gpu::WinCounters>()> counters_;
    std::unique_ptr<Batch_Sink<result_t>> sink_;
    uint64_t step_id_ = 0;
    bool closed_ = false;
};

// WinBatchEngine driven directly (numerics tests): tuples is an [n,5] int64 array of 40-byte tuples processed in batches of
// batch_len; returns an [m,4] int64 array of 32-byte results (key, gwid, ts, value bits) + the device counters.
// func_mode: see WinFuncByMode; wave_reduce: evaluate sum / newest / oldest through wave_reduce(lift, CombByMode)
static py::tuple win_batch_run(py::array_t<int64_t, py::array::c_style | py::array::forcecast> tuples, uint64_t win, uint64_t slide, uint64_t delay,
                               bool time_based, size_t max_keys, size_t batch_len, size_t carry_capacity, size_t out_capacity, int func_mode,
                               bool wave_reduce, bool renumber) {
    auto buf = tuples.request();
    if (buf.ndim != 2 || buf.shape[1] != 5) throw std::invalid_argument("tuples must be [n,5] int64");
    const size_t n = (size_t)buf.shape[0];
    const tuple_t* host = static_cast<const tuple_t*>(buf.ptr);
    WF_HIP_CHECK(hipSetDevice(0));
    gpu::Stream stream;
    WinParams p;
    p.win_len = win;
    p.slide_len = slide;
    p.triggering_delay = delay;
    p.winType = time_based ? TB : CB;
    p.config = WinOperatorConfig(0, 1, slide, 0, 1, slide);
    gpu::WinGPUOptions opt;
    opt.max_keys = max_keys;
    opt.carry_capacity = carry_capacity;
    opt.out_capacity = out_capacity;
    opt.renumber = renumber;
    std::vector<result_t> results;
    uint64_t kernels = 0;
    gpu::WinCounters ctr{};
    size_t dev_bytes = 0;
    auto run = [&](auto& engine) {
        gpu::BatchPool in_pool(gpu::MemKind::DEVICE, 0, sizeof(tuple_t), batch_len, 2);
        auto drain = [&](gpu::Batch* out) {
            out->ready.record(stream);
            stream.sync();
            const uint32_t cnt = *out->count_mirror;
            const size_t old = results.size();
            results.resize(old + cnt);
            if (cnt) WF_HIP_CHECK(hipMemcpy(results.data() + old, out->data<result_t>(), cnt * sizeof(result_t), hipMemcpyDeviceToHost));
            out->release(stream);
        };
        for (size_t off = 0; off < n; off += batch_len) {
            const size_t cnt = std::min(batch_len, n - off);
            gpu::Batch* d = in_pool.acquire(stream);
            d->host_hdr.count = (uint32_t)cnt;
            WF_HIP_CHECK(hipMemcpyAsync(d->header(), &d->host_hdr, sizeof(gpu::BatchHeader), hipMemcpyHostToDevice, stream));
            WF_HIP_CHECK(hipMemcpyAsync(d->data<tuple_t>(), host + off, cnt * sizeof(tuple_t), hipMemcpyHostToDevice, stream));
            gpu::Batch* out = engine.process(d, stream, kernels);
            d->release(stream);
            drain(out);
        }
        size_t cursor = 0;
        gpu::Batch* out = nullptr;
        while (engine.flush_range(cursor, stream, kernels, out)) drain(out);
        ctr = engine.read_counters(stream);
        dev_bytes = engine.device_bytes();
    };
    if (wave_reduce) {
        auto f = gpu::wave_reduce(wfbench::LiftValue{}, CombByMode{func_mode});
        gpu::WinBatchEngine<tuple_t, result_t, decltype(f)> engine(f, p, opt, batch_len, 0, 3, 0);
        run(engine);
    } else {
        gpu::WinBatchEngine<tuple_t, result_t, WinFuncByMode> engine(WinFuncByMode{func_mode}, p, opt, batch_len, 0, 3, 0);
        run(engine);
    }
    py::array_t<int64_t> arr({(py::ssize_t)results.size(), (py::ssize_t)4});
    if (!results.empty()) std::memcpy(arr.mutable_data(), results.data(), results.size() * sizeof(result_t));
    py::dict counters;
    counters["dropped"] = ctr.dropped;
    counters["deferred"] = ctr.deferred;
    counters["table_full"] = ctr.table_full;
    counters["carry_overflow"] = ctr.carry_overflow;
    counters["work_overflow"] = ctr.work_overflow;
    counters["kernels"] = kernels;
    counters["device_bytes"] = dev_bytes;
    return py::make_tuple(arr, counters);
}

// ---------------------------------------------------------------------------
// Yahoo! Streaming Benchmark on the device (BASELINE.json config 5), driven step by step:
//   S gated event sources -> Filter_GPU -> Map_GPU (join) -> [Exchange_GPU across ranks] -> device KeyBy
//     -> Key_FFAT_GPU (10 s tumbling windows per campaign) -> D2H -> batch sinks
// Events carry a synthetic clock (`period_us` between the events of a source), so a run is reproducible and the number
// of views every window must count is known in closed form.
// ---------------------------------------------------------------------------
class GatedEventSource {
public:
    GatedEventSource(wfbench::StepGate* gates, const ysb::Campaigns* c, uint64_t period_us) : gates_(gates), period_(period_us), n_ads_(c->n_ads()) {
        ads_ = std::make_shared<std::vector<uint64_t>>(n_ads_);
        for (size_t i = 0; i < n_ads_; i++) (*ads_)[i] = c->ad(i);
    }
    bool operator()(ysb::Event* buf, size_t cap, size_t& n, wf::RuntimeContext& ctx) {
        wfbench::StepGate& g = gates_[ctx.getReplicaIndex()];
        punct_ = 0;
        int64_t take = 0;
        {
            std::unique_lock<std::mutex> l(g.m);
            g.cv.wait(l, [&] { return g.quota > 0 || g.stop; });
            if (g.quota <= 0 && g.stop) {
                n = 0;
                return false;
            }
            take = g.quota < (int64_t)cap ? g.quota : (int64_t)cap;
            g.quota -= take;
            if (g.quota == 0) punct_ = g.step_id;
        }
        if (skew_) skew_->wait_turn(ctx.getReplicaIndex(), g);
        opened_ns_ = wf::current_time_nsecs();
        uint64_t v = value_;
        const uint64_t* ads = ads_->data();
        for (int64_t i = 0; i < take; i++, v++) {
            ysb::Event& e = buf[i];
            const uint64_t x = v % 100000;
            e.ts = v * period_;
            e.seq = v;
            e.ad_id = ads[x % n_ads_];
            e.user_id = 0;
            e.page_id = 0;
            e.ad_type = (uint32_t)(x % 5);
            e.event_type = (uint32_t)(x % 3);
            e.ip = 1;
        }
        value_ = v;
        n = (size_t)take;
        if (skew_) skew_->batch_done(ctx.getReplicaIndex());
        return true;
    }
    uint64_t punctuation() const { return punct_; }
    uint64_t opened_ns() const { return opened_ns_; }
    // every source emits every campaign: the sources of a process advance together (bench_common.hpp SkewBound)
    void bound_skew(std::shared_ptr<wfbench::SkewBound> s) { skew_ = std::move(s); }

private:
    std::shared_ptr<wfbench::SkewBound> skew_;
    wfbench::StepGate* gates_;
    uint64_t period_;
    size_t n_ads_;
    std::shared_ptr<std::vector<uint64_t>> ads_;
    uint64_t value_ = 0, punct_ = 0, opened_ns_ = 0;
};

struct YsbTotals {
    std::mutex m;
    uint64_t results = 0, views = 0, batches = 0;
};

class YsbPipeline {
public:
    // split_merge: the events are split on the device by ad type into two branches, each with its own Filter_GPU -> Map_GPU(join),
    // and the branches are merged again in front of the exchange / aggregation (BASELINE.json config 5)
    YsbPipeline(int gpu_id, size_t n_sources, size_t batch_len, uint64_t period_us, uint64_t delay_us, size_t inflight, size_t agg_replicas,
                std::shared_ptr<dist::Transport> transport = nullptr, bool split_merge = false, bool deterministic = false)
        : n_sources_(n_sources), n_sinks_(agg_replicas ? agg_replicas : 1), gates_(new wfbench::StepGate[n_sources]),
          graph_("ysb_gpu", deterministic ? Mode::DETERMINISTIC : Mode::DEFAULT) {
        auto totals = totals_ = std::make_shared<YsbTotals>();
        auto barrier = barrier_ = std::make_shared<wfbench::StepBarrier>();
        const auto us = [](uint64_t v) { return std::chrono::microseconds(v); };
        GatedEventSource events(gates_.get(), &campaigns_, period_us);
        events.bound_skew(make_skew_bound(transport, n_sources));
        source_ = BatchSource_Builder(events).withName("ysb_source").withParallelism(n_sources)
                      .withBatch(batch_len).withInFlightBatches(inflight).withGPUConfiguration(gpu_id).build_unique();
        MultiPipe* mp = &graph_.add_source(*source_);
        const size_t branches = split_merge ? 2 : 1;
        for (size_t b = 0; b < branches; b++) {
            filter_.push_back(FilterGPU_Builder(ysb::ViewFilterGPU()).withName("ysb_filter_gpu").withParallelism(n_sources).withInFlightBatches(inflight)
                                  .withGPUConfiguration(gpu_id).build_unique());
            join_.push_back(MapGPU_Builder(ysb::JoinGPU(tables_)).withName("ysb_join_gpu").withParallelism(n_sources).withInFlightBatches(inflight)
                                .withGPUConfiguration(gpu_id).build_unique());
        }
        if (split_merge) {
            GPUConfig scfg;
            scfg.gpu_id = gpu_id;
            scfg.pool_slabs = inflight;
            split_gpu(*mp, ysb::AdTypeSplit(), 2, scfg);
            MultiPipe& b0 = mp->select(0);
            b0.chain(*filter_[0]);
            b0.chain(*join_[0]);
            MultiPipe& b1 = mp->select(1);
            b1.chain(*filter_[1]);
            b1.chain(*join_[1]);
            mp = &b0.merge(b1);
        } else {
            mp->add(*filter_[0]);
            mp->chain(*join_[0]);
        }
        // the device KeyBy / the exchange merge the punctuations of a step into one per aggregation replica; a single replica
        // without an exchange in front is fed by every lane directly and sees the punctuation of each
        punct_expected_ = (transport || n_sinks_ > 1) ? n_sinks_ : n_sources * branches;
        if (transport) {
            exchange_ = ExchangeGPU_Builder<ysb::Joined>(transport).withName("exchange").withParallelism(n_sources).withLocalLanes(n_sources * branches)
                            .withOutputLanes(n_sinks_).withGPUConfiguration(gpu_id).withInFlightBatches(inflight).build_unique();
            mp->add(*exchange_);
        }
        agg_ = KeyFFATGPU_Builder(ysb::CountLift(), ysb::CountCombine()).withTBWindows(us(ysb::WINDOW_US), us(ysb::WINDOW_US), us(delay_us))
                   .withName("ysb_kff_gpu").withParallelism(n_sinks_).withGPUConfiguration(gpu_id).withInFlightBatches(inflight).withMaxKeys(256)
                   .withOutputCapacity(8192).withKeyDisjointLanes(transport != nullptr).build_unique();
        mp->add(*agg_);
        auto sink_f = [totals, barrier](const ysb::WinResult* data, size_t n, const gpu::BatchHeader& h) {
            uint64_t res = 0, views = 0;
            for (size_t i = 0; i < n; i++) {
                if (data[i].count == 0) continue;
                res++;
                views += data[i].count;
            }
            {
                std::lock_guard<std::mutex> l(totals->m);
                totals->results += res;
                totals->views += views;
                totals->batches += (data != nullptr);
            }
            if (h.reserved[0] != 0) barrier->arrive(h.reserved[0]);
        };
        sink_ = BatchSink_Builder(sink_f).withName("ysb_sink").withParallelism(n_sinks_).build_unique();
        mp->chain_sink(*sink_);
        latency_ = sink_->latency();
        graph_.start();
    }
    ~YsbPipeline() { close(); }
    void step(int64_t events_per_source) { steps(1, events_per_source); }
    void steps(int64_t k, int64_t events_per_source) {
        py::gil_scoped_release rel;
        step_id_ += (uint64_t)k;
        for (size_t i = 0; i < n_sources_; i++) gates_[i].release(step_id_, k * events_per_source);
        barrier_->wait(step_id_, (int)punct_expected_);
    }
    void close() {
        if (closed_) return;
        closed_ = true;
        py::gil_scoped_release rel;
        for (size_t i = 0; i < n_sources_; i++) gates_[i].shutdown();
        graph_.wait_end();
    }
    py::dict stats() {
        py::dict d;
        {
            std::lock_guard<std::mutex> l(totals_->m);
            d["sink_tuples"] = totals_->results;
            d["counted_views"] = totals_->views;
            d["sink_batches"] = totals_->batches;
        }
        d["latency_p50_us"] = latency_->percentile_us(50);
        d["latency_p99_us"] = latency_->percentile_us(99);
        d["latency_mean_us"] = latency_->mean_us();
        d["latency_samples"] = latency_->count();
        d["threads"] = graph_.getNumThreads();
        uint64_t dropped = 0, overflow = 0, full = 0;
        for (auto& c : agg_->getDeviceCounters()) {
            dropped += c.dropped;
            overflow += c.overflow;
            full += c.table_full;
        }
        d["dropped"] = dropped;
        d["overflow"] = overflow;
        d["table_full"] = full;
        if (exchange_) exchange_stats(d, exchange_->stats(), exchange_->getOutputLanes());
        return d;
    }
    void reset_latency() { latency_->clear(); }
    std::string dot() { return graph_.get_DOTGraph(); }

private:
    size_t n_sources_, n_sinks_, punct_expected_ = 0;
    ysb::Campaigns campaigns_;
    ysb::DeviceCampaignTables tables_{campaigns_};  // the join's hash table + record table on the device
    std::unique_ptr<wfbench::StepGate[]> gates_;
    PipeGraph graph_;
    std::shared_ptr<YsbTotals> totals_;
    std::shared_ptr<wfbench::StepBarrier> barrier_;
    std::shared_ptr<LatencyRecorder> latency_;
    std::unique_ptr<Batch_Source<ysb::Event>> source_;
    std::vector<std::unique_ptr<Filter_GPU<ysb::Event>>> filter_;
    std::vector<std::unique_ptr<Map_GPU<ysb::Event, ysb::Joined>>> join_;
    std::unique_ptr<Exchange_GPU<ysb::Joined>> exchange_;
    std::unique_ptr<Key_FFAT_GPU<ysb::Joined, ysb::WinResult>> agg_;
    std::unique_ptr<Batch_Sink<ysb::WinResult>> sink_;
    uint64_t step_id_ = 0;
    bool closed_ = false;
};

// ---------------------------------------------------------------------------
// distributed layer
// ---------------------------------------------------------------------------
static py::bytes rccl_create_id() { return py::bytes(dist::RcclTransport::create_id()); }

static std::shared_ptr<dist::Transport> make_rccl_transport(int rank, int world, const std::string& id, int device) {
    py::gil_scoped_release rel;
    return std::make_shared<dist::RcclTransport>(rank, world, id, device);
}

// sum of a device buffer of doubles across ranks (in place), synchronous
static void transport_all_reduce_sum(dist::Transport& t, uintptr_t dev, size_t n) {
    check_ptr(dev, "dev");
    py::gil_scoped_release rel;
    gpu::Stream s;
    t.all_reduce_sum(reinterpret_cast<double*>(dev), n, s);
    s.sync();
}

static std::vector<uint64_t> transport_all_gather(dist::Transport& t, const std::vector<uint64_t>& mine) {
    std::vector<uint64_t> out(mine.size() * (size_t)t.world());
    py::gil_scoped_release rel;
    gpu::Stream s;
    t.all_gather_host(mine.data(), out.data(), mine.size(), s);
    return out;
}

// Exchange_GPU conservation check: `world` in-process ranks (threads) on one device, each runs
//   Batch_Source x n_sources (n tuples each, keys of their own) -> Exchange_GPU(LocalTransport, `lanes` output lanes) -> Batch_Sink x lanes
// returns per rank {tuples received, value sum, tuples not owned by the (rank, lane) that saw them, per-key order violations,
// rounds, input batches, batches delivered}
static std::vector<std::vector<double>> exchange_local_run(int gpu_id, int world, size_t n_per_rank, size_t n_keys, size_t batch_len, size_t n_sources,
                                                           size_t lanes, size_t max_batches_per_round) {
    py::gil_scoped_release rel;
    auto fabric = std::make_shared<dist::LocalFabric>(world);
    std::vector<std::vector<double>> res((size_t)world, std::vector<double>(7, 0.0));
    std::vector<std::thread> threads;
    for (int r = 0; r < world; r++) {
        threads.emplace_back([&, r] {
            auto tr = std::make_shared<dist::LocalTransport>(fabric, r);
            PipeGraph g("exchange_rank" + std::to_string(r), Mode::DEFAULT);
            // rank r produces (r+1)/world of n_per_rank tuples per source: uneven streams exercise the end-of-stream rounds
            const size_t n = std::max<size_t>(1, n_per_rank * (size_t)(r + 1) / (size_t)world);
            const size_t S = n_sources;
            auto gen = [=, sent = (size_t)0](tuple_t& t, RuntimeContext& rc) mutable {
                // keys of different sources and ranks are disjoint: per-key order is the order of one producer
                t.key = ((sent % n_keys) * S + rc.getReplicaIndex()) * (size_t)world + (size_t)r;
                t.id = sent / n_keys;
                t.ts = sent;
                t.value = (double)(sent % 97);
                t.aux = (double)r;
                sent++;
                return sent < n;
            };
            auto src = BatchSource_Builder(gen).withName("src").withParallelism(S).withBatch(batch_len).withGPUConfiguration(gpu_id).build_unique();
            MultiPipe& mp = g.add_source(*src);
            auto ex = ExchangeGPU_Builder<tuple_t>(tr).withParallelism(S).withLocalLanes(S).withOutputLanes(lanes).withMaxBatchesPerRound(max_batches_per_round)
                          .withGPUConfiguration(gpu_id).build_unique();
            mp.add(*ex);
            auto m = std::make_shared<std::mutex>();
            std::vector<double>* out = &res[(size_t)r];
            const uint32_t W = (uint32_t)world, L = (uint32_t)lanes;
            auto sink_f = [out, m, W, L, r, last_id = std::map<uint64_t, uint64_t>()](const tuple_t* data, size_t cnt, RuntimeContext& rc) mutable {
                double n_in = 0, sum = 0, foreign = 0, disorder = 0;
                for (size_t i = 0; i < cnt; i++) {
                    n_in += 1;
                    sum += data[i].value;
                    if (gpu::keyby_destination(data[i].key, W, gpu::HASH_SHIFT_RANKS) != (uint32_t)r ||
                        gpu::keyby_destination(data[i].key, L, gpu::HASH_SHIFT_LANES) != (uint32_t)rc.getReplicaIndex())
                        foreign += 1;
                    auto it = last_id.find(data[i].key);
                    if (it != last_id.end() && data[i].id <= it->second) disorder += 1;
                    last_id[data[i].key] = data[i].id;
                }
                std::lock_guard<std::mutex> l(*m);
                (*out)[0] += n_in;
                (*out)[1] += sum;
                (*out)[2] += foreign;
                (*out)[3] += disorder;
            };
            auto sink = BatchSink_Builder(sink_f).withName("sink").withParallelism(lanes).build_unique();
            mp.add_sink(*sink);
            g.run();
            const ExchangeStats x = ex->stats();
            (*out)[4] = (double)x.rounds;
            (*out)[5] = (double)x.bundles;
            (*out)[6] = (double)x.lane_batches;
        });
    }
    for (auto& t : threads) t.join();
    return res;
}

PYBIND11_MODULE(_C, m) {
    m.doc() = "windflow_amd native extension (gfx950)";
    m.def("device_count", &gpu::device_count);
    m.def("device_pci_bus_id", [](int device) {
        char buf[64] = {0};
        if (hipDeviceGetPCIBusId(buf, (int)sizeof(buf), device) != hipSuccess) {
            (void)hipGetLastError();  // a probe of a device that does not exist must not leave a sticky error behind
            return std::string();
        }
        return std::string(buf);
    }, py::arg("device"), "PCI address (domain:bus:device.function) of a GPU, without creating a context on it");
    m.def("tuple_size", []() { return sizeof(tuple_t); });
    m.def("map_scale_inplace", &map_scale_inplace, py::arg("tuples"), py::arg("n"), py::arg("factor"));
    m.def("map_scale_out_of_place", &map_scale_out_of_place, py::arg("tuples"), py::arg("n"), py::arg("results"), py::arg("factor"));
    py::class_<FlatFatSession>(m, "FlatFatGPU")
        .def(py::init<size_t, size_t, int>(), py::arg("capacity"), py::arg("max_windows"), py::arg("comb_mode") = 0)
        .def_property_readonly("capacity", &FlatFatSession::capacity)
        .def_property_readonly("size", &FlatFatSession::size)
        .def_property_readonly("kernels", &FlatFatSession::kernels)
        .def("build", &FlatFatSession::build)
        .def("update", &FlatFatSession::update, py::arg("values"), py::arg("n_remove"))
        .def("results", &FlatFatSession::results, py::arg("n_windows"), py::arg("win"), py::arg("slide"));
    m.def("filter_compact", &filter_compact, py::arg("inp"), py::arg("n"), py::arg("out"), py::arg("mode"), py::arg("threshold"));
    m.def("key_ffat_run", &key_ffat_run, py::arg("tuples"), py::arg("win"), py::arg("slide"), py::arg("delay"), py::arg("time_based"),
          py::arg("max_keys"), py::arg("batch_len"), py::arg("out_capacity") = 0, py::arg("comb_mode") = 0, py::arg("skip_gaps") = false);
    m.def("radix_sort_pairs", &radix_sort_pairs, py::arg("keys"), py::arg("vals"), py::arg("n"), py::arg("bits"), py::arg("mode") = 0);
    m.def("exclusive_scan", &exclusive_scan, py::arg("inp"), py::arg("out"), py::arg("n"));
    py::class_<dist::Transport, std::shared_ptr<dist::Transport>>(m, "Transport")
        .def_property_readonly("rank", &dist::Transport::rank)
        .def_property_readonly("world", &dist::Transport::world)
        .def_property_readonly("name", [](dist::Transport& t) { return std::string(t.name()); })
        .def("all_reduce_sum", &transport_all_reduce_sum, py::arg("dev_ptr"), py::arg("n"))
        .def("all_gather", &transport_all_gather, py::arg("words"))
        .def("barrier", [](dist::Transport& t) {
            py::gil_scoped_release rel;
            gpu::Stream s;
            t.barrier(s);
        });
    py::class_<dist::LocalFabric, std::shared_ptr<dist::LocalFabric>>(m, "LocalFabric").def(py::init<int>(), py::arg("world"));
    m.def("rccl_create_id", &rccl_create_id);
    m.def("rccl_transport", &make_rccl_transport, py::arg("rank"), py::arg("world"), py::arg("unique_id"), py::arg("device"));
    m.def("ipc_transport", [](int rank, int world, const std::string& token, int device) -> std::shared_ptr<dist::Transport> {
        py::gil_scoped_release rel;
        return std::make_shared<dist::IpcTransport>(rank, world, token, device);
    }, py::arg("rank"), py::arg("world"), py::arg("token"), py::arg("device"));
    m.def("local_transport", [](std::shared_ptr<dist::LocalFabric> f, int rank) -> std::shared_ptr<dist::Transport> {
        return std::make_shared<dist::LocalTransport>(std::move(f), rank);
    }, py::arg("fabric"), py::arg("rank"));
    m.def("exchange_local_run", &exchange_local_run, py::arg("gpu_id"), py::arg("world"), py::arg("n_per_rank"), py::arg("n_keys"),
          py::arg("batch_len"), py::arg("n_sources") = 1, py::arg("lanes") = 1, py::arg("max_batches_per_round") = 0);
    py::class_<WinAggPipeline>(m, "WinAggPipeline")
        .def(py::init<int, size_t, size_t, size_t, uint64_t, uint64_t, bool, uint64_t, size_t, std::shared_ptr<dist::Transport>, size_t, int, double,
                      bool, bool, bool, bool>(),
             py::arg("gpu_id"), py::arg("n_sources"), py::arg("batch_len"), py::arg("n_keys"), py::arg("win"), py::arg("slide"),
             py::arg("time_based") = false, py::arg("delay") = 0, py::arg("inflight") = 6, py::arg("transport") = nullptr,
             py::arg("agg_replicas") = 0, py::arg("key_dist") = 0, py::arg("zipf_alpha") = 1.3, py::arg("shared_keys") = false,
             py::arg("deterministic") = false, py::arg("device_source") = false, py::arg("combiner") = false)
        .def("step", &WinAggPipeline::step)
        .def("steps", &WinAggPipeline::steps)
        .def("close", &WinAggPipeline::close)
        .def("stats", &WinAggPipeline::stats)
        .def("reset_latency", &WinAggPipeline::reset_latency)
        .def("dot", &WinAggPipeline::dot);
    m.def("win_batch_run", &win_batch_run, py::arg("tuples"), py::arg("win"), py::arg("slide"), py::arg("delay"), py::arg("time_based"),
          py::arg("max_keys"), py::arg("batch_len"), py::arg("carry_capacity") = 0, py::arg("out_capacity") = 0, py::arg("func_mode") = 0,
          py::arg("wave_reduce") = false, py::arg("renumber") = false);
    py::class_<WinSeqPipeline>(m, "WinSeqPipeline")
        .def(py::init<int, size_t, size_t, size_t, uint64_t, uint64_t, bool, size_t, size_t, bool, int, double, bool>(), py::arg("gpu_id"),
             py::arg("n_sources"), py::arg("batch_len"), py::arg("n_keys"), py::arg("win"), py::arg("slide"), py::arg("time_based") = false,
             py::arg("inflight") = 4, py::arg("replicas") = 0, py::arg("wave_reduce") = false, py::arg("key_dist") = 0, py::arg("zipf_alpha") = 1.3,
             py::arg("shared_keys") = false)
        .def("step", &WinSeqPipeline::step)
        .def("steps", &WinSeqPipeline::steps)
        .def("close", &WinSeqPipeline::close)
        .def("stats", &WinSeqPipeline::stats)
        .def("reset_latency", &WinSeqPipeline::reset_latency)
        .def("dot", &WinSeqPipeline::dot);
    py::class_<YsbPipeline>(m, "YsbPipeline")
        .def(py::init<int, size_t, size_t, uint64_t, uint64_t, size_t, size_t, std::shared_ptr<dist::Transport>, bool, bool>(), py::arg("gpu_id"),
             py::arg("n_sources"), py::arg("batch_len"), py::arg("period_us") = 1, py::arg("delay_us") = 1000000, py::arg("inflight") = 4,
             py::arg("agg_replicas") = 1, py::arg("transport") = nullptr, py::arg("split_merge") = false, py::arg("deterministic") = false)
        .def("step", &YsbPipeline::step)
        .def("steps", &YsbPipeline::steps)
        .def("close", &YsbPipeline::close)
        .def("stats", &YsbPipeline::stats)
        .def("reset_latency", &YsbPipeline::reset_latency)
        .def("dot", &YsbPipeline::dot);
    py::class_<MapFilterPipeline>(m, "MapFilterPipeline")
        .def(py::init<int, size_t, size_t, size_t, size_t>(), py::arg("gpu_id"), py::arg("n_sources"), py::arg("batch_len"), py::arg("n_keys"),
             py::arg("inflight") = 6)
        .def("step", &MapFilterPipeline::step)
        .def("steps", &MapFilterPipeline::steps)
        .def("close", &MapFilterPipeline::close)
        .def("stats", &MapFilterPipeline::stats)
        .def("reset_latency", &MapFilterPipeline::reset_latency)
        .def("dot", &MapFilterPipeline::dot);
}
