// Throughput of the window operators fed tuple by tuple (the reference's programming model for its GPU window
// operators: wf/win_seq_gpu.hpp batches fired windows, the tuples reach the operator one at a time from CPU
// operators).  The reference publishes no numbers for them; this program gives ours, CPU and GPU
// flavours side by side, on count-based sliding windows with a window sum.
//
//   bench_win_gpu -o kf|kf_gpu|wf_gpu|pf_gpu|wmr_gpu|pf_gg|wmr_gg|kff|kff_gpu [-s sources] [-n replicas] [-k keys] [-w win] [-l slide]
//                 [-b tuples per GPU batch] [-t seconds] [-B] [-D [-d delay]]
// -D runs the tuple-fed graph in DEFAULT mode with TIME-based windows of the same geometry (a key ticks once per round of its
// source; the sources are unsynchronised threads, so the windows get a triggering delay, -d, default 4 windows).
// -B feeds the flat GPU operators (kf_gpu, wf_gpu, kff_gpu) with batches instead of tuples: Batch_Source lanes that own
// disjoint key sets -> operator -> Batch_Sink, the stream never leaves pinned / device memory.
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <iostream>
#include <string>

#include <wf/windflow_gpu.hpp>

using namespace wf;

struct tuple_t {
    size_t key = 0;
    uint64_t id = 0;
    uint64_t ts = 0;
    int64_t value = 0;
    WF_HOST_DEVICE tuple_t() {}
    WF_HOST_DEVICE std::tuple<size_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    WF_HOST_DEVICE void setControlFields(size_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};
struct result_t {
    size_t key = 0;
    uint64_t id = 0;
    uint64_t ts = 0;
    int64_t value = 0;
    WF_HOST_DEVICE result_t() {}
    WF_HOST_DEVICE std::tuple<size_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    WF_HOST_DEVICE void setControlFields(size_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};

// window sum in every flavour the operators accept
struct SumGPU {
    __host__ __device__ void operator()(uint64_t, const tuple_t* data, size_t n, result_t* res, char*, size_t) const {
        int64_t sum = 0;
        for (size_t i = 0; i < n; i++) sum += data[i].value;
        res->value = sum;
    }
};
struct Sum2GPU {
    __host__ __device__ void operator()(uint64_t, const result_t* data, size_t n, result_t* res, char*, size_t) const {
        int64_t sum = 0;
        for (size_t i = 0; i < n; i++) sum += data[i].value;
        res->value = sum;
    }
};
static void sum_nic(uint64_t, const Iterable<tuple_t>& win, result_t& res) {
    int64_t sum = 0;
    for (const auto& t : win) sum += t.value;
    res.value = sum;
}
static void sum2_nic(uint64_t, const Iterable<result_t>& win, result_t& res) {
    int64_t sum = 0;
    for (const auto& t : win) sum += t.value;
    res.value = sum;
}
static void lift_cpu(const tuple_t& t, result_t& r) { r.value = t.value; }
static void comb_cpu(const result_t& a, const result_t& b, result_t& out) { out.value = a.value + b.value; }
struct LiftGPU {
    __host__ __device__ void operator()(const tuple_t& t, result_t& r) const { r.value = t.value; }
};
struct CombGPU {
    __host__ __device__ void operator()(const result_t& a, const result_t& b, result_t& out) const { out.value = a.value + b.value; }
};

// every source emits every key round-robin with progressive ids per key, and stops after `seconds`.  (Sources with
// disjoint key sets would be a trap in DETERMINISTIC mode: the ordering nodes release a key's tuples up to the
// smallest id seen on EVERY input channel, as in the reference, so a key that only one source produces is held
// back until the end of the stream.)
struct TimedSource {
    size_t n_keys, n_sources;
    uint64_t stop_at_us;
    std::atomic<uint64_t>* generated;
    uint64_t seq = 0;
    bool operator()(tuple_t& t, RuntimeContext&) {
        const uint64_t s = seq++;
        t.key = s % n_keys;
        t.id = s / n_keys;
        t.ts = t.id;
        t.value = (int64_t)(s & 1023);
        if ((s & 0xfff) == 0xfff) {
            generated->fetch_add(0x1000, std::memory_order_relaxed);
            return current_time_usecs() < stop_at_us;
        }
        return true;
    }
};

// -B: every source owns `n_keys` keys (key % n_sources == source), round-robin with progressive ids, filled a batch at a time
struct TimedBatchSource {
    size_t n_keys, n_sources;
    uint64_t stop_at_us;
    std::atomic<uint64_t>* generated;
    uint64_t seq = 0;
    bool operator()(tuple_t* buf, size_t cap, size_t& n, RuntimeContext& ctx) {
        const uint64_t src = ctx.getReplicaIndex();
        uint64_t s = seq;
        for (size_t i = 0; i < cap; i++, s++) {
            buf[i].key = (s % n_keys) * n_sources + src;
            buf[i].id = s / n_keys;
            buf[i].ts = buf[i].id;
            buf[i].value = (int64_t)(s & 1023);
        }
        seq = s;
        n = cap;
        generated->fetch_add(cap, std::memory_order_relaxed);
        return current_time_usecs() < stop_at_us;
    }
};

int main(int argc, char** argv) {
    std::string op = "kf_gpu";
    bool batch_fed = false, default_mode = false;
    uint64_t delay = 0;
    size_t sources = 2, replicas = 4, keys = 64, batch = 4096;
    uint64_t win = 1024, slide = 128;
    double seconds = 3.0;
    int c;
    while ((c = getopt(argc, argv, "o:s:n:k:w:l:b:t:BDd:")) != -1) {
        switch (c) {
            case 'o': op = optarg; break;
            case 's': sources = std::strtoull(optarg, nullptr, 10); break;
            case 'n': replicas = std::strtoull(optarg, nullptr, 10); break;
            case 'k': keys = std::strtoull(optarg, nullptr, 10); break;
            case 'w': win = std::strtoull(optarg, nullptr, 10); break;
            case 'l': slide = std::strtoull(optarg, nullptr, 10); break;
            case 'b': batch = std::strtoull(optarg, nullptr, 10); break;
            case 't': seconds = std::atof(optarg); break;
            case 'B': batch_fed = true; break;
            case 'D': default_mode = true; break;
            case 'd': delay = std::strtoull(optarg, nullptr, 10); break;
            default:
                std::cout << argv[0] << " -o kf|kf_gpu|wf_gpu|pf_gpu|wmr_gpu|pf_gg|wmr_gg|kff|kff_gpu [-s sources] [-n replicas] [-k keys] [-w win] [-l slide] "
                          << "[-b batch] [-t seconds]" << std::endl;
                return 0;
        }
    }
    const bool needs_gpu = op.find("_gpu") != std::string::npos;
    if (needs_gpu && gpu::device_count() < 1) {
        std::cerr << "bench_win_gpu: " << op << " needs a GPU" << std::endl;
        return 3;
    }
    std::atomic<uint64_t> generated{0}, results{0};
    if (batch_fed) {
        if (op != "kf_gpu" && op != "wf_gpu" && op != "kff_gpu" && op != "pf_gpu" && op != "wmr_gpu" && op != "pf_hy" && op != "wmr_hy") {
            std::cerr << "-B needs a GPU operator with a batch path (kf_gpu, wf_gpu, kff_gpu, pf_gpu, wmr_gpu, pf_hy, wmr_hy)" << std::endl;
            return 2;
        }
        const size_t keys_per_source = std::max<size_t>(1, keys / sources);
        if (batch < 4096) batch = 1 << 17;
        PipeGraph graph("bench_" + op + "_batches", Mode::DEFAULT);
        const uint64_t t0 = current_time_usecs();
        TimedBatchSource gen{keys_per_source, sources, t0 + (uint64_t)(seconds * 1e6), &generated};
        auto src = BatchSource_Builder(gen).withName("source").withParallelism(sources).withBatch(batch).build_unique();
        MultiPipe& mp = graph.add_source(*src);
        const bool disjoint = replicas == sources;
        const size_t max_keys = keys_per_source * sources / (disjoint ? sources : 1) * 5 / 4 + 64;
        const size_t out_cap = batch / slide + max_keys + 8192;
        std::unique_ptr<Basic_Operator> oper;
        if (op == "kf_gpu")
            oper = KeyFarmGPU_Builder(SumGPU()).withCBWindows(win, slide).withName(op).withParallelism(replicas).withMaxKeys(max_keys)
                       .withOutputCapacity(out_cap).withKeyDisjointLanes(disjoint).build_unique();
        else if (op == "wf_gpu")
            oper = WinFarmGPU_Builder(SumGPU()).withCBWindows(win, slide).withName(op).withParallelism(replicas).withMaxKeys(max_keys)
                       .withOutputCapacity(out_cap).withKeyDisjointLanes(disjoint).build_unique();
        else if (op == "pf_gpu")  // both stages on the device: PLQ -> WLQ inside every replica
            oper = PaneFarmGPU_Builder(SumGPU(), Sum2GPU()).withCBWindows(win, slide).withName(op).withParallelism(replicas, replicas).withMaxKeys(max_keys)
                       .withOutputCapacity(out_cap).withKeyDisjointLanes(disjoint).build_unique();
        else if (op == "wmr_gpu")  // every window cut into 4 stripes: map per stripe, reduce over the 4 partial results
            oper = WinMapReduceGPU_Builder(SumGPU(), Sum2GPU()).withCBWindows(win, slide).withName(op).withParallelism(4, replicas).withMaxKeys(max_keys)
                       .withOutputCapacity(out_cap).withKeyDisjointLanes(disjoint).build_unique();
        else if (op == "pf_hy")  // the reference's form, fed with batches: PLQ on the device, WLQ on the host (results leave as tuples)
            oper = PaneFarmGPU_Builder(SumGPU(), sum2_nic).withCBWindows(win, slide).withName(op).withParallelism(replicas, 2).withMaxKeys(max_keys)
                       .withKeyDisjointLanes(disjoint).build_unique();
        else if (op == "wmr_hy")  // MAP on the device (4 stripes), REDUCE on the host
            oper = WinMapReduceGPU_Builder(SumGPU(), sum2_nic).withCBWindows(win, slide).withName(op).withParallelism(4, replicas).withMaxKeys(max_keys)
                       .withKeyDisjointLanes(disjoint).build_unique();
        else
            oper = KeyFFATGPU_Builder(LiftGPU(), CombGPU()).withCBWindows(win, slide).withName(op).withParallelism(replicas).withMaxKeys(max_keys)
                       .withOutputCapacity(out_cap).withKeyDisjointLanes(disjoint).build_unique();
        mp.add(*oper);
        std::unique_ptr<Basic_Operator> sink;
        if (op == "pf_hy" || op == "wmr_hy")
            sink = Sink_Builder([&results](std::optional<result_t>& r) {
                       if (r) results.fetch_add(1, std::memory_order_relaxed);
                   }).withName("sink").withParallelism(1).build_unique();
        else
            sink = BatchSink_Builder([&results](const result_t*, size_t n) { results.fetch_add(n, std::memory_order_relaxed); })
                       .withName("sink").withParallelism(replicas).build_unique();
        if (op == "pf_hy" || op == "wmr_hy")
            mp.add_sink(*sink);
        else
            mp.chain_sink(*sink);
        graph.run();
        const double elapsed = (current_time_usecs() - t0) / 1e6;
        std::cout << op << " [batch-fed] sources=" << sources << " replicas=" << replicas << " keys=" << keys_per_source * sources << " win=" << win
                  << " slide=" << slide << " batch=" << batch << " : " << (double)generated.load() / elapsed << " tuples/s ("
                  << (double)results.load() / elapsed << " windows/s), elapsed " << elapsed << " s" << std::endl;
        return 0;
    }
    const size_t keys_per_source = std::max<size_t>(1, keys);  // every source produces all the keys
    PipeGraph graph("bench_" + op, default_mode ? Mode::DEFAULT : Mode::DETERMINISTIC);  // count-based windows need ordered input
    if (default_mode && delay == 0) delay = 4 * win;
    using us = std::chrono::microseconds;
    // time-based windows keep what the open windows (and the triggering delay) still need: every source ticks every key once per time unit
    const size_t archive = default_mode ? (size_t)(2 * (win + delay) * sources + 2 * batch / std::max<size_t>(1, keys) + 1024) : 0;
    auto windows = [&](auto b) {
        if (default_mode)
            b.withTBWindows(us(win), us(slide), us(delay));
        else
            b.withCBWindows(win, slide);
        return b;
    };
    const uint64_t t0 = current_time_usecs();
    TimedSource gen{keys_per_source, sources, t0 + (uint64_t)(seconds * 1e6), &generated};
    auto src = Source_Builder(gen).withName("source").withParallelism(sources).build_unique();
    MultiPipe& mp = graph.add_source(*src);
    std::unique_ptr<Basic_Operator> oper;
    if (op == "kf") {
        oper = windows(KeyFarm_Builder(sum_nic)).withName(op).withParallelism(replicas).build_unique();
    } else if (op == "kf_gpu") {
        oper = windows(KeyFarmGPU_Builder(SumGPU())).withName(op).withParallelism(replicas).withBatch(batch).withMaxKeys(keys_per_source + 64)
                   .withArchiveCapacity(archive).build_unique();
    } else if (op == "wf_gpu") {
        oper = windows(WinFarmGPU_Builder(SumGPU())).withName(op).withParallelism(replicas).withBatch(batch).withMaxKeys(keys_per_source + 64)
                   .withArchiveCapacity(archive).build_unique();
    } else if (op == "pf_gpu") {
        oper = windows(PaneFarmGPU_Builder(SumGPU(), sum2_nic)).withName(op).withParallelism(replicas, 1).withBatch(batch)
                   .withMaxKeys(keys_per_source + 64).withArchiveCapacity(archive).build_unique();
    } else if (op == "wmr_gpu") {
        oper = windows(WinMapReduceGPU_Builder(SumGPU(), sum2_nic)).withName(op).withParallelism(std::max<size_t>(2, replicas), 1)
                   .withBatch(batch).withMaxKeys(keys_per_source + 64).withArchiveCapacity(archive).build_unique();
    } else if (op == "pf_gg") {  // both stages on the device: the producers fill batches, PLQ -> WLQ inside every replica
        oper = windows(PaneFarmGPU_Builder(SumGPU(), Sum2GPU())).withName(op).withParallelism(replicas, replicas).withBatch(batch)
                   .withMaxKeys(keys_per_source + 64).withArchiveCapacity(archive).build_unique();
    } else if (op == "wmr_gg") {
        oper = windows(WinMapReduceGPU_Builder(SumGPU(), Sum2GPU())).withName(op).withParallelism(4, replicas).withBatch(batch)
                   .withMaxKeys(keys_per_source + 64).withArchiveCapacity(archive).build_unique();
    } else if (op == "kff") {
        oper = windows(KeyFFAT_Builder(lift_cpu, comb_cpu)).withName(op).withParallelism(replicas).build_unique();
    } else if (op == "kff_gpu") {
        oper = windows(KeyFFATGPU_Builder(LiftGPU(), CombGPU())).withName(op).withParallelism(replicas).withBatch(64 * batch)
                   .withMaxKeys(keys_per_source + 64).withOutputCapacity(1 << 18).build_unique();
    } else {
        std::cerr << "unknown operator " << op << std::endl;
        return 2;
    }
    mp.add(*oper);
    auto sink = Sink_Builder([&results](std::optional<result_t>& r) {
                    if (r) results.fetch_add(1, std::memory_order_relaxed);
                }).withName("sink").withParallelism(1).build_unique();
    mp.add_sink(*sink);
    graph.run();
    const double elapsed = (current_time_usecs() - t0) / 1e6;
    std::cout << op << " sources=" << sources << " replicas=" << replicas << " keys=" << keys_per_source << " win=" << win << " slide=" << slide
              << " batch=" << batch << (default_mode ? " mode=DEFAULT/TB" : " mode=DETERMINISTIC/CB") << " : " << (double)generated.load() / elapsed << " tuples/s (" << (double)results.load() / elapsed
              << " windows/s), elapsed " << elapsed << " s" << std::endl;
    return 0;
}
