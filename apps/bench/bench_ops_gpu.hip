// Throughput benchmarks of the batched GPU operators, mirroring the reference's
// operator studies (/root/reference/src/GPU_Tests/new_tests/benchmarks/
// gpu_map_stateless.cpp, gpu_filter_stateless.cpp, gpu_map_stateful.cpp,
// gpu_filter_stateful.cpp and their run_tests.py sweeps over -s -k -b -n; the
// published numbers are in BASELINE.md: 147.5 M tuples/s stateless map,
// 118.6 M tuples/s stateful map) but built from the library's operators:
//
//   Batch_Source x S  ->  <operator> x N  ->  Batch_Sink x N
//
//   -o map_stateless     in-place map doing a lookup in a device hash table
//   -o filter_stateless  predicate + on-device stable compaction
//   -o map_stateful      keyed map: per-key window of the last 1000 values, incremental average (8 KB of state per key)
//   -o map_sum           keyed map with the cheapest state there is (a running sum per key)
//   -o map_scan          the same running sum in the associative form keyed_scan(lift, combine, apply): a wave folds 64 tuples of a key per step
//   -o filter_stateful   keyed filter on the same state: keeps tuples above the running average
//   -o keyby             device KeyBy only (scattering study): S lanes re-partitioned into N
//
//   bench_ops_gpu -o <op> [-s sources] [-n operator replicas] [-k keys] [-b batch length] [-t seconds] [-i batches in flight per source (6)]
//
// prints one line:  <op> sources=S replicas=N keys=K batch=B : X tuples/s (Y results/s), elapsed Z s
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <unistd.h>

#include <atomic>
#include <iostream>
#include <string>

#include <wf/windflow_gpu.hpp>

#include "../../windflow_amd/csrc/bench_common.hpp"

using namespace wf;
using wfbench::tuple_t;

// ---- stateless functors ---------------------------------------------------------------------------
struct TableEntry {
    unsigned long long key, value;
};
struct LookupMap {
    const TableEntry* table;
    uint32_t mask;
    __host__ __device__ void operator()(tuple_t& t) const {
        uint32_t h = (uint32_t)(t.key * 0x9E3779B97F4A7C15ull >> 40) & mask;
        for (uint32_t probes = 0; probes <= mask; probes++) {
            if (table[h].key == t.key) {
                t.id = table[h].value;
                return;
            }
            if (table[h].key == ~0ull) break;
            h = (h + 1) & mask;
        }
        t.id = 0;
    }
};
struct HalfFilter {
    __host__ __device__ bool operator()(const tuple_t& t) const { return ((uint64_t)t.value & 1) == 0; }
};

// ---- stateful functors: the reference's Window_State (gpu_map_stateful.cpp:371-403) ------------------
struct WindowState {
    static constexpr uint32_t CAP = 1000;
    double values[CAP];
    double sum;
    uint32_t first, last, count, pad;
    // the header is read once and written once per tuple (independent loads, one round trip), the
    // ring buffer touched at most twice: the function is latency-bound, one tuple of a key after the other
    __host__ __device__ double push(double v) {
        uint32_t f = first, l = last, c = count;
        double s = sum;
        if (c == CAP) {
            s -= values[f];
            f = f + 1 == CAP ? 0 : f + 1;
        } else {
            c++;
        }
        values[l] = v;
        l = l + 1 == CAP ? 0 : l + 1;
        s += v;
        first = f;
        last = l;
        count = c;
        sum = s;
        return s / c;
    }
};
struct RunningAverageMap {
    __host__ __device__ void operator()(tuple_t& t, char* state, size_t) const { t.aux = reinterpret_cast<WindowState*>(state)->push(t.value); }
};
// the cheapest stateful function there is (a running sum): what is left is the cost of the keyed machinery itself
struct RunningSumMap {
    __host__ __device__ void operator()(tuple_t& t, char* state, size_t) const {
        double& s = *reinterpret_cast<double*>(state);
        s += t.value;
        t.aux = s;
    }
};
// the same running sum in the associative form: keyed_scan(lift, combine, apply)
struct SumLift {
    __host__ __device__ void operator()(const tuple_t& t, double& s) const { s = t.value; }
};
struct SumCombine {
    __host__ __device__ void operator()(const double& a, const double& b, double& out) const { out = a + b; }
};
struct SumApply {
    __host__ __device__ void operator()(tuple_t& t, const double& s) const { t.aux = s; }
};
struct AboveAverageFilter {
    __host__ __device__ void operator()(const tuple_t& t, bool& keep, char* state, size_t) const {
        keep = t.value >= reinterpret_cast<WindowState*>(state)->push(t.value);
    }
};

// endless generator: keys round-robin inside a per-source key range, stops after `seconds`
struct TimedGenerator {
    size_t n_keys, n_sources;
    uint64_t stop_at_us;
    uint64_t seq = 0;
    std::atomic<uint64_t>* generated;
    bool operator()(tuple_t* buf, size_t cap, size_t& n, RuntimeContext& ctx) {
        const uint64_t src = ctx.getReplicaIndex();
        uint64_t s = seq;
        for (size_t i = 0; i < cap; i++, s++) {
            buf[i].key = (s % n_keys) * n_sources + src;
            buf[i].id = s / n_keys;
            buf[i].ts = s;
            buf[i].value = (double)((s * 2654435761ull >> 7) & 1023);
            buf[i].aux = 0.0;
        }
        seq = s;
        n = cap;
        generated->fetch_add(cap, std::memory_order_relaxed);
        return current_time_usecs() < stop_at_us;
    }
};

// the same stream generated on the device (-d): no host->device copies, shows what the operators sustain when the
// data is already in HBM
struct DeviceGenerator {
    uint64_t n_keys, n_sources;
    __host__ __device__ void operator()(uint64_t s, uint32_t src, tuple_t& t) const {
        t.key = (s % n_keys) * n_sources + src;
        t.id = s / n_keys;
        t.ts = s;
        t.value = (double)((s * 2654435761ull >> 7) & 1023);
        t.aux = 0.0;
    }
};

// WF_SEGV_TRACE=1: print the faulting address and a backtrace with library names (profilers swallow the default one)
static void segv_trace(int, siginfo_t* si, void*) {
    void* frames[48];
    const int n = backtrace(frames, 48);
    dprintf(2, "SIGSEGV at address %p\n", si->si_addr);
    backtrace_symbols_fd(frames, n, 2);
    _exit(139);
}

int main(int argc, char** argv) {
    if (std::getenv("WF_SEGV_TRACE")) {
        struct sigaction sa {};
        sa.sa_sigaction = segv_trace;
        sa.sa_flags = SA_SIGINFO;
        sigaction(SIGSEGV, &sa, nullptr);
    }
    std::string op = "map_stateless";
    size_t sources = 4, replicas = 0, keys = 1 << 14, batch = 1 << 17;
    double seconds = 3.0;
    bool device_source = false;
    size_t inflight = 0;  // pinned slabs per source: batches that can be queued in front of a replica (0: 6, or 12 for small batches)
    int c;
    while ((c = getopt(argc, argv, "o:s:n:k:b:t:di:")) != -1) {
        switch (c) {
            case 'o': op = optarg; break;
            case 's': sources = std::strtoull(optarg, nullptr, 10); break;
            case 'n': replicas = std::strtoull(optarg, nullptr, 10); break;
            case 'k': keys = std::strtoull(optarg, nullptr, 10); break;
            case 'b': batch = std::strtoull(optarg, nullptr, 10); break;
            case 't': seconds = std::atof(optarg); break;
            case 'd': device_source = true; break;
            case 'i': inflight = std::strtoull(optarg, nullptr, 10); break;
            default:
                std::cout << argv[0] << " -o map_stateless|filter_stateless|map_stateful|map_sum|map_scan|filter_stateful|keyby|win_agg [-s sources] [-n replicas] "
                          << "[-k keys] [-b batch] [-t seconds] [-d (generate on the device)] [-i batches in flight per source]" << std::endl;
                return 0;
        }
    }
    if (replicas == 0) replicas = sources;
    if (gpu::device_count() < 1) {
        std::cerr << "bench_ops_gpu needs a GPU" << std::endl;
        return 3;
    }
    const size_t keys_per_source = std::max<size_t>(1, keys / sources);
    std::atomic<uint64_t> generated{0}, results{0};

    // lookup table for the stateless map
    gpu::DeviceBuffer<TableEntry> table;
    uint32_t mask = 0;
    if (op == "map_stateless") {
        size_t cap = 2;
        while (cap < 2 * keys_per_source * sources) cap <<= 1;
        mask = (uint32_t)(cap - 1);
        std::vector<TableEntry> host(cap, TableEntry{~0ull, 0});
        for (size_t k = 0; k < keys_per_source * sources; k++) {
            uint32_t h = (uint32_t)(k * 0x9E3779B97F4A7C15ull >> 40) & mask;
            while (host[h].key != ~0ull) h = (h + 1) & mask;
            host[h] = TableEntry{k, k * 7 + 1};
        }
        table.resize(cap);
        WF_HIP_CHECK(hipMemcpy(table.data(), host.data(), cap * sizeof(TableEntry), hipMemcpyHostToDevice));
    }

    PipeGraph graph("bench_" + op, Mode::DEFAULT);
    const uint64_t t0 = current_time_usecs();
    TimedGenerator gen{keys_per_source, sources, t0 + (uint64_t)(seconds * 1e6), 0, &generated};
    std::unique_ptr<Basic_Operator> src;
    if (device_source) {
        const uint64_t stop_at = t0 + (uint64_t)(seconds * 1e6);
        auto* gcount = &generated;
        src = DeviceSource_Builder(DeviceGenerator{keys_per_source, sources})
                  .withName("device_source")
                  .withParallelism(sources)
                  .withBatch(batch)
                  .withStopCondition([stop_at, gcount, batch](uint64_t) {
                      if (current_time_usecs() >= stop_at) return false;
                      gcount->fetch_add(batch, std::memory_order_relaxed);
                      return true;
                  })
                  .build_unique();
    } else {
        // a replica groups at most half of its producer's slabs (WF_COALESCE_SHARE): small batches get a deeper ring so that a
        // group still amortises the launches (profiles/r3_small_batch/coalescing_share.txt)
        if (inflight == 0) inflight = batch < 32768 ? 12 : 6;
        src = BatchSource_Builder(gen).withName("source").withParallelism(sources).withBatch(batch).withInFlightBatches(inflight).build_unique();
    }
    MultiPipe& mp = graph.add_source(*src);
    std::unique_ptr<Basic_Operator> oper;
    const bool disjoint = (replicas == sources);  // lane i carries exactly the keys of replica i
    // distinct keys a replica can see, plus slack: the device table is twice this, and its size decides the number of
    // sort passes per batch (one 11-bit pass up to 1024 slots)
    const size_t keys_per_replica = keys_per_source * sources / (disjoint ? sources : 1);
    const size_t max_keys = keys_per_replica + keys_per_replica / 4 + 16;
    if (op == "map_stateless") {
        oper = MapGPU_Builder(LookupMap{table.data(), mask}).withName(op).withParallelism(replicas).build_unique();
    } else if (op == "filter_stateless") {
        oper = FilterGPU_Builder(HalfFilter()).withName(op).withParallelism(replicas).build_unique();
    } else if (op == "map_stateful" || op == "keyby") {
        auto b = MapGPU_Builder(RunningAverageMap()).withName(op).withParallelism(replicas).withScratchpad(sizeof(WindowState))
                     .withMaxKeys(max_keys);
        if (disjoint && op != "keyby") b.withKeyDisjointLanes();
        oper = b.build_unique();
    } else if (op == "map_sum") {
        auto b = MapGPU_Builder(RunningSumMap()).withName(op).withParallelism(replicas).withScratchpad(sizeof(double)).withMaxKeys(max_keys);
        if (disjoint) b.withKeyDisjointLanes();
        oper = b.build_unique();
    } else if (op == "map_scan") {
        auto b = ScanMapGPU_Builder(keyed_scan(SumLift(), SumCombine(), SumApply())).withName(op).withParallelism(replicas).withMaxKeys(max_keys);
        if (disjoint) b.withKeyDisjointLanes();
        oper = b.build_unique();
    } else if (op == "filter_stateful") {
        auto b = FilterGPU_Builder(AboveAverageFilter()).withName(op).withParallelism(replicas).withScratchpad(sizeof(WindowState))
                     .withMaxKeys(max_keys);
        if (disjoint) b.withKeyDisjointLanes();
        oper = b.build_unique();
    } else if (op == "win_agg") {
        // the flagship of bench.py: keyed sliding-window sum, time-based window of 16 / slide of 4 tuples of a key
        if (!disjoint) {
            std::cerr << "win_agg needs as many replicas as sources" << std::endl;
            return 2;
        }
        KeyFFATGPUConfig wcfg;
        wcfg.win_len = 16 * keys_per_source;
        wcfg.slide_len = 4 * keys_per_source;
        wcfg.delay = 0;
        wcfg.type = TB;
        wcfg.max_keys = keys_per_source;
        wcfg.out_capacity = batch;
        wcfg.lanes_are_key_disjoint = true;
        GPUConfig cfg;
        cfg.batch_len = batch;
        oper.reset(new Key_FFAT_GPU<tuple_t, wfbench::result_t>(wfbench::LiftValue{}, wfbench::CombSum{}, wcfg, replicas, "key_ffat_gpu", cfg));
    } else {
        std::cerr << "unknown operator " << op << std::endl;
        return 2;
    }
    mp.add(*oper);
    std::unique_ptr<Basic_Operator> sink;
    if (op == "win_agg")
        sink = BatchSink_Builder([&results](const wfbench::result_t*, size_t n) { results.fetch_add(n, std::memory_order_relaxed); })
                   .withName("sink").withParallelism(replicas).build_unique();
    else
        sink = BatchSink_Builder([&results](const tuple_t*, size_t n) { results.fetch_add(n, std::memory_order_relaxed); })
                   .withName("sink").withParallelism(replicas).build_unique();
    mp.chain_sink(*sink);
    graph.run();
    const double elapsed = (current_time_usecs() - t0) / 1e6;
    std::cout << op << (device_source ? " [device source]" : "") << " sources=" << sources << " replicas=" << replicas << " keys=" << keys_per_source * sources << " batch=" << batch << " : "
              << (double)generated.load() / elapsed << " tuples/s (" << (double)results.load() / elapsed << " results/s), elapsed " << elapsed
              << " s" << std::endl;
    return 0;
}
