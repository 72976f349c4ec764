// GPU flavour of test_mp_windows.cpp: Source -> Filter -> FlatMap -> Map -> <GPU
// window operator> -> Sink, randomised parallelism and batch length, checked
// against the sequential oracle.  Mirrors the reference's src/mp_test_gpu suite
// (test_mp_gpu_{kf,wf,pf,wmr,kf+pf,wf+wmr,...}_{cb,tb}) plus Key_FFAT_GPU.
//
//   test_mp_windows_gpu -t <op> -m <cb|tb|tbd> [-r runs] [-l len] [-k keys] [-w win] [-s slide]
//     op: seq kf wf pf_plq pf_wlq wmr_map wmr_red kff kf+pf wf+wmr kf+wmr wf+pf (one stage on the CPU, as in the reference)
//         pf_gg wmr_gg kf+pf_gg wf+pf_gg kf+wmr_gg wf+wmr_gg             (both stages on the device: batch path)
//     -c: "chaining" variant (src/mp_test_gpu/test_mp_gpu_*_chaining.cpp): equal parallelism everywhere, every operator chain()ed
#include <unistd.h>

#include <chrono>
#include <iostream>
#include <random>

#include <wf/windflow_gpu.hpp>

#include "mp_common.hpp"

using namespace wf;
using namespace std;
using us = std::chrono::microseconds;

// device window functors (6-argument form)
struct Win_GPU {
    __host__ __device__ void operator()(uint64_t, const tuple_t* data, size_t n, output_t* res, char*, size_t) const {
        int64_t sum = 0;
        for (size_t i = 0; i < n; i++) sum += data[i].value;
        res->value = sum;
    }
};
struct Win2_GPU {
    __host__ __device__ void operator()(uint64_t, const output_t* data, size_t n, output_t* res, char* scratch, size_t sz) const {
        int64_t sum = 0;
        for (size_t i = 0; i < n; i++) sum += data[i].value;
        if (sz >= sizeof(int64_t)) *reinterpret_cast<int64_t*>(scratch) = sum;  // exercise the scratchpad
        res->value = sum;
    }
};
struct Lift_GPU {
    __host__ __device__ void operator()(const tuple_t& t, output_t& r) const { r.value = t.value; }
};
struct Comb_GPU {
    __host__ __device__ void operator()(const output_t& a, const output_t& b, output_t& out) const { out.value = a.value + b.value; }
};

struct Cfg {
    string op = "kf", mode = "cb";
    size_t runs = 3, len = 2000, keys = 4;
    uint64_t win = 10, slide = 3;
    bool chaining = false;
};

template <typename B>
static B& set_windows(B& b, const Cfg& c) {
    if (c.mode == "cb")
        b.withCBWindows(c.win, c.slide);
    else
        b.withTBWindows(us(c.win), us(c.slide), c.mode == "tbd" ? us(50000000) : us(0));
    return b;
}
// non-incremental engines keep the tuples of open windows on the device: with a 50 s triggering delay that is the whole stream
template <typename B>
static B& set_nic_windows(B& b, const Cfg& c) {
    set_windows(b, c);
    b.withMaxKeys(64);
    if (c.mode == "tbd") b.withArchiveCapacity(3 * c.len);
    return b;
}

static void add_window_op(MultiPipe& mp, const Cfg& c, std::mt19937& rng, vector<shared_ptr<Basic_Operator>>& keep, string& descr) {
    std::uniform_int_distribution<int> d(1, 4);
    std::uniform_int_distribution<int> db(1, 300);
    size_t p1 = (size_t)d(rng), p2 = (size_t)d(rng), p3 = (size_t)d(rng);
    if (c.chaining) p1 = p2 = p3 = 2;
    const size_t batch = (size_t)db(rng);
    auto hold = [&keep](auto* op) {
        keep.emplace_back(op);
        return op;
    };
    descr = c.op + " batch=" + to_string(batch) + " par=(" + to_string(p1) + "," + to_string(p2) + "," + to_string(p3) + ")";
    if (c.op == "seq") {
        auto b = WinSeqGPU_Builder(Win_GPU());
        mp.add(*hold(set_nic_windows(b, c).withName("seq_gpu").withBatch(batch).build_ptr()));
    } else if (c.op == "kf") {
        auto b = KeyFarmGPU_Builder(Win_GPU());
        mp.add(*hold(set_nic_windows(b, c).withName("kf_gpu").withParallelism(p1).withBatch(batch).build_ptr()));
    } else if (c.op == "wf") {
        auto b = WinFarmGPU_Builder(Win_GPU());
        mp.add(*hold(set_nic_windows(b, c).withName("wf_gpu").withParallelism(p1).withBatch(batch).build_ptr()));
    } else if (c.op == "pf_plq") {
        auto b = PaneFarmGPU_Builder(Win_GPU(), win2_nic);
        mp.add(*hold(set_nic_windows(b, c).withName("pf_gpu").withParallelism(p1, p2).withBatch(batch).build_ptr()));
    } else if (c.op == "pf_wlq") {
        auto b = PaneFarmGPU_Builder(win_nic, Win2_GPU());
        mp.add(*hold(set_nic_windows(b, c).withName("pf_gpu").withParallelism(p1, p2).withBatch(batch).withScratchpad(16).build_ptr()));
    } else if (c.op == "wmr_map") {
        auto b = WinMapReduceGPU_Builder(Win_GPU(), win2_inc);
        mp.add(*hold(set_nic_windows(b, c).withName("wmr_gpu").withParallelism(p1 + 1, p2).withBatch(batch).build_ptr()));
    } else if (c.op == "wmr_red") {
        auto b = WinMapReduceGPU_Builder(win_inc, Win2_GPU());
        mp.add(*hold(set_nic_windows(b, c).withName("wmr_gpu").withParallelism(p1 + 1, p2).withBatch(batch).build_ptr()));
    } else if (c.op == "kf_wr") {  // associative form: a wave64 per window
        auto b = KeyFarmGPU_Builder(gpu::wave_reduce(Lift_GPU(), Comb_GPU()));
        mp.add(*hold(set_nic_windows(b, c).withName("kf_wr_gpu").withParallelism(p1).withBatch(batch).build_ptr()));
    } else if (c.op == "wf_wr") {
        auto b = WinFarmGPU_Builder(gpu::wave_reduce(Lift_GPU(), Comb_GPU()));
        mp.add(*hold(set_nic_windows(b, c).withName("wf_wr_gpu").withParallelism(p1).withBatch(batch).build_ptr()));
    } else if (c.op == "kff") {
        auto b = KeyFFATGPU_Builder(Lift_GPU(), Comb_GPU());
        mp.add(*hold(set_windows(b, c).withName("kff_gpu").withParallelism(p1).withBatch(64 * batch).withMaxKeys(64).withOutputCapacity(1 << 17).build_ptr()));
    } else if (c.op == "seqffat") {
        auto b = WinSeqFFATGPU_Builder(Lift_GPU(), Comb_GPU());
        mp.add(*hold(set_windows(b, c).withName("seqffat_gpu").withBatch(64 * batch).withMaxKeys(64).withOutputCapacity(1 << 17).withRebuild(false)
                         .build_ptr()));
    } else if (c.op == "kf+pf") {
        auto b = PaneFarmGPU_Builder(Win_GPU(), win2_nic);
        auto* pf = hold(set_nic_windows(b, c).withName("pf_inner").withParallelism(p2, p3).withBatch(batch).prepare4Nesting().build_ptr());
        mp.add(*hold(KeyFarmGPU_Builder(*pf).withName("kf_pf_gpu").withParallelism(p1).build_ptr()));
    } else if (c.op == "wf+wmr") {
        auto b = WinMapReduceGPU_Builder(win_nic, Win2_GPU());
        auto* wmr = hold(set_nic_windows(b, c).withName("wmr_inner").withParallelism(p2 + 1, p3).withBatch(batch).prepare4Nesting().build_ptr());
        mp.add(*hold(WinFarmGPU_Builder(*wmr).withName("wf_wmr_gpu").withParallelism(p1).build_ptr()));
    } else if (c.op == "kf+wmr") {
        auto b = WinMapReduceGPU_Builder(Win_GPU(), win2_nic);
        auto* wmr = hold(set_nic_windows(b, c).withName("wmr_inner").withParallelism(p2 + 1, p3).withBatch(batch).prepare4Nesting().build_ptr());
        mp.add(*hold(KeyFarmGPU_Builder(*wmr).withName("kf_wmr_gpu").withParallelism(p1).build_ptr()));
    } else if (c.op == "wf+pf") {
        auto b = PaneFarmGPU_Builder(win_nic, Win2_GPU());
        auto* pf = hold(set_nic_windows(b, c).withName("pf_inner").withParallelism(p2, p3).withBatch(batch).prepare4Nesting().build_ptr());
        mp.add(*hold(WinFarmGPU_Builder(*pf).withName("wf_pf_gpu").withParallelism(p1).build_ptr()));
    } else if (c.op == "pf_gg") {  // both stages on the device: replicas divide the keys, PLQ -> WLQ without leaving HBM
        auto b = PaneFarmGPU_Builder(Win_GPU(), Win2_GPU());
        mp.add(*hold(set_nic_windows(b, c).withName("pf_gg").withParallelism(p1, p2).withBatch(batch).withScratchpad(16).build_ptr()));
    } else if (c.op == "wmr_gg") {
        auto b = WinMapReduceGPU_Builder(Win_GPU(), Win2_GPU());
        mp.add(*hold(set_nic_windows(b, c).withName("wmr_gg").withParallelism(p1 + 1, p2).withBatch(batch).build_ptr()));
    } else if (c.op == "kf+pf_gg" || c.op == "wf+pf_gg") {
        auto b = PaneFarmGPU_Builder(Win_GPU(), Win2_GPU());
        auto* pf = hold(set_nic_windows(b, c).withName("pf_inner").withParallelism(p2, p3).withBatch(batch).prepare4Nesting().build_ptr());
        if (c.op[0] == 'k')
            mp.add(*hold(KeyFarmGPU_Builder(*pf).withName("kf_pf_gg").withParallelism(p1).build_ptr()));
        else
            mp.add(*hold(WinFarmGPU_Builder(*pf).withName("wf_pf_gg").withParallelism(p1).build_ptr()));
    } else if (c.op == "kf+wmr_gg" || c.op == "wf+wmr_gg") {
        auto b = WinMapReduceGPU_Builder(Win_GPU(), Win2_GPU());
        auto* wmr = hold(set_nic_windows(b, c).withName("wmr_inner").withParallelism(p2 + 1, p3).withBatch(batch).prepare4Nesting().build_ptr());
        if (c.op[0] == 'k')
            mp.add(*hold(KeyFarmGPU_Builder(*wmr).withName("kf_wmr_gg").withParallelism(p1).build_ptr()));
        else
            mp.add(*hold(WinFarmGPU_Builder(*wmr).withName("wf_wmr_gg").withParallelism(p1).build_ptr()));
    } else {
        cerr << "unknown operator " << c.op << endl;
        exit(2);
    }
}

int main(int argc, char* argv[]) {
    Cfg c;
    int o;
    while ((o = getopt(argc, argv, "t:m:r:l:k:w:s:c")) != -1) {
        switch (o) {
            case 't': c.op = optarg; break;
            case 'm': c.mode = optarg; break;
            case 'r': c.runs = (size_t)atoi(optarg); break;
            case 'l': c.len = (size_t)atoi(optarg); break;
            case 'k': c.keys = (size_t)atoi(optarg); break;
            case 'w': c.win = (uint64_t)atoll(optarg); break;
            case 's': c.slide = (uint64_t)atoll(optarg); break;
            case 'c': c.chaining = true; break;
            default: return 2;
        }
    }
    if (gpu::device_count() < 1) {
        cerr << "no GPU" << endl;
        return 3;
    }
    const auto stream = oracle_stream(c.len, c.keys);
    const OracleResult expect = (c.mode == "cb") ? oracle_cb(stream, c.win, c.slide) : oracle_tb(stream, c.win, c.slide);
    cout << "oracle: " << expect.windows << " windows, sum " << expect.sum << endl;
    std::mt19937 rng(std::random_device{}());
    std::uniform_int_distribution<int> d(1, 4);
    int failures = 0;
    for (size_t run = 0; run < c.runs; run++) {
        size_t fp = (size_t)d(rng), mpp = (size_t)d(rng);
        if (c.chaining) fp = mpp = 2;  // equal parallelism: chain() really fuses the CPU operators
        SinkTotals totals;
        PipeGraph graph("test_mp_gpu_" + c.op + "_" + c.mode, c.mode == "tbd" ? Mode::DEFAULT : Mode::DETERMINISTIC);
        Source source = Source_Builder(Source_Functor(c.len, c.keys)).withName("source").withParallelism(1).build();
        MultiPipe& mp = graph.add_source(source);
        Filter filter = Filter_Builder(Filter_Functor()).withName("filter").withParallelism(fp).build();
        mp.chain(filter);
        FlatMap flatmap = FlatMap_Builder(FlatMap_Functor()).withName("flatmap").withParallelism(2).build();
        mp.chain(flatmap);
        Map map = Map_Builder(Map_Functor()).withName("map").withParallelism(mpp).build();
        mp.chain(map);
        vector<shared_ptr<Basic_Operator>> keep;
        string descr;
        add_window_op(mp, c, rng, keep, descr);
        Sink sink = Sink_Builder(Sink_Functor(&totals)).withName("sink").withParallelism(1).build();
        mp.chain_sink(sink);
        if (c.chaining) descr += " chaining, " + to_string(graph.getNumThreads()) + " threads";
        graph.run();
        bool ok = totals.received == expect.windows;
        if (c.mode != "tbd") ok = ok && totals.sum == expect.sum;
        cout << "run " << run << ": " << descr << " -> received " << totals.received << " sum " << totals.sum << (ok ? "  OK" : "  FAILED") << endl;
        if (!ok) failures++;
    }
    return failures == 0 ? 0 : 1;
}
