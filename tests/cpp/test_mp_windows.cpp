// MultiPipe regression program: Source -> Filter -> FlatMap -> Map -> <window
// operator> -> Sink with randomised parallelism, checked against the oracle of
// mp_common.hpp.  One executable covers the matrix of the reference's
// src/mp_test_cpu suite (test_mp_{wf,kf,pf,wmr,kf+pf,kf+wmr,wf+pf,wf+wmr}_{cb,tb,tb_2}
// and the _chaining variants) plus Key_FFAT / Win_Seq / Win_SeqFFAT.
//
//   test_mp_windows -t <op> -m <cb|tb|tbd> [-r runs] [-l len] [-k keys] [-w win] [-s slide] [-i] [-c]
//     op  : seq ffat wf kf kff pf wmr kf+pf kf+wmr wf+pf wf+wmr
//     mode: cb  = count-based, DETERMINISTIC graph
//           tb  = time-based,  DETERMINISTIC graph (exact sum check)
//           tbd = time-based,  DEFAULT graph with a large triggering delay (count check)
//     -i  : incremental window functions     -c : equal parallelism + chain() everywhere
#include <unistd.h>

#include <chrono>
#include <iostream>
#include <random>

#include "mp_common.hpp"

using namespace wf;
using namespace std;
using us = std::chrono::microseconds;

struct Cfg {
    string op = "wf", mode = "cb";
    size_t runs = 3, len = 2000, keys = 4;
    uint64_t win = 10, slide = 3;
    bool inc = false, chaining = false;
};

template <typename B>
static B& set_windows(B& b, const Cfg& c) {
    if (c.mode == "cb")
        b.withCBWindows(c.win, c.slide);
    else
        b.withTBWindows(us(c.win), us(c.slide), c.mode == "tbd" ? us(50000000) : us(0));
    return b;
}

// builds the window operator, adds it to the pipe; objects are kept alive in `keep`
static void add_window_op(MultiPipe& mp, const Cfg& c, std::mt19937& rng, vector<shared_ptr<Basic_Operator>>& keep, string& descr) {
    std::uniform_int_distribution<int> d(1, 6);
    const size_t p1 = c.chaining ? 2 : (size_t)d(rng), p2 = (size_t)d(rng), p3 = (size_t)d(rng);
    auto hold = [&keep](auto* op) {
        keep.emplace_back(op);
        return op;
    };
    if (c.op == "seq") {
        auto b = c.inc ? nullptr : nullptr;
        (void)b;
        if (c.inc) {
            auto bb = WinSeq_Builder(win_inc);
            mp.add(*hold(set_windows(bb, c).withName("win_seq").build_ptr()));
        } else {
            auto bb = WinSeq_Builder(win_nic);
            mp.add(*hold(set_windows(bb, c).withName("win_seq").build_ptr()));
        }
        descr = "SEQ(1)";
    } else if (c.op == "ffat") {
        auto bb = WinSeqFFAT_Builder(lift_f, comb_f);
        mp.add(*hold(set_windows(bb, c).withName("win_seqffat").build_ptr()));
        descr = "FFAT(1)";
    } else if (c.op == "wf") {
        if (c.inc) {
            auto bb = WinFarm_Builder(win_inc);
            mp.add(*hold(set_windows(bb, c).withName("wf").withParallelism(p1).build_ptr()));
        } else {
            auto bb = WinFarm_Builder(Rich_Win_NIC());
            mp.add(*hold(set_windows(bb, c).withName("wf").withParallelism(p1).build_ptr()));
        }
        descr = "WF(" + to_string(p1) + ")";
    } else if (c.op == "kf") {
        if (c.inc) {
            auto bb = KeyFarm_Builder(win_inc);
            mp.add(*hold(set_windows(bb, c).withName("kf").withParallelism(p1).build_ptr()));
        } else {
            auto bb = KeyFarm_Builder(win_nic);
            mp.add(*hold(set_windows(bb, c).withName("kf").withParallelism(p1).build_ptr()));
        }
        descr = "KF(" + to_string(p1) + ")";
    } else if (c.op == "kff") {
        auto bb = KeyFFAT_Builder(lift_f, comb_f);
        mp.add(*hold(set_windows(bb, c).withName("kff").withParallelism(p1).build_ptr()));
        descr = "KFF(" + to_string(p1) + ")";
    } else if (c.op == "pf") {
        if (c.inc) {
            auto bb = PaneFarm_Builder(win_inc, win2_inc);
            mp.add(*hold(set_windows(bb, c).withName("pf").withParallelism(p1, p2).build_ptr()));
        } else {
            auto bb = PaneFarm_Builder(win_nic, win2_nic);
            mp.add(*hold(set_windows(bb, c).withName("pf").withParallelism(p1, p2).build_ptr()));
        }
        descr = "PF(" + to_string(p1) + "," + to_string(p2) + ")";
    } else if (c.op == "wmr") {
        const size_t m = std::max<size_t>(2, p1);
        if (c.inc) {
            auto bb = WinMapReduce_Builder(win_inc, win2_inc);
            mp.add(*hold(set_windows(bb, c).withName("wmr").withParallelism(m, p2).build_ptr()));
        } else {
            auto bb = WinMapReduce_Builder(win_nic, win2_nic);
            mp.add(*hold(set_windows(bb, c).withName("wmr").withParallelism(m, p2).build_ptr()));
        }
        descr = "WMR(" + to_string(m) + "," + to_string(p2) + ")";
    } else if (c.op == "kf+pf" || c.op == "wf+pf") {
        auto bb = PaneFarm_Builder(win_nic, win2_inc);
        auto* pf = hold(set_windows(bb, c).withName("pf_inner").withParallelism(p2, p3).prepare4Nesting().build_ptr());
        if (c.op == "kf+pf")
            mp.add(*hold(KeyFarm_Builder(*pf).withName("kf_pf").withParallelism(p1).build_ptr()));
        else
            mp.add(*hold(WinFarm_Builder(*pf).withName("wf_pf").withParallelism(p1).build_ptr()));
        descr = c.op + "(" + to_string(p1) + " x [" + to_string(p2) + "," + to_string(p3) + "])";
    } else if (c.op == "kf+wmr" || c.op == "wf+wmr") {
        const size_t m = std::max<size_t>(2, p2);
        auto bb = WinMapReduce_Builder(win_inc, win2_nic);
        auto* wmr = hold(set_windows(bb, c).withName("wmr_inner").withParallelism(m, p3).prepare4Nesting().build_ptr());
        if (c.op == "kf+wmr")
            mp.add(*hold(KeyFarm_Builder(*wmr).withName("kf_wmr").withParallelism(p1).build_ptr()));
        else
            mp.add(*hold(WinFarm_Builder(*wmr).withName("wf_wmr").withParallelism(p1).build_ptr()));
        descr = c.op + "(" + to_string(p1) + " x [" + to_string(m) + "," + to_string(p3) + "])";
    } else {
        cerr << "unknown operator " << c.op << endl;
        exit(2);
    }
}

int main(int argc, char* argv[]) {
    Cfg c;
    int o;
    while ((o = getopt(argc, argv, "t:m:r:l:k:w:s:ic")) != -1) {
        switch (o) {
            case 't': c.op = optarg; break;
            case 'm': c.mode = optarg; break;
            case 'r': c.runs = (size_t)atoi(optarg); break;
            case 'l': c.len = (size_t)atoi(optarg); break;
            case 'k': c.keys = (size_t)atoi(optarg); break;
            case 'w': c.win = (uint64_t)atoll(optarg); break;
            case 's': c.slide = (uint64_t)atoll(optarg); break;
            case 'i': c.inc = true; break;
            case 'c': c.chaining = true; break;
            default: return 2;
        }
    }
    const auto stream = oracle_stream(c.len, c.keys);
    const OracleResult expect = (c.mode == "cb") ? oracle_cb(stream, c.win, c.slide) : oracle_tb(stream, c.win, c.slide);
    cout << "oracle: " << expect.windows << " windows, sum " << expect.sum << endl;
    std::mt19937 rng(std::random_device{}());
    std::uniform_int_distribution<int> d(1, 6);
    int failures = 0;
    for (size_t run = 0; run < c.runs; run++) {
        const size_t fp = c.chaining ? 2 : (size_t)d(rng), fmp = c.chaining ? 2 : (size_t)d(rng), mpp = c.chaining ? 2 : (size_t)d(rng);
        SinkTotals totals;
        PipeGraph graph("test_mp_" + c.op + "_" + c.mode, c.mode == "tbd" ? Mode::DEFAULT : Mode::DETERMINISTIC);
        Source source = Source_Builder(Source_Functor(c.len, c.keys)).withName("source").withParallelism(1).build();
        MultiPipe& mp = graph.add_source(source);
        Filter filter = Filter_Builder(Filter_Functor()).withName("filter").withParallelism(fp).build();
        mp.chain(filter);
        FlatMap flatmap = FlatMap_Builder(FlatMap_Functor()).withName("flatmap").withParallelism(fmp).build();
        mp.chain(flatmap);
        Map map = Map_Builder(Map_Functor()).withName("map").withParallelism(mpp).build();
        mp.chain(map);
        vector<shared_ptr<Basic_Operator>> keep;
        string descr;
        add_window_op(mp, c, rng, keep, descr);
        Sink sink = Sink_Builder(Sink_Functor(&totals)).withName("sink").withParallelism(1).build();
        mp.chain_sink(sink);
        const size_t threads = graph.getNumThreads();
        graph.run();
        bool ok = totals.received == expect.windows;
        if (c.mode != "tbd") ok = ok && totals.sum == expect.sum;
        cout << "run " << run << ": S(1) F(" << fp << ") FM(" << fmp << ") M(" << mpp << ") " << descr << " threads=" << threads
             << " -> received " << totals.received << " sum " << totals.sum << (ok ? "  OK" : "  FAILED") << endl;
        if (!ok) failures++;
    }
    return failures == 0 ? 0 : 1;
}
