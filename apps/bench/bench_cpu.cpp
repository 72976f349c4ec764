// Throughput of the CPU runtime on simple pipelines (no GPU): how much a stage boundary costs
// when it is a thread hop through an SPSC ring (add) and when it is a fused call (chain).
//   bench_cpu [-n tuples per source] [-s sources] [-p parallelism of the middle operators]
#include <unistd.h>

#include <atomic>
#include <iostream>
#include <optional>
#include <tuple>

#include <wf/windflow.hpp>

using namespace wf;

struct rec_t {
    size_t key = 0;
    uint64_t id = 0, ts = 0;
    int64_t value = 0;
    std::tuple<size_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    void setControlFields(size_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};

static size_t sinks = 1;
static double run(const char* what, size_t n, size_t sources, size_t par, bool chained, bool windowed) {
    PipeGraph g(what, windowed ? Mode::DETERMINISTIC : Mode::DEFAULT);
    Source src = Source_Builder([n, v = (uint64_t)0](rec_t& t) mutable {
                     t.setControlFields(v % 64, v / 64, v);
                     t.value = (int64_t)v;
                     return ++v < n;
                 }).withParallelism(sources).build();
    MultiPipe& mp = g.add_source(src);
    Map m = Map_Builder([](rec_t& t) { t.value = t.value * 3 + 1; }).withParallelism(chained ? sources : par).build();
    Filter f = Filter_Builder([](rec_t& t) { return (t.value & 3) != 0; }).withParallelism(chained ? sources : par).build();
    if (chained) {
        mp.chain(m);
        mp.chain(f);
    } else {
        mp.add(m);
        mp.add(f);
    }
    std::unique_ptr<Basic_Operator> win;
    std::atomic<uint64_t> got{0};
    if (windowed) {
        win = KeyFFAT_Builder([](const rec_t& t, rec_t& r) { r.value = t.value; }, [](const rec_t& a, const rec_t& b, rec_t& o) { o.value = a.value + b.value; })
                  .withCBWindows(64, 16).withParallelism(par).build_unique();
        mp.add(*win);
    }
    Sink s = Sink_Builder([&got](std::optional<rec_t>& r) {
                 if (r) got.fetch_add(1, std::memory_order_relaxed);
             }).withParallelism(sinks).build();
    mp.add_sink(s);
    const uint64_t t0 = current_time_usecs();
    g.run();
    const double sec = (current_time_usecs() - t0) / 1e6;
    const double rate = (double)(n * sources) / sec / 1e6;
    std::cout << what << ": " << rate << " M tuples/s (" << g.getNumThreads() << " threads, " << got.load() << " results)" << std::endl;
    return rate;
}

int main(int argc, char** argv) {
    size_t n = 4000000, sources = 2, par = 2;
    int c;
    while ((c = getopt(argc, argv, "n:s:p:k:")) != -1) {
        if (c == 'n') n = std::strtoull(optarg, nullptr, 10);
        if (c == 's') sources = std::strtoull(optarg, nullptr, 10);
        if (c == 'p') par = std::strtoull(optarg, nullptr, 10);
        if (c == 'k') sinks = std::strtoull(optarg, nullptr, 10);
    }
    run("source->map->filter->sink, every stage a thread", n, sources, par, false, false);
    run("source->map->filter chained, sink a thread", n, sources, par, true, false);
    run("source->map->filter->Key_FFAT(CB 64/16)->sink", n, sources, par, true, true);
    return 0;
}
