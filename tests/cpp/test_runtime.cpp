// Unit tests of the runtime substrate and of the window building blocks (the
// reference has none: SURVEY.md §4 "no unit-test framework"; §7 asks for SPSC /
// graph / FlatFAT / ordering tests of our own).  Also built with
// -fsanitize=thread by tests/test_cpu_runtime.py::test_runtime_tsan.
//
//   test_runtime            run every case, exit 0 iff all pass
//   test_runtime illegal N  run illegal-topology case N (must die with a WindFlow Error)
#include "../../windflow_amd/csrc/bench_common.hpp"  // SkewBound (the benchmark sources that share keys)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <functional>
#include <iostream>
#include <optional>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include <wf/windflow.hpp>

using namespace wf;
using namespace std;

static int failures = 0;
#define CHECK(cond)                                                                         \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            cerr << "CHECK failed at " << __FILE__ << ":" << __LINE__ << ": " #cond << endl; \
            failures++;                                                                     \
        }                                                                                   \
    } while (0)

// ---- SPSC ring ---------------------------------------------------------------------------------
static void test_spsc() {
    rt::SPSCQueue q(8);
    CHECK(q.capacity() == 8);
    void* out = nullptr;
    CHECK(!q.try_pop(out));
    for (uintptr_t i = 1; i <= 8; i++) CHECK(q.try_push(reinterpret_cast<void*>(i)));
    CHECK(!q.try_push(reinterpret_cast<void*>(9)));  // full
    CHECK(q.size_approx() == 8);
    void* bulk[5];
    CHECK(q.pop_bulk(bulk, 5) == 5);
    for (uintptr_t i = 0; i < 5; i++) CHECK(bulk[i] == reinterpret_cast<void*>(i + 1));
    CHECK(q.try_pop(out) && out == reinterpret_cast<void*>(6));
    // two threads, 2M items through a ring of 64 slots: FIFO and nothing lost
    rt::SPSCQueue ring(64);
    const uintptr_t N = 2000000;
    atomic<bool> ok{true};
    thread cons([&] {
        uintptr_t expect = 1;
        void* items[16];
        while (expect <= N) {
            const size_t n = ring.pop_bulk(items, 16);
            for (size_t i = 0; i < n; i++)
                if (items[i] != reinterpret_cast<void*>(expect++)) ok = false;
            if (n == 0) this_thread::yield();
        }
    });
    for (uintptr_t i = 1; i <= N; i++)
        while (!ring.try_push(reinterpret_cast<void*>(i))) this_thread::yield();
    cons.join();
    CHECK(ok.load());
}

// ---- graph: fan-out / fan-in, EOS propagation, back-pressure with tiny rings --------------------
struct Producer : rt::Node {
    long n, sent = 0;
    explicit Producer(long n_) : n(n_) {}
    bool is_source() const override { return true; }
    bool generate() override {
        emit_to(new long(sent), (size_t)(sent % (long)num_out()));
        return ++sent < n;
    }
};
struct Adder : rt::Node {
    long add;
    explicit Adder(long a) : add(a) {}
    void svc(void* m, int) override {
        long* v = static_cast<long*>(m);
        *v += add;
        emit(v);
    }
};
struct Collector : rt::Node {
    atomic<long>* sum;
    atomic<long>* cnt;
    atomic<int>* eos_seen;
    Collector(atomic<long>* s, atomic<long>* c, atomic<int>* e) : sum(s), cnt(c), eos_seen(e) {}
    void svc(void* m, int) override {
        long* v = static_cast<long*>(m);
        *sum += *v;
        (*cnt)++;
        delete v;
    }
    void eosnotify(int) override { (*eos_seen)++; }
};
static void test_graph() {
    rt::RuntimeConfig cfg;
    cfg.queue_capacity = 4;  // forces the producer to block on full rings all the time
    rt::Graph g(cfg);
    atomic<long> sum{0}, cnt{0};
    atomic<int> eos{0};
    const long N = 200000;
    auto* p = g.adopt(new Producer(N));
    auto* c = g.adopt(new Collector(&sum, &cnt, &eos));
    for (int i = 0; i < 3; i++) {
        auto* a = g.adopt(new Adder(1));
        auto* b = g.adopt(new Adder(10));  // fused behind a: same thread
        g.connect_queue(p, p->add_out_port(), a);
        g.connect_fused(a, a->add_out_port(), b);
        g.connect_queue(b, b->add_out_port(), c);
    }
    CHECK(g.num_threads() == 5);  // producer, 3 workers (adder pairs), collector
    g.run();
    g.wait();
    CHECK(cnt == N);
    CHECK(sum == N * (N - 1) / 2 + 11 * N);
    CHECK(eos == 3);  // one notification per input channel
}

// svc_bulk: a head node sees the messages found queued on a channel together, in order, never the end-of-stream token, and
// every message exactly once (what a GPU replica builds its coalescing of small host batches on)
struct BulkCollector : rt::Node {
    long next = 0, groups = 0, largest = 0, total = 0;
    bool in_order = true;
    void svc(void*, int) override { in_order = false; }  // (a head node is always fed through svc_bulk)
    void svc_bulk(void** msgs, size_t n, int ch) override {
        if (ch != 0 || n == 0) in_order = false;
        groups++;
        largest = std::max<long>(largest, (long)n);
        for (size_t i = 0; i < n; i++) {
            long* v = static_cast<long*>(msgs[i]);
            if (*v != next++) in_order = false;
            total++;
            delete v;
        }
        if (groups % 50 == 0) this_thread::sleep_for(chrono::microseconds(200));  // fall behind: the ring fills up
    }
};
static void test_svc_bulk() {
    rt::RuntimeConfig cfg;
    cfg.queue_capacity = 64;
    rt::Graph g(cfg);
    const long N = 100000;
    auto* p = g.adopt(new Producer(N));
    auto* c = static_cast<BulkCollector*>(g.adopt(new BulkCollector()));
    g.connect_queue(p, p->add_out_port(), c);
    g.run();
    g.wait();
    CHECK(c->in_order);
    CHECK(c->total == N);
    CHECK(c->groups <= N);
    CHECK(c->largest > 1 && c->largest <= 64);  // the consumer did fall behind, and a group never exceeds the bulk pop
}

// blocking mode: consumers sleep on their futex when the rings are empty, producers when a ring is full
static void test_graph_blocking() {
    rt::RuntimeConfig cfg;
    cfg.queue_capacity = 4;
    cfg.blocking = true;
    rt::Graph g(cfg);
    atomic<long> sum{0}, cnt{0};
    atomic<int> eos{0};
    const long N = 100000;
    auto* p = g.adopt(new Producer(N));
    auto* c = g.adopt(new Collector(&sum, &cnt, &eos));
    rt::Node* adders[2];
    for (int i = 0; i < 2; i++) {
        auto* a = g.adopt(new Adder(1));
        adders[i] = a;
        g.connect_queue(p, p->add_out_port(), a);
        g.connect_queue(a, a->add_out_port(), c);
    }
    g.run();
    g.wait();
    CHECK(cnt == N);
    CHECK(sum == N * (N - 1) / 2 + N);
    CHECK(eos == 2);
    // with 4-slot rings somebody must have slept, and the time spent in svc() was measured
    CHECK(p->counters().parks + c->counters().parks + adders[0]->counters().parks + adders[1]->counters().parks > 0);
    CHECK(c->counters().busy_ns > 0);
}

// ---- tuples with string keys -----------------------------------------------------------------------
struct skey_t {
    string key;
    uint64_t id = 0, ts = 0;
    long value = 0;
    tuple<string, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    void setControlFields(string k, uint64_t i, uint64_t t) {
        key = std::move(k);
        id = i;
        ts = t;
    }
};
static void test_string_keys() {
    // Source -> Accumulator (keyed) -> Key_Farm CB windows -> Sink, keys are strings (reference: test_mp_*_string.cpp)
    const vector<string> names = {"alpha", "beta", "gamma", "delta", "epsilon"};
    const long per_key = 3000;
    for (int run = 0; run < 2; run++) {
        PipeGraph graph("string_keys", Mode::DETERMINISTIC);  // count-based windows need DETERMINISTIC mode
        long sent = 0;
        vector<uint64_t> ids(names.size(), 0);
        Source src = Source_Builder([=](skey_t& t) mutable {
                         const size_t k = (size_t)sent % names.size();
                         t.setControlFields(names[k], ids[k]++, (uint64_t)sent);
                         t.value = 1;
                         sent++;
                         return sent < per_key * (long)names.size();
                     }).withParallelism(1).build();
        MultiPipe& mp = graph.add_source(src);
        Accumulator acc = Accumulator_Builder([](const skey_t& t, skey_t& a) { a.value += t.value; }).withParallelism(run == 0 ? 1 : 3).build();
        mp.add(acc);  // emits the running count per key: 1, 2, 3 ...
        Key_Farm kf = KeyFarm_Builder([](uint64_t, const skey_t& t, skey_t& out) { out.value += t.value; })
                          .withCBWindows(10, 10).withParallelism(2).build();
        mp.add(kf);
        atomic<long> total{0}, windows{0};
        Sink sink = Sink_Builder([&](optional<skey_t>& r) {
                        if (r) {
                            total += r->value;
                            windows++;
                        }
                    }).withParallelism(1).build();
        mp.add_sink(sink);
        graph.run();
        CHECK(windows == (long)names.size() * per_key / 10);
        CHECK(total == (long)names.size() * per_key * (per_key + 1) / 2);
    }
}

// ---- FlatFAT against a brute-force fold (non-commutative combine) -------------------------------------
static void test_flatfat() {
    struct S {
        string s;
        size_t key = 0;
        uint64_t id = 0, ts = 0;
        S() = default;
        explicit S(string v) : s(std::move(v)) {}
        tuple<size_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
        void setControlFields(size_t k, uint64_t i, uint64_t t) {
            key = k;
            id = i;
            ts = t;
        }
    };
    using comb_t = std::function<void(const S&, const S&, S&)>;
    const comb_t concat = [](const S& a, const S& b, S& out) { out.s = a.s + b.s; };
    FlatFAT<S, comb_t> fat(concat, 10);
    CHECK(fat.capacity() == 16);
    string model;
    S r;
    CHECK(!fat.getResult(r));
    unsigned seed = 7;
    for (int step = 0; step < 5000; step++) {
        seed = seed * 1103515245u + 12345u;
        const bool can_insert = fat.size() < fat.capacity();
        if (can_insert && ((seed >> 16) % 3 != 0 || fat.is_Empty())) {
            const char c = (char)('a' + (seed >> 8) % 26);
            fat.insert(S{string(1, c)});
            model.push_back(c);
        } else if (!fat.is_Empty()) {
            fat.remove();
            model.erase(model.begin());
        }
        const bool has = fat.getResult(r);
        CHECK(has == !model.empty());
        if (has && r.s != model) {
            CHECK(r.s == model);
            break;
        }
    }
    // bulk insert keeps the order too
    FlatFAT<S, comb_t> fat2(concat, 32);
    vector<S> many;
    string exp;
    for (int i = 0; i < 20; i++) {
        many.push_back(S{string(1, (char)('A' + i))});
        exp.push_back((char)('A' + i));
    }
    fat2.insert(many);
    CHECK(fat2.getResult(r) && r.s == exp);
    fat2.remove(5);
    CHECK(fat2.getResult(r) && r.s == exp.substr(5));
}

// ---- statistics + DOT dump ---------------------------------------------------------------------------
struct rec_t {
    size_t key = 0;
    uint64_t id = 0, ts = 0;
    long value = 0;
    tuple<size_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    void setControlFields(size_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};
static void test_stats_and_dot() {
    PipeGraph graph("stats", Mode::DEFAULT);
    long v = 0;
    Source src = Source_Builder([=](rec_t& t) mutable {
                     t.setControlFields((size_t)v % 3, (uint64_t)v / 3, (uint64_t)v);
                     t.value = v++;
                     return v < 9000;
                 }).withName("numbers").withParallelism(1).build();
    MultiPipe& mp = graph.add_source(src);
    Filter f = Filter_Builder([](rec_t& t) { return t.value % 3 == 0; }).withName("thirds").withParallelism(2).build();
    mp.add(f);
    Map m = Map_Builder([](rec_t& t) { t.value *= 2; }).withName("double").withParallelism(2).build();
    mp.chain(m);
    atomic<long> n{0};
    Sink s = Sink_Builder([&](optional<rec_t>& r) {
                 if (r) n++;
             }).withName("count").withParallelism(1).build();
    mp.add_sink(s);
    const string dot = graph.get_DOTGraph();
    for (const char* name : {"numbers", "thirds", "double", "count"}) CHECK(dot.find(name) != string::npos);
    CHECK(dot.find("digraph") != string::npos);
    CHECK(graph.get_ListOperators().size() == 4);
    CHECK(graph.getNumThreads() == 1 + 2 + 1);  // the Map is chained to the Filter
    graph.run();
    CHECK(n == 3000);
    long in_filter = 0, out_filter = 0, in_map = 0;
    for (auto& r : f.get_StatsRecords()) {
        in_filter += (long)r.inputs_received;
        out_filter += (long)r.outputs_sent;
    }
    for (auto& r : m.get_StatsRecords()) in_map += (long)r.inputs_received;
    CHECK(f.get_StatsRecords().size() == 2);
    CHECK(in_filter == 9000);
    CHECK(out_filter == 3000);
    CHECK(in_map == 3000);
    // the filter replicas are thread heads: their service time is measured; the chained map shares their threads
    for (auto& r : f.get_StatsRecords()) CHECK(r.service_time_measured && r.service_time_us > 0.0 && r.busy_time_us <= r.end_time_us - r.start_time_us);
    for (auto& r : m.get_StatsRecords()) CHECK(!r.service_time_measured);
}

// ---- late tuples are dropped and accounted (DEFAULT mode, time-based windows) -----------------------------------
static void test_dropped_tuples() {
    PipeGraph graph("late", Mode::DEFAULT);
    long v = 0;
    // timestamps advance by 10; every 100th tuple is 5000 time units late (several windows in the past)
    Source src = Source_Builder([=](rec_t& t) mutable {
                     const uint64_t ts = (v % 100 == 99 && v > 1000) ? (uint64_t)(v * 10 - 5000) : (uint64_t)(v * 10);
                     t.setControlFields(0, (uint64_t)v, ts);
                     t.value = 1;
                     return ++v < 5000;
                 }).withParallelism(1).build();
    MultiPipe& mp = graph.add_source(src);
    Key_Farm kf = KeyFarm_Builder([](uint64_t, const rec_t& t, rec_t& out) { out.value += t.value; })
                      .withTBWindows(std::chrono::microseconds(200), std::chrono::microseconds(100)).withParallelism(1).build();
    mp.add(kf);
    atomic<long> counted{0};
    Sink s = Sink_Builder([&](optional<rec_t>& r) {
                 if (r) counted += r->value;
             }).withParallelism(1).build();
    mp.add_sink(s);
    graph.run();
    const long late = (5000 - 1000) / 100;  // tuples 1099, 1199, ... 4999
    CHECK((long)kf.getNumDroppedTuples() == late);
    CHECK(counted == 2 * (5000 - late) - 10);  // every on-time tuple is in two windows, except the first 10 (window [-100, 100) does not exist)
}

// ---- illegal topologies must be refused -----------------------------------------------
static int illegal(int which) {
    PipeGraph graph("illegal", Mode::DEFAULT);
    auto gen = [](rec_t& t) {
        t.value = 1;
        return false;
    };
    Source src = Source_Builder(gen).withParallelism(1).build();
    Source src2 = Source_Builder(gen).withParallelism(1).build();
    Map m = Map_Builder([](rec_t& t) { t.value++; }).withParallelism(1).build();
    Sink s = Sink_Builder([](optional<rec_t>&) {}).withParallelism(1).build();
    MultiPipe& mp = graph.add_source(src);
    switch (which) {
        case 0:  // operator used twice
            mp.add(m);
            mp.add(m);
            break;
        case 1:  // operator after a sink
            mp.add_sink(s);
            mp.add(m);
            break;
        case 2:  // select on a MultiPipe that was not split
            mp.select(0);
            break;
        case 3:  // adding to a MultiPipe that has been split
            mp.split([](const rec_t&) -> size_t { return 0; }, 2);
            mp.add(m);
            break;
        case 4:  // running a graph with an unterminated MultiPipe
            mp.add(m);
            graph.run();
            break;
        case 5: {  // adding to a MultiPipe that has been merged
            MultiPipe& other = graph.add_source(src2);
            mp.merge(other);
            mp.add(m);
            break;
        }
        case 6: {  // type mismatch between consecutive operators
            Map ms = Map_Builder([](skey_t& t) { t.value++; }).withParallelism(1).build();
            mp.add(ms);
            break;
        }
        case 7:  // Key_FFAT on a tumbling window is refused like in the reference (sliding windows only)... CB window of length zero
        {
            Key_Farm kf = KeyFarm_Builder([](uint64_t, const rec_t& t, rec_t& out) { out.value += t.value; }).withCBWindows(0, 1).build();
            mp.add(kf);
            break;
        }
        default: return 0;
    }
    return 0;  // reaching this point means the library accepted the topology
}

// tuple pool: producers allocate, consumers destroy (the pipeline pattern), blocks are recycled, values survive,
// destructors run exactly once, over-aligned / large types fall back to the heap
struct Tracked {
    static std::atomic<long> alive;
    long v;
    std::string s;
    explicit Tracked(long x) : v(x), s("tuple " + std::to_string(x)) { alive++; }
    Tracked(Tracked&& o) : v(o.v), s(std::move(o.s)) { alive++; }
    Tracked(const Tracked& o) : v(o.v), s(o.s) { alive++; }
    ~Tracked() { alive--; }
};
std::atomic<long> Tracked::alive{0};
struct alignas(64) OverAligned {
    long v;
};
struct Huge {
    char bytes[8192];
};

static void test_tuple_pool() {
    static_assert(mem::poolable<Tracked>, "small types are pooled");
    static_assert(!mem::poolable<OverAligned> && !mem::poolable<Huge>, "these go to the heap");
    constexpr long N = 200000;
    constexpr int PAIRS = 3;
    std::vector<std::unique_ptr<rt::SPSCQueue>> qs;
    for (int i = 0; i < PAIRS; i++) qs.emplace_back(new rt::SPSCQueue(256));
    std::atomic<long> sum{0};
    std::vector<std::thread> th;
    for (int i = 0; i < PAIRS; i++) {
        th.emplace_back([&, i] {
            for (long k = 0; k < N; k++) {
                Tracked* t = tuple_new<Tracked>(k);
                while (!qs[i]->try_push(t)) std::this_thread::yield();
            }
        });
        th.emplace_back([&, i] {
            long local = 0;
            for (long k = 0; k < N; k++) {
                void* p = nullptr;
                while (!qs[i]->try_pop(p)) std::this_thread::yield();
                Tracked* t = static_cast<Tracked*>(p);
                if (t->s != "tuple " + std::to_string(t->v)) local -= 1000000;
                local += t->v;
                tuple_delete(t);
            }
            sum += local;
        });
    }
    for (auto& t : th) t.join();
    CHECK(sum.load() == PAIRS * (N * (N - 1) / 2));
    CHECK(Tracked::alive.load() == 0);
    // same thread, interleaved lifetimes across block boundaries
    std::vector<Tracked*> held;
    for (long k = 0; k < 5000; k++) {
        held.push_back(tuple_new<Tracked>(k));
        if (k % 3 == 0) {
            tuple_delete(held.front());
            held.erase(held.begin());
        }
    }
    long s2 = 0;
    for (Tracked* t : held) {
        s2 += t->v;
        tuple_delete(t);
    }
    CHECK(s2 > 0 && Tracked::alive.load() == 0);
    OverAligned* oa = tuple_new<OverAligned>();
    CHECK(reinterpret_cast<uintptr_t>(oa) % 64 == 0);
    tuple_delete(oa);
    tuple_delete(tuple_new<Huge>());
    // a user-allocated object handed over is moved into a pooled one
    Tracked* mine = new Tracked(77);
    Tracked* adopted = tuple_adopt(mine);
    CHECK(adopted->v == 77 && adopted->s == "tuple 77" && Tracked::alive.load() == 1);
    tuple_delete(adopted);
    CHECK(Tracked::alive.load() == 0);
}

// four "sources" of very different speed advance at most `slack` batches apart; a shutdown releases the waiters
static void test_skew_bound() {
    constexpr size_t N = 4;
    constexpr uint64_t SLACK = 3, BATCHES = 400;
    wfbench::SkewBound sb(N, SLACK);
    wfbench::StepGate gates[N];
    std::atomic<uint64_t> progress[N];
    for (auto& p : progress) p = 0;
    std::atomic<uint64_t> worst{0};
    vector<thread> ts;
    for (size_t i = 0; i < N; i++)
        ts.emplace_back([&, i] {
            for (uint64_t b = 0; b < BATCHES; b++) {
                sb.wait_turn(i, gates[i]);
                uint64_t slowest = UINT64_MAX;
                for (auto& p : progress) slowest = std::min<uint64_t>(slowest, p.load());
                const uint64_t ahead = progress[i].load() - slowest;  // (slowest may have moved on since: an upper bound is fine)
                uint64_t w = worst.load();
                while (ahead > w && !worst.compare_exchange_weak(w, ahead)) {}
                if (i == 0 && b % 8 == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));  // the slow one
                progress[i]++;
                sb.batch_done(i);
            }
        });
    for (auto& t : ts) t.join();
    for (auto& p : progress) CHECK(p.load() == BATCHES);
    CHECK(worst.load() <= SLACK);
    // a source blocked behind a peer that never moves again leaves when its gate shuts down
    wfbench::SkewBound stuck(2, 1);
    wfbench::StepGate g2[2];
    stuck.batch_done(1);
    thread waiter([&] { stuck.wait_turn(1, g2[1]); });
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    g2[1].shutdown();
    waiter.join();
    // sources of three "ranks": every rank has a SkewBound of its own over the same shared rows (the control board's progress
    // table between processes) and advances its row only; the bound holds over all ranks' sources
    constexpr size_t RANKS = 3, PER = 2;
    std::atomic<uint64_t> table[RANKS][PER];
    for (auto& row : table)
        for (auto& c : row) c = 0;
    vector<std::unique_ptr<wfbench::SkewBound>> bounds;
    for (size_t r = 0; r < RANKS; r++) bounds.emplace_back(new wfbench::SkewBound({table[0], table[1], table[2]}, r, PER, SLACK));
    wfbench::StepGate g3[RANKS * PER];
    std::atomic<uint64_t> worst3{0};
    vector<thread> ts3;
    for (size_t r = 0; r < RANKS; r++)
        for (size_t i = 0; i < PER; i++)
            ts3.emplace_back([&, r, i] {
                for (uint64_t b = 0; b < BATCHES; b++) {
                    bounds[r]->wait_turn(i, g3[r * PER + i]);
                    uint64_t slowest = UINT64_MAX;
                    for (auto& row : table)
                        for (auto& c : row) slowest = std::min<uint64_t>(slowest, c.load());
                    const uint64_t ahead = table[r][i].load() - slowest;
                    uint64_t w = worst3.load();
                    while (ahead > w && !worst3.compare_exchange_weak(w, ahead)) {}
                    if (r == 2 && i == 1 && b % 8 == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
                    bounds[r]->batch_done(i);
                }
            });
    for (auto& t : ts3) t.join();
    for (auto& row : table)
        for (auto& c : row) CHECK(c.load() == BATCHES);
    CHECK(worst3.load() < SLACK);
}

int main(int argc, char** argv) {
    if (argc == 3 && strcmp(argv[1], "illegal") == 0) return illegal(atoi(argv[2]));
    test_skew_bound();
    test_spsc();
    test_graph();
    test_svc_bulk();
    test_graph_blocking();
    test_string_keys();
    test_flatfat();
    test_stats_and_dot();
    test_dropped_tuples();
    test_tuple_pool();
    cout << (failures == 0 ? "ALL OK" : "FAILED") << endl;
    return failures == 0 ? 0 : 1;
}
