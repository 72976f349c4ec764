// Split / merge / DAG regression program.
//
// Covers what the reference's src/merge_test (4 programs), src/split_test (5) and
// src/graph_test (9) cover - merging independent MultiPipes (incl. merge -> operator
// -> merge again), splitting by a user function into 2-3 branches with single and
// multiple destinations, nested splits, partial / full / general merges of split
// branches, thread-count accounting - with two differences: every scenario draws
// random parallelism degrees per run AND is checked against a value computed
// sequentially here (the reference only checks run-to-run equality), and the
// non-contiguous partial merge the reference rejects (pipegraph.hpp:901-908) is
// legal here and tested.
//
//   test_graphs [-r runs] [-n stream length] [-t scenario]    (default: all scenarios)
#include <atomic>
#include <functional>
#include <iostream>
#include <map>
#include <numeric>
#include <optional>
#include <random>
#include <string>
#include <tuple>
#include <vector>

#include <unistd.h>

#include <wf/windflow.hpp>

using namespace wf;
using namespace std;

struct rec_t {
    size_t key = 0;
    uint64_t id = 0, ts = 0;
    int64_t value = 0;
    rec_t() = default;
    tuple<size_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    void setControlFields(size_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};

constexpr size_t N_KEYS = 4;

// emits sign * v for v in [1, n]; key = v % N_KEYS; ids progressive per key; ts = 3 v + phase (distinct across sources)
struct Gen {
    int64_t n, sign;
    uint64_t phase;
    int64_t v = 0;
    vector<uint64_t> ids = vector<uint64_t>(N_KEYS, 0);
    Gen(int64_t n_, int64_t sign_, uint64_t phase_) : n(n_), sign(sign_), phase(phase_) {}
    bool operator()(rec_t& r) {
        v++;
        r.key = (size_t)(v % (int64_t)N_KEYS);
        r.id = ids[r.key]++;
        r.ts = 3 * (uint64_t)v + phase;
        r.value = sign * v;
        return v < n;
    }
};

// ---- sequential model of a stream: the vector of records in source order ----------------------
using vec = vector<rec_t>;
static vec model(int64_t n, int64_t sign, uint64_t phase) {
    Gen g(n, sign, phase);
    vec out;
    bool more = true;
    while (more) {
        rec_t r;
        more = g(r);
        out.push_back(r);
    }
    return out;
}
static vec vmap(vec v, function<int64_t(int64_t)> f) {
    for (auto& r : v) r.value = f(r.value);
    return v;
}
static vec vfilter(const vec& v, function<bool(int64_t)> f) {
    vec o;
    for (auto& r : v)
        if (f(r.value)) o.push_back(r);
    return o;
}
static vec vcat(vec a, const vec& b) {
    a.insert(a.end(), b.begin(), b.end());
    return a;
}
static long vsum(const vec& v) {
    long s = 0;
    for (auto& r : v) s += r.value;
    return s;
}

// ---- operators ------------------------------------------------------------------------------------
static int64_t f_dbl(int64_t x) { return 2 * x; }
static int64_t f_inc(int64_t x) { return x + 1; }
static int64_t f_neg(int64_t x) { return -x; }
static bool p_even(int64_t x) { return x % 2 == 0; }
static bool p_m3(int64_t x) { return x % 3 == 0; }

struct Totals {
    atomic<long> sum{0}, count{0}, order_violations{0};
};

struct Rig {
    PipeGraph graph;
    mt19937& rng;
    vector<unique_ptr<Basic_Operator>> ops;
    size_t replicas = 0;       // sum of the parallelism of every operator
    size_t chained_replicas = 0;
    Rig(const string& name, Mode m, mt19937& r) : graph(name, m), rng(r) {}
    size_t par() { return uniform_int_distribution<size_t>(1, 4)(rng); }
    template <typename T>
    T& keep(T* p, size_t n) {
        ops.emplace_back(p);
        replicas += n;
        return *p;
    }
    MultiPipe& source(int64_t n, int64_t sign, uint64_t phase, size_t p = 0) {
        if (p == 0) p = 1;  // one generator per logical stream: the model is a single sequence
        return graph.add_source(keep(Source_Builder(Gen(n, sign, phase)).withName("src").withParallelism(p).build_ptr(), p));
    }
    MultiPipe& map(MultiPipe& mp, int64_t (*f)(int64_t), const string& name = "map") {
        const size_t p = par();
        return mp.add(keep(Map_Builder([f](rec_t& r) { r.value = f(r.value); }).withName(name).withParallelism(p).build_ptr(), p));
    }
    MultiPipe& filter(MultiPipe& mp, bool (*f)(int64_t), const string& name = "filter") {
        const size_t p = par();
        return mp.add(keep(Filter_Builder([f](rec_t& r) { return f(r.value); }).withName(name).withParallelism(p).build_ptr(), p));
    }
    // a sink of parallelism 1 can also check that timestamps never go backwards (DETERMINISTIC mode)
    MultiPipe& sink(MultiPipe& mp, Totals& t, bool check_order = false) {
        const size_t p = check_order ? 1 : par();
        uint64_t last_ts = 0;
        Totals* tp = &t;
        return mp.add_sink(keep(Sink_Builder([tp, last_ts, check_order](optional<rec_t>& r) mutable {
                                    if (!r) return;
                                    tp->sum += r->value;
                                    tp->count++;
                                    if (check_order && r->ts < last_ts) tp->order_violations++;
                                    last_ts = r->ts;
                                }).withName("sink").withParallelism(p).build_ptr(),
                                p));
    }
};

struct Expect {
    long sum = 0, count = 0;
    Expect() = default;
    Expect(const vec& v) : sum(vsum(v)), count((long)v.size()) {}
    Expect operator+(const Expect& o) const {
        Expect e;
        e.sum = sum + o.sum;
        e.count = count + o.count;
        return e;
    }
};

struct Scenario {
    string name;
    // builds and runs the graph; returns {measured, expected, thread check ok}
    function<tuple<Expect, Expect, bool>(int64_t n, mt19937& rng)> run;
};

static Expect got(const Totals& t) {
    Expect e;
    e.sum = t.sum;
    e.count = t.count;
    return e;
}

static vector<Scenario> scenarios() {
    vector<Scenario> S;
    // ---------------- merges ------------------------------------------------------------------
    S.push_back({"merge_1", [](int64_t n, mt19937& rng) {
                     Rig g("merge_1", Mode::DEFAULT, rng);
                     Totals t;
                     MultiPipe& a = g.map(g.source(n, 1, 0), f_dbl);
                     MultiPipe& b = g.filter(g.source(n, -1, 1), p_even);
                     MultiPipe& m = a.merge(b);
                     g.sink(g.map(m, f_inc), t);
                     g.graph.run();
                     Expect e(vmap(vcat(vmap(model(n, 1, 0), f_dbl), vfilter(model(n, -1, 1), p_even)), f_inc));
                     return make_tuple(got(t), e, g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"merge_2", [](int64_t n, mt19937& rng) {
                     Rig g("merge_2", Mode::DEFAULT, rng);
                     Totals t;
                     MultiPipe& a = g.source(n, 1, 0);
                     MultiPipe& b = g.source(2 * n, 1, 1);
                     MultiPipe& c = g.source(n / 2 + 1, -1, 2);
                     MultiPipe& m = a.merge(b, c);
                     g.sink(g.filter(m, p_even), t);
                     g.graph.run();
                     Expect e(vfilter(vcat(vcat(model(n, 1, 0), model(2 * n, 1, 1)), model(n / 2 + 1, -1, 2)), p_even));
                     return make_tuple(got(t), e, g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"merge_3", [](int64_t n, mt19937& rng) {
                     Rig g("merge_3", Mode::DEFAULT, rng);
                     Totals t;
                     MultiPipe& ab = g.source(n, 1, 0).merge(g.source(n, -1, 1));
                     MultiPipe& abm = g.map(ab, f_dbl);
                     MultiPipe& c = g.map(g.source(n, 1, 2), f_inc);
                     g.sink(abm.merge(c), t);
                     g.graph.run();
                     Expect e(vcat(vmap(vcat(model(n, 1, 0), model(n, -1, 1)), f_dbl), vmap(model(n, 1, 2), f_inc)));
                     return make_tuple(got(t), e, g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"merge_4", [](int64_t n, mt19937& rng) {  // 3-way merge -> filter -> merge again (reference test_merge_4)
                     Rig g("merge_4", Mode::DETERMINISTIC, rng);
                     Totals t;
                     MultiPipe& m3 = g.source(n, 1, 0).merge(g.source(n, 1, 1), g.source(n, -1, 2));
                     MultiPipe& f = g.filter(m3, p_m3);
                     MultiPipe& d = g.source(n, -1, 1000000007);
                     MultiPipe& m = f.merge(d);
                     g.sink(g.map(m, f_inc), t, true);
                     g.graph.run();
                     Expect e(vmap(vcat(vfilter(vcat(vcat(model(n, 1, 0), model(n, 1, 1)), model(n, -1, 2)), p_m3), model(n, -1, 0)), f_inc));
                     auto r = got(t);
                     if (t.order_violations != 0) r.count = -1;  // DETERMINISTIC: the sink must see timestamps in order
                     return make_tuple(r, e, g.graph.getNumThreads() == g.replicas);
                 }});
    // ---------------- splits ------------------------------------------------------------------
    S.push_back({"split_1", [](int64_t n, mt19937& rng) {
                     Rig g("split_1", Mode::DEFAULT, rng);
                     Totals t0, t1;
                     MultiPipe& s = g.source(n, 1, 0);
                     s.split([](const rec_t& r) -> size_t { return r.value % 2 == 0 ? 0 : 1; }, 2);
                     g.sink(g.map(s.select(0), f_dbl), t0);
                     g.sink(g.map(s.select(1), f_inc), t1);
                     g.graph.run();
                     const vec a = model(n, 1, 0);
                     Expect e0(vmap(vfilter(a, p_even), f_dbl)), e1(vmap(vfilter(a, [](int64_t x) { return x % 2 != 0; }), f_inc));
                     const bool each = got(t0).sum == e0.sum && got(t1).sum == e1.sum;
                     return make_tuple(got(t0) + got(t1), e0 + e1, each && g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"split_2", [](int64_t n, mt19937& rng) {  // three branches, multiples of 5 are copied to branch 2 as well
                     Rig g("split_2", Mode::DEFAULT, rng);
                     Totals t0, t1, t2;
                     MultiPipe& s = g.map(g.source(n, 1, 0), f_inc);
                     s.split([](const rec_t& r) -> vector<size_t> {
                         const size_t b = (size_t)(r.value % 3);
                         if (r.value % 5 == 0 && b != 2) return {b, 2};
                         return {b};
                     }, 3);
                     g.sink(s.select(0), t0);
                     g.sink(g.map(s.select(1), f_dbl), t1);
                     g.sink(g.filter(s.select(2), p_even), t2);
                     g.graph.run();
                     const vec a = vmap(model(n, 1, 0), f_inc);
                     Expect e0(vfilter(a, [](int64_t x) { return x % 3 == 0; }));
                     Expect e1(vmap(vfilter(a, [](int64_t x) { return x % 3 == 1; }), f_dbl));
                     Expect e2(vfilter(vfilter(a, [](int64_t x) { return x % 3 == 2 || x % 5 == 0; }), p_even));
                     const bool each = got(t0).sum == e0.sum && got(t1).sum == e1.sum && got(t2).sum == e2.sum;
                     return make_tuple(got(t0) + got(t1) + got(t2), e0 + e1 + e2, each && g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"split_3", [](int64_t n, mt19937& rng) {  // nested split of a branch (reference test_split_3)
                     Rig g("split_3", Mode::DEFAULT, rng);
                     Totals t00, t01, t1;
                     MultiPipe& s = g.source(n, 1, 0);
                     s.split([](const rec_t& r) -> size_t { return r.value % 2 == 0 ? 0 : 1; }, 2);
                     MultiPipe& b0 = g.map(s.select(0), f_inc);
                     b0.split([](const rec_t& r) -> size_t { return r.value % 3 == 0 ? 0 : 1; }, 2);
                     g.sink(b0.select(0), t00);
                     g.sink(g.map(b0.select(1), f_dbl), t01);
                     g.sink(g.map(s.select(1), f_neg), t1);
                     g.graph.run();
                     const vec a = model(n, 1, 0);
                     const vec ev = vmap(vfilter(a, p_even), f_inc);
                     Expect e00(vfilter(ev, p_m3)), e01(vmap(vfilter(ev, [](int64_t x) { return x % 3 != 0; }), f_dbl));
                     Expect e1(vmap(vfilter(a, [](int64_t x) { return x % 2 != 0; }), f_neg));
                     const bool each = got(t00).sum == e00.sum && got(t01).sum == e01.sum && got(t1).sum == e1.sum;
                     return make_tuple(got(t00) + got(t01) + got(t1), e00 + e01 + e1, each && g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"split_4", [](int64_t n, mt19937& rng) {  // keyed stateful operators on the branches
                     Rig g("split_4", Mode::DETERMINISTIC, rng);
                     Totals tw, ta;
                     MultiPipe& s = g.source(n, 1, 0);
                     s.split([](const rec_t& r) -> size_t { return r.key < 2 ? 0 : 1; }, 2);
                     const size_t pw = g.par(), pa = g.par();
                     s.select(0).add(g.keep(KeyFarm_Builder([](uint64_t, const rec_t& r, rec_t& out) { out.value += r.value; })
                                                .withCBWindows(6, 2).withName("kf").withParallelism(pw).build_ptr(), pw));
                     g.sink(s.select(0), tw);
                     s.select(1).add(g.keep(Accumulator_Builder([](const rec_t& r, rec_t& acc) { acc.value += r.value; })
                                                .withName("acc").withParallelism(pa).build_ptr(), pa));
                     g.sink(s.select(1), ta);
                     g.graph.run();
                     const vec a = model(n, 1, 0);
                     Expect ew, ea;
                     for (size_t k = 0; k < N_KEYS; k++) {
                         vector<int64_t> v;
                         for (auto& r : a)
                             if (r.key == k) v.push_back(r.value);
                         if (k < 2) {
                             for (size_t w = 0; w * 2 < v.size(); w++) {
                                 for (size_t i = w * 2; i < min(v.size(), w * 2 + 6); i++) ew.sum += v[i];
                                 ew.count++;
                             }
                         } else {
                             int64_t run = 0;
                             for (auto x : v) {
                                 run += x;
                                 ea.sum += run;
                                 ea.count++;
                             }
                         }
                     }
                     const bool each = got(tw).sum == ew.sum && got(ta).sum == ea.sum;
                     return make_tuple(got(tw) + got(ta), ew + ea, each && g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"split_5", [](int64_t n, mt19937& rng) {  // DETERMINISTIC: order survives split + shuffles
                     Rig g("split_5", Mode::DETERMINISTIC, rng);
                     Totals t0, t1;
                     MultiPipe& s = g.map(g.source(n, 1, 0), f_dbl);
                     s.split([](const rec_t& r) -> size_t { return r.value % 4 == 0 ? 0 : 1; }, 2);
                     g.sink(g.filter(g.map(s.select(0), f_inc), p_m3), t0, true);
                     g.sink(g.map(s.select(1), f_inc), t1, true);
                     g.graph.run();
                     const vec a = vmap(model(n, 1, 0), f_dbl);
                     Expect e0(vfilter(vmap(vfilter(a, [](int64_t x) { return x % 4 == 0; }), f_inc), p_m3));
                     Expect e1(vmap(vfilter(a, [](int64_t x) { return x % 4 != 0; }), f_inc));
                     auto r = got(t0) + got(t1);
                     if (t0.order_violations != 0 || t1.order_violations != 0) r.count = -1;
                     return make_tuple(r, e0 + e1, g.graph.getNumThreads() == g.replicas);
                 }});
    // ---------------- DAGs --------------------------------------------------------------------
    S.push_back({"graph_1", [](int64_t n, mt19937& rng) {  // split 3, partial merge of two siblings
                     Rig g("graph_1", Mode::DEFAULT, rng);
                     Totals tm, t2;
                     MultiPipe& s = g.source(n, 1, 0);
                     s.split([](const rec_t& r) -> size_t { return (size_t)(r.value % 3); }, 3);
                     MultiPipe& b0 = g.map(s.select(0), f_dbl);
                     MultiPipe& b1 = g.map(s.select(1), f_inc);
                     MultiPipe& b2 = g.map(s.select(2), f_neg);
                     g.sink(g.map(b0.merge(b1), f_inc), tm);
                     g.sink(b2, t2);
                     g.graph.run();
                     const vec a = model(n, 1, 0);
                     auto br = [&](int64_t m) { return vfilter(a, [m](int64_t x) { return x % 3 == m; }); };
                     Expect em(vmap(vcat(vmap(br(0), f_dbl), vmap(br(1), f_inc)), f_inc)), e2(vmap(br(2), f_neg));
                     const bool each = got(tm).sum == em.sum && got(t2).sum == e2.sum;
                     return make_tuple(got(tm) + got(t2), em + e2, each && g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"graph_2", [](int64_t n, mt19937& rng) {  // full merge of all the branches of a split
                     Rig g("graph_2", Mode::DEFAULT, rng);
                     Totals t;
                     MultiPipe& s = g.source(n, 1, 0);
                     s.split([](const rec_t& r) -> size_t { return r.value % 2 == 0 ? 0 : 1; }, 2);
                     MultiPipe& b0 = g.map(s.select(0), f_dbl);
                     MultiPipe& b1 = g.filter(s.select(1), p_m3);
                     g.sink(g.map(b0.merge(b1), f_inc), t);
                     g.graph.run();
                     const vec a = model(n, 1, 0);
                     Expect e(vmap(vcat(vmap(vfilter(a, p_even), f_dbl), vfilter(vfilter(a, [](int64_t x) { return x % 2 != 0; }), p_m3)), f_inc));
                     return make_tuple(got(t), e, g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"graph_3", [](int64_t n, mt19937& rng) {  // a split branch merged with an independent source
                     Rig g("graph_3", Mode::DEFAULT, rng);
                     Totals tm, t1;
                     MultiPipe& s = g.source(n, 1, 0);
                     s.split([](const rec_t& r) -> size_t { return r.value % 2 == 0 ? 0 : 1; }, 2);
                     MultiPipe& b0 = g.map(s.select(0), f_dbl);
                     MultiPipe& other = g.map(g.source(n, -1, 1), f_inc);
                     g.sink(b0.merge(other), tm);
                     g.sink(s.select(1), t1);
                     g.graph.run();
                     const vec a = model(n, 1, 0);
                     Expect em(vcat(vmap(vfilter(a, p_even), f_dbl), vmap(model(n, -1, 1), f_inc)));
                     Expect e1(vfilter(a, [](int64_t x) { return x % 2 != 0; }));
                     const bool each = got(tm).sum == em.sum && got(t1).sum == e1.sum;
                     return make_tuple(got(tm) + got(t1), em + e1, each && g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"graph_4", [](int64_t n, mt19937& rng) {  // nested split whose children are merged back, then the pipe goes on
                     Rig g("graph_4", Mode::DEFAULT, rng);
                     Totals tm, t1;
                     MultiPipe& s = g.source(n, 1, 0);
                     s.split([](const rec_t& r) -> size_t { return r.value % 2 == 0 ? 0 : 1; }, 2);
                     MultiPipe& b0 = g.map(s.select(0), f_inc);  // (a branch must hold an operator before it can be split again)
                     b0.split([](const rec_t& r) -> size_t { return r.value % 4 == 1 ? 0 : 1; }, 2);
                     MultiPipe& c0 = g.map(b0.select(0), f_dbl);
                     MultiPipe& c1 = g.map(b0.select(1), f_neg);
                     g.sink(g.filter(c0.merge(c1), p_m3), tm);
                     g.sink(g.map(s.select(1), f_inc), t1);
                     g.graph.run();
                     const vec a = model(n, 1, 0);
                     const vec ev = vmap(vfilter(a, p_even), f_inc);
                     Expect em(vfilter(vcat(vmap(vfilter(ev, [](int64_t x) { return x % 4 == 1; }), f_dbl),
                                            vmap(vfilter(ev, [](int64_t x) { return x % 4 != 1; }), f_neg)), p_m3));
                     Expect e1(vmap(vfilter(a, [](int64_t x) { return x % 2 != 0; }), f_inc));
                     const bool each = got(tm).sum == em.sum && got(t1).sum == e1.sum;
                     return make_tuple(got(tm) + got(t1), em + e1, each && g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"graph_5", [](int64_t n, mt19937& rng) {  // two split sources, branches merged pairwise
                     Rig g("graph_5", Mode::DEFAULT, rng);
                     Totals te, to;
                     MultiPipe& a = g.source(n, 1, 0);
                     MultiPipe& b = g.source(n, -1, 1);
                     auto parity = [](const rec_t& r) -> size_t { return r.value % 2 == 0 ? 0 : 1; };
                     a.split(parity, 2);
                     b.split(parity, 2);
                     g.sink(g.map(a.select(0).merge(b.select(0)), f_dbl), te);
                     g.sink(g.map(a.select(1).merge(b.select(1)), f_inc), to);
                     g.graph.run();
                     const vec all = vcat(model(n, 1, 0), model(n, -1, 1));
                     Expect ee(vmap(vfilter(all, p_even), f_dbl)), eo(vmap(vfilter(all, [](int64_t x) { return x % 2 != 0; }), f_inc));
                     const bool each = got(te).sum == ee.sum && got(to).sum == eo.sum;
                     return make_tuple(got(te) + got(to), ee + eo, each && g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"graph_6", [](int64_t n, mt19937& rng) {  // non-contiguous partial merge (branches 0 and 2 of 3)
                     Rig g("graph_6", Mode::DEFAULT, rng);
                     Totals tm, t1;
                     MultiPipe& s = g.source(n, 1, 0);
                     s.split([](const rec_t& r) -> size_t { return (size_t)(r.value % 3); }, 3);
                     MultiPipe& b0 = g.map(s.select(0), f_dbl);
                     MultiPipe& b2 = g.map(s.select(2), f_inc);
                     g.sink(b0.merge(b2), tm);
                     g.sink(g.map(s.select(1), f_neg), t1);
                     g.graph.run();
                     const vec a = model(n, 1, 0);
                     auto br = [&](int64_t m) { return vfilter(a, [m](int64_t x) { return x % 3 == m; }); };
                     Expect em(vcat(vmap(br(0), f_dbl), vmap(br(2), f_inc))), e1(vmap(br(1), f_neg));
                     const bool each = got(tm).sum == em.sum && got(t1).sum == e1.sum;
                     return make_tuple(got(tm) + got(t1), em + e1, each && g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"graph_7", [](int64_t n, mt19937& rng) {  // five branches, two partial merges, one more merge (reference test_graph_7)
                     Rig g("graph_7", Mode::DEFAULT, rng);
                     Totals ta, tb;
                     MultiPipe& s = g.map(g.source(n, 1, 0), f_inc);
                     s.split([](const rec_t& r) -> size_t { return (size_t)(r.value % 5); }, 5);
                     vector<MultiPipe*> b;
                     for (size_t i = 0; i < 5; i++) b.push_back(&g.map(s.select(i), i % 2 ? f_dbl : f_neg));
                     MultiPipe& m01 = g.filter(b[0]->merge(*b[1]), p_even);
                     MultiPipe& m23 = g.map(b[2]->merge(*b[3]), f_inc);
                     g.sink(m01.merge(m23), ta);
                     g.sink(*b[4], tb);
                     g.graph.run();
                     const vec a = vmap(model(n, 1, 0), f_inc);
                     auto br = [&](int64_t m) { return vmap(vfilter(a, [m](int64_t x) { return x % 5 == m; }), m % 2 ? f_dbl : f_neg); };
                     Expect ea(vcat(vfilter(vcat(br(0), br(1)), p_even), vmap(vcat(br(2), br(3)), f_inc))), eb(br(4));
                     const bool each = got(ta).sum == ea.sum && got(tb).sum == eb.sum;
                     return make_tuple(got(ta) + got(tb), ea + eb, each && g.graph.getNumThreads() == g.replicas);
                 }});
    S.push_back({"graph_8", [](int64_t n, mt19937& rng) {  // chained operators inside branches: fewer threads than replicas
                     Rig g("graph_8", Mode::DEFAULT, rng);
                     Totals t0, t1;
                     const size_t p = g.par();
                     MultiPipe& s = g.source(n, 1, 0);
                     s.split([](const rec_t& r) -> size_t { return r.value % 2 == 0 ? 0 : 1; }, 2);
                     auto add_chain = [&](MultiPipe& mp, int64_t (*f)(int64_t), Totals& t) {
                         mp.add(g.keep(Map_Builder([f](rec_t& r) { r.value = f(r.value); }).withParallelism(p).build_ptr(), p));
                         mp.chain(g.keep(Filter_Builder([](rec_t& r) { return r.value % 3 != 0; }).withParallelism(p).build_ptr(), p));
                         Totals* tp = &t;
                         mp.chain_sink(g.keep(Sink_Builder([tp](optional<rec_t>& r) {
                                                  if (r) {
                                                      tp->sum += r->value;
                                                      tp->count++;
                                                  }
                                              }).withParallelism(p).build_ptr(), p));
                         g.chained_replicas += 2 * p;
                     };
                     add_chain(s.select(0), f_dbl, t0);
                     add_chain(s.select(1), f_inc, t1);
                     g.graph.run();
                     const vec a = model(n, 1, 0);
                     auto not3 = [](int64_t x) { return x % 3 != 0; };
                     Expect e0(vfilter(vmap(vfilter(a, p_even), f_dbl), not3));
                     Expect e1(vfilter(vmap(vfilter(a, [](int64_t x) { return x % 2 != 0; }), f_inc), not3));
                     const bool each = got(t0).sum == e0.sum && got(t1).sum == e1.sum;
                     return make_tuple(got(t0) + got(t1), e0 + e1, each && g.graph.getNumThreads() == g.replicas - g.chained_replicas);
                 }});
    S.push_back({"graph_9", [](int64_t n, mt19937& rng) {  // DETERMINISTIC: merged sources feed time-based keyed windows
                     Rig g("graph_9", Mode::DETERMINISTIC, rng);
                     Totals t;
                     MultiPipe& m = g.source(n, 1, 0).merge(g.source(n, 1, 1));
                     const size_t pw = g.par();
                     MultiPipe& f = g.filter(m, p_even);
                     f.add(g.keep(KeyFarm_Builder([](uint64_t, const Iterable<rec_t>& in, rec_t& out) {
                                      for (const rec_t& r : in) out.value += r.value;
                                  }).withTBWindows(chrono::microseconds(90), chrono::microseconds(30)).withName("kf_tb").withParallelism(pw).build_ptr(),
                                  pw));
                     g.sink(f, t);
                     g.graph.run();
                     vec all = vfilter(vcat(model(n, 1, 0), model(n, 1, 1)), p_even);
                     Expect e;
                     for (size_t k = 0; k < N_KEYS; k++) {
                         vector<pair<uint64_t, int64_t>> v;
                         for (auto& r : all)
                             if (r.key == k) v.push_back({r.ts, r.value});
                         if (v.empty()) continue;
                         sort(v.begin(), v.end());
                         const uint64_t last = v.back().first;
                         for (uint64_t w = 0; w * 30 <= last; w++) {
                             for (auto& x : v)
                                 if (x.first >= w * 30 && x.first < w * 30 + 90) e.sum += x.second;
                             e.count++;
                         }
                     }
                     return make_tuple(got(t), e, g.graph.getNumThreads() == g.replicas);
                 }});
    return S;
}

int main(int argc, char** argv) {
    size_t runs = 3;
    int64_t n = 20000;
    string only;
    int c;
    while ((c = getopt(argc, argv, "r:n:t:")) != -1) {
        if (c == 'r') runs = (size_t)atol(optarg);
        if (c == 'n') n = atol(optarg);
        if (c == 't') only = optarg;
    }
    mt19937 rng(random_device{}());
    int failures = 0;
    for (auto& s : scenarios()) {
        if (!only.empty() && only != s.name) continue;
        for (size_t r = 0; r < runs; r++) {
            auto [g, e, threads_ok] = s.run(n, rng);
            const bool ok = g.sum == e.sum && g.count == e.count && threads_ok;
            cout << s.name << " run " << r << ": sum " << g.sum << " count " << g.count << " (expected " << e.sum << " " << e.count << ")"
                 << (threads_ok ? "" : " thread count mismatch") << (ok ? "  OK" : "  FAILED") << endl;
            failures += !ok;
        }
    }
    return failures == 0 ? 0 : 1;
}
