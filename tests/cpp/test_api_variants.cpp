// Every functor signature the builders accept (reference: the API file), each checked
// against a closed-form value: plain and "rich" (RuntimeContext&) forms, Source item / loop mode, the
// three Filter modes, in-place / out-of-place Map, FlatMap, Accumulator with an initial value, the
// two Sink forms, non-incremental / incremental window functions, lift / combine, closing functions,
// LocalStorage, custom KeyBy routing.
#include <atomic>
#include <functional>
#include <iostream>
#include <optional>
#include <tuple>

#include <wf/windflow.hpp>

using namespace wf;
using namespace std;

struct in_t {
    size_t key = 0;
    uint64_t id = 0, ts = 0;
    long v = 0;
    tuple<size_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    void setControlFields(size_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};
struct out_t {
    size_t key = 0;
    uint64_t id = 0, ts = 0;
    long w = 0;
    tuple<size_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    void setControlFields(size_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};

static int failures = 0;
static void expect(const char* what, long got, long want) {
    const bool ok = got == want;
    cout << what << ": " << got << (ok ? "  OK" : "  FAILED (expected " + to_string(want) + ")") << endl;
    failures += !ok;
}

constexpr long N = 6000;  // values 0..N-1, key = v % 3
static in_t make(long v) {
    in_t t;
    t.setControlFields((size_t)(v % 3), (uint64_t)(v / 3), (uint64_t)v);
    t.v = v;
    return t;
}
static long sum_to(long n) { return n * (n - 1) / 2; }

// runs Source(variant) -> op chain built by `body` -> Sink(variant); returns the sink's sum
template <typename Body>
static long pipeline(const char* name, int source_variant, int sink_variant, Mode mode, Body body, atomic<long>* closings = nullptr) {
    PipeGraph g(name, mode);
    unique_ptr<Basic_Operator> src;
    long next = 0;
    switch (source_variant) {
        case 0:
            src = Source_Builder([next](in_t& t) mutable {
                      t = make(next);
                      return ++next < N;
                  }).build_unique();
            break;
        case 1:
            src = Source_Builder([next](in_t& t, RuntimeContext& ctx) mutable {
                      t = make(next);
                      (void)ctx.getParallelism();
                      return ++next < N;
                  }).build_unique();
            break;
        case 2:  // loop mode: the function is called once and pushes everything
            src = Source_Builder([](Shipper<in_t>& out) {
                      for (long v = 0; v < N; v++) out.push(make(v));
                      return false;
                  }).build_unique();
            break;
        default:
            src = Source_Builder([](Shipper<in_t>& out, RuntimeContext& ctx) {
                      for (long v = 0; v < N; v++) out.push(make(v));
                      return ctx.getReplicaIndex() != 0;  // false: end of stream
                  }).build_unique();
    }
    MultiPipe& mp = g.add_source(*src);
    vector<unique_ptr<Basic_Operator>> ops;
    const bool out_is_out_t = body(mp, ops);
    atomic<long> sum{0};
    unique_ptr<Basic_Operator> sink;
    auto closing = [closings](RuntimeContext&) {
        if (closings) (*closings)++;
    };
    if (out_is_out_t) {
        if (sink_variant % 2 == 0)
            sink = Sink_Builder([&sum](optional<out_t>& r) {
                       if (r) sum += r->w;
                   }).withClosingFunction(closing).build_unique();
        else
            sink = Sink_Builder([&sum](optional<reference_wrapper<out_t>> r, RuntimeContext&) {
                       if (r) sum += r->get().w;
                   }).withClosingFunction(closing).build_unique();
    } else {
        if (sink_variant % 2 == 0)
            sink = Sink_Builder([&sum](optional<in_t>& r, RuntimeContext&) {
                       if (r) sum += r->v;
                   }).withClosingFunction(closing).build_unique();
        else
            sink = Sink_Builder([&sum](optional<reference_wrapper<in_t>> r) {
                       if (r) sum += r->get().v;
                   }).withClosingFunction(closing).build_unique();
    }
    mp.add_sink(*sink);
    g.run();
    return sum.load();
}

int main() {
    using ops_t = vector<unique_ptr<Basic_Operator>>;
    // ---- sources x sinks ------------------------------------------------------------------------
    for (int sv = 0; sv < 4; sv++)
        for (int kv = 0; kv < 2; kv++)
            expect(("source variant " + to_string(sv) + " / sink variant " + to_string(kv)).c_str(),
                   pipeline("src_sink", sv, kv, Mode::DEFAULT, [](MultiPipe&, ops_t&) { return false; }), sum_to(N));
    // ---- Filter: in place, optional<R>, optional<R*> (+ rich) -----------------------------------
    expect("filter bool(T&)", pipeline("f0", 0, 0, Mode::DEFAULT, [](MultiPipe& mp, ops_t& o) {
               o.push_back(Filter_Builder([](in_t& t) { return t.v % 2 == 0; }).withParallelism(2).build_unique());
               mp.add(*o.back());
               return false;
           }), 2 * sum_to(N / 2));
    expect("filter bool(T&, ctx)", pipeline("f1", 0, 1, Mode::DEFAULT, [](MultiPipe& mp, ops_t& o) {
               o.push_back(Filter_Builder([](in_t& t, RuntimeContext&) { return t.v % 2 == 0; }).build_unique());
               mp.chain(*o.back());
               return false;
           }), 2 * sum_to(N / 2));
    expect("filter optional<R>(const T&)", pipeline("f2", 0, 0, Mode::DEFAULT, [](MultiPipe& mp, ops_t& o) {
               o.push_back(Filter_Builder([](const in_t& t) -> optional<out_t> {
                               if (t.v % 2) return {};
                               out_t r;
                               r.setControlFields(t.key, t.id, t.ts);
                               r.w = 10 * t.v;
                               return r;
                           }).withParallelism(3).build_unique());
               mp.add(*o.back());
               return true;
           }), 20 * sum_to(N / 2));
    expect("filter optional<R*>(const T&, ctx)", pipeline("f3", 0, 1, Mode::DEFAULT, [](MultiPipe& mp, ops_t& o) {
               o.push_back(Filter_Builder([](const in_t& t, RuntimeContext&) -> optional<out_t*> {
                               if (t.v % 2) return {};
                               out_t* r = new out_t();
                               r->setControlFields(t.key, t.id, t.ts);
                               r->w = t.v;
                               return r;
                           }).build_unique());
               mp.add(*o.back());
               return true;
           }), 2 * sum_to(N / 2));
    // ---- Map: 4 forms ---------------------------------------------------------------------------
    expect("map void(T&)", pipeline("m0", 0, 0, Mode::DEFAULT, [](MultiPipe& mp, ops_t& o) {
               o.push_back(Map_Builder([](in_t& t) { t.v += 1; }).withParallelism(2).build_unique());
               mp.add(*o.back());
               return false;
           }), sum_to(N) + N);
    expect("map void(T&, ctx) + LocalStorage", pipeline("m1", 0, 0, Mode::DEFAULT, [](MultiPipe& mp, ops_t& o) {
               o.push_back(Map_Builder([](in_t& t, RuntimeContext& ctx) {
                               long& seen = ctx.getLocalStorage().get<long>("seen");  // per-replica counter
                               seen++;
                               t.v = 1;
                           }).withParallelism(2).build_unique());
               mp.add(*o.back());
               return false;
           }), N);
    expect("map void(const T&, R&)", pipeline("m2", 0, 0, Mode::DEFAULT, [](MultiPipe& mp, ops_t& o) {
               o.push_back(Map_Builder([](const in_t& t, out_t& r) { r.w = 2 * t.v; }).withParallelism(2).build_unique());
               mp.add(*o.back());
               return true;
           }), 2 * sum_to(N));
    expect("map void(const T&, R&, ctx) keyed", pipeline("m3", 0, 1, Mode::DEFAULT, [](MultiPipe& mp, ops_t& o) {
               o.push_back(Map_Builder([](const in_t& t, out_t& r, RuntimeContext& ctx) {
                               r.w = (t.key % ctx.getParallelism() == ctx.getReplicaIndex()) ? 1 : 1000000;  // routed by key?
                           }).withParallelism(3).enable_KeyBy().build_unique());
               mp.add(*o.back());
               return true;
           }), N);
    expect("map keyed with a custom routing function", pipeline("m4", 0, 0, Mode::DEFAULT, [](MultiPipe& mp, ops_t& o) {
               o.push_back(Map_Builder([](const in_t& t, out_t& r, RuntimeContext& ctx) {
                               r.w = ((t.key + 1) % ctx.getParallelism() == ctx.getReplicaIndex()) ? 1 : 1000000;
                           }).withParallelism(3).enable_KeyBy([](size_t hash, size_t n) { return (hash + 1) % n; }).build_unique());
               mp.add(*o.back());
               return true;
           }), N);
    // ---- FlatMap, Accumulator ----------------------------------------------------------------------
    expect("flatmap (+ctx)", pipeline("fm", 0, 0, Mode::DEFAULT, [](MultiPipe& mp, ops_t& o) {
               o.push_back(FlatMap_Builder([](const in_t& t, Shipper<out_t>& s, RuntimeContext&) {
                               for (int i = 0; i < (int)(t.v % 3); i++) {
                                   out_t r;
                                   r.setControlFields(t.key, t.id, t.ts);
                                   r.w = 1;
                                   s.push(r);
                               }
                           }).withParallelism(2).build_unique());
               mp.add(*o.back());
               return true;
           }), (N / 3) * 3);
    {
        // running sum per key starting from 100: sum over keys of sum_{j=1..m} (100 + prefix_j)
        long want = 0;
        for (long k = 0; k < 3; k++) {
            long acc = 100;
            for (long v = k; v < N; v += 3) {
                acc += v;
                want += acc;
            }
        }
        out_t init;
        init.w = 100;
        expect("accumulator with initial value", pipeline("acc", 0, 0, Mode::DEFAULT, [init](MultiPipe& mp, ops_t& o) {
                   o.push_back(Accumulator_Builder([](const in_t& t, out_t& a) { a.w += t.v; }).withInitialValue(init).withParallelism(2).build_unique());
                   mp.add(*o.back());
                   return true;
               }), want);
        expect("accumulator rich", pipeline("acc2", 0, 1, Mode::DEFAULT, [init](MultiPipe& mp, ops_t& o) {
                   o.push_back(Accumulator_Builder([](const in_t& t, out_t& a, RuntimeContext&) { a.w += t.v; }).withInitialValue(init).build_unique());
                   mp.add(*o.back());
                   return true;
               }), want);
    }
    // ---- windows: NIC / INC, plain / rich; lift + combine plain / rich ----------------------------------
    {
        // CB windows of 10 slide 5 per key over values v = key, key+3, ... : every opened window is emitted
        long want = 0;
        for (long k = 0; k < 3; k++) {
            vector<long> v;
            for (long x = k; x < N; x += 3) v.push_back(x);
            for (size_t w = 0; w * 5 < v.size(); w++)
                for (size_t i = w * 5; i < min(v.size(), w * 5 + 10); i++) want += v[i];
        }
        auto add_win = [&](const char* what, auto builder, bool sliding_only = false) {
            (void)sliding_only;
            expect(what, pipeline(what, 0, 0, Mode::DETERMINISTIC, [&](MultiPipe& mp, ops_t& o) {
                       o.push_back(builder());
                       mp.add(*o.back());
                       return true;
                   }), want);
        };
        add_win("Win_Seq NIC", [] {
            return WinSeq_Builder([](uint64_t, const Iterable<in_t>& in, out_t& r) {
                       for (const in_t& t : in) r.w += t.v;
                   }).withCBWindows(10, 5).build_unique();
        });
        add_win("Win_Seq NIC rich", [] {
            return WinSeq_Builder([](uint64_t, const Iterable<in_t>& in, out_t& r, RuntimeContext&) {
                       for (size_t i = 0; i < in.size(); i++) r.w += in[i].v;
                   }).withCBWindows(10, 5).build_unique();
        });
        add_win("Key_Farm INC", [] {
            return KeyFarm_Builder([](uint64_t, const in_t& t, out_t& r) { r.w += t.v; }).withCBWindows(10, 5).withParallelism(3).build_unique();
        });
        add_win("Win_Farm INC rich", [] {
            return WinFarm_Builder([](uint64_t, const in_t& t, out_t& r, RuntimeContext&) { r.w += t.v; }).withCBWindows(10, 5).withParallelism(2)
                .build_unique();
        });
        add_win("Key_FFAT lift/comb", [] {
            return KeyFFAT_Builder([](const in_t& t, out_t& r) { r.w = t.v; }, [](const out_t& a, const out_t& b, out_t& o) { o.w = a.w + b.w; })
                .withCBWindows(10, 5).withParallelism(2).build_unique();
        });
        add_win("Win_SeqFFAT lift/comb rich", [] {
            return WinSeqFFAT_Builder([](const in_t& t, out_t& r, RuntimeContext&) { r.w = t.v; },
                                      [](const out_t& a, const out_t& b, out_t& o, RuntimeContext&) { o.w = a.w + b.w; })
                .withCBWindows(10, 5).build_unique();
        });
        add_win("Pane_Farm INC + NIC", [] {
            return PaneFarm_Builder([](uint64_t, const in_t& t, out_t& r) { r.w += t.v; },
                                    [](uint64_t, const Iterable<out_t>& in, out_t& r) {
                                        for (const out_t& p : in) r.w += p.w;
                                    })
                .withCBWindows(10, 5).withParallelism(2, 2).build_unique();
        });
        add_win("Win_MapReduce NIC + INC", [] {
            return WinMapReduce_Builder([](uint64_t, const Iterable<in_t>& in, out_t& r) {
                                            for (const in_t& t : in) r.w += t.v;
                                        },
                                        [](uint64_t, const out_t& p, out_t& r) { r.w += p.w; })
                .withCBWindows(10, 5).withParallelism(2, 1).build_unique();
        });
    }
    // ---- closing functions run once per replica ------------------------------------------------------
    {
        atomic<long> closings{0};
        pipeline("closing", 0, 0, Mode::DEFAULT, [&](MultiPipe& mp, ops_t& o) {
            auto c = [&closings](RuntimeContext&) { closings++; };
            o.push_back(Map_Builder([](in_t& t) { t.v++; }).withParallelism(3).withClosingFunction(c).build_unique());
            mp.add(*o.back());
            return false;
        }, &closings);
        expect("closing functions (3 map replicas + 1 sink)", closings.load(), 4);
    }
    cout << (failures == 0 ? "ALL OK" : "FAILED") << endl;
    return failures == 0 ? 0 : 1;
}
