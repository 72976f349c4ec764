// Keyed (stateful) GPU operators and the device KeyBy emitter inside MultiPipes.
//   A: Source (tuples) -> keyed Map_GPU (per-key running sum) -> Sink
//   B: Batch_Source -> [KeyBy_EmitterGPU] keyed Map_GPU -> [KeyBy_EmitterGPU] keyed Filter_GPU -> Sink
//   C: Batch_Source -> [KeyBy_EmitterGPU] Key_FFAT_GPU -> Sink
// every run draws random parallelism / batch sizes and is checked against host oracles.
#include <iostream>
#include <map>
#include <random>
#include <thread>

#include <wf/windflow_gpu.hpp>

#include "mp_common.hpp"

using namespace wf;
using namespace std;

// per-key running sum kept in the scratchpad
struct RunningSum {
    __host__ __device__ void operator()(tuple_t& t, char* state, size_t) const {
        int64_t* acc = reinterpret_cast<int64_t*>(state);
        *acc += t.value;
        t.value = *acc;
    }
};
// the associative form of a keyed map: the history of a key is (sum, newest value) - the second field makes the combine
// non-commutative, so a wrong operand order shows
struct Hist {
    int64_t sum, newest;
};
struct HistLift {
    __host__ __device__ void operator()(const tuple_t& t, Hist& h) const {
        h.sum = t.value;
        h.newest = t.value;
    }
};
struct HistCombine {
    __host__ __device__ void operator()(const Hist& older, const Hist& newer, Hist& out) const {
        out.sum = older.sum + newer.sum;
        out.newest = newer.newest;
    }
};
struct HistApply {
    __host__ __device__ void operator()(tuple_t& t, const Hist& h) const { t.value = h.sum + 1000 * (h.newest - t.value); }  // newest == own value
};
// out-of-place flavour
struct RunningSumOut {
    __host__ __device__ void operator()(const tuple_t& t, output_t& r, char* state, size_t) const {
        int64_t* acc = reinterpret_cast<int64_t*>(state);
        *acc += t.value;
        r.setControlFields(t.key, t.id, t.ts);
        r.value = *acc;
    }
};
// keeps the 1st, 4th, 7th ... tuple of every key
struct EveryThird {
    __host__ __device__ void operator()(const tuple_t& t, bool& keep, char* state, size_t) const {
        uint32_t* cnt = reinterpret_cast<uint32_t*>(state);
        keep = (*cnt % 3) == 0;
        (*cnt)++;
        (void)t;
    }
};
struct EveryThirdOut {
    __host__ __device__ void operator()(const output_t&, bool& keep, char* state, size_t) const {
        uint32_t* cnt = reinterpret_cast<uint32_t*>(state);
        keep = (*cnt % 3) == 0;
        (*cnt)++;
    }
};
struct Lift_GPU {
    __host__ __device__ void operator()(const tuple_t& t, output_t& r) const { r.value = t.value; }
};
struct Comb_GPU {
    __host__ __device__ void operator()(const output_t& a, const output_t& b, output_t& out) const { out.value = a.value + b.value; }
};

// deterministic stream: n tuples, keys drawn by an LCG, value = small ints
struct Gen {
    size_t n, keys, sent = 0;
    uint64_t state = 99;
    Gen(size_t n_, size_t k_) : n(n_), keys(k_) {}
    bool operator()(tuple_t& t) {
        state = state * 6364136223846793005ULL + 1442695040888963407ULL;
        t.key = (state >> 33) % keys;
        t.id = sent;
        t.ts = sent;
        t.value = (int64_t)((state >> 20) % 17) - 8;
        sent++;
        return sent < n;
    }
};

// source replica r of `stride` emits elements r, r + stride, r + 2 stride ... of a materialised stream: every lane is in
// timestamp order, the lanes overlap in time
struct StridedGen {
    const vector<tuple_t>* st;
    size_t stride;
    size_t offset = 0;  // lanes of other ranks come first
    size_t pos = (size_t)-1;
    bool operator()(tuple_t& t, RuntimeContext& rc) {
        if (pos == (size_t)-1) pos = offset + rc.getReplicaIndex();
        t = (*st)[pos];
        pos += stride;
        return pos < st->size();
    }
};

static vector<tuple_t> materialize(size_t n, size_t keys) {
    Gen g(n, keys);
    vector<tuple_t> v;
    bool more = true;
    while (more) {
        tuple_t t;
        more = g(t);
        v.push_back(t);
    }
    return v;
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 60000;
    const size_t keys = argc > 2 ? (size_t)atol(argv[2]) : 37;
    const size_t runs = argc > 3 ? (size_t)atol(argv[3]) : 2;
    if (gpu::device_count() < 1) return 3;
    const auto stream = materialize(n, keys);
    // oracles
    long exp_a = 0, exp_b_sum = 0, exp_b_cnt = 0;
    {
        map<size_t, int64_t> acc;
        map<size_t, uint32_t> cnt;
        for (auto& t : stream) {
            acc[t.key] += t.value;
            exp_a += acc[t.key];
            if (cnt[t.key]++ % 3 == 0) {
                exp_b_sum += acc[t.key];
                exp_b_cnt++;
            }
        }
    }
    const uint64_t win = 12, slide = 4;
    long exp_c_sum = 0, exp_c_cnt = 0;
    {
        map<size_t, vector<int64_t>> per;
        for (auto& t : stream) per[t.key].push_back(t.value);
        for (auto& kv : per) {
            const auto& v = kv.second;
            for (size_t w = 0; w * slide < v.size(); w++) {
                // weighted by the window number: WHICH tuples fall into which window matters (a plain sum of all windows would
                // hardly notice tuples of a key arriving in another order)
                for (size_t i = w * slide; i < min(v.size(), w * slide + win); i++) exp_c_sum += v[i] * (long)(w % 7 + 1);
                exp_c_cnt++;
            }
        }
    }
    std::mt19937 rng(std::random_device{}());
    std::uniform_int_distribution<int> dp(1, 4), db(1, 40);
    int failures = 0;
    for (size_t run = 0; run < runs; run++) {
        const size_t p1 = (size_t)dp(rng), p2 = (size_t)dp(rng);
        const size_t batch = 64 * (size_t)db(rng);
        // ---- A ----------------------------------------------------------------
        {
            PipeGraph g("keyed_a", Mode::DEFAULT);
            Source src = Source_Builder(Gen(n, keys)).withParallelism(1).build();
            MultiPipe& mp = g.add_source(src);
            auto map = MapGPU_Builder(RunningSum()).withName("kmap").withParallelism(p1).withBatch(batch).withScratchpad(16).withMaxKeys(keys)
                           .build_unique();
            mp.add(*map);
            long sum = 0, cnt = 0;
            Sink sink = Sink_Builder([&](std::optional<tuple_t>& o) {
                            if (o) {
                                sum += o->value;
                                cnt++;
                            }
                        }).withParallelism(1).build();
            mp.add_sink(sink);
            g.run();
            const bool ok = (sum == exp_a && cnt == (long)n);
            cout << "A par=" << p1 << " batch=" << batch << " sum " << sum << " cnt " << cnt << (ok ? "  OK" : "  FAILED") << endl;
            failures += !ok;
        }
        // ---- H: the same running sum as a parallel scan (keyed_scan), fed with tuples and with batches ----------------------
        for (int batched = 0; batched < 2; batched++) {
            PipeGraph g("keyed_h", Mode::DEFAULT);
            Source tsrc = Source_Builder(Gen(n, keys)).withParallelism(1).build();
            auto bsrc = BatchSource_Builder(Gen(n, keys)).withParallelism(1).withBatch(batch).build_unique();
            MultiPipe& mp = batched ? g.add_source(*bsrc) : g.add_source(tsrc);
            auto map = ScanMapGPU_Builder(keyed_scan(HistLift(), HistCombine(), HistApply())).withName("scanmap").withParallelism(p1).withBatch(batch)
                           .withMaxKeys(keys).build_unique();
            mp.add(*map);
            long sum = 0, cnt = 0;
            Sink sink = Sink_Builder([&](std::optional<tuple_t>& o) {
                            if (o) {
                                sum += o->value;
                                cnt++;
                            }
                        }).withParallelism(1).build();
            mp.add_sink(sink);
            g.run();
            const bool ok = (sum == exp_a && cnt == (long)n);
            cout << "H keyed_scan " << (batched ? "batches" : "tuples") << " par=" << p1 << " batch=" << batch << " sum " << sum << " cnt " << cnt
                 << " (expected " << exp_a << ")" << (ok ? "  OK" : "  FAILED") << endl;
            failures += !ok;
        }
        // ---- B ----------------------------------------------------------------
        {
            PipeGraph g("keyed_b", Mode::DEFAULT);
            auto src = BatchSource_Builder(Gen(n, keys)).withParallelism(1).withBatch(batch).build_unique();
            MultiPipe& mp = g.add_source(*src);
            auto map = MapGPU_Builder(RunningSumOut()).withName("kmap").withParallelism(p1).withScratchpad(8).withMaxKeys(keys).build_unique();
            mp.add(*map);
            auto fil = FilterGPU_Builder([] __host__ __device__(const output_t&, bool& keep, char* st, size_t) {
                           uint32_t* c = reinterpret_cast<uint32_t*>(st);
                           keep = (*c % 3) == 0;
                           (*c)++;
                       }).withName("kfilter").withParallelism(p2).withScratchpad(8).withMaxKeys(keys).build_unique();
            mp.add(*fil);
            long sum = 0, cnt = 0;
            Sink sink = Sink_Builder([&](std::optional<output_t>& o) {
                            if (o) {
                                sum += o->value;
                                cnt++;
                            }
                        }).withParallelism(1).build();
            mp.add_sink(sink);
            g.run();
            const bool ok = (sum == exp_b_sum && cnt == exp_b_cnt);
            cout << "B par=" << p1 << "," << p2 << " batch=" << batch << " sum " << sum << " cnt " << cnt << " (expected " << exp_b_sum << " "
                 << exp_b_cnt << ")" << (ok ? "  OK" : "  FAILED") << endl;
            failures += !ok;
        }
        // ---- C ----------------------------------------------------------------
        {
            PipeGraph g("keyed_c", Mode::DEFAULT);
            auto src = BatchSource_Builder(Gen(n, keys)).withParallelism(1).withBatch(batch).build_unique();
            MultiPipe& mp = g.add_source(*src);
            auto kff = KeyFFATGPU_Builder(Lift_GPU(), Comb_GPU()).withName("kff").withCBWindows(win, slide).withParallelism(p1).withMaxKeys(keys)
                           .withOutputCapacity(1 << 16).build_unique();
            mp.add(*kff);
            long sum = 0, cnt = 0;
            Sink sink = Sink_Builder([&](std::optional<output_t>& o) {
                            if (o) {
                                sum += o->value * (long)(o->id % 7 + 1);
                                cnt++;
                            }
                        }).withParallelism(1).build();
            mp.add_sink(sink);
            g.run();
            const bool ok = (sum == exp_c_sum && cnt == exp_c_cnt);
            cout << "C par=" << p1 << " batch=" << batch << " sum " << sum << " cnt " << cnt << " (expected " << exp_c_sum << " " << exp_c_cnt
                 << ")" << (ok ? "  OK" : "  FAILED") << endl;
            failures += !ok;
        }
    }
    // ---- D: GPU operators around split and merge: batches are turned back into tuples where the graph forks ----
    for (size_t run = 0; run < runs; run++) {
        const size_t p1 = (size_t)dp(rng), p2 = (size_t)dp(rng);
        const size_t batch = 64 * (size_t)db(rng);
        PipeGraph g("keyed_d", Mode::DEFAULT);
        auto srcA = BatchSource_Builder(Gen(n, keys)).withName("srcA").withParallelism(1).withBatch(batch).build_unique();
        auto srcB = BatchSource_Builder(Gen(n, keys)).withName("srcB").withParallelism(1).withBatch(batch).build_unique();
        MultiPipe& a = g.add_source(*srcA);
        MultiPipe& b = g.add_source(*srcB);
        auto mapA = MapGPU_Builder([] __host__ __device__(tuple_t & t) { t.value = t.value * 2; }).withName("dbl").withParallelism(p1)
                        .withGPUPlacement({0, 0, 0}).build_unique();  // placement per replica (all on the one GPU of the test box)
        auto mapB = MapGPU_Builder([] __host__ __device__(tuple_t & t) { t.value = t.value + 100; }).withName("plus").withParallelism(p2).build_unique();
        a.add(*mapA);
        b.add(*mapB);
        MultiPipe& m = a.merge(b);
        auto fil = FilterGPU_Builder([] __host__ __device__(const tuple_t& t) { return t.value % 3 == 0; }).withName("thirds").withParallelism(p1)
                       .withBatch(batch).build_unique();
        m.add(*fil);
        m.split([](const tuple_t& t) -> size_t { return t.value < 0 ? 0 : 1; }, 2);
        long neg_sum = 0, neg_cnt = 0, pos_sum = 0, pos_cnt = 0;
        auto neg_map = MapGPU_Builder([] __host__ __device__(tuple_t & t) { t.value = -t.value; }).withName("abs").withParallelism(p2).withBatch(batch)
                           .build_unique();
        m.select(0).add(*neg_map);
        Sink s0 = Sink_Builder([&](std::optional<tuple_t>& o) {
                      if (o) {
                          neg_sum += o->value;
                          neg_cnt++;
                      }
                  }).withParallelism(1).build();
        m.select(0).add_sink(s0);
        Sink s1 = Sink_Builder([&](std::optional<tuple_t>& o) {
                      if (o) {
                          pos_sum += o->value;
                          pos_cnt++;
                      }
                  }).withParallelism(1).build();
        m.select(1).add_sink(s1);
        g.run();
        long e_neg_sum = 0, e_neg_cnt = 0, e_pos_sum = 0, e_pos_cnt = 0;
        for (auto& t : stream)
            for (int64_t v : {t.value * 2, t.value + 100}) {
                if (v % 3 != 0) continue;
                if (v < 0) {
                    e_neg_sum += -v;
                    e_neg_cnt++;
                } else {
                    e_pos_sum += v;
                    e_pos_cnt++;
                }
            }
        const bool ok = neg_sum == e_neg_sum && neg_cnt == e_neg_cnt && pos_sum == e_pos_sum && pos_cnt == e_pos_cnt;
        cout << "D par=" << p1 << "," << p2 << " batch=" << batch << " neg " << neg_sum << "/" << neg_cnt << " pos " << pos_sum << "/" << pos_cnt
             << (ok ? "  OK" : "  FAILED") << endl;
        failures += !ok;
    }
    // ---- F: the stream is born on the device (Device_Source): 2 source replicas -> device KeyBy -> keyed Map_GPU ----
    for (size_t run = 0; run < runs; run++) {
        struct DevGen {
            uint64_t keys;
            __host__ __device__ void operator()(uint64_t seq, uint32_t replica, tuple_t& t) const {
                t.key = (size_t)((seq % keys) * 2 + replica);
                t.id = seq;
                t.ts = seq;
                t.value = (int64_t)((seq * 7 + replica * 3) % 17) - 8;
            }
        };
        const size_t p1 = (size_t)dp(rng);
        const size_t batch = 64 * (size_t)db(rng);
        const uint64_t per_replica = ((n / 2 + batch - 1) / batch) * batch;  // whole batches
        long exp_sum = 0;
        {
            DevGen g{keys};
            map<size_t, int64_t> acc;
            for (uint32_t r = 0; r < 2; r++)
                for (uint64_t q = 0; q < per_replica; q++) {
                    tuple_t t;
                    g(q, r, t);
                    acc[t.key] += t.value;
                    exp_sum += acc[t.key];
                }
        }
        PipeGraph g("device_source", Mode::DEFAULT);
        auto src = DeviceSource_Builder(DevGen{keys}).withName("dsrc").withParallelism(2).withBatch(batch).withLimit(per_replica).build_unique();
        MultiPipe& mp = g.add_source(*src);
        auto map = MapGPU_Builder(RunningSum()).withName("kmap").withParallelism(p1).withScratchpad(16).withMaxKeys(2 * keys + 2).build_unique();
        mp.add(*map);
        long sum = 0, cnt = 0;
        Sink sink = Sink_Builder([&](std::optional<tuple_t>& o) {
                        if (o) {
                            sum += o->value;
                            cnt++;
                        }
                    }).withParallelism(1).build();
        mp.add_sink(sink);
        g.run();
        const bool ok = (sum == exp_sum && cnt == (long)(2 * per_replica));
        cout << "F device source par=" << p1 << " batch=" << batch << " sum " << sum << " cnt " << cnt << " (expected " << exp_sum << " "
             << 2 * per_replica << ")" << (ok ? "  OK" : "  FAILED") << endl;
        failures += !ok;
    }
    // ---- G: split and merge WITHOUT leaving the device: split_gpu partitions every batch by a device functor, the branches
    //      run their own GPU operators on device batches and are merged again in front of a GPU filter ----
    for (size_t run = 0; run < runs; run++) {
        struct BySign {
            __host__ __device__ size_t operator()(const tuple_t& t) const { return t.value < 0 ? 0 : 1; }
        };
        const size_t p0 = (size_t)dp(rng), p1 = (size_t)dp(rng), p2 = (size_t)dp(rng);
        const size_t batch = 64 * (size_t)db(rng);
        PipeGraph g("keyed_g", Mode::DEFAULT);
        auto src = BatchSource_Builder(Gen(n, keys)).withName("src").withParallelism(2).withBatch(batch).build_unique();
        MultiPipe& mp = g.add_source(*src);
        split_gpu(mp, BySign(), 2);
        auto neg = MapGPU_Builder([] __host__ __device__(tuple_t & t) { t.value = -3 * t.value; }).withName("neg").withParallelism(p0).build_unique();
        auto pos = MapGPU_Builder([] __host__ __device__(tuple_t & t) { t.value = t.value + 7; }).withName("pos").withParallelism(p1).build_unique();
        MultiPipe& b0 = mp.select(0);
        b0.add(*neg);
        MultiPipe& b1 = mp.select(1);
        b1.add(*pos);
        MultiPipe& m = b0.merge(b1);
        auto fil = FilterGPU_Builder([] __host__ __device__(const tuple_t& t) { return t.value % 2 == 0; }).withName("even").withParallelism(p2).build_unique();
        m.add(*fil);
        std::atomic<long> sum{0}, cnt{0};
        Sink sink = Sink_Builder([&](std::optional<tuple_t>& o) {
                        if (o) {
                            sum += o->value;
                            cnt++;
                        }
                    }).withParallelism(1).build();
        m.add_sink(sink);
        g.run();
        long e_sum = 0, e_cnt = 0;
        for (auto& t : stream) {
            const int64_t v = t.value < 0 ? -3 * t.value : t.value + 7;
            if (v % 2 == 0) {
                e_sum += 2 * v;  // two source replicas emit the stream
                e_cnt += 2;
            }
        }
        const bool ok = sum == e_sum && cnt == e_cnt;
        cout << "G device split/merge par=" << p0 << "," << p1 << "," << p2 << " batch=" << batch << " sum " << sum << " cnt " << cnt << " (expected " << e_sum
             << " " << e_cnt << ")" << (ok ? "  OK" : "  FAILED") << endl;
        failures += !ok;
    }
    // ---- I: DETERMINISTIC graphs on batch edges: 4 source lanes that overlap in time feed keyed GPU operators, whose device
    //      KeyBy endpoint merges the lanes in timestamp order (the watermark rule on batches); the results are those of the
    //      sequential stream whatever the parallelism ----
    for (size_t run = 0; run < runs; run++) {
        const size_t p1 = (size_t)dp(rng), p2 = (size_t)dp(rng);
        const size_t batch = 64 * (size_t)db(rng);
        {
            PipeGraph g("keyed_i1", Mode::DETERMINISTIC);
            auto src = BatchSource_Builder(StridedGen{&stream, 4}).withName("lanes").withParallelism(4).withBatch(batch).build_unique();
            MultiPipe& mp = g.add_source(*src);
            auto kff = KeyFFATGPU_Builder(Lift_GPU(), Comb_GPU()).withCBWindows(win, slide).withName("kff").withParallelism(p1).withMaxKeys(keys)
                           .withOutputCapacity(1 << 16).build_unique();
            mp.add(*kff);
            long sum = 0, cnt = 0;
            Sink sink = Sink_Builder([&](std::optional<output_t>& o) {
                            if (o) {
                                sum += o->value * (long)(o->id % 7 + 1);
                                cnt++;
                            }
                        }).withParallelism(1).build();
            mp.add_sink(sink);
            g.run();
            const bool ok = (sum == exp_c_sum && cnt == exp_c_cnt);
            cout << "I deterministic windows par=" << p1 << " batch=" << batch << " sum " << sum << " cnt " << cnt << " (expected " << exp_c_sum << " "
                 << exp_c_cnt << ")" << (ok ? "  OK" : "  FAILED") << endl;
            failures += !ok;
        }
        {
            // a keyed in-place map alone: every tuple must carry the running sum its key had at that point of the SEQUENTIAL stream
            vector<int64_t> exp_val(stream.size());
            {
                map<size_t, int64_t> acc;
                for (size_t i = 0; i < stream.size(); i++) exp_val[i] = (acc[stream[i].key] += stream[i].value);
            }
            PipeGraph g("keyed_i3", Mode::DETERMINISTIC);
            auto src = BatchSource_Builder(StridedGen{&stream, 4}).withName("lanes").withParallelism(4).withBatch(batch).build_unique();
            MultiPipe& mp = g.add_source(*src);
            auto map = MapGPU_Builder(RunningSum()).withName("kmap").withParallelism(p1).withScratchpad(8).withMaxKeys(keys).build_unique();
            mp.add(*map);
            long wrong = 0, cnt = 0, first_wrong = -1;
            Sink sink = Sink_Builder([&](std::optional<tuple_t>& o) {
                            if (o) {
                                if (o->value != exp_val[o->id]) {
                                    if (wrong == 0) first_wrong = (long)o->id;
                                    wrong++;
                                }
                                cnt++;
                            }
                        }).withParallelism(1).build();
            mp.add_sink(sink);
            g.run();
            const bool ok = (wrong == 0 && cnt == (long)n);
            cout << "I deterministic map par=" << p1 << " batch=" << batch << " wrong " << wrong << " (first at " << first_wrong << ") cnt " << cnt
                 << (ok ? "  OK" : "  FAILED") << endl;
            failures += !ok;
        }
        {
            // the same with timestamps 2^24 apart: a release spans more than 2^32 time units, the ordered merge sorts in two rounds
            vector<tuple_t> wide = stream;
            for (auto& t : wide) t.ts <<= 24;
            vector<int64_t> exp_val(wide.size());
            {
                map<size_t, int64_t> acc;
                for (size_t i = 0; i < wide.size(); i++) exp_val[i] = (acc[wide[i].key] += wide[i].value);
            }
            PipeGraph g("keyed_i4", Mode::DETERMINISTIC);
            auto src = BatchSource_Builder(StridedGen{&wide, 4}).withName("lanes").withParallelism(4).withBatch(batch).build_unique();
            MultiPipe& mp = g.add_source(*src);
            auto map = MapGPU_Builder(RunningSum()).withName("kmap").withParallelism(p2).withScratchpad(8).withMaxKeys(keys).build_unique();
            mp.add(*map);
            long wrong = 0, cnt = 0;
            Sink sink = Sink_Builder([&](std::optional<tuple_t>& o) {
                            if (o) {
                                wrong += o->value != exp_val[o->id];
                                cnt++;
                            }
                        }).withParallelism(1).build();
            mp.add_sink(sink);
            g.run();
            const bool ok = (wrong == 0 && cnt == (long)n);
            cout << "I deterministic map, wide timestamps par=" << p2 << " batch=" << batch << " wrong " << wrong << " cnt " << cnt
                 << (ok ? "  OK" : "  FAILED") << endl;
            failures += !ok;
        }
        {
            PipeGraph g("keyed_i2", Mode::DETERMINISTIC);
            auto src = BatchSource_Builder(StridedGen{&stream, 4}).withName("lanes").withParallelism(4).withBatch(batch).build_unique();
            MultiPipe& mp = g.add_source(*src);
            auto map = MapGPU_Builder(RunningSumOut()).withName("kmap").withParallelism(p1).withScratchpad(8).withMaxKeys(keys).build_unique();
            mp.add(*map);
            auto fil = FilterGPU_Builder(EveryThirdOut()).withName("kfilter").withParallelism(p2).withScratchpad(8).withMaxKeys(keys).build_unique();
            mp.add(*fil);
            long sum = 0, cnt = 0;
            Sink sink = Sink_Builder([&](std::optional<output_t>& o) {
                            if (o) {
                                sum += o->value;
                                cnt++;
                            }
                        }).withParallelism(1).build();
            mp.add_sink(sink);
            g.run();
            const bool ok = (sum == exp_b_sum && cnt == exp_b_cnt);
            cout << "I deterministic map+filter par=" << p1 << "," << p2 << " batch=" << batch << " sum " << sum << " cnt " << cnt << " (expected " << exp_b_sum
                 << " " << exp_b_cnt << ")" << (ok ? "  OK" : "  FAILED") << endl;
            failures += !ok;
        }
    }
    // ---- J: DETERMINISTIC graphs ACROSS ranks: `world` ranks (threads over LocalTransport, one GPU), each with 2 source lanes that
    //      overlap in time with every other lane of every rank; Exchange_GPU orders its output lanes by timestamp under the
    //      low-watermark rule over all the lanes of all the ranks, so the keyed map behind it computes, for every tuple, the
    //      running sum its key had at that point of the SEQUENTIAL stream ----
    for (int world : {2, 3}) {
        const size_t lanes_per_rank = 2, out_lanes = (size_t)dp(rng);
        const size_t batch = 64 * (size_t)db(rng);
        vector<int64_t> exp_val(stream.size());
        {
            map<size_t, int64_t> acc;
            for (size_t i = 0; i < stream.size(); i++) exp_val[i] = (acc[stream[i].key] += stream[i].value);
        }
        auto fabric = std::make_shared<dist::LocalFabric>(world);
        std::atomic<long> wrong{0}, cnt{0}, foreign{0};
        vector<std::thread> threads;
        for (int r = 0; r < world; r++) {
            threads.emplace_back([&, r] {
                auto tr = std::make_shared<dist::LocalTransport>(fabric, r);
                PipeGraph g("keyed_j_rank" + to_string(r), Mode::DETERMINISTIC);
                auto src = BatchSource_Builder(StridedGen{&stream, lanes_per_rank * (size_t)world, lanes_per_rank * (size_t)r}).withName("lanes")
                               .withParallelism(lanes_per_rank).withBatch(batch).build_unique();
                MultiPipe& mp = g.add_source(*src);
                auto ex = ExchangeGPU_Builder<tuple_t>(tr).withName("exchange").withParallelism(lanes_per_rank).withLocalLanes(lanes_per_rank)
                              .withOutputLanes(out_lanes).build_unique();
                mp.add(*ex);
                auto map = MapGPU_Builder(RunningSum()).withName("kmap").withParallelism(out_lanes).withScratchpad(8).withMaxKeys(keys).withKeyDisjointLanes()
                               .build_unique();
                mp.add(*map);
                Sink sink = Sink_Builder([&, r](std::optional<tuple_t>& o) {
                                if (o) {
                                    if (o->value != exp_val[o->id]) wrong++;
                                    if (gpu::keyby_destination(o->key, (uint32_t)world, gpu::HASH_SHIFT_RANKS) != (uint32_t)r) foreign++;
                                    cnt++;
                                }
                            }).withParallelism(1).build();
                mp.add_sink(sink);
                g.run();
            });
        }
        for (auto& t : threads) t.join();
        const bool ok = (wrong == 0 && foreign == 0 && cnt == (long)n);
        cout << "J deterministic across ranks world=" << world << " lanes=" << out_lanes << " batch=" << batch << " wrong " << wrong << " foreign " << foreign
             << " cnt " << cnt << (ok ? "  OK" : "  FAILED") << endl;
        failures += !ok;
    }
    // ---- E (whenever 2+ GPUs are visible; WF_SKIP_MULTI_GPU=1 opts out): scenario B with the replicas of the two keyed
    //      operators placed on different GPUs; the sub-batches of the device KeyBy cross GPUs with peer copies ----
    if (gpu::device_count() >= 2 && std::getenv("WF_SKIP_MULTI_GPU") == nullptr) {
        const size_t batch = 1024;
        PipeGraph g("keyed_e", Mode::DEFAULT);
        auto src = BatchSource_Builder(Gen(n, keys)).withParallelism(1).withBatch(batch).build_unique();
        MultiPipe& mp = g.add_source(*src);
        auto map = MapGPU_Builder(RunningSumOut()).withName("kmap").withParallelism(2).withScratchpad(8).withMaxKeys(keys)
                       .withGPUPlacement({0, 1}).build_unique();
        mp.add(*map);
        auto fil = FilterGPU_Builder(EveryThirdOut()).withName("kfilter").withParallelism(2).withScratchpad(8).withMaxKeys(keys)
                       .withGPUPlacement({1, 0}).build_unique();
        mp.add(*fil);
        long sum = 0, cnt = 0;
        Sink sink = Sink_Builder([&](std::optional<output_t>& o) {
                        if (o) {
                            sum += o->value;
                            cnt++;
                        }
                    }).withParallelism(1).build();
        mp.add_sink(sink);
        g.run();
        const bool ok = (sum == exp_b_sum && cnt == exp_b_cnt);
        cout << "E two GPUs sum " << sum << " cnt " << cnt << (ok ? "  OK" : "  FAILED") << endl;
        failures += !ok;
    }
    (void)EveryThird{};
    return failures == 0 ? 0 : 1;
}
