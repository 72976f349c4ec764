// Shared fixtures of the MultiPipe regression programs: tuple types, a
// deterministic source, the Filter / FlatMap / Map functors, window functions in
// every accepted flavour, a sink, and a brute-force oracle.
//
// The workload mirrors the reference's suite (/root/reference/src/mp_test_cpu/
// mp_common.hpp:45-314: keys round-robin, per-key progressive ids, Pareto
// inter-arrival times with a fixed seed, "drop odd", "x3", "double", window sum)
// but unlike the reference - whose only oracle is run-to-run equality under random
// parallelism - every run is also checked against values computed sequentially
// here (SURVEY.md §4 "we must add a CPU reference oracle").
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <map>
#include <numeric>
#include <optional>
#include <string>
#include <tuple>
#include <vector>

#include <wf/windflow.hpp>

// Key type of the suite.  -DWF_TEST_FIXED_KEY builds the "string key" twins of the programs (reference:
// src/mp_test_cpu/mp_common_string.hpp, src/mp_test_gpu/mp_common_string.hpp:44-84 and the four test_mp_gpu_*_cb_string.cpp):
// keys "key_0", "key_1" ... as 16-byte fixed-size keys, which - unlike std::string - can live in HBM.
#ifdef WF_TEST_FIXED_KEY
using test_key_t = wf::FixedKey<16>;
inline test_key_t make_key(size_t k) { return test_key_t("key_" + std::to_string(k)); }
#else
using test_key_t = size_t;
inline test_key_t make_key(size_t k) { return k; }
#endif

struct tuple_t {
    test_key_t key{};
    uint64_t id = 0;
    uint64_t ts = 0;
    int64_t value = 0;
    WF_HOST_DEVICE tuple_t() {}
    WF_HOST_DEVICE tuple_t(test_key_t k, uint64_t i, uint64_t t, int64_t v) : key(k), id(i), ts(t), value(v) {}
    WF_HOST_DEVICE std::tuple<test_key_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    WF_HOST_DEVICE void setControlFields(test_key_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};

struct output_t {
    test_key_t key{};
    uint64_t id = 0;
    uint64_t ts = 0;
    int64_t value = 0;
    WF_HOST_DEVICE output_t() {}
    WF_HOST_DEVICE std::tuple<test_key_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    WF_HOST_DEVICE void setControlFields(test_key_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};

// deterministic Pareto inter-arrival generator (own LCG: independent of rand())
struct ArrivalClock {
    uint64_t state = 12345;
    uint64_t next_gap() {
        state = state * 6364136223846793005ULL + 1442695040888963407ULL;
        const double u = ((state >> 11) + 1) * (1.0 / 9007199254740993.0);  // (0,1]
        const double kappa = (1000 * 0.05) / 1.05;
        const double g = kappa / std::pow(u, 1.0 / 1.05);
        return (uint64_t)std::ceil(g > 1e7 ? 1e7 : g);
    }
};

// generates len tuples per key, keys round-robin, ids progressive per key
class Source_Functor {
public:
    Source_Functor(size_t len, size_t keys) : len_(len), keys_(keys), ids_(keys, 0) {}
    bool operator()(tuple_t& t) {
        t.setControlFields(make_key(k_), ids_[k_], next_ts_);
        t.value = (int64_t)ids_[k_]++;
        sent_++;
        k_ = (k_ + 1) % keys_;
        next_ts_ += clock_.next_gap();
        return sent_ < len_ * keys_;
    }

private:
    size_t len_, keys_, k_ = 0, sent_ = 0;
    std::vector<uint64_t> ids_;
    uint64_t next_ts_ = 0;
    ArrivalClock clock_;
};

struct Filter_Functor {
    bool operator()(tuple_t& t) { return t.value % 2 == 0; }
};
struct FlatMap_Functor {
    void operator()(const tuple_t& t, wf::Shipper<tuple_t>& shipper) {
        for (size_t i = 0; i < 3; i++) {
            tuple_t t2 = t;
            t2.value = t.value + (int64_t)i;
            t2.ts = t.ts + i;  // keeps timestamps distinct inside a key
            shipper.push(t2);
        }
    }
};
struct Map_Functor {
    void operator()(tuple_t& t) { t.value = t.value * 2; }
};

// ---- window functions ---------------------------------------------------------
inline void win_nic(uint64_t, const wf::Iterable<tuple_t>& input, output_t& result) {
    int64_t sum = 0;
    for (const auto& t : input) sum += t.value;
    result.value = sum;
}
inline void win_inc(uint64_t, const tuple_t& t, output_t& result) { result.value += t.value; }
inline void win2_nic(uint64_t, const wf::Iterable<output_t>& input, output_t& result) {
    int64_t sum = 0;
    for (const auto& t : input) sum += t.value;
    result.value = sum;
}
inline void win2_inc(uint64_t, const output_t& t, output_t& result) { result.value += t.value; }
struct Rich_Win_NIC {
    void operator()(uint64_t, const wf::Iterable<tuple_t>& input, output_t& result, wf::RuntimeContext& ctx) const {
        (void)ctx.getReplicaIndex();
        int64_t sum = 0;
        for (size_t i = 0; i < input.size(); i++) sum += input[i].value;
        result.value = sum;
    }
};
inline void lift_f(const tuple_t& t, output_t& r) { r.value = t.value; }
inline void comb_f(const output_t& a, const output_t& b, output_t& out) { out.value = a.value + b.value; }

// ---- sink -----------------------------------------------------------------------
struct SinkTotals {
    std::atomic<long> sum{0};
    std::atomic<long> received{0};
    std::atomic<long> order_violations{0};
};
class Sink_Functor {
public:
    explicit Sink_Functor(SinkTotals* t) : totals_(t) {}
    void operator()(std::optional<output_t>& out) {
        if (out) {
            totals_->received++;
            totals_->sum += out->value;
            auto it = last_id_.find(out->key);
            if (it != last_id_.end() && out->id <= it->second) totals_->order_violations++;
            last_id_[out->key] = out->id;
        }
    }

private:
    SinkTotals* totals_;
    std::map<test_key_t, uint64_t> last_id_;
};

// ---- oracle ---------------------------------------------------------------------
// the stream after Source -> Filter(even) -> FlatMap(x3) -> Map(x2), per key, in
// timestamp order
inline std::map<test_key_t, std::vector<tuple_t>> oracle_stream(size_t len, size_t keys) {
    Source_Functor src(len, keys);
    Filter_Functor fil;
    Map_Functor mp;
    std::map<test_key_t, std::vector<tuple_t>> per_key;
    bool more = true;
    while (more) {
        tuple_t t;
        more = src(t);
        if (!fil(t)) continue;
        for (size_t i = 0; i < 3; i++) {
            tuple_t t2 = t;
            t2.value = t.value + (int64_t)i;
            t2.ts = t.ts + i;
            mp(t2);
            per_key[t2.key].push_back(t2);
        }
    }
    for (auto& kv : per_key)
        std::stable_sort(kv.second.begin(), kv.second.end(), [](const tuple_t& a, const tuple_t& b) { return a.ts < b.ts; });
    return per_key;
}

struct OracleResult {
    long sum = 0;
    long windows = 0;
};
// count-based windows over the arrival (timestamp) order of each key; every opened
// window is emitted (partial ones at end of stream)
inline OracleResult oracle_cb(const std::map<test_key_t, std::vector<tuple_t>>& s, uint64_t win, uint64_t slide) {
    OracleResult r;
    for (auto& kv : s) {
        const auto& v = kv.second;
        const uint64_t n = v.size();
        if (n == 0) continue;
        uint64_t nw;
        if (win >= slide) {
            nw = (n + slide - 1) / slide;
        } else {  // hopping: window w covers [w*slide, w*slide+win); opened up to the one reached by the last tuple
            nw = (n - 1) / slide + 1;
        }
        for (uint64_t w = 0; w < nw; w++) {
            for (uint64_t i = w * slide; i < std::min<uint64_t>(n, w * slide + win); i++) r.sum += v[i].value;
            r.windows++;
        }
    }
    return r;
}
// time-based windows [w*slide, w*slide+win)
inline OracleResult oracle_tb(const std::map<test_key_t, std::vector<tuple_t>>& s, uint64_t win, uint64_t slide) {
    OracleResult r;
    for (auto& kv : s) {
        const auto& v = kv.second;
        if (v.empty()) continue;
        const uint64_t last_ts = v.back().ts;
        uint64_t nw;
        if (win >= slide)
            nw = (last_ts + 1 + slide - 1) / slide;
        else
            nw = last_ts / slide + 1;
        size_t lo = 0;
        for (uint64_t w = 0; w < nw; w++) {
            const uint64_t a = w * slide, b = a + win;
            while (lo < v.size() && v[lo].ts < a) lo++;
            for (size_t i = lo; i < v.size() && v[i].ts < b; i++) r.sum += v[i].value;
            r.windows++;
        }
    }
    return r;
}
