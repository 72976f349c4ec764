// Shared pieces of the benchmark / flagship pipelines: the 40-byte sensor tuple of
// the reference benchmarks (/root/reference/src/GPU_Tests/new_tests/benchmarks/
// gpu_map_stateful.cpp:80-85), a fast synthetic generator, and the step gate /
// barrier that lets a driver time "K steps" on a persistent streaming graph.
#pragma once
#include <cstdlib>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <map>
#include <mutex>
#include <tuple>

#include <wf/basic.hpp>
#include <wf/context.hpp>

namespace wfbench {

// 40 bytes, same field mix as the reference benchmark tuple
struct tuple_t {
    uint64_t key;
    uint64_t id;
    uint64_t ts;
    double value;
    double aux;
    __host__ __device__ std::tuple<uint64_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    __host__ __device__ void setControlFields(uint64_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};
static_assert(sizeof(tuple_t) == 40, "benchmark tuple must be 40 bytes");

// 32-byte window result
struct result_t {
    uint64_t key;
    uint64_t id;
    uint64_t ts;
    double value;
    __host__ __device__ std::tuple<uint64_t, uint64_t, uint64_t> getControlFields() const { return {key, id, ts}; }
    __host__ __device__ void setControlFields(uint64_t k, uint64_t i, uint64_t t) {
        key = k;
        id = i;
        ts = t;
    }
};
static_assert(sizeof(result_t) == 32, "benchmark result must be 32 bytes");

// lift / combine of the flagship sliding-window sum
struct LiftValue {
    __host__ __device__ void operator()(const tuple_t& t, result_t& r) const { r.value = t.value; }
};
struct CombSum {
    __host__ __device__ void operator()(const result_t& a, const result_t& b, result_t& out) const { out.value = a.value + b.value; }
};

// A driver releases `quota` tuples per step; the source closes the step with a
// punctuated batch and sleeps until the next step (or stop).
struct StepGate {
    std::mutex m;
    std::condition_variable cv;
    int64_t quota = 0;      // tuples still to generate in the current step
    uint64_t step_id = 0;   // id of the step the quota belongs to
    bool stop = false;

    void release(uint64_t step, int64_t tuples) {
        {
            std::lock_guard<std::mutex> l(m);
            step_id = step;
            quota = tuples;
        }
        cv.notify_all();
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> l(m);
            stop = true;
        }
        cv.notify_all();
    }
};

struct StepBarrier {
    std::mutex m;
    std::condition_variable cv;
    std::map<uint64_t, int> arrivals;
    void arrive(uint64_t step) {
        {
            std::lock_guard<std::mutex> l(m);
            arrivals[step]++;
        }
        cv.notify_all();
    }
    void wait(uint64_t step, int expected) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return arrivals[step] >= expected; });
        arrivals.erase(step);
    }
};

// Bulk generator functor for Batch_Source:  bool(tuple_t*, size_t cap, size_t& n, RuntimeContext&)
// keys are drawn round-robin-with-stride so every key receives in-order ids, ts is
// a monotone per-source microsecond clock.
class GatedGenerator {
public:
    // src_offset / src_stride: position of this process's sources among all the sources of the job
    // (rank * n_sources, world * n_sources), so keys stay disjoint across ranks as well
    GatedGenerator(StepGate* gates, size_t n_keys, uint64_t ts_step_us, size_t n_sources, size_t src_offset = 0, size_t src_stride = 0)
        : gates_(gates), n_keys_(n_keys), ts_step_(ts_step_us), n_sources_(src_stride ? src_stride : n_sources), src_offset_(src_offset) {}

    bool operator()(tuple_t* buf, size_t cap, size_t& n, wf::RuntimeContext& ctx) {
        StepGate& g = gates_[ctx.getReplicaIndex()];
        punct_ = 0;
        int64_t take = 0;
        {
            std::unique_lock<std::mutex> l(g.m);
            g.cv.wait(l, [&] { return g.quota > 0 || g.stop; });
            if (g.quota <= 0 && g.stop) {
                n = 0;
                return false;
            }
            take = g.quota < (int64_t)cap ? g.quota : (int64_t)cap;
            g.quota -= take;
            if (g.quota == 0) punct_ = g.step_id;  // last batch of the step
        }
        opened_ns_ = wf::current_time_nsecs();
        const uint64_t src = ctx.getReplicaIndex() + src_offset_;
        uint64_t seq = seq_, kc = kc_, round = round_;
        const uint64_t nk = n_keys_, ns = n_sources_, tstep = ts_step_;
        const bool nt = nt_stores_;
        // keys of different sources are disjoint (key % n_sources == src): per-key order is
        // then well defined without cross-source ordering
        for (int64_t i = 0; i < take; i++, seq++) {
            if (nt) {
                // non-temporal stores: the slab is only read back by the DMA engine; not pulling its lines into the
                // cache first saves the read-for-ownership traffic when many ranks share the host memory system
                __builtin_nontemporal_store(kc * ns + src, &buf[i].key);
                __builtin_nontemporal_store(round, &buf[i].id);
                __builtin_nontemporal_store(seq * tstep, &buf[i].ts);
                __builtin_nontemporal_store((double)(seq & 1023), &buf[i].value);
                __builtin_nontemporal_store(1.0, &buf[i].aux);
            } else {
                buf[i].key = kc * ns + src;
                buf[i].id = round;
                buf[i].ts = seq * tstep;
                buf[i].value = (double)(seq & 1023);
                buf[i].aux = 1.0;
            }
            if (++kc == nk) {
                kc = 0;
                round++;
            }
        }
        seq_ = seq;
        kc_ = kc;
        round_ = round;
        n = (size_t)take;
        return true;
    }
    // Batch_Source hook: non-zero closes a step
    uint64_t punctuation() const { return punct_; }
    // Batch_Source hook: when the first tuple of the batch was generated
    uint64_t opened_ns() const { return opened_ns_; }

private:
    StepGate* gates_;
    size_t n_keys_;
    uint64_t ts_step_;
    size_t n_sources_;
    size_t src_offset_ = 0;
    uint64_t seq_ = 0, kc_ = 0, round_ = 0;
    uint64_t punct_ = 0;
    uint64_t opened_ns_ = 0;
    // on by default (single GPU: same throughput, lower p99 latency; several ranks: less host memory traffic); WF_BENCH_NT_STORES=0 disables
    bool nt_stores_ = std::getenv("WF_BENCH_NT_STORES") == nullptr || std::atoi(std::getenv("WF_BENCH_NT_STORES")) != 0;
};

}  // namespace wfbench
