// Device side of the Yahoo! Streaming Benchmark: functors callable on the GPU, the bulk event source that writes
// straight into pinned slabs, and the batch sink.  Shared by apps/ysb/ysb_gpu.hip (one process, one or more local
// replicas), apps/ysb/ysb_gpu_dist.hip (one process per GPU, campaigns partitioned across ranks through Exchange_GPU)
// and the pipeline bench.py drives.  The whole query stays in HBM:
//   Batch_Source -> Filter_GPU (event_type == view) -> Map_GPU (ad -> campaign) -> [KeyBy / Exchange_GPU]
//                -> Key_FFAT_GPU (10 s tumbling time window per campaign: count + last update) -> Batch_Sink
// Reference harness: /root/reference/src/yahoo_test_cpu/ysb_nodes.hpp:54-220 (per-tuple CPU operators, Key_Farm or
// Win_MapReduce aggregation).
#pragma once
#include <wf/windflow_gpu.hpp>

#include "ysb_common.hpp"

namespace ysb {

struct ViewFilterGPU {
    __host__ __device__ bool operator()(const Event& e) const { return e.event_type == 0; }
};
// split/merge variant of the query: the events are split by ad type into two branches (each filters and joins its share)
// that are merged again in front of the aggregation
struct AdTypeSplit {
    __host__ __device__ size_t operator()(const Event& e) const { return e.ad_type & 1u; }
};
// stream-table join without a table: ad ids are a fixed permutation of [0, n_ads) (ysb::Campaigns), inverted arithmetically
struct JoinGPU {
    uint64_t n_ads, per_campaign, inv_a, b;
    explicit JoinGPU(const Campaigns& c) : n_ads(c.n_ads()), per_campaign(c.ads_per_campaign()), inv_a(c.inverse_multiplier()), b(Campaigns::B) {}
    __host__ __device__ void operator()(const Event& e, Joined& j) const {
        j.ts = e.ts;
        j.seq = e.seq;
        j.ad_id = e.ad_id;
        const uint64_t pos = ((e.ad_id % n_ads + n_ads - b % n_ads) % n_ads) * inv_a % n_ads;
        j.cmp_id = e.ad_id < n_ads ? pos / per_campaign : 0xffffffffull;
    }
};
// incremental aggregation of a window: number of views and the latest of them
struct CountLift {
    __host__ __device__ void operator()(const Joined& j, WinResult& r) const {
        r.count = 1;
        r.last_update = j.ts;
    }
};
struct CountCombine {
    __host__ __device__ void operator()(const WinResult& a, const WinResult& b, WinResult& out) const {
        out.count = a.count + b.count;
        out.last_update = a.last_update > b.last_update ? a.last_update : b.last_update;
    }
};
// the same aggregation as an opaque non-incremental window function (Key_Farm_GPU)
struct CountWindowGPU {
    __host__ __device__ void operator()(uint64_t, const Joined* data, size_t n, WinResult* res, char*, size_t) const {
        uint64_t last = 0;
        for (size_t i = 0; i < n; i++) last = data[i].ts > last ? data[i].ts : last;
        res->count = n;
        res->last_update = last;
    }
};

// Bulk event source (Batch_Source form bool(Event*, size_t cap, size_t& n)): same event mix as ysb::EventSource.
// timed mode: ts = wall clock since the start of the run, read once per 256 events; counted mode: `events` events
// `period_us` apart on a synthetic clock (deterministic).  `index` / `stride` interleave the sequence numbers of several
// sources of a job (each source draws every ad).
class BulkEventSource {
public:
    BulkEventSource(const Campaigns* c, Metrics* m, uint64_t seconds, uint64_t events, uint64_t period_us)
        : m_(m), seconds_(seconds), events_(events), period_(period_us), n_ads_(c->n_ads()) {
        ads_.resize(n_ads_);
        for (size_t i = 0; i < n_ads_; i++) ads_[i] = c->ad(i);
    }
    bool operator()(Event* buf, size_t cap, size_t& n) {
        uint64_t v = value_;
        bool more = true;
        size_t i = 0;
        uint64_t now_ts = 0;
        for (; i < cap && more; i++, v++) {
            Event& e = buf[i];
            if (events_ > 0) {
                e.ts = v * period_;
                more = v + 1 < events_;
            } else {
                if ((i & 255) == 0) now_ts = wf::current_time_usecs() - m_->start_us;
                e.ts = now_ts;
                more = now_ts < seconds_ * 1000000ull;
            }
            const uint64_t x = v % 100000;
            e.seq = v;
            e.ad_id = ads_[x % n_ads_];
            e.user_id = 0;
            e.page_id = 0;
            e.ad_type = (uint32_t)(x % 5);
            e.event_type = (uint32_t)(x % 3);
            e.ip = 1;
        }
        value_ = v;
        n = i;
        if (!more) m_->generated += value_;
        return more;
    }

private:
    Metrics* m_;
    uint64_t seconds_, events_, period_;
    size_t n_ads_;
    std::vector<uint64_t> ads_;
    uint64_t value_ = 0;
};

// Batch_Sink functor: window results arrive a batch at a time
class BatchResultSink {
public:
    explicit BatchResultSink(Metrics* m) : m_(m) {}
    void operator()(const WinResult* r, size_t n) {
        if (r == nullptr) {  // end of stream
            m_->results += received_;
            m_->counted_events += events_;
            std::lock_guard<std::mutex> l(m_->m);
            m_->latencies_us.insert(m_->latencies_us.end(), local_.begin(), local_.end());
            return;
        }
        const uint64_t now = wf::current_time_usecs();
        for (size_t i = 0; i < n; i++) {
            if (r[i].count == 0) continue;  // windows without views are not reported (reference: ysb_nodes.hpp:200)
            received_++;
            events_ += r[i].count;
            const uint64_t born = r[i].last_update + m_->start_us;
            local_.push_back(now > born ? now - born : 0);
        }
    }

private:
    Metrics* m_;
    uint64_t received_ = 0, events_ = 0;
    std::vector<uint64_t> local_;
};

}  // namespace ysb
