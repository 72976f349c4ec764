// Device side of the Yahoo! Streaming Benchmark: functors callable on the GPU, the bulk event source that writes
// straight into pinned slabs, and the batch sink.  Shared by apps/ysb/ysb_gpu.hip (one process, one or more local
// replicas), apps/ysb/ysb_gpu_dist.hip (one process per GPU, campaigns partitioned across ranks through Exchange_GPU)
// and the pipeline bench.py drives.  The whole query stays in HBM:
//   Batch_Source -> Filter_GPU (event_type == view) -> Map_GPU (hash join ad -> campaign record) -> [KeyBy / Exchange_GPU]
//                -> Key_FFAT_GPU (10 s tumbling time window per campaign: count + last update) -> Batch_Sink
// Reference harness: /root/reference/src/yahoo_test_cpu/ysb_nodes.hpp:54-220 (per-tuple CPU operators, Key_Farm or
// Win_MapReduce aggregation).
#pragma once
#include <map>
#include <memory>
#include <mutex>

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
// ---- stream-table join on the device ------------------------------------------------------------------------------------
// Reference: YSBJoin (/root/reference/src/yahoo_test_cpu/ysb_nodes.hpp:147-175) probes an unordered_map ad id -> row and reads
// the campaign_record of that row from a relational table, per view.  The same two steps here, in HBM: an open-addressing
// hash table (ad id -> row, linear probing, load factor <= 1/2) and the table of records; both are a few tens of KB and stay
// in the vector caches of the CUs.  One copy per GPU, uploaded by the replica that first runs there (Map_GPU bind_device).
struct CampaignRecord {
    uint64_t ad_id, cmp_id;
};
class DeviceCampaignTables {
public:
    static constexpr unsigned long long EMPTY = ~0ull;
    struct View {
        const unsigned long long* ad_keys = nullptr;  // [mask + 1] ad id or EMPTY
        const uint32_t* rows = nullptr;               // [mask + 1] row of the ad in `records`
        const CampaignRecord* records = nullptr;
        uint32_t mask = 0;
    };
    explicit DeviceCampaignTables(const Campaigns& c) {
        size_t cap = 16;
        while (cap < 2 * c.n_ads()) cap <<= 1;
        keys_.assign(cap, EMPTY);
        rows_.assign(cap, 0);
        for (const auto& kv : c.ad_to_campaign()) {
            uint32_t h = (uint32_t)wf::gpu::mix64(kv.first) & (uint32_t)(cap - 1);
            while (keys_[h] != EMPTY) h = (h + 1) & (uint32_t)(cap - 1);
            keys_[h] = kv.first;
            rows_[h] = (uint32_t)records_.size();
            records_.push_back(CampaignRecord{kv.first, kv.second});
        }
    }
    View view(int device) {
        std::lock_guard<std::mutex> l(m_);
        auto it = per_device_.find(device);
        if (it == per_device_.end()) {
            WF_HIP_CHECK(hipSetDevice(device));
            auto d = std::make_unique<OnDevice>();
            d->keys.resize(keys_.size());
            d->rows.resize(rows_.size());
            d->records.resize(records_.size());
            WF_HIP_CHECK(hipMemcpy(d->keys.data(), keys_.data(), keys_.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
            WF_HIP_CHECK(hipMemcpy(d->rows.data(), rows_.data(), rows_.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            WF_HIP_CHECK(hipMemcpy(d->records.data(), records_.data(), records_.size() * sizeof(CampaignRecord), hipMemcpyHostToDevice));
            it = per_device_.emplace(device, std::move(d)).first;
        }
        return View{it->second->keys.data(), it->second->rows.data(), it->second->records.data(), (uint32_t)(keys_.size() - 1)};
    }

private:
    struct OnDevice {
        wf::gpu::DeviceBuffer<unsigned long long> keys;
        wf::gpu::DeviceBuffer<uint32_t> rows;
        wf::gpu::DeviceBuffer<CampaignRecord> records;
    };
    std::vector<unsigned long long> keys_;
    std::vector<uint32_t> rows_;
    std::vector<CampaignRecord> records_;
    std::mutex m_;
    std::map<int, std::unique_ptr<OnDevice>> per_device_;
};
// the join functor: `tables` must outlive the graph (it owns the device copies)
struct JoinGPU {
    DeviceCampaignTables* tables;
    DeviceCampaignTables::View v;
    explicit JoinGPU(DeviceCampaignTables& t) : tables(&t) {}
    void bind_device(int gpu_id) { v = tables->view(gpu_id); }
    __host__ __device__ void operator()(const Event& e, Joined& j) const {
        j.ts = e.ts;
        j.seq = e.seq;
        j.ad_id = e.ad_id;
        j.cmp_id = 0xffffffffull;  // unknown ad: belongs to no campaign (the reference drops the event, ysb_nodes.hpp:163-165)
        uint32_t h = (uint32_t)wf::gpu::mix64(e.ad_id) & v.mask;
        for (uint32_t probes = 0; probes <= v.mask; probes++) {
            const unsigned long long k = v.ad_keys[h];
            if (k == (unsigned long long)e.ad_id) {
                const CampaignRecord rec = v.records[v.rows[h]];
                j.ad_id = rec.ad_id;
                j.cmp_id = rec.cmp_id;
                return;
            }
            if (k == DeviceCampaignTables::EMPTY) return;
            h = (h + 1) & v.mask;
        }
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
