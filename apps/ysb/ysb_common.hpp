// Yahoo! Streaming Benchmark on windflow_amd: shared types, campaign table and
// functors.
//
// Behavioural parity with the reference harness (/root/reference/src/
// yahoo_test_cpu: yahoo_app.hpp event/result records, campaign_generator.hpp
// 100 campaigns x 10 ads + ad->campaign table, ysb_nodes.hpp:54-220 source ->
// filter(event_type == 0) -> join(ad -> campaign) -> 10 s tumbling time window
// per campaign counting events and tracking the last update -> sink measuring
// latency = now - (lastUpdate + start)).  Written from scratch; additions: a
// deterministic "counted" source mode (-e N events on a synthetic clock) so the
// applications can be verified against a closed-form oracle, and p50/p99
// latencies instead of the mean only (BASELINE.json asks for p99).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <mutex>
#include <optional>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include <unistd.h>

#include <wf/windflow.hpp>

namespace ysb {

constexpr uint64_t WINDOW_US = 10ull * 1000 * 1000;  // 10 s tumbling windows

struct Event {
    uint64_t ts = 0;          // microseconds since the start of the run
    uint64_t seq = 0;         // progressive id (per source)
    uint64_t ad_id = 0;
    uint64_t user_id = 0, page_id = 0;
    uint32_t ad_type = 0;     // 0..4
    uint32_t event_type = 0;  // 0 = view (kept), 1 = click, 2 = purchase
    uint32_t ip = 1;
    WF_HOST_DEVICE Event() {}
    WF_HOST_DEVICE std::tuple<uint64_t, uint64_t, uint64_t> getControlFields() const { return {ad_id, seq, ts}; }
    WF_HOST_DEVICE void setControlFields(uint64_t k, uint64_t i, uint64_t t) {
        ad_id = k;
        seq = i;
        ts = t;
    }
};

struct Joined {
    uint64_t ts = 0, seq = 0;
    uint64_t ad_id = 0;
    uint64_t cmp_id = 0;
    WF_HOST_DEVICE Joined() {}
    WF_HOST_DEVICE std::tuple<uint64_t, uint64_t, uint64_t> getControlFields() const { return {cmp_id, seq, ts}; }
    WF_HOST_DEVICE void setControlFields(uint64_t k, uint64_t i, uint64_t t) {
        cmp_id = k;
        seq = i;
        ts = t;
    }
};

struct WinResult {
    uint64_t cmp_id = 0, wid = 0, ts = 0;
    uint64_t count = 0;
    uint64_t last_update = 0;  // greatest event timestamp seen by the window
    WF_HOST_DEVICE WinResult() {}
    WF_HOST_DEVICE std::tuple<uint64_t, uint64_t, uint64_t> getControlFields() const { return {cmp_id, wid, ts}; }
    WF_HOST_DEVICE void setControlFields(uint64_t k, uint64_t i, uint64_t t) {
        cmp_id = k;
        wid = i;
        ts = t;
    }
};

// campaigns and their ads.  Ad ids are a fixed permutation of [0, n_ads): ad(i) = (A*i + B) mod n_ads.  The CPU versions
// join through the hash map below (as the reference does), the GPU versions through a device hash table + record table built
// from it (ysb_gpu_common.hpp DeviceCampaignTables).
class Campaigns {
public:
    static constexpr uint64_t A = 7919, B = 123;
    explicit Campaigns(size_t n_campaigns = 100, size_t ads_per_campaign = 10) : n_cmp_(n_campaigns), per_(ads_per_campaign) {
        const uint64_t n = n_cmp_ * per_;
        ads_.resize(n);
        for (uint64_t i = 0; i < n; i++) {
            ads_[i] = (A * i + B) % n;
            by_ad_[ads_[i]] = i / per_;
        }
        if (by_ad_.size() != n) wf::fatal_error("Campaigns: the number of ads must be coprime with 7919");
        for (uint64_t x = 1; x < n; x++)
            if ((A % n) * x % n == 1) inv_a_ = x;
        if (n == 1) inv_a_ = 0;
    }
    size_t n_campaigns() const { return n_cmp_; }
    size_t n_ads() const { return ads_.size(); }
    size_t ads_per_campaign() const { return per_; }
    uint64_t ad(size_t i) const { return ads_[i % ads_.size()]; }
    uint64_t inverse_multiplier() const { return inv_a_; }  // position of ad a = ((a + n - B % n) % n) * inverse_multiplier % n
    const std::unordered_map<uint64_t, uint64_t>& ad_to_campaign() const { return by_ad_; }

private:
    size_t n_cmp_, per_;
    std::vector<uint64_t> ads_;
    std::unordered_map<uint64_t, uint64_t> by_ad_;
    uint64_t inv_a_ = 0;
};

// run-wide counters
struct Metrics {
    uint64_t start_us = 0;
    std::atomic<uint64_t> generated{0}, results{0}, counted_events{0};
    std::mutex m;
    std::vector<uint64_t> latencies_us;
    void report(double elapsed_s, const char* title) {
        std::lock_guard<std::mutex> l(m);
        std::sort(latencies_us.begin(), latencies_us.end());
        auto pct = [&](double p) -> uint64_t {
            if (latencies_us.empty()) return 0;
            return latencies_us[std::min(latencies_us.size() - 1, (size_t)(p * (latencies_us.size() - 1)))];
        };
        long double sum = 0;
        for (auto v : latencies_us) sum += v;
        std::cout << "[" << title << "] generated events " << generated.load() << "\n"
                  << "[" << title << "] received results " << results.load() << "\n"
                  << "[" << title << "] events counted in windows " << counted_events.load() << "\n"
                  << "[" << title << "] latency usec mean " << (latencies_us.empty() ? 0.0 : (double)(sum / latencies_us.size())) << " p50 "
                  << pct(0.50) << " p99 " << pct(0.99) << "\n"
                  << "[" << title << "] elapsed seconds " << elapsed_s << "\n"
                  << "[" << title << "] throughput events/s " << (elapsed_s > 0 ? generated.load() / elapsed_s : 0.0) << std::endl;
    }
};

// Event source.  timed mode: runs for `seconds` of wall clock, ts = now - start
// (reference behaviour).  counted mode: exactly `events` events, ts advances by
// `period_us` per event (deterministic).
class EventSource {
public:
    EventSource(const Campaigns* c, Metrics* m, uint64_t seconds, uint64_t events, uint64_t period_us)
        : c_(c), m_(m), seconds_(seconds), events_(events), period_(period_us) {}
    bool operator()(Event& e) {
        const uint64_t v = value_++;
        bool more;
        if (events_ > 0) {
            e.ts = v * period_;
            more = value_ < events_;
        } else {
            const uint64_t now = wf::current_time_usecs();
            e.ts = now - m_->start_us;
            more = e.ts < seconds_ * 1000000ull;
        }
        e.seq = v;
        e.ad_id = c_->ad((v % 100000) % c_->n_ads());
        e.ad_type = (uint32_t)((v % 100000) % 5);
        e.event_type = (uint32_t)((v % 100000) % 3);
        if (!more) m_->generated += value_;
        return more;
    }

private:
    const Campaigns* c_;
    Metrics* m_;
    uint64_t seconds_, events_, period_;
    uint64_t value_ = 0;
};

struct ViewFilter {
    bool operator()(Event& e) const { return e.event_type == 0; }
};

// stream-table join: ad -> campaign; events of unknown ads are dropped
class CampaignJoin {
public:
    explicit CampaignJoin(const Campaigns* c) : c_(c) {}
    void operator()(const Event& e, wf::Shipper<Joined>& out) const {
        auto it = c_->ad_to_campaign().find(e.ad_id);
        if (it == c_->ad_to_campaign().end()) return;
        Joined j;
        j.ts = e.ts;
        j.seq = e.seq;
        j.ad_id = e.ad_id;
        j.cmp_id = it->second;
        out.push(std::move(j));
    }

private:
    const Campaigns* c_;
};

// incremental window function: count + last update
inline void aggregate_inc(uint64_t, const Joined& j, WinResult& r) {
    r.count++;
    r.last_update = std::max(r.last_update, j.ts);
}
// non-incremental REDUCE over the partial results of a window
inline void reduce_nic(uint64_t, const wf::Iterable<WinResult>& parts, WinResult& r) {
    for (const WinResult& p : parts) {
        r.count += p.count;
        r.last_update = std::max(r.last_update, p.last_update);
    }
}

class ResultSink {
public:
    explicit ResultSink(Metrics* m) : m_(m) {}
    void operator()(std::optional<WinResult>& r) {
        if (r) {
            if (r->count > 0) {
                received_++;
                events_ += r->count;
                // counted mode runs on a synthetic clock that may be ahead of the wall clock: clamp at zero
                const uint64_t now = wf::current_time_usecs(), born = r->last_update + m_->start_us;
                local_.push_back(now > born ? now - born : 0);
            }
        } else {
            m_->results += received_;
            m_->counted_events += events_;
            std::lock_guard<std::mutex> l(m_->m);
            m_->latencies_us.insert(m_->latencies_us.end(), local_.begin(), local_.end());
        }
    }

private:
    Metrics* m_;
    uint64_t received_ = 0, events_ = 0;
    std::vector<uint64_t> local_;
};

struct Options {
    uint64_t seconds = 0, events = 0, period_us = 100;
    size_t par1 = 1, par2 = 1, batch = 4096, gpus = 1;
    bool ok = true;
};
inline Options parse(int argc, char** argv, const char* extra_usage = "") {
    Options o;
    int c;
    while ((c = getopt(argc, argv, "l:e:p:n:m:b:g:")) != -1) {
        switch (c) {
            case 'l': o.seconds = std::strtoull(optarg, nullptr, 10); break;
            case 'e': o.events = std::strtoull(optarg, nullptr, 10); break;
            case 'p': o.period_us = std::strtoull(optarg, nullptr, 10); break;
            case 'n': o.par1 = std::strtoull(optarg, nullptr, 10); break;
            case 'm': o.par2 = std::strtoull(optarg, nullptr, 10); break;
            case 'b': o.batch = std::strtoull(optarg, nullptr, 10); break;
            case 'g': o.gpus = std::strtoull(optarg, nullptr, 10); break;
            default: o.ok = false;
        }
    }
    if (!o.ok || (o.seconds == 0 && o.events == 0)) {
        std::cout << argv[0] << " (-l <seconds> | -e <events per source> [-p <usec between events>]) -n <source/filter/join parallelism> "
                  << "-m <aggregation parallelism>" << extra_usage << std::endl;
        o.ok = false;
    }
    return o;
}

}  // namespace ysb
