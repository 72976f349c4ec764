// Ordering_Node: K-way merge of the input channels of a replica; this is the
// low-watermark mechanism of DETERMINISTIC mode.
//
// Semantics follow /root/reference/wf/ordering_node.hpp:124-279 (SURVEY.md A.7,
// A.11): buffer tuples in a priority queue, remember the greatest id/ts seen per
// input channel, release everything <= min over channels; mode ID orders per key
// on the id field, TS orders globally on the timestamp, TS_RENUMBERING
// additionally rewrites ids as a per-key progressive counter so count-based
// windows work after a shuffle.  End-of-stream markers are held back and the
// most recent one per key is released last.
//
// Differences by design: (1) ties are broken by (channel, arrival sequence), not
// by pointer value, so N ordering nodes fed by the same broadcast agree on one
// total order even though each receives its own copy of the tuples; (2) a
// channel that ended stops holding back the watermark.
#pragma once
#include <algorithm>
#include <cstdint>
#include <limits>
#include <queue>
#include <unordered_map>
#include <vector>

#include "basic.hpp"
#include "rt/node.hpp"

namespace wf {

template <typename tuple_t>
class Ordering_Node : public rt::Node {
public:
    explicit Ordering_Node(ordering_mode_t mode = ID) : mode_(mode) {}

    void svc_init() override { global_maxs_.assign(num_in(), 0); }

    void svc(void* msg, int channel) override {
        tuple_t* t = untag<tuple_t>(msg);
        const auto key = key_of(*t);
        const uint64_t ord = (mode_ == ID) ? id_of(*t) : ts_of(*t);
        auto it = keys_.find(key);
        if (it == keys_.end()) {
            it = keys_.emplace(key, KeyState(num_in())).first;
            for (int c : ended_) it->second.maxs[c] = std::numeric_limits<uint64_t>::max();
        }
        KeyState& ks = it->second;
        if (is_marker(msg)) {
            if (ks.marker == nullptr) {
                ks.marker = t;
            } else {
                const uint64_t old = (mode_ == ID) ? id_of(*ks.marker) : ts_of(*ks.marker);
                if (ord > old) {
                    tuple_delete(ks.marker);
                    ks.marker = t;
                } else {
                    tuple_delete(t);
                }
            }
            return;
        }
        Entry e{ord, channel, seq_++, t};
        if (mode_ == ID) {
            ks.maxs[channel] = std::max(ks.maxs[channel], ord);
            ks.queue.push(e);
            drain(ks.queue, *std::min_element(ks.maxs.begin(), ks.maxs.end()));
        } else {
            global_maxs_[channel] = std::max(global_maxs_[channel], ord);
            queue_.push(e);
            drain(queue_, *std::min_element(global_maxs_.begin(), global_maxs_.end()));
        }
    }

    void eosnotify(int channel) override {
        // an ended channel no longer constrains the watermark
        constexpr uint64_t inf = std::numeric_limits<uint64_t>::max();
        if (mode_ == ID) {
            for (auto& kv : keys_) {
                kv.second.maxs[channel] = inf;
                drain(kv.second.queue, *std::min_element(kv.second.maxs.begin(), kv.second.maxs.end()));
            }
            ended_.push_back(channel);
        } else {
            if ((size_t)channel < global_maxs_.size()) global_maxs_[channel] = inf;
            if (!global_maxs_.empty()) drain(queue_, *std::min_element(global_maxs_.begin(), global_maxs_.end()));
        }
    }

    void flush() override {
        constexpr uint64_t inf = std::numeric_limits<uint64_t>::max();
        if (mode_ == ID) {
            for (auto& kv : keys_) drain(kv.second.queue, inf);
        } else {
            drain(queue_, inf);
        }
        // release the most recent marker of each key, flagged as marker again
        for (auto& kv : keys_) {
            KeyState& ks = kv.second;
            if (ks.marker != nullptr) {
                if (mode_ == TS_RENUMBERING) ks.marker->setControlFields(kv.first, ks.emit_counter++, ts_of(*ks.marker));
                emit(tag_marker(ks.marker));
                ks.marker = nullptr;
            }
        }
    }

private:
    struct Entry {
        uint64_t ord;
        int ch;
        uint64_t seq;
        tuple_t* t;
    };
    struct Later {
        bool operator()(const Entry& a, const Entry& b) const {
            if (a.ord != b.ord) return a.ord > b.ord;
            if (a.ch != b.ch) return a.ch > b.ch;
            return a.seq > b.seq;
        }
    };
    using pq_t = std::priority_queue<Entry, std::vector<Entry>, Later>;
    struct KeyState {
        explicit KeyState(size_t n) : maxs(n, 0) {}
        uint64_t emit_counter = 0;
        std::vector<uint64_t> maxs;
        tuple_t* marker = nullptr;
        pq_t queue;
    };

    void drain(pq_t& q, uint64_t watermark) {
        while (!q.empty() && q.top().ord <= watermark) {
            tuple_t* t = q.top().t;
            q.pop();
            if (mode_ == TS_RENUMBERING) {
                const auto key = key_of(*t);
                KeyState& ks = keys_.find(key)->second;
                t->setControlFields(key, ks.emit_counter++, ts_of(*t));
            }
            emit(t);
        }
    }

    ordering_mode_t mode_;
    std::unordered_map<key_of_t<tuple_t>, KeyState> keys_;
    pq_t queue_;
    std::vector<uint64_t> global_maxs_;
    std::vector<int> ended_;
    uint64_t seq_ = 0;
};

}  // namespace wf
