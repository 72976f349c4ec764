// Routing and collecting nodes of the window-parallel operators.
//
//   WF_Emitter        window parallelism: a tuple goes to every replica owning a
//                     window that contains it (reference wf/wf_nodes.hpp:110-227)
//   WinMap_Emitter    intra-window striping: tuples of a key are dealt round-robin
//                     to the MAP replicas (reference wf/wm_nodes.hpp:91-159)
//   WinMap_Dropper    count-based WMR: every MAP replica sees the (renumbered)
//                     broadcast stream and keeps its own stripe (wm_nodes.hpp:219-251)
//   Ordered_Collector per-key reordering of results by progressive id; one class
//                     covers the reference's WF_Collector, KF_Collector and
//                     WinMap_Collector (wf_nodes.hpp:279-312, kf_nodes.hpp:116-180,
//                     wm_nodes.hpp:285-318), which are the same algorithm
// At end-of-stream the emitters send the last tuple of every key to all replicas
// as a *marker* (tagged pointer, basic.hpp) so that every replica opens and
// flushes the same set of windows regardless of parallelism.
#pragma once
#include <deque>
#include <unordered_map>
#include <vector>

#include "basic.hpp"
#include "rt/node.hpp"

namespace wf {

template <typename tuple_t>
class WF_Emitter : public rt::Node {
public:
    using key_t = key_of_t<tuple_t>;
    WF_Emitter(win_type_t type, uint64_t win_len, uint64_t slide_len, size_t pardegree, size_t id_outer, size_t n_outer, uint64_t slide_outer,
               role_t role)
        : type_(type), win_(win_len), slide_(slide_len), n_(pardegree), id_outer_(id_outer), n_outer_(n_outer), slide_outer_(slide_outer),
          role_(role) {
        for (size_t i = 0; i < n_; i++) add_out_port();
    }

    void svc(void* msg, int) override {
        tuple_t* t = untag<tuple_t>(msg);
        const key_t key = key_of(*t);
        const size_t hashcode = std::hash<key_t>()(key);
        const uint64_t ord = (type_ == CB) ? id_of(*t) : ts_of(*t);
        auto it = keys_.find(key);
        if (it == keys_.end()) {
            it = keys_.emplace(key, KeyD{}).first;
            it->second.last = *t;
            it->second.seen = true;
        } else {
            const uint64_t last = (type_ == CB) ? id_of(it->second.last) : ts_of(it->second.last);
            if (ord > last) it->second.last = *t;
        }
        if (is_marker(msg)) {  // markers from upstream only refresh the last tuple
            tuple_delete(t);
            return;
        }
        const uint64_t first_gwid_key = (id_outer_ + n_outer_ - (hashcode % n_outer_)) % n_outer_;
        const uint64_t initial_id = (role_ == WLQ || role_ == REDUCE) ? 0 : first_gwid_key * slide_outer_;
        if (ord < initial_id) {
            tuple_delete(t);
            return;
        }
        const uint64_t rel = ord - initial_id;
        uint64_t first_w, last_w;
        if (win_ >= slide_) {
            first_w = (rel + 1 < win_) ? 0 : (rel + 1 - win_ + slide_ - 1) / slide_;
            last_w = (rel + 1 + slide_ - 1) / slide_ - 1;
        } else {
            const uint64_t n = rel / slide_;
            if (rel >= n * slide_ + win_) {  // gap between hopping windows
                tuple_delete(t);
                return;
            }
            first_w = last_w = n;
        }
        // window w of the key belongs to replica (hash % N + w) % N
        const size_t start = hashcode % n_;
        size_t sent = 0;
        for (uint64_t w = first_w; w <= last_w && sent < n_; w++, sent++) {
            const size_t dst = (start + w) % n_;
            const bool last_copy = (w == last_w) || (sent + 1 == n_);
            emit_to(last_copy ? t : tuple_new<tuple_t>(*t), dst);
        }
    }

    void flush() override {
        for (auto& kv : keys_) {
            if (!kv.second.seen) continue;
            for (size_t i = 0; i < n_; i++) emit_to(tag_marker(tuple_new<tuple_t>(kv.second.last)), i);
        }
    }

private:
    struct KeyD {
        tuple_t last{};
        bool seen = false;
    };
    win_type_t type_;
    uint64_t win_, slide_;
    size_t n_, id_outer_, n_outer_;
    uint64_t slide_outer_;
    role_t role_;
    std::unordered_map<key_t, KeyD> keys_;
};

template <typename tuple_t>
class WinMap_Emitter : public rt::Node {
public:
    using key_t = key_of_t<tuple_t>;
    WinMap_Emitter(win_type_t type, size_t map_degree) : type_(type), n_(map_degree) {
        for (size_t i = 0; i < n_; i++) add_out_port();
    }
    void svc(void* msg, int) override {
        tuple_t* t = untag<tuple_t>(msg);
        const key_t key = key_of(*t);
        const uint64_t ord = (type_ == CB) ? id_of(*t) : ts_of(*t);
        auto it = keys_.find(key);
        if (it == keys_.end()) {
            it = keys_.emplace(key, KeyD{}).first;
            it->second.next = std::hash<key_t>()(key) % n_;
            it->second.last = *t;
        } else {
            const uint64_t last = (type_ == CB) ? id_of(it->second.last) : ts_of(it->second.last);
            if (ord > last) it->second.last = *t;
        }
        if (is_marker(msg)) {
            tuple_delete(t);
            return;
        }
        emit_to(t, it->second.next);
        it->second.next = (it->second.next + 1) % n_;
    }
    void flush() override {
        for (auto& kv : keys_)
            for (size_t i = 0; i < n_; i++) emit_to(tag_marker(tuple_new<tuple_t>(kv.second.last)), i);
    }

private:
    struct KeyD {
        size_t next = 0;
        tuple_t last{};
    };
    win_type_t type_;
    size_t n_;
    std::unordered_map<key_t, KeyD> keys_;
};

// fused in front of MAP replica `my_id` when the MAP stage is fed by a broadcast
template <typename tuple_t>
class WinMap_Dropper : public rt::Node {
public:
    using key_t = key_of_t<tuple_t>;
    WinMap_Dropper(size_t my_id, size_t map_degree) : my_id_(my_id), n_(map_degree) {}
    void svc(void* msg, int) override {
        tuple_t* t = untag<tuple_t>(msg);
        const key_t key = key_of(*t);
        auto it = next_.find(key);
        if (it == next_.end()) it = next_.emplace(key, std::hash<key_t>()(key) % n_).first;
        if (is_marker(msg)) {
            emit(msg);
            return;
        }
        if (it->second == my_id_)
            emit(t);
        else
            tuple_delete(t);
        it->second = (it->second + 1) % n_;
    }

private:
    size_t my_id_, n_;
    std::unordered_map<key_t, size_t> next_;
};

// emits the results of each key in progressive id order (0, 1, 2, ...)
template <typename result_t>
class Ordered_Collector : public rt::Node {
public:
    using key_t = key_of_t<result_t>;
    void svc(void* msg, int) override {
        result_t* r = static_cast<result_t*>(msg);
        KeyD& kd = keys_[key_of(*r)];
        const uint64_t wid = id_of(*r);
        if (wid < kd.next) {  // duplicate / out of range: forward as is
            emit(r);
            return;
        }
        const uint64_t off = wid - kd.next;
        if (off >= kd.pending.size()) kd.pending.resize(off + 1, nullptr);
        kd.pending[off] = r;
        while (!kd.pending.empty() && kd.pending.front() != nullptr) {
            emit(kd.pending.front());
            kd.pending.pop_front();
            kd.next++;
        }
    }
    void flush() override {
        // nothing should be left; forward stragglers in id order rather than losing them
        for (auto& kv : keys_)
            for (result_t* r : kv.second.pending)
                if (r != nullptr) emit(r);
    }

private:
    struct KeyD {
        uint64_t next = 0;
        std::deque<result_t*> pending;
    };
    std::unordered_map<key_t, KeyD> keys_;
};

}  // namespace wf
