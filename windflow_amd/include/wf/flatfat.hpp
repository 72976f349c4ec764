// FlatFAT: Flat Fixed-size Aggregator Tree (Tangwongsan et al., VLDB 2015) for
// incremental sliding-window aggregation with an associative, possibly
// non-commutative combine function.
//
// Capability parity with /root/reference/wf/flatfat.hpp:51-396: circular leaves
// in a flat 2n array (n rounded to a power of two), insert(one | vector),
// remove(one | count), getResult() that respects arrival order when the live
// range wraps around.  Implementation differences: nodes carry a validity flag,
// so no identity element is required from the user, and getResult() is an
// ordered range query that works for any live range, wrapped or not.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "basic.hpp"

namespace wf {

template <typename result_t, typename comb_t>
class FlatFAT {
public:
    // comb(a, b, out): out = a (+) b, a older than b
    FlatFAT(comb_t comb, size_t capacity) : comb_(std::move(comb)) {
        n_ = 1;
        while (n_ < capacity) n_ <<= 1;
        tree_.resize(2 * n_);
        valid_.assign(2 * n_, 0);
    }

    size_t capacity() const { return n_; }
    size_t size() const { return count_; }
    bool is_Empty() const { return count_ == 0; }

    void insert(const result_t& v) {
        if (count_ == n_) fatal_error("FlatFAT is full");
        const size_t pos = (front_ + count_) & (n_ - 1);
        tree_[n_ + pos] = v;
        valid_[n_ + pos] = 1;
        count_++;
        update_path(n_ + pos);
    }
    void insert(const std::vector<result_t>& vs) {
        if (vs.size() <= 4) {
            for (const auto& v : vs) insert(v);
            return;
        }
        if (count_ + vs.size() > n_) fatal_error("FlatFAT is full");
        // write the leaves, then rebuild the touched ancestors level by level
        std::vector<size_t> dirty;
        dirty.reserve(vs.size());
        for (const auto& v : vs) {
            const size_t pos = (front_ + count_) & (n_ - 1);
            tree_[n_ + pos] = v;
            valid_[n_ + pos] = 1;
            count_++;
            dirty.push_back((n_ + pos) >> 1);
        }
        rebuild(dirty);
    }
    void remove() { remove(1); }
    // drop the `cnt` oldest leaves
    void remove(size_t cnt) {
        if (cnt > count_) cnt = count_;
        if (cnt <= 4) {
            for (size_t i = 0; i < cnt; i++) {
                valid_[n_ + front_] = 0;
                update_path(n_ + front_);
                front_ = (front_ + 1) & (n_ - 1);
                count_--;
            }
            return;
        }
        std::vector<size_t> dirty;
        dirty.reserve(cnt);
        for (size_t i = 0; i < cnt; i++) {
            valid_[n_ + front_] = 0;
            dirty.push_back((n_ + front_) >> 1);
            front_ = (front_ + 1) & (n_ - 1);
            count_--;
        }
        rebuild(dirty);
    }

    // aggregate of every live leaf in arrival order; returns false if empty
    bool getResult(result_t& out) const {
        if (count_ == 0) return false;
        const size_t end = front_ + count_;
        if (end <= n_) {
            if (front_ == 0 && end == n_ && valid_[1]) {
                out = tree_[1];
                return true;
            }
            return query(front_, end, out);
        }
        // wrapped: older part [front, n) then newer part [0, end - n)
        result_t a, b;
        const bool ha = query(front_, n_, a);
        const bool hb = query(0, end - n_, b);
        if (ha && hb)
            combine(a, b, out);
        else
            out = ha ? a : b;
        return ha || hb;
    }

private:
    void combine(const result_t& a, const result_t& b, result_t& out) const {
        // control fields of a combined value: key of the operands, id 0, latest timestamp
        out.setControlFields(key_of(a), 0, ts_of(a) > ts_of(b) ? ts_of(a) : ts_of(b));
        comb_(a, b, out);
    }
    void recompute(size_t node) {
        const size_t l = node << 1, r = l | 1;
        if (valid_[l] && valid_[r]) {
            result_t tmp;
            combine(tree_[l], tree_[r], tmp);
            tree_[node] = tmp;
            valid_[node] = 1;
        } else if (valid_[l]) {
            tree_[node] = tree_[l];
            valid_[node] = 1;
        } else if (valid_[r]) {
            tree_[node] = tree_[r];
            valid_[node] = 1;
        } else {
            valid_[node] = 0;
        }
    }
    void update_path(size_t leaf) {
        for (size_t node = leaf >> 1; node >= 1; node >>= 1) recompute(node);
    }
    void rebuild(std::vector<size_t>& dirty) {
        while (!dirty.empty()) {
            std::vector<size_t> next;
            size_t last = 0;
            for (size_t node : dirty) {
                if (node == last || node == 0) continue;  // consecutive duplicates (leaves are written in order)
                last = node;
                recompute(node);
                if (node > 1) next.push_back(node >> 1);
            }
            dirty.swap(next);
        }
    }
    // ordered aggregate of leaves [l, r), 0 <= l < r <= n
    bool query(size_t l, size_t r, result_t& out) const {
        result_t left, right;
        bool hl = false, hr = false;
        for (l += n_, r += n_; l < r; l >>= 1, r >>= 1) {
            if (l & 1) {
                if (valid_[l]) {
                    if (hl) {
                        result_t tmp;
                        combine(left, tree_[l], tmp);
                        left = tmp;
                    } else {
                        left = tree_[l];
                        hl = true;
                    }
                }
                l++;
            }
            if (r & 1) {
                --r;
                if (valid_[r]) {
                    if (hr) {
                        result_t tmp;
                        combine(tree_[r], right, tmp);
                        right = tmp;
                    } else {
                        right = tree_[r];
                        hr = true;
                    }
                }
            }
        }
        if (hl && hr)
            combine(left, right, out);
        else if (hl)
            out = left;
        else if (hr)
            out = right;
        return hl || hr;
    }

    comb_t comb_;
    size_t n_ = 1;
    size_t front_ = 0;
    size_t count_ = 0;
    std::vector<result_t> tree_;
    std::vector<uint8_t> valid_;
};

}  // namespace wf
