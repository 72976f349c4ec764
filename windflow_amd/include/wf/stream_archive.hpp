// StreamArchive: per-key buffer of tuples kept sorted on the windowing field (id
// for count-based windows, timestamp for time-based ones).
// Capability parity: /root/reference/wf/stream_archive.hpp:43-160 (insert by
// binary search, purge, getWinRange, getDistance).  Ranges are addressed by
// *ordinal bounds* instead of by sample tuples: a window [lo, hi) is located
// with two binary searches, so the engine does not have to remember the first
// and last tuple of every open window.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <utility>

namespace wf {

template <typename tuple_t, typename ord_fn_t>
class StreamArchive {
public:
    using container_t = std::deque<tuple_t>;
    using const_iterator = typename container_t::const_iterator;
    explicit StreamArchive(ord_fn_t ord) : ord_(ord) {}

    // insert keeping the order; equal ordinals keep arrival order (stable)
    void insert(const tuple_t& t) {
        const uint64_t o = ord_(t);
        if (buf_.empty() || ord_(buf_.back()) <= o) {
            buf_.push_back(t);
            return;
        }
        auto it = std::upper_bound(buf_.begin(), buf_.end(), o, [this](uint64_t v, const tuple_t& x) { return v < ord_(x); });
        buf_.insert(it, t);
    }
    // remove every tuple with ordinal < lo; returns how many were removed
    size_t purge(uint64_t lo) {
        auto it = lower(lo);
        const size_t n = (size_t)std::distance(buf_.cbegin(), it);
        buf_.erase(buf_.begin(), buf_.begin() + (std::ptrdiff_t)n);
        return n;
    }
    // tuples with ordinal in [lo, hi)
    std::pair<const_iterator, const_iterator> getWinRange(uint64_t lo, uint64_t hi) const { return {lower(lo), lower(hi)}; }
    // tuples with ordinal >= lo
    std::pair<const_iterator, const_iterator> getWinRange(uint64_t lo) const { return {lower(lo), buf_.cend()}; }
    size_t getDistance(uint64_t lo, uint64_t hi) const {
        auto r = getWinRange(lo, hi);
        return (size_t)std::distance(r.first, r.second);
    }
    size_t size() const { return buf_.size(); }
    const_iterator begin() const { return buf_.cbegin(); }
    const_iterator end() const { return buf_.cend(); }

private:
    const_iterator lower(uint64_t v) const {
        return std::lower_bound(buf_.cbegin(), buf_.cend(), v, [this](const tuple_t& x, uint64_t val) { return ord_(x) < val; });
    }
    ord_fn_t ord_;
    container_t buf_;
};

}  // namespace wf
