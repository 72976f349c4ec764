// Iterable<T>: read-only random-access view over the tuples of one window, handed
// to non-incremental window functions (API parity: /root/reference/wf/
// iterable.hpp:51-241 - begin/end/size/operator[]/at/front/back).
#pragma once
#include <cstddef>
#include <deque>
#include <iterator>
#include <stdexcept>

namespace wf {

template <typename tuple_t>
class Iterable {
public:
    using const_iterator = typename std::deque<tuple_t>::const_iterator;
    Iterable() = default;
    Iterable(const_iterator first, const_iterator last) : first_(first), last_(last), n_((size_t)std::distance(first, last)) {}
    const_iterator begin() const { return first_; }
    const_iterator end() const { return last_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    const tuple_t& operator[](size_t i) const { return *(first_ + (std::ptrdiff_t)i); }
    const tuple_t& at(size_t i) const {
        if (i >= n_) throw std::out_of_range("Invalid index of the Iterable");
        return *(first_ + (std::ptrdiff_t)i);
    }
    const tuple_t& front() const {
        if (n_ == 0) throw std::out_of_range("Invalid index of the Iterable");
        return *first_;
    }
    const tuple_t& back() const {
        if (n_ == 0) throw std::out_of_range("Invalid index of the Iterable");
        return *(last_ - 1);
    }

private:
    const_iterator first_{};
    const_iterator last_{};
    size_t n_ = 0;
};

}  // namespace wf
