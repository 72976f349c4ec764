// Bounded lock-free single-producer/single-consumer ring of opaque pointers.
//
// Replaces the FastFlow SPSC channels the reference is built on (SURVEY.md
// §2.7 item 3/5).  Producer and consumer indices live on separate cache lines
// and each side keeps a cached copy of the other side's index, so the common
// case touches only thread-local lines.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <new>

namespace wf {
namespace rt {

class SPSCQueue {
public:
    explicit SPSCQueue(size_t capacity) {
        size_t c = 2;
        while (c < capacity) c <<= 1;
        mask_ = c - 1;
        buf_.reset(new void*[c]);
    }
    SPSCQueue(const SPSCQueue&) = delete;
    SPSCQueue& operator=(const SPSCQueue&) = delete;

    size_t capacity() const { return mask_ + 1; }

    // producer side
    bool try_push(void* p) {
        const size_t t = tail_.load(std::memory_order_relaxed);
        if (t - head_cache_ > mask_) {
            head_cache_ = head_.load(std::memory_order_acquire);
            if (t - head_cache_ > mask_) return false;
        }
        buf_[t & mask_] = p;
        tail_.store(t + 1, std::memory_order_release);
        return true;
    }

    // consumer side
    bool try_pop(void*& p) {
        const size_t h = head_.load(std::memory_order_relaxed);
        if (h == tail_cache_) {
            tail_cache_ = tail_.load(std::memory_order_acquire);
            if (h == tail_cache_) return false;
        }
        p = buf_[h & mask_];
        head_.store(h + 1, std::memory_order_release);
        return true;
    }

    // consumer side: pop up to n items at once (one release store for the batch)
    size_t pop_bulk(void** out, size_t n) {
        const size_t h = head_.load(std::memory_order_relaxed);
        size_t avail = tail_cache_ - h;
        if (avail == 0) {
            tail_cache_ = tail_.load(std::memory_order_acquire);
            avail = tail_cache_ - h;
            if (avail == 0) return 0;
        }
        if (avail < n) n = avail;
        for (size_t i = 0; i < n; i++) out[i] = buf_[(h + i) & mask_];
        head_.store(h + n, std::memory_order_release);
        return n;
    }

    // approximate occupancy (either side)
    size_t size_approx() const {
        return tail_.load(std::memory_order_acquire) - head_.load(std::memory_order_acquire);
    }

private:
    alignas(64) std::atomic<size_t> head_{0};
    size_t tail_cache_ = 0;  // consumer-private
    alignas(64) std::atomic<size_t> tail_{0};
    size_t head_cache_ = 0;  // producer-private
    alignas(64) size_t mask_ = 0;
    std::unique_ptr<void*[]> buf_;
};

}  // namespace rt
}  // namespace wf
