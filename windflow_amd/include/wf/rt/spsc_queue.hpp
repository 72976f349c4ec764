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

#include <linux/futex.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

namespace wf {
namespace rt {

// Where a runtime thread sleeps in blocking mode (WF_BLOCKING=1): one futex word per thread.  The sleeper
// announces itself (word = 1), re-checks its rings and only then waits; whoever makes progress possible for it
// (a push into one of its input rings, a pop from a ring it found full) clears the word and wakes it.  Every wait
// carries a timeout, so a lost wake-up costs latency, never liveness.
class Parker {
public:
    // call before the final re-check of the condition
    void announce() { word_.store(1, std::memory_order_seq_cst); }
    void cancel() { word_.store(0, std::memory_order_relaxed); }
    // sleeps until unpark() or the timeout; a no-op if unpark() already came
    void park(unsigned timeout_us) {
        struct timespec ts;
        ts.tv_sec = timeout_us / 1000000u;
        ts.tv_nsec = (long)(timeout_us % 1000000u) * 1000L;
        (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&word_), FUTEX_WAIT_PRIVATE, 1u, &ts, nullptr, 0);
        word_.store(0, std::memory_order_relaxed);
        parks_++;
    }
    // called by the other side after it changed a ring
    void unpark() {
        if (word_.load(std::memory_order_seq_cst) == 1 && word_.exchange(0, std::memory_order_seq_cst) == 1)
            (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&word_), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
    }
    uint64_t parks() const { return parks_; }

private:
    std::atomic<uint32_t> word_{0};
    uint64_t parks_ = 0;  // owner thread only
};
static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "futex word must be a plain 32-bit word");

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

    // blocking mode: the threads to wake when the ring stops being empty / full (set by the sleeper itself)
    std::atomic<Parker*> consumer_parker{nullptr};
    std::atomic<Parker*> producer_parker{nullptr};
    void wake_consumer() {
        std::atomic_thread_fence(std::memory_order_seq_cst);
        if (Parker* p = consumer_parker.load(std::memory_order_relaxed)) p->unpark();
    }
    void wake_producer() {
        std::atomic_thread_fence(std::memory_order_seq_cst);
        if (Parker* p = producer_parker.load(std::memory_order_relaxed)) p->unpark();
    }
    bool empty_approx() const { return tail_.load(std::memory_order_acquire) == head_.load(std::memory_order_relaxed); }

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
