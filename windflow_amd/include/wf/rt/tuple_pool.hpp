// Tuple memory for the CPU dataflow path.
//
// Tuples travel between operator threads as heap pointers (the reference's FastFlow model): one allocation in the
// producer and one deallocation in a consumer thread per tuple and hop.  With the general-purpose allocator that
// cross-thread pattern costs a lock or an atomic on the producer's arena per operation and dominates the per-tuple
// cost of short operators.  Here every thread bump-allocates from a private 64 KiB block (no atomics on the
// allocation path); a block carries one counter, decremented by whoever destroys one of its tuples, and goes back
// to a process-wide cache of empty blocks when the owner has moved on and the last tuple is gone.
//
//   live = OPEN_BIAS - frees                      while the owner still allocates from the block
//   live = allocated - frees                      after the owner closed it (one fetch_sub of OPEN_BIAS - allocated)
//   whoever brings it to 0 returns the block
//
// A block is recycled only when ALL its tuples are gone, so a long-lived tuple pins 64 KiB.  The runtime holds
// tuples transiently (rings, ordering buffers, end-of-stream markers, all allocated back to back); operators that
// keep data for long (window archives, per-key state) store copies by value, not pooled pointers.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

namespace wf {
namespace mem {

constexpr size_t BLOCK_BYTES = 64 * 1024;
constexpr uint32_t OPEN_BIAS = 1u << 30;
constexpr size_t MAX_POOLED_SIZE = 2048;

struct alignas(64) BlockHeader {
    std::atomic<uint32_t> live;
};

// process-wide cache of empty blocks (touched once per ~64 KiB of tuples); never destroyed: threads may exit late
class BlockCache {
public:
    static BlockCache& instance() {
        static BlockCache* c = new BlockCache();
        return *c;
    }
    void* get() {
        {
            std::lock_guard<std::mutex> l(m_);
            if (!free_.empty()) {
                void* b = free_.back();
                free_.pop_back();
                return b;
            }
        }
        void* b = std::aligned_alloc(BLOCK_BYTES, BLOCK_BYTES);
        if (!b) throw std::bad_alloc();
        return b;
    }
    void put(void* b) {
        {
            std::lock_guard<std::mutex> l(m_);
            if (free_.size() < MAX_CACHED) {
                free_.push_back(b);
                return;
            }
        }
        std::free(b);
    }

private:
    static constexpr size_t MAX_CACHED = 2048;  // 128 MiB
    std::mutex m_;
    std::vector<void*> free_;
};

inline void release_block_ref(BlockHeader* h, uint32_t n) {
    if (h->live.fetch_sub(n, std::memory_order_acq_rel) == n) BlockCache::instance().put(h);
}

class ThreadArena {
public:
    ~ThreadArena() { close(); }
    void* alloc(size_t bytes) {
        bytes = (bytes + 15) & ~size_t(15);
        if (cur_ == nullptr || bump_ + bytes > BLOCK_BYTES) {
            close();
            cur_ = static_cast<char*>(BlockCache::instance().get());
            new (cur_) BlockHeader();
            reinterpret_cast<BlockHeader*>(cur_)->live.store(OPEN_BIAS, std::memory_order_relaxed);
            bump_ = sizeof(BlockHeader);
            allocated_ = 0;
        }
        void* p = cur_ + bump_;
        bump_ += bytes;
        allocated_++;
        return p;
    }
    // an allocation whose constructor threw: the slot is simply not counted
    void unalloc() { allocated_--; }

private:
    void close() {
        if (cur_ == nullptr) return;
        release_block_ref(reinterpret_cast<BlockHeader*>(cur_), OPEN_BIAS - allocated_);
        cur_ = nullptr;
    }
    char* cur_ = nullptr;
    size_t bump_ = 0;
    uint32_t allocated_ = 0;
};

inline ThreadArena& thread_arena() {
    static thread_local ThreadArena a;
    return a;
}

inline void release(void* p) {
    release_block_ref(reinterpret_cast<BlockHeader*>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(BLOCK_BYTES - 1)), 1);
}

template <typename T>
constexpr bool poolable = sizeof(T) <= MAX_POOLED_SIZE && alignof(T) <= 16;

}  // namespace mem

// every tuple that flows on a tuple edge is created with tuple_new and destroyed with tuple_delete
template <typename T, typename... A>
inline T* tuple_new(A&&... a) {
    if constexpr (mem::poolable<T>) {
        mem::ThreadArena& ar = mem::thread_arena();
        void* p = ar.alloc(sizeof(T));
        try {
            return new (p) T(std::forward<A>(a)...);
        } catch (...) {
            ar.unalloc();
            throw;
        }
    } else {
        return new T(std::forward<A>(a)...);
    }
}

template <typename T>
inline void tuple_delete(T* t) {
    if (t == nullptr) return;
    if constexpr (mem::poolable<T>) {
        t->~T();
        mem::release(static_cast<void*>(t));
    } else {
        delete t;
    }
}

// a heap object the USER created with new and hands over (Shipper::push(T*), Filter returning optional<T*>):
// moved into a pooled tuple, the user's object is destroyed
template <typename T>
inline T* tuple_adopt(T* user) {
    if constexpr (mem::poolable<T>) {
        T* t = tuple_new<T>(std::move(*user));
        delete user;
        return t;
    } else {
        return user;
    }
}

}  // namespace wf
