// Node: the unit of computation of the native runtime.
//
// Behavioural contract taken from what the reference needs from FastFlow
// (SURVEY.md §2.7): a node has svc_init / svc / eosnotify / svc_end, can send to
// "the next stage" or to a specific destination, and several nodes can be fused
// into one thread (the reference's ff_comb / combine_with_laststage).  The design
// is ours: a node owns a vector of output *ports*; a port is either a lock-free
// SPSC ring to a node running in another thread, or a direct call into a node
// fused in the same thread.  Fusion is therefore a wiring decision, not a
// different node type, and emitters / ordering nodes / replicas / collectors are
// all plain nodes.
#pragma once
#include <atomic>
#include <cassert>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <sched.h>

#include "config.hpp"
#include "spsc_queue.hpp"
#include "tuple_pool.hpp"

namespace wf {
namespace rt {

// sentinel pushed on every queue by a terminating producer
inline void* eos_token() {
    static char token;
    return &token;
}

inline void cpu_relax() {
#if defined(__x86_64__)
    _mm_pause();
#else
    std::this_thread::yield();
#endif
}

// the Parker of the calling runtime thread (set by Graph::thread_main; nullptr outside the runtime)
inline Parker*& this_thread_parker() {
    static thread_local Parker* p = nullptr;
    return p;
}

// spin -> yield -> sleep back-off helper
class Backoff {
public:
    explicit Backoff(const RuntimeConfig* cfg) : cfg_(cfg) {}
    void reset() { n_ = 0; }
    void pause() {
        const unsigned spin = cfg_->blocking ? 0 : cfg_->spin_iters;
        if (n_ < spin) {
            cpu_relax();
        } else if (n_ < spin + cfg_->yield_iters) {
            sched_yield();
        } else {
            std::this_thread::sleep_for(std::chrono::microseconds(cfg_->sleep_us));
        }
        n_++;
    }

private:
    const RuntimeConfig* cfg_;
    unsigned n_ = 0;
};

// cheap always-on counters (the reference only has them under -DTRACE_WINDFLOW)
struct NodeCounters {
    uint64_t inputs = 0;
    uint64_t outputs = 0;
    uint64_t busy_ns = 0;  // head nodes: time spent inside svc()/generate(), fused successors included (thread loop)
    uint64_t parks = 0;    // head nodes, blocking mode: times the thread went to sleep on its futex
};

class Graph;

class Node {
public:
    Node() = default;
    virtual ~Node() = default;
    Node(const Node&) = delete;
    Node& operator=(const Node&) = delete;

    // ---- user-overridable behaviour -------------------------------------
    virtual void svc_init() {}
    // process one message received on input channel `channel`
    virtual void svc(void* /*msg*/, int /*channel*/) {}
    // the messages the thread found queued on one input channel, oldest first (head nodes only): a node that can do several
    // at once for less than the sum (a GPU replica coalescing small host batches into one device batch) overrides this
    virtual void svc_bulk(void** msgs, size_t n, int channel) {
        for (size_t i = 0; i < n; i++) svc(msgs[i], channel);
    }
    // source nodes: produce some output, return false when the stream is over
    virtual bool generate() { return false; }
    virtual bool is_source() const { return false; }
    // one input channel reached end-of-stream
    virtual void eosnotify(int /*channel*/) {}
    // every input channel reached end-of-stream: emit whatever is still buffered
    virtual void flush() {}
    virtual void svc_end() {}
    // nodes with asynchronous work in flight (device copies...) can ask to be polled
    // whenever their thread has no input to process
    // true when the node is fed by ONE producer of its own thread (fused onto it): the two then run strictly one after the
    // other, which GPU nodes use to share the producer's stream
    bool fed_in_thread_by_one() const { return in_.empty() && n_in_ == 1; }
    // false for a node fused onto its producers: it has no thread of its own, so nobody polls it (on_idle) between inputs
    bool has_input_rings() const { return !in_.empty(); }
    virtual bool wants_idle() const { return false; }
    virtual void on_idle() {}
    // A node that only needs the VALUE of its input (it copies the tuple somewhere: a batch under construction) can say so;
    // a producer fused onto it then hands it a reference to a tuple of its own (svc_value) instead of a pooled heap
    // tuple: no allocation, no hand-over of ownership.  See emit_value() and wf::Batcher.
    virtual bool accepts_values() const { return false; }
    virtual void svc_value(const void* /*tuple*/, int /*channel*/) {}

    // ---- output ---------------------------------------------------------
    size_t add_out_port() {
        out_.emplace_back();
        return out_.size() - 1;
    }
    size_t num_out() const { return out_.size(); }
    size_t num_in() const { return n_in_; }

    // send to port `p` (blocks with back-off while the ring is full)
    inline void emit_to(void* m, size_t p) {
        assert(p < out_.size());
        Port& o = out_[p];
        counters_.outputs++;
        if (o.fused != nullptr) {
            o.fused->counters_.inputs++;
            o.fused->svc(m, o.fused_ch);
        } else if (o.q != nullptr) {
            if (!o.q->try_push(m)) push_slow(o.q, m);
            if (cfg_->blocking) o.q->wake_consumer();
        }
        // unconnected port: message is dropped by design (caller keeps ownership rules)
    }
    // send to "the next stage": the only port, or round-robin over the ports
    inline void emit(void* m) {
        if (out_.size() == 1) {
            emit_to(m, 0);
        } else if (!out_.empty()) {
            emit_to(m, rr_);
            rr_ = (rr_ + 1 == out_.size()) ? 0 : rr_ + 1;
        }
    }
    bool has_output() const { return !out_.empty(); }
    bool feeds_value_consumer() const { return out_.size() == 1 && out_[0].fused != nullptr && out_[0].fused->accepts_values(); }
    // send a COPY of `t` to the next stage: by reference when the only successor is fused and takes values, as a pooled
    // heap tuple otherwise
    template <typename T>
    inline void emit_value(const T& t) {
        if (out_.size() == 1 && out_[0].fused != nullptr && out_[0].fused->accepts_values()) {
            counters_.outputs++;
            out_[0].fused->counters_.inputs++;
            out_[0].fused->svc_value(&t, out_[0].fused_ch);
        } else {
            emit(::wf::tuple_new<T>(t));
        }
    }

    const NodeCounters& counters() const { return counters_; }
    const std::string& node_name() const { return node_name_; }
    void set_node_name(const std::string& n) { node_name_ = n; }
    const RuntimeConfig& runtime_config() const { return *cfg_; }

    // called by the runtime (or by a fused predecessor) when input `ch` is over
    void input_ended(int ch) {
        eosnotify(ch);
        eos_seen_++;
        if (eos_seen_ >= n_in_) terminate();
    }
    // flush + propagate end-of-stream on every port
    void terminate() {
        if (terminated_) return;
        terminated_ = true;
        flush();
        for (Port& o : out_) {
            if (o.fused != nullptr)
                o.fused->input_ended(o.fused_ch);
            else if (o.q != nullptr) {
                push_slow(o.q, eos_token());
                if (cfg_->blocking) o.q->wake_consumer();
            }
        }
    }

private:
    friend class Graph;
    struct Port {
        Node* fused = nullptr;
        int fused_ch = 0;
        SPSCQueue* q = nullptr;
    };

    void push_slow(SPSCQueue* q, void* m) {
        Parker* me = cfg_->blocking ? this_thread_parker() : nullptr;
        if (me != nullptr) {
            // blocking mode: sleep until the consumer pops (it wakes the producer it finds registered on the ring)
            q->producer_parker.store(me, std::memory_order_relaxed);
            while (true) {
                me->announce();
                if (q->try_push(m)) {
                    me->cancel();
                    return;
                }
                me->park(cfg_->park_timeout_us);
                counters_.parks++;
            }
        }
        Backoff b(cfg_);
        while (!q->try_push(m)) b.pause();
    }

    std::vector<Port> out_;
    std::vector<SPSCQueue*> in_;  // queue inputs (head nodes only); channel id == index
    size_t n_in_ = 0;             // number of input channels (queue or fused)
    size_t eos_seen_ = 0;
    size_t rr_ = 0;
    bool terminated_ = false;
    NodeCounters counters_;
    std::string node_name_;
    static const RuntimeConfig* default_cfg() {
        static RuntimeConfig c = RuntimeConfig::from_env();
        return &c;
    }
    const RuntimeConfig* cfg_ = default_cfg();
};

}  // namespace rt
}  // namespace wf
