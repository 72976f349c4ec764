// Graph: owns nodes and rings, wires ports, and runs one thread per "head" node
// (a node fed by rings, or a source).  Every node reachable from a head through
// fused ports runs inside the head's thread.  Replaces ff_pipeline / ff_farm /
// ff_a2a composition + run/wait of the reference's substrate.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <memory>
#include <set>
#include <stdexcept>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <pthread.h>

#include "node.hpp"

namespace wf {
namespace rt {

class Graph {
public:
    explicit Graph(RuntimeConfig cfg = RuntimeConfig::from_env()) : cfg_(cfg) {}
    ~Graph() {
        if (running_) wait();
    }
    Graph(const Graph&) = delete;
    Graph& operator=(const Graph&) = delete;

    RuntimeConfig& config() { return cfg_; }

    // take ownership of a node
    template <typename N>
    N* adopt(N* n) {
        nodes_.emplace_back(n);
        static_cast<Node*>(n)->cfg_ = &cfg_;
        return n;
    }

    // port `port` of `from` feeds `to` through a new SPSC ring; returns the channel id at `to`
    int connect_queue(Node* from, size_t port, Node* to) {
        check_port(from, port);
        queues_.emplace_back(new SPSCQueue(cfg_.queue_capacity));
        SPSCQueue* q = queues_.back().get();
        from->out_[port].q = q;
        to->in_.push_back(q);
        if (to->in_.size() != to->n_in_ + 1)
            throw std::logic_error("wf::rt::Graph: node mixes fused and queue inputs");
        return (int)to->n_in_++;
    }
    // port `port` of `from` calls `to` directly (same thread); returns the channel id at `to`
    int connect_fused(Node* from, size_t port, Node* to) {
        check_port(from, port);
        if (!to->in_.empty()) throw std::logic_error("wf::rt::Graph: node mixes fused and queue inputs");
        from->out_[port].fused = to;
        from->out_[port].fused_ch = (int)to->n_in_;
        return (int)to->n_in_++;
    }

    // heads = sources + nodes with ring inputs
    std::vector<Node*> heads() const {
        std::vector<Node*> h;
        for (auto& n : nodes_)
            if (n->is_source() || !n->in_.empty()) h.push_back(n.get());
        return h;
    }
    size_t num_threads() const { return heads().size(); }
    size_t num_nodes() const { return nodes_.size(); }
    size_t num_queues() const { return queues_.size(); }

    void run() {
        if (running_ || done_) throw std::logic_error("wf::rt::Graph: already started");
        auto hs = heads();
        // every node must be reachable from exactly one head
        std::unordered_set<Node*> seen;
        std::vector<std::vector<Node*>> groups;
        for (Node* h : hs) {
            std::vector<Node*> g;
            collect(h, g, seen);
            groups.push_back(std::move(g));
        }
        running_ = true;
        parkers_.clear();
        for (size_t i = 0; i < hs.size(); i++) parkers_.emplace_back(new Parker());
        pin_list_ = cfg_.pin_threads ? cpu_list() : std::vector<int>();
        threads_.reserve(hs.size());
        for (size_t i = 0; i < hs.size(); i++) {
            threads_.emplace_back([this, i, h = hs[i], g = groups[i]]() { thread_main(i, h, g); });
        }
    }

    void wait() {
        for (auto& t : threads_)
            if (t.joinable()) t.join();
        threads_.clear();
        running_ = false;
        done_ = true;
    }

    bool finished() const { return done_; }

    const std::vector<std::unique_ptr<Node>>& nodes() const { return nodes_; }

private:
    static void check_port(Node* from, size_t port) {
        if (port >= from->out_.size()) throw std::out_of_range("wf::rt::Graph: bad port");
        if (from->out_[port].q != nullptr || from->out_[port].fused != nullptr)
            throw std::logic_error("wf::rt::Graph: port already connected");
    }

    static void collect(Node* n, std::vector<Node*>& g, std::unordered_set<Node*>& seen) {
        if (!seen.insert(n).second) return;
        g.push_back(n);
        for (auto& o : n->out_)
            if (o.fused != nullptr) collect(o.fused, g, seen);
    }

    // CPUs the threads are pinned to, round-robin: WF_PIN_CPUS ("0-15,32-47") or, by default, the CPUs the process
    // is allowed to run on - a launcher that binds each rank to the NUMA node of its GPU (bench.py does) thereby
    // keeps every runtime thread, and the pinned slabs it first touches, on that node
    std::vector<int> cpu_list() const {
        std::vector<int> cpus;
        const std::string& spec = cfg_.pin_cpus;
        size_t pos = 0;
        while (pos < spec.size()) {
            size_t end = spec.find(',', pos);
            if (end == std::string::npos) end = spec.size();
            const std::string part = spec.substr(pos, end - pos);
            pos = end + 1;
            if (part.empty()) continue;
            const size_t dash = part.find('-');
            const int lo = std::atoi(part.substr(0, dash).c_str());
            const int hi = dash == std::string::npos ? lo : std::atoi(part.substr(dash + 1).c_str());
            for (int c = lo; c <= hi && c < CPU_SETSIZE; c++) cpus.push_back(c);
        }
        if (cpus.empty()) {
            cpu_set_t set;
            CPU_ZERO(&set);
            if (sched_getaffinity(0, sizeof(set), &set) == 0)
                for (int c = 0; c < CPU_SETSIZE; c++)
                    if (CPU_ISSET(c, &set)) cpus.push_back(c);
        }
        if (cpus.empty()) cpus.push_back(0);
        return cpus;
    }

    static uint64_t now_ns() {
        return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

    void thread_main(size_t idx, Node* head, const std::vector<Node*>& group) {
        if (!pin_list_.empty()) {
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(pin_list_[idx % pin_list_.size()], &set);
            pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
        }
        Parker* parker = parkers_[idx].get();
        this_thread_parker() = parker;
        for (Node* n : group) n->svc_init();
        if (head->is_source()) {
            const uint64_t t0 = now_ns();
            while (head->generate()) {
            }
            head->counters_.busy_ns += now_ns() - t0;  // a source is always busy (time blocked on a full ring included)
            head->terminate();
        } else {
            constexpr size_t BULK = 64;
            void* buf[BULK];
            std::vector<int> active(head->in_.size());
            for (size_t i = 0; i < active.size(); i++) active[i] = (int)i;
            Backoff backoff(&cfg_);
            std::vector<Node*> idlers;
            for (Node* n : group)
                if (n->wants_idle()) idlers.push_back(n);
            const bool blocking = cfg_.blocking;
            if (blocking)
                for (SPSCQueue* q : head->in_) q->consumer_parker.store(parker, std::memory_order_relaxed);
            unsigned idle_rounds = 0;
            while (!active.empty()) {
                bool progress = false;
                for (size_t a = 0; a < active.size();) {
                    const int ch = active[a];
                    SPSCQueue* q = head->in_[ch];
                    const size_t n = q->pop_bulk(buf, BULK);
                    bool ended = false;
                    if (n > 0) {
                        if (blocking) q->wake_producer();
                        // service time is measured per bulk (two clock reads per up to 64 messages), successors fused in
                        // this thread included
                        const uint64_t t0 = now_ns();
                        size_t m = 0;
                        while (m < n && buf[m] != eos_token()) m++;
                        ended = m < n;  // producer never pushes after EOS
                        if (m > 0) {
                            head->counters_.inputs += m;
                            head->svc_bulk(buf, m, ch);
                        }
                        head->counters_.busy_ns += now_ns() - t0;
                    }
                    if (n > 0) progress = true;
                    if (ended) {
                        active[a] = active.back();
                        active.pop_back();
                        head->input_ended(ch);
                    } else {
                        a++;
                    }
                }
                if (progress) {
                    backoff.reset();
                    idle_rounds = 0;
                } else {
                    for (Node* n : idlers) n->on_idle();
                    if (!blocking) {
                        backoff.pause();
                    } else if (++idle_rounds < 4) {
                        sched_yield();
                    } else {
                        // sleep on the futex until a producer pushes into one of the input rings (nodes with work in
                        // flight on the device are polled every idle_poll_us instead)
                        parker->announce();
                        bool any = false;
                        for (int ch : active) any = any || !head->in_[ch]->empty_approx();
                        if (any) {
                            parker->cancel();
                        } else {
                            parker->park(idlers.empty() ? cfg_.park_timeout_us : std::min(cfg_.park_timeout_us, 100u));
                            head->counters_.parks++;
                        }
                    }
                }
            }
            if (head->in_.empty()) head->terminate();
        }
        for (Node* n : group) n->svc_end();
    }

    RuntimeConfig cfg_;
    std::vector<std::unique_ptr<Node>> nodes_;
    std::vector<std::unique_ptr<SPSCQueue>> queues_;
    std::vector<std::thread> threads_;
    std::vector<std::unique_ptr<Parker>> parkers_;
    std::vector<int> pin_list_;
    bool running_ = false;
    bool done_ = false;
};

}  // namespace rt
}  // namespace wf
