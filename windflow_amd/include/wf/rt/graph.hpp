// Graph: owns nodes and rings, wires ports, and runs one thread per "head" node
// (a node fed by rings, or a source).  Every node reachable from a head through
// fused ports runs inside the head's thread.  Replaces ff_pipeline / ff_farm /
// ff_a2a composition + run/wait of the reference's substrate.
#pragma once
#include <algorithm>
#include <cstdio>
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

    void thread_main(size_t idx, Node* head, const std::vector<Node*>& group) {
        if (cfg_.pin_threads) {
            unsigned nc = std::max(1u, std::thread::hardware_concurrency());
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(idx % nc, &set);
            pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
        }
        for (Node* n : group) n->svc_init();
        if (head->is_source()) {
            while (head->generate()) {
            }
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
            while (!active.empty()) {
                bool progress = false;
                for (size_t a = 0; a < active.size();) {
                    const int ch = active[a];
                    SPSCQueue* q = head->in_[ch];
                    const size_t n = q->pop_bulk(buf, BULK);
                    bool ended = false;
                    for (size_t i = 0; i < n; i++) {
                        if (buf[i] == eos_token()) {
                            ended = true;  // producer never pushes after EOS
                            break;
                        }
                        head->counters_.inputs++;
                        head->svc(buf[i], ch);
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
                } else {
                    for (Node* n : idlers) n->on_idle();
                    backoff.pause();
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
    bool running_ = false;
    bool done_ = false;
};

}  // namespace rt
}  // namespace wf
