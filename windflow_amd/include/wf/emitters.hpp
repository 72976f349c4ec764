// Routing nodes for the basic operators: Standard (FORWARD / KEYBY), Broadcast,
// Splitting; plus the helpers that turn them into EmitterAssembly objects and
// compose two levels of emitters (the reference's Tree_Emitter).
//
// Behaviour parity: /root/reference/wf/standard_emitter.hpp:85-110,
// wf/broadcast_emitter.hpp:71-84, wf/splitting_emitter.hpp:100-126,
// wf/tree_emitter.hpp:119-144.  Multicast copies the tuple per destination
// instead of sharing a ref-counted wrapper (see basic.hpp "message tagging").
#pragma once
#include <functional>
#include <type_traits>
#include <vector>

#include "basic.hpp"
#include "basic_operator.hpp"
#include "meta.hpp"
#include "rt/graph.hpp"

namespace wf {

using routing_func_t = std::function<size_t(size_t, size_t)>;
inline size_t default_routing(size_t k, size_t n) { return k % n; }

template <typename tuple_t>
class Standard_Emitter : public rt::Node {
public:
    // FORWARD
    explicit Standard_Emitter(size_t n_dest) : keyed_(false), n_dest_(n_dest) { make_ports(); }
    // KEYBY
    Standard_Emitter(routing_func_t routing, size_t n_dest) : keyed_(true), routing_(std::move(routing)), n_dest_(n_dest) { make_ports(); }

    void svc(void* msg, int) override {
        if (keyed_) {
            const tuple_t* t = untag<tuple_t>(msg);
            emit_to(msg, routing_(hash_of(*t), n_dest_));
        } else {
            emit(msg);
        }
    }

private:
    void make_ports() {
        for (size_t i = 0; i < n_dest_; i++) add_out_port();
    }
    bool keyed_;
    routing_func_t routing_;
    size_t n_dest_;
};

template <typename tuple_t>
class Broadcast_Emitter : public rt::Node {
public:
    explicit Broadcast_Emitter(size_t n_dest) : n_dest_(n_dest) {
        for (size_t i = 0; i < n_dest_; i++) add_out_port();
    }
    void svc(void* msg, int) override {
        tuple_t* t = untag<tuple_t>(msg);
        const bool mk = is_marker(msg);
        for (size_t i = 0; i + 1 < n_dest_; i++) {
            tuple_t* c = tuple_new<tuple_t>(*t);
            emit_to(mk ? tag_marker(c) : (void*)c, i);
        }
        emit_to(msg, n_dest_ - 1);
    }

private:
    size_t n_dest_;
};

// calls the user split function; an empty result drops the tuple, several
// indices multicast a copy to each branch
template <typename tuple_t, typename F>
class Splitting_Emitter : public rt::Node {
public:
    Splitting_Emitter(F func, size_t n_branches) : func_(std::move(func)), n_(n_branches) {
        for (size_t i = 0; i < n_; i++) add_out_port();
    }
    void svc(void* msg, int) override {
        tuple_t* t = static_cast<tuple_t*>(msg);
        if constexpr (meta::split_sig<F>::multi) {
            auto dests = func_(*t);
            if (dests.empty()) {
                tuple_delete(t);
                return;
            }
            for (size_t i = 0; i + 1 < dests.size(); i++) {
                check((size_t)dests[i]);
                emit_to(tuple_new<tuple_t>(*t), (size_t)dests[i]);
            }
            check((size_t)dests.back());
            emit_to(t, (size_t)dests.back());
        } else {
            const size_t d = (size_t)func_(*t);
            check(d);
            emit_to(t, d);
        }
    }

private:
    void check(size_t d) const {
        if (d >= n_) fatal_error("splitting function returned a branch index out of range");
    }
    F func_;
    size_t n_;
};

// ---- assembly helpers -------------------------------------------------------
// single-node emitter with one port per destination
template <typename E>
inline EmitterAssembly single_emitter(rt::Graph& g, E* e, size_t n_dest, size_t dest_offset = 0) {
    g.adopt(e);
    EmitterAssembly a;
    a.root = e;
    for (size_t i = 0; i < n_dest; i++) a.leaves.push_back({dest_offset + i, e, i});
    return a;
}

// two-level composition: `root` has one port per child; child c routes inside
// its own destination range.  Leaves are flattened in child order.
inline EmitterAssembly tree_emitter(rt::Graph& g, rt::Node* root, std::vector<EmitterAssembly>& children) {
    EmitterAssembly a;
    a.root = root;
    for (size_t c = 0; c < children.size(); c++) {
        g.connect_fused(root, c, children[c].root);
        for (auto& l : children[c].leaves) a.leaves.push_back(l);
    }
    return a;
}

}  // namespace wf
