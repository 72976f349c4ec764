// Accumulator operator: keyed fold.  Accepted signatures
// (/root/reference/API:39-42): void(const tuple_t&, result_t&) [+ RuntimeContext&].
// Always KEYBY; one running result per key, a copy is emitted per input
// (/root/reference/wf/accumulator.hpp:160-190).
#pragma once
#include <unordered_map>

#include "simple_operator.hpp"

namespace wf {

template <typename tuple_t, typename result_t, typename F>
class Accumulator_Node : public Replica_Base {
public:
    Accumulator_Node(F func, result_t init, std::string name, size_t par, size_t idx, meta::closing_func_t closing,
                     std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub)), func_(std::move(func)), init_(std::move(init)) {
        set_io_bytes(sizeof(tuple_t), sizeof(result_t));
    }
    void svc(void* msg, int) override {
        tuple_t* t = static_cast<tuple_t*>(msg);
        const auto key = key_of(*t);
        auto it = state_.find(key);
        if (it == state_.end()) it = state_.emplace(key, init_).first;
        result_t& acc = it->second;
        // control fields of the running result are the user function's business
        meta::invoke_rich(func_, context, static_cast<const tuple_t&>(*t), acc);
        tuple_delete(t);
        emit(tuple_new<result_t>(acc));
    }

private:
    F func_;
    result_t init_;
    std::unordered_map<key_of_t<tuple_t>, result_t> state_;
};

template <typename tuple_t, typename result_t>
class Accumulator : public Simple_Operator<tuple_t, result_t> {
public:
    template <typename F>
    Accumulator(F func, result_t init_value, size_t parallelism, std::string name, meta::closing_func_t closing = nullptr,
                routing_func_t routing = default_routing)
        : Simple_Operator<tuple_t, result_t>(name, parallelism, KEYBY, std::move(routing),
                                             [func, init_value, name, closing](size_t i, size_t n, std::shared_ptr<StatsHub> hub) -> rt::Node* {
                                                 return new Accumulator_Node<tuple_t, result_t, F>(func, init_value, name, n, i, closing,
                                                                                                    std::move(hub));
                                             }) {}
};

}  // namespace wf
