// Sink operator.  Accepted signatures (/root/reference/API:154-159):
//   void(std::optional<tuple_t>&) | void(std::optional<std::reference_wrapper<tuple_t>>)
//   [+ RuntimeContext&]
// The function is called once more with an empty optional at end-of-stream
// (/root/reference/wf/sink.hpp:218-241).
#pragma once
#include <functional>
#include <optional>

#include "simple_operator.hpp"

namespace wf {

template <typename tuple_t, typename F>
class Sink_Node : public Replica_Base {
public:
    Sink_Node(F func, std::string name, size_t par, size_t idx, meta::closing_func_t closing, std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub)), func_(std::move(func)) {
        set_io_bytes(sizeof(tuple_t), 0);
    }
    void svc(void* msg, int) override {
        tuple_t* t = untag<tuple_t>(msg);
        if (is_marker(msg)) {  // markers never reach user code
            tuple_delete(t);
            return;
        }
        if constexpr (meta::sink_sig<F>::by_ref) {
            std::optional<std::reference_wrapper<tuple_t>> o = std::ref(*t);
            meta::invoke_rich(func_, context, o);
        } else {
            std::optional<tuple_t> o = std::move(*t);
            meta::invoke_rich(func_, context, o);
        }
        tuple_delete(t);
    }
    void flush() override {
        if constexpr (meta::sink_sig<F>::by_ref) {
            std::optional<std::reference_wrapper<tuple_t>> o;
            meta::invoke_rich(func_, context, o);
        } else {
            std::optional<tuple_t> o;
            meta::invoke_rich(func_, context, o);
        }
    }

private:
    F func_;
};

template <typename tuple_t>
class Sink : public Simple_Operator<tuple_t, void> {
public:
    template <typename F>
    Sink(F func, size_t parallelism, std::string name, meta::closing_func_t closing = nullptr)
        : Simple_Operator<tuple_t, void>(name, parallelism, FORWARD, nullptr, make_factory(func, name, closing)) {}
    template <typename F>
    Sink(F func, size_t parallelism, std::string name, meta::closing_func_t closing, routing_func_t routing)
        : Simple_Operator<tuple_t, void>(name, parallelism, KEYBY, std::move(routing), make_factory(func, name, closing)) {}

private:
    template <typename F>
    static typename Simple_Operator<tuple_t, void>::replica_factory_t make_factory(F func, std::string name, meta::closing_func_t closing) {
        return [func, name, closing](size_t i, size_t n, std::shared_ptr<StatsHub> hub) -> rt::Node* {
            return new Sink_Node<tuple_t, F>(func, name, n, i, closing, std::move(hub));
        };
    }
};

}  // namespace wf
