// Map operator.  Accepted signatures (/root/reference/API:27-32):
//   void(tuple_t&) in place | void(const tuple_t&, result_t&)  [+ RuntimeContext&]
// The in-place form reuses the incoming heap tuple
// (/root/reference/wf/map.hpp:180-198).
#pragma once
#include "simple_operator.hpp"

namespace wf {

template <typename tuple_t, typename result_t, typename F>
class Map_Node : public Replica_Base {
public:
    Map_Node(F func, std::string name, size_t par, size_t idx, meta::closing_func_t closing, std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub)), func_(std::move(func)) {
        set_io_bytes(sizeof(tuple_t), sizeof(result_t));
    }
    void svc(void* msg, int) override {
        tuple_t* t = static_cast<tuple_t*>(msg);
        if constexpr (meta::map_sig<F>::in_place) {
            meta::invoke_rich(func_, context, *t);
            emit(t);
        } else {
            if (this->feeds_value_consumer()) {  // the consumer copies the result (Batcher): no heap tuple for it
                result_t r{};
                meta::invoke_rich(func_, context, static_cast<const tuple_t&>(*t), r);
                tuple_delete(t);
                this->emit_value(r);
                return;
            }
            result_t* r = tuple_new<result_t>();
            meta::invoke_rich(func_, context, static_cast<const tuple_t&>(*t), *r);
            tuple_delete(t);
            emit(r);
        }
    }

private:
    F func_;
};

template <typename tuple_t, typename result_t>
class Map : public Simple_Operator<tuple_t, result_t> {
public:
    // ctor I: FORWARD
    template <typename F>
    Map(F func, size_t parallelism, std::string name, meta::closing_func_t closing = nullptr)
        : Simple_Operator<tuple_t, result_t>(name, parallelism, FORWARD, nullptr, make_factory(func, name, closing)) {}
    // ctor II: KEYBY
    template <typename F>
    Map(F func, size_t parallelism, std::string name, meta::closing_func_t closing, routing_func_t routing)
        : Simple_Operator<tuple_t, result_t>(name, parallelism, KEYBY, std::move(routing), make_factory(func, name, closing)) {}

private:
    template <typename F>
    static typename Simple_Operator<tuple_t, result_t>::replica_factory_t make_factory(F func, std::string name,
                                                                                        meta::closing_func_t closing) {
        return [func, name, closing](size_t i, size_t n, std::shared_ptr<StatsHub> hub) -> rt::Node* {
            return new Map_Node<tuple_t, result_t, F>(func, name, n, i, closing, std::move(hub));
        };
    }
};

}  // namespace wf
