// FlatMap operator.  Accepted signatures (/root/reference/API:34-37):
//   void(const tuple_t&, Shipper<result_t>&)  [+ RuntimeContext&]
// (/root/reference/wf/flatmap.hpp:127-160).
#pragma once
#include "simple_operator.hpp"

namespace wf {

template <typename tuple_t, typename result_t, typename F>
class FlatMap_Node : public Replica_Base {
public:
    FlatMap_Node(F func, std::string name, size_t par, size_t idx, meta::closing_func_t closing, std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub)), func_(std::move(func)), shipper_(this) {
        set_io_bytes(sizeof(tuple_t), sizeof(result_t));
    }
    void svc(void* msg, int) override {
        tuple_t* t = static_cast<tuple_t*>(msg);
        meta::invoke_rich(func_, context, static_cast<const tuple_t&>(*t), shipper_);
        tuple_delete(t);
    }

private:
    F func_;
    Shipper<result_t> shipper_;
};

template <typename tuple_t, typename result_t>
class FlatMap : public Simple_Operator<tuple_t, result_t> {
public:
    template <typename F>
    FlatMap(F func, size_t parallelism, std::string name, meta::closing_func_t closing = nullptr)
        : Simple_Operator<tuple_t, result_t>(name, parallelism, FORWARD, nullptr, make_factory(func, name, closing)) {}
    template <typename F>
    FlatMap(F func, size_t parallelism, std::string name, meta::closing_func_t closing, routing_func_t routing)
        : Simple_Operator<tuple_t, result_t>(name, parallelism, KEYBY, std::move(routing), make_factory(func, name, closing)) {}

private:
    template <typename F>
    static typename Simple_Operator<tuple_t, result_t>::replica_factory_t make_factory(F func, std::string name,
                                                                                        meta::closing_func_t closing) {
        return [func, name, closing](size_t i, size_t n, std::shared_ptr<StatsHub> hub) -> rt::Node* {
            return new FlatMap_Node<tuple_t, result_t, F>(func, name, n, i, closing, std::move(hub));
        };
    }
};

}  // namespace wf
