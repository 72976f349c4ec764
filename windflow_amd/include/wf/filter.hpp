// Filter operator.  Accepted signatures (/root/reference/API:18-25):
//   bool(tuple_t&) | std::optional<result_t>(const tuple_t&) |
//   std::optional<result_t*>(const tuple_t&)   [+ RuntimeContext&]
// (three modes, /root/reference/wf/filter.hpp:198-310).
#pragma once
#include <optional>

#include "simple_operator.hpp"

namespace wf {

template <typename tuple_t, typename result_t, typename F>
class Filter_Node : public Replica_Base {
public:
    Filter_Node(F func, std::string name, size_t par, size_t idx, meta::closing_func_t closing, std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub)), func_(std::move(func)) {
        set_io_bytes(sizeof(tuple_t), sizeof(result_t));
    }
    void svc(void* msg, int) override {
        tuple_t* t = static_cast<tuple_t*>(msg);
        constexpr int mode = meta::filter_sig<F>::mode;
        if constexpr (mode == 0) {
            if (meta::invoke_rich(func_, context, *t)) {
                emit(t);
            } else {
                ignored_++;
                tuple_delete(t);
            }
        } else if constexpr (mode == 1) {
            std::optional<result_t> out = meta::invoke_rich(func_, context, static_cast<const tuple_t&>(*t));
            tuple_delete(t);
            if (out)
                emit(tuple_new<result_t>(std::move(*out)));
            else
                ignored_++;
        } else {
            std::optional<result_t*> out = meta::invoke_rich(func_, context, static_cast<const tuple_t&>(*t));
            tuple_delete(t);
            if (out)
                emit(tuple_adopt(*out));
            else
                ignored_++;
        }
    }

protected:
    void finalize_stats(Stats_Record& s) override { s.inputs_ignored = ignored_; }

private:
    F func_;
    uint64_t ignored_ = 0;
};

template <typename tuple_t, typename result_t = tuple_t>
class Filter : public Simple_Operator<tuple_t, result_t> {
public:
    template <typename F>
    Filter(F func, size_t parallelism, std::string name, meta::closing_func_t closing = nullptr)
        : Simple_Operator<tuple_t, result_t>(name, parallelism, FORWARD, nullptr, make_factory(func, name, closing)) {}
    template <typename F>
    Filter(F func, size_t parallelism, std::string name, meta::closing_func_t closing, routing_func_t routing)
        : Simple_Operator<tuple_t, result_t>(name, parallelism, KEYBY, std::move(routing), make_factory(func, name, closing)) {}

private:
    template <typename F>
    static typename Simple_Operator<tuple_t, result_t>::replica_factory_t make_factory(F func, std::string name,
                                                                                        meta::closing_func_t closing) {
        return [func, name, closing](size_t i, size_t n, std::shared_ptr<StatsHub> hub) -> rt::Node* {
            return new Filter_Node<tuple_t, result_t, F>(func, name, n, i, closing, std::move(hub));
        };
    }
};

}  // namespace wf
