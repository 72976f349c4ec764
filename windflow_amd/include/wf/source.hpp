// Source operator.  Accepted signatures (/root/reference/API:11-16):
//   bool(tuple_t&) | bool(Shipper<tuple_t>&)  [+ RuntimeContext&]
// Itemized mode emits one tuple per call, including the one filled by the call
// that returns false; loop mode pushes through a Shipper
// (/root/reference/wf/source.hpp:168-237).
#pragma once
#include "simple_operator.hpp"

namespace wf {

template <typename tuple_t, typename F>
class Source_Node : public Replica_Base {
public:
    Source_Node(F func, std::string name, size_t par, size_t idx, meta::closing_func_t closing, std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub)), func_(std::move(func)), shipper_(this) {
        set_io_bytes(0, sizeof(tuple_t));
    }
    bool is_source() const override { return true; }
    bool generate() override {
        if constexpr (meta::source_sig<F>::loop) {
            return meta::invoke_rich(func_, context, shipper_);
        } else {
            if (this->feeds_value_consumer()) {
                // the consumer copies the tuple (a Batcher in front of a GPU operator): no heap tuple at all
                tuple_t t{};
                const bool more = meta::invoke_rich(func_, context, t);
                this->emit_value(t);
                return more;
            }
            tuple_t* t = tuple_new<tuple_t>();
            const bool more = meta::invoke_rich(func_, context, *t);
            emit(t);
            return more;
        }
    }

private:
    F func_;
    Shipper<tuple_t> shipper_;
};

template <typename tuple_t>
class Source : public Simple_Operator<void, tuple_t> {
public:
    template <typename F>
    Source(F func, size_t parallelism, std::string name, meta::closing_func_t closing = nullptr)
        : Simple_Operator<void, tuple_t>(name, parallelism, NONE, nullptr,
                                         [func, name, closing](size_t i, size_t n, std::shared_ptr<StatsHub> hub) -> rt::Node* {
                                             return new Source_Node<tuple_t, F>(func, name, n, i, closing, std::move(hub));
                                         }) {}
};

}  // namespace wf
