// Shipper: output handle given to loop-mode Source functions and to FlatMap
// functions (API parity: /root/reference/wf/shipper.hpp:51-104).
#pragma once
#include <cstddef>
#include <utility>

#include "rt/node.hpp"
#include "rt/tuple_pool.hpp"

namespace wf {

template <typename result_t>
class Shipper {
public:
    explicit Shipper(rt::Node* node) : node_(node) {}
    // copy semantics
    void push(const result_t& r) {
        node_->emit_value(r);  // (no heap tuple when the consumer only copies it, e.g. a Batcher in front of a GPU operator)
        delivered_++;
    }
    // move semantics
    void push(result_t&& r) {
        node_->emit(tuple_new<result_t>(std::move(r)));
        delivered_++;
    }
    // ownership transfer of a heap object created with new: tuples on the edges live in the runtime's tuple
    // pool (rt/tuple_pool.hpp), so the object is moved into a pooled one and the caller's is destroyed here
    void push(result_t* r) {
        node_->emit(tuple_adopt(r));
        delivered_++;
    }
    unsigned long delivered() const { return delivered_; }

private:
    rt::Node* node_;
    unsigned long delivered_ = 0;
};

}  // namespace wf
