// RuntimeContext + LocalStorage: what "rich" user functions receive.
// API parity: /root/reference/wf/context.hpp:49-102, wf/local_storage.hpp:49-135.
#pragma once
#include <cstddef>
#include <memory>
#include <string>
#include <typeindex>
#include <unordered_map>

namespace wf {

// Per-replica keyed store of arbitrary typed objects.  The reference keeps raw
// void* and leaks on mismatch; here each slot owns its object and remembers its
// type so a mismatching get<T>() is detected.
class LocalStorage {
public:
    template <typename T>
    T& get(const std::string& name) {
        auto it = slots_.find(name);
        if (it == slots_.end() || it->second.type != std::type_index(typeid(T))) {
            Slot s{std::type_index(typeid(T)), std::shared_ptr<void>(new T(), [](void* p) { delete static_cast<T*>(p); })};
            it = slots_.insert_or_assign(name, std::move(s)).first;
        }
        return *static_cast<T*>(it->second.obj.get());
    }
    template <typename T>
    void put(const std::string& name, const T& value) {
        Slot s{std::type_index(typeid(T)), std::shared_ptr<void>(new T(value), [](void* p) { delete static_cast<T*>(p); })};
        slots_.insert_or_assign(name, std::move(s));
    }
    template <typename T>
    void remove(const std::string& name) {
        slots_.erase(name);
    }
    bool isContained(const std::string& name) const { return slots_.count(name) != 0; }
    size_t getSize() const { return slots_.size(); }

private:
    struct Slot {
        std::type_index type;
        std::shared_ptr<void> obj;
    };
    std::unordered_map<std::string, Slot> slots_;
};

class RuntimeContext {
public:
    RuntimeContext() = default;
    RuntimeContext(size_t parallelism, size_t index) : parallelism_(parallelism), index_(index) {}
    size_t getParallelism() const { return parallelism_; }
    size_t getReplicaIndex() const { return index_; }
    LocalStorage& getLocalStorage() { return storage_; }

private:
    size_t parallelism_ = 1;
    size_t index_ = 0;
    LocalStorage storage_;
};

}  // namespace wf
