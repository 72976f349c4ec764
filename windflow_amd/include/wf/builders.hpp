// Fluent builders of the CPU operators.  Method names and defaults follow
// /root/reference/wf/builders.hpp (Source_Builder :49, Filter_Builder :168,
// Map_Builder :332, FlatMap_Builder :494, Accumulator_Builder :654,
// WinSeq_Builder :796, WinSeqFFAT_Builder :957, WinFarm_Builder :1127,
// KeyFarm_Builder :1350, KeyFFAT_Builder :1576, PaneFarm_Builder :1762,
// WinMapReduce_Builder :1982, Sink_Builder :2202): parallelism 1, names
// "source"/"filter"/..., routing k % n, window 1/1 CB.
#pragma once
#include <chrono>
#include <memory>
#include <string>

#include "accumulator.hpp"
#include "filter.hpp"
#include "flatmap.hpp"
#include "map.hpp"
#include "meta.hpp"
#include "sink.hpp"
#include "source.hpp"

namespace wf {

// CRTP base with the options every builder has
template <typename Derived>
class Builder_Base {
public:
    Derived& withName(std::string n) {
        name_ = std::move(n);
        return self();
    }
    Derived& withParallelism(size_t p) {
        parallelism_ = p;
        return self();
    }
    template <typename C>
    Derived& withClosingFunction(C c) {
        closing_ = meta::closing_func_t(std::move(c));
        return self();
    }

protected:
    explicit Builder_Base(std::string default_name) : name_(std::move(default_name)) {}
    Derived& self() { return static_cast<Derived&>(*this); }
    std::string name_;
    size_t parallelism_ = 1;
    meta::closing_func_t closing_;
};

// adds enable_KeyBy() for operators that can be FORWARD or KEYBY
template <typename Derived>
class Keyable_Builder : public Builder_Base<Derived> {
public:
    Derived& enable_KeyBy(routing_func_t r = default_routing) {
        keyed_ = true;
        routing_ = std::move(r);
        return this->self();
    }

protected:
    using Builder_Base<Derived>::Builder_Base;
    bool keyed_ = false;
    routing_func_t routing_ = default_routing;
};

template <typename F>
class Source_Builder : public Builder_Base<Source_Builder<F>> {
public:
    using tuple_t = typename meta::source_sig<F>::tuple_t;
    using source_t = Source<tuple_t>;
    explicit Source_Builder(F func) : Builder_Base<Source_Builder<F>>("source"), func_(std::move(func)) {}
    source_t build() { return source_t(func_, this->parallelism_, this->name_, this->closing_); }
    source_t* build_ptr() { return new source_t(func_, this->parallelism_, this->name_, this->closing_); }
    std::unique_ptr<source_t> build_unique() { return std::unique_ptr<source_t>(build_ptr()); }

private:
    F func_;
};

template <typename F>
class Filter_Builder : public Keyable_Builder<Filter_Builder<F>> {
public:
    using tuple_t = typename meta::filter_sig<F>::tuple_t;
    using result_t = typename meta::filter_sig<F>::result_t;
    using filter_t = Filter<tuple_t, result_t>;
    explicit Filter_Builder(F func) : Keyable_Builder<Filter_Builder<F>>("filter"), func_(std::move(func)) {}
    filter_t build() { return *std::unique_ptr<filter_t>(build_ptr()); }
    filter_t* build_ptr() {
        return this->keyed_ ? new filter_t(func_, this->parallelism_, this->name_, this->closing_, this->routing_)
                            : new filter_t(func_, this->parallelism_, this->name_, this->closing_);
    }
    std::unique_ptr<filter_t> build_unique() { return std::unique_ptr<filter_t>(build_ptr()); }

private:
    F func_;
};

template <typename F>
class Map_Builder : public Keyable_Builder<Map_Builder<F>> {
public:
    using tuple_t = typename meta::map_sig<F>::tuple_t;
    using result_t = typename meta::map_sig<F>::result_t;
    using map_t = Map<tuple_t, result_t>;
    explicit Map_Builder(F func) : Keyable_Builder<Map_Builder<F>>("map"), func_(std::move(func)) {}
    map_t build() { return *std::unique_ptr<map_t>(build_ptr()); }
    map_t* build_ptr() {
        return this->keyed_ ? new map_t(func_, this->parallelism_, this->name_, this->closing_, this->routing_)
                            : new map_t(func_, this->parallelism_, this->name_, this->closing_);
    }
    std::unique_ptr<map_t> build_unique() { return std::unique_ptr<map_t>(build_ptr()); }

private:
    F func_;
};

template <typename F>
class FlatMap_Builder : public Keyable_Builder<FlatMap_Builder<F>> {
public:
    using tuple_t = typename meta::flatmap_sig<F>::tuple_t;
    using result_t = typename meta::flatmap_sig<F>::result_t;
    using flatmap_t = FlatMap<tuple_t, result_t>;
    explicit FlatMap_Builder(F func) : Keyable_Builder<FlatMap_Builder<F>>("flatmap"), func_(std::move(func)) {}
    flatmap_t build() { return *std::unique_ptr<flatmap_t>(build_ptr()); }
    flatmap_t* build_ptr() {
        return this->keyed_ ? new flatmap_t(func_, this->parallelism_, this->name_, this->closing_, this->routing_)
                            : new flatmap_t(func_, this->parallelism_, this->name_, this->closing_);
    }
    std::unique_ptr<flatmap_t> build_unique() { return std::unique_ptr<flatmap_t>(build_ptr()); }

private:
    F func_;
};

template <typename F>
class Accumulator_Builder : public Builder_Base<Accumulator_Builder<F>> {
public:
    using tuple_t = typename meta::acc_sig<F>::tuple_t;
    using result_t = typename meta::acc_sig<F>::result_t;
    using accumulator_t = Accumulator<tuple_t, result_t>;
    explicit Accumulator_Builder(F func) : Builder_Base<Accumulator_Builder<F>>("accumulator"), func_(std::move(func)) {}
    Accumulator_Builder& withInitialValue(const result_t& v) {
        init_ = v;
        return *this;
    }
    Accumulator_Builder& set_KeyBy(routing_func_t r) {
        routing_ = std::move(r);
        return *this;
    }
    accumulator_t build() { return accumulator_t(func_, init_, this->parallelism_, this->name_, this->closing_, routing_); }
    accumulator_t* build_ptr() { return new accumulator_t(func_, init_, this->parallelism_, this->name_, this->closing_, routing_); }
    std::unique_ptr<accumulator_t> build_unique() { return std::unique_ptr<accumulator_t>(build_ptr()); }

private:
    F func_;
    result_t init_{};
    routing_func_t routing_ = default_routing;
};

template <typename F>
class Sink_Builder : public Keyable_Builder<Sink_Builder<F>> {
public:
    using tuple_t = typename meta::sink_sig<F>::tuple_t;
    using sink_t = Sink<tuple_t>;
    explicit Sink_Builder(F func) : Keyable_Builder<Sink_Builder<F>>("sink"), func_(std::move(func)) {}
    sink_t build() { return *std::unique_ptr<sink_t>(build_ptr()); }
    sink_t* build_ptr() {
        return this->keyed_ ? new sink_t(func_, this->parallelism_, this->name_, this->closing_, this->routing_)
                            : new sink_t(func_, this->parallelism_, this->name_, this->closing_);
    }
    std::unique_ptr<sink_t> build_unique() { return std::unique_ptr<sink_t>(build_ptr()); }

private:
    F func_;
};

}  // namespace wf
