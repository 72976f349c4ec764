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
#include "key_farm.hpp"
#include "key_ffat.hpp"
#include "pane_farm.hpp"
#include "win_farm.hpp"
#include "win_mapreduce.hpp"
#include "win_seq.hpp"
#include "win_seqffat.hpp"

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


// ---- window builders ---------------------------------------------------------
// options shared by every window builder: withCBWindows / withTBWindows
template <typename Derived>
class Window_Builder_Base : public Builder_Base<Derived> {
public:
    Derived& withCBWindows(uint64_t win_len, uint64_t slide_len) {
        win_ = win_len;
        slide_ = slide_len;
        type_ = CB;
        delay_ = 0;
        return this->self();
    }
    Derived& withTBWindows(std::chrono::microseconds win_len, std::chrono::microseconds slide_len,
                           std::chrono::microseconds triggering_delay = std::chrono::microseconds::zero()) {
        win_ = (uint64_t)win_len.count();
        slide_ = (uint64_t)slide_len.count();
        delay_ = (uint64_t)triggering_delay.count();
        type_ = TB;
        return this->self();
    }

protected:
    using Builder_Base<Derived>::Builder_Base;
    uint64_t win_ = 1, slide_ = 1, delay_ = 0;
    win_type_t type_ = CB;
};

template <typename F>
class WinSeq_Builder : public Window_Builder_Base<WinSeq_Builder<F>> {
public:
    using tuple_t = typename meta::win_sig<F>::tuple_t;
    using result_t = typename meta::win_sig<F>::result_t;
    using winseq_t = Win_Seq<tuple_t, result_t>;
    explicit WinSeq_Builder(F func) : Window_Builder_Base<WinSeq_Builder<F>>("win_seq"), func_(std::move(func)) {}
    winseq_t build() { return winseq_t(func_, this->win_, this->slide_, this->delay_, this->type_, this->name_, this->closing_); }
    winseq_t* build_ptr() { return new winseq_t(func_, this->win_, this->slide_, this->delay_, this->type_, this->name_, this->closing_); }
    std::unique_ptr<winseq_t> build_unique() { return std::unique_ptr<winseq_t>(build_ptr()); }

private:
    F func_;
};

template <typename L, typename C>
class WinSeqFFAT_Builder : public Window_Builder_Base<WinSeqFFAT_Builder<L, C>> {
public:
    using tuple_t = typename meta::lift_sig<L>::tuple_t;
    using result_t = typename meta::lift_sig<L>::result_t;
    static_assert(std::is_same_v<result_t, typename meta::comb_sig<C>::result_t>,
                  "WindFlow Compilation Error - lift and combine functions must agree on result_t");
    using winffat_t = Win_SeqFFAT<tuple_t, result_t>;
    WinSeqFFAT_Builder(L lift, C comb) : Window_Builder_Base<WinSeqFFAT_Builder<L, C>>("win_seqffat"), lift_(std::move(lift)), comb_(std::move(comb)) {}
    winffat_t build() { return winffat_t(lift_, comb_, this->win_, this->slide_, this->delay_, this->type_, this->name_, this->closing_); }
    winffat_t* build_ptr() { return new winffat_t(lift_, comb_, this->win_, this->slide_, this->delay_, this->type_, this->name_, this->closing_); }
    std::unique_ptr<winffat_t> build_unique() { return std::unique_ptr<winffat_t>(build_ptr()); }

private:
    L lift_;
    C comb_;
};

template <typename L, typename C>
class KeyFFAT_Builder : public Window_Builder_Base<KeyFFAT_Builder<L, C>> {
public:
    using tuple_t = typename meta::lift_sig<L>::tuple_t;
    using result_t = typename meta::lift_sig<L>::result_t;
    static_assert(std::is_same_v<result_t, typename meta::comb_sig<C>::result_t>,
                  "WindFlow Compilation Error - lift and combine functions must agree on result_t");
    using keyffat_t = Key_FFAT<tuple_t, result_t>;
    KeyFFAT_Builder(L lift, C comb) : Window_Builder_Base<KeyFFAT_Builder<L, C>>("key_ffat"), lift_(std::move(lift)), comb_(std::move(comb)) {}
    KeyFFAT_Builder& set_KeyBy(routing_func_t r) {
        routing_ = std::move(r);
        return *this;
    }
    keyffat_t build() { return *std::unique_ptr<keyffat_t>(build_ptr()); }
    keyffat_t* build_ptr() {
        return new keyffat_t(lift_, comb_, this->win_, this->slide_, this->delay_, this->type_, this->parallelism_, this->name_, this->closing_, routing_);
    }
    std::unique_ptr<keyffat_t> build_unique() { return std::unique_ptr<keyffat_t>(build_ptr()); }

private:
    L lift_;
    C comb_;
    routing_func_t routing_ = default_routing;
};

namespace detail {
// tuple/result types of a WinFarm/KeyFarm builder input: a window function, or a
// Pane_Farm / Win_MapReduce to be nested
template <typename T, bool IsOp>
struct farm_input_types {
    using tuple_t = typename meta::win_sig<T>::tuple_t;
    using result_t = typename meta::win_sig<T>::result_t;
};
template <typename A, typename B>
struct farm_input_types<Pane_Farm<A, B>, true> {
    using tuple_t = A;
    using result_t = B;
};
template <typename A, typename B>
struct farm_input_types<Win_MapReduce<A, B>, true> {
    using tuple_t = A;
    using result_t = B;
};
}  // namespace detail

template <typename T>
class WinFarm_Builder : public Window_Builder_Base<WinFarm_Builder<T>> {
public:
    static constexpr bool nested = std::is_base_of_v<Basic_Operator, T>;
    using tuple_t = typename detail::farm_input_types<T, nested>::tuple_t;
    using result_t = typename detail::farm_input_types<T, nested>::result_t;
    using winfarm_t = Win_Farm<tuple_t, result_t>;
    // a window function (by value) or a Pane_Farm / Win_MapReduce (by reference: it must outlive the builder)
    template <typename U>
    explicit WinFarm_Builder(U&& input) : Window_Builder_Base<WinFarm_Builder<T>>("win_farm"), input_(store(std::forward<U>(input))) {
        if constexpr (nested) {
            this->win_ = input.getWinLen();
            this->slide_ = input.getSlideLen();
            this->delay_ = input.getTriggeringDelay();
            this->type_ = input.getWinType();
        }
    }
    WinFarm_Builder& withOrdering(bool ordered) {
        ordered_ = ordered;
        return *this;
    }
    WinFarm_Builder& withOptLevel(opt_level_t o) {
        opt_ = o;
        return *this;
    }
    winfarm_t build() { return *std::unique_ptr<winfarm_t>(build_ptr()); }
    winfarm_t* build_ptr() {
        if constexpr (nested)
            return new winfarm_t(*input_, this->win_, this->slide_, this->delay_, this->type_, this->parallelism_, this->name_, this->closing_, ordered_, opt_);
        else
            return new winfarm_t(input_, this->win_, this->slide_, this->delay_, this->type_, this->parallelism_, this->name_, this->closing_, ordered_, opt_);
    }
    std::unique_ptr<winfarm_t> build_unique() { return std::unique_ptr<winfarm_t>(build_ptr()); }

private:
    using stored_t = std::conditional_t<nested, T*, T>;
    template <typename U>
    static stored_t store(U&& u) {
        if constexpr (nested)
            return &u;
        else
            return stored_t(std::forward<U>(u));
    }
    stored_t input_;
    bool ordered_ = true;
    opt_level_t opt_ = LEVEL2;  // reference default (builders.hpp:1150)
};
template <typename U>
WinFarm_Builder(U&&) -> WinFarm_Builder<std::decay_t<U>>;

template <typename T>
class KeyFarm_Builder : public Window_Builder_Base<KeyFarm_Builder<T>> {
public:
    static constexpr bool nested = std::is_base_of_v<Basic_Operator, T>;
    using tuple_t = typename detail::farm_input_types<T, nested>::tuple_t;
    using result_t = typename detail::farm_input_types<T, nested>::result_t;
    using keyfarm_t = Key_Farm<tuple_t, result_t>;
    template <typename U>
    explicit KeyFarm_Builder(U&& input) : Window_Builder_Base<KeyFarm_Builder<T>>("key_farm"), input_(store(std::forward<U>(input))) {
        if constexpr (nested) {
            this->win_ = input.getWinLen();
            this->slide_ = input.getSlideLen();
            this->delay_ = input.getTriggeringDelay();
            this->type_ = input.getWinType();
        }
    }
    KeyFarm_Builder& set_KeyBy(routing_func_t r) {
        routing_ = std::move(r);
        return *this;
    }
    KeyFarm_Builder& withOptLevel(opt_level_t o) {
        opt_ = o;
        return *this;
    }
    keyfarm_t build() { return *std::unique_ptr<keyfarm_t>(build_ptr()); }
    keyfarm_t* build_ptr() {
        if constexpr (nested)
            return new keyfarm_t(*input_, this->win_, this->slide_, this->delay_, this->type_, this->parallelism_, this->name_, this->closing_, routing_, opt_);
        else
            return new keyfarm_t(input_, this->win_, this->slide_, this->delay_, this->type_, this->parallelism_, this->name_, this->closing_, routing_, opt_);
    }
    std::unique_ptr<keyfarm_t> build_unique() { return std::unique_ptr<keyfarm_t>(build_ptr()); }

private:
    using stored_t = std::conditional_t<nested, T*, T>;
    template <typename U>
    static stored_t store(U&& u) {
        if constexpr (nested)
            return &u;
        else
            return stored_t(std::forward<U>(u));
    }
    stored_t input_;
    routing_func_t routing_ = default_routing;
    opt_level_t opt_ = LEVEL2;
};
template <typename U>
KeyFarm_Builder(U&&) -> KeyFarm_Builder<std::decay_t<U>>;

// common part of PaneFarm_Builder and WinMapReduce_Builder
template <typename Derived>
class TwoStage_Builder_Base : public Window_Builder_Base<Derived> {
public:
    Derived& withParallelism(size_t first, size_t second) {
        par1_ = first;
        par2_ = second;
        return this->self();
    }
    Derived& withOrdering(bool ordered) {
        ordered_ = ordered;
        return this->self();
    }
    Derived& withOptLevel(opt_level_t o) {
        opt_ = o;
        return this->self();
    }
    // the operator will be replicated inside a Win_Farm or a Key_Farm
    Derived& prepare4Nesting() {
        opt_ = LEVEL2;
        nesting_ = true;
        return this->self();
    }

protected:
    using Window_Builder_Base<Derived>::Window_Builder_Base;
    size_t par1_ = 1, par2_ = 1;
    bool ordered_ = true;
    opt_level_t opt_ = LEVEL0;  // reference default (builders.hpp:1796)
    bool nesting_ = false;
};

template <typename F, typename G>
class PaneFarm_Builder : public TwoStage_Builder_Base<PaneFarm_Builder<F, G>> {
public:
    using tuple_t = typename meta::win_sig<F>::tuple_t;
    using result_t = typename meta::win_sig<F>::result_t;
    static_assert(std::is_same_v<result_t, typename meta::win_sig<G>::tuple_t> && std::is_same_v<result_t, typename meta::win_sig<G>::result_t>,
                  "WindFlow Compilation Error - the WLQ function must consume and produce the result type of the PLQ function");
    using panefarm_t = Pane_Farm<tuple_t, result_t>;
    PaneFarm_Builder(F plq, G wlq) : TwoStage_Builder_Base<PaneFarm_Builder<F, G>>("pane_farm"), plq_(std::move(plq)), wlq_(std::move(wlq)) {}
    panefarm_t build() { return *std::unique_ptr<panefarm_t>(build_ptr()); }
    panefarm_t* build_ptr() {
        auto* p = new panefarm_t(plq_, wlq_, this->win_, this->slide_, this->delay_, this->type_, this->par1_, this->par2_, this->name_,
                                 this->closing_, this->ordered_, this->opt_);
        if (this->nesting_) p->setPrepared4Nesting();
        return p;
    }
    std::unique_ptr<panefarm_t> build_unique() { return std::unique_ptr<panefarm_t>(build_ptr()); }

private:
    F plq_;
    G wlq_;
};

template <typename F, typename G>
class WinMapReduce_Builder : public TwoStage_Builder_Base<WinMapReduce_Builder<F, G>> {
public:
    using tuple_t = typename meta::win_sig<F>::tuple_t;
    using result_t = typename meta::win_sig<F>::result_t;
    static_assert(std::is_same_v<result_t, typename meta::win_sig<G>::tuple_t> && std::is_same_v<result_t, typename meta::win_sig<G>::result_t>,
                  "WindFlow Compilation Error - the REDUCE function must consume and produce the result type of the MAP function");
    using winmapreduce_t = Win_MapReduce<tuple_t, result_t>;
    WinMapReduce_Builder(F map, G reduce) : TwoStage_Builder_Base<WinMapReduce_Builder<F, G>>("win_mapreduce"), map_(std::move(map)), reduce_(std::move(reduce)) {
        this->par1_ = 2;
    }
    winmapreduce_t build() { return *std::unique_ptr<winmapreduce_t>(build_ptr()); }
    winmapreduce_t* build_ptr() {
        auto* p = new winmapreduce_t(map_, reduce_, this->win_, this->slide_, this->delay_, this->type_, this->par1_, this->par2_, this->name_,
                                     this->closing_, this->ordered_, this->opt_);
        if (this->nesting_) p->setPrepared4Nesting();
        return p;
    }
    std::unique_ptr<winmapreduce_t> build_unique() { return std::unique_ptr<winmapreduce_t>(build_ptr()); }

private:
    F map_;
    G reduce_;
};

}  // namespace wf
