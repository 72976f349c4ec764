// Signature traits: deduce tuple_t / result_t and the "flavour" of a user
// callable (plain function, functor, lambda) for every operator.
//
// The reference does this with ~700 lines of overloaded "black-hole" function
// declarations (/root/reference/wf/meta.hpp:47-765) plus a legacy duplicate
// (meta_utils.hpp).  Here one generic callable_traits<> exposes return type and
// argument list; each operator classifies a callable with a handful of
// constexpr predicates.  The accepted signatures are exactly those listed in
// /root/reference/API and a wrong signature still dies in a static_assert with a
// readable message.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <optional>
#include <tuple>
#include <type_traits>
#include <vector>

#include "context.hpp"
#include "shipper.hpp"

namespace wf {

template <typename T>
class Iterable;

namespace meta {

template <typename T>
struct callable_traits : callable_traits<decltype(&std::remove_reference_t<T>::operator())> {};
template <typename R, typename... A>
struct callable_traits<R (*)(A...)> {
    using ret = R;
    using args = std::tuple<A...>;
    static constexpr size_t arity = sizeof...(A);
};
template <typename R, typename... A>
struct callable_traits<R(A...)> : callable_traits<R (*)(A...)> {};
template <typename R, typename... A>
struct callable_traits<R (&)(A...)> : callable_traits<R (*)(A...)> {};
template <typename C, typename R, typename... A>
struct callable_traits<R (C::*)(A...)> : callable_traits<R (*)(A...)> {};
template <typename C, typename R, typename... A>
struct callable_traits<R (C::*)(A...) const> : callable_traits<R (*)(A...)> {};
template <typename C, typename R, typename... A>
struct callable_traits<R (C::*)(A...) noexcept> : callable_traits<R (*)(A...)> {};
template <typename C, typename R, typename... A>
struct callable_traits<R (C::*)(A...) const noexcept> : callable_traits<R (*)(A...)> {};

template <typename F>
using ret_t = typename callable_traits<std::decay_t<F>>::ret;
template <typename F>
constexpr size_t arity_v = callable_traits<std::decay_t<F>>::arity;
template <typename F, size_t I>
using arg_t = std::tuple_element_t<I, typename callable_traits<std::decay_t<F>>::args>;
template <typename F, size_t I>
using arg_bare_t = std::remove_cv_t<std::remove_reference_t<arg_t<F, I>>>;

template <typename T>
constexpr bool is_const_ref_v = std::is_lvalue_reference_v<T>&& std::is_const_v<std::remove_reference_t<T>>;
template <typename T>
constexpr bool is_mut_ref_v = std::is_lvalue_reference_v<T> && !std::is_const_v<std::remove_reference_t<T>>;

template <typename T>
struct is_optional : std::false_type {};
template <typename T>
struct is_optional<std::optional<T>> : std::true_type {
    using inner = T;
};
template <typename T>
struct is_shipper : std::false_type {};
template <typename T>
struct is_shipper<Shipper<T>> : std::true_type {
    using inner = T;
};
template <typename T>
struct is_iterable : std::false_type {};
template <typename T>
struct is_iterable<Iterable<T>> : std::true_type {
    using inner = T;
};
template <typename T>
struct is_refwrap : std::false_type {};
template <typename T>
struct is_refwrap<std::reference_wrapper<T>> : std::true_type {
    using inner = T;
};
template <typename T>
struct is_vector : std::false_type {};
template <typename T>
struct is_vector<std::vector<T>> : std::true_type {
    using inner = T;
};

// true iff the last argument of F is RuntimeContext&
template <typename F>
constexpr bool last_is_context() {
    constexpr size_t n = arity_v<F>;
    if constexpr (n == 0)
        return false;
    else
        return std::is_same_v<arg_bare_t<F, n - 1>, RuntimeContext>;
}
template <typename F>
constexpr bool is_rich_v = last_is_context<F>();
// arity without the trailing context
template <typename F>
constexpr size_t base_arity_v = arity_v<F> - (is_rich_v<F> ? 1 : 0);

// call f(args...) or f(args..., ctx) depending on whether F is rich
template <typename F, typename... A>
inline decltype(auto) invoke_rich(F& f, RuntimeContext& ctx, A&&... a) {
    if constexpr (is_rich_v<F>)
        return f(std::forward<A>(a)..., ctx);
    else
        return f(std::forward<A>(a)...);
}

// ---- Source: bool(T&) | bool(Shipper<T>&) [+ctx] ---------------------------
template <typename F>
struct source_sig {
    static_assert(base_arity_v<F> == 1 && std::is_same_v<ret_t<F>, bool>,
                  "WindFlow Compilation Error - unknown signature passed to the Source_Builder:\n"
                  "  Candidate 1 : bool(tuple_t &)\n  Candidate 2 : bool(tuple_t &, RuntimeContext &)\n"
                  "  Candidate 3 : bool(Shipper<tuple_t> &)\n  Candidate 4 : bool(Shipper<tuple_t> &, RuntimeContext &)");
    using a0 = arg_bare_t<F, 0>;
    static constexpr bool loop = is_shipper<a0>::value;
    template <bool L, typename X>
    struct pick {
        using type = X;
    };
    template <typename X>
    struct pick<true, X> {
        using type = typename is_shipper<X>::inner;
    };
    using tuple_t = typename pick<loop, a0>::type;
};

// ---- Filter: bool(T&) | optional<R>(const T&) | optional<R*>(const T&) ----
template <typename F>
struct filter_sig {
    static_assert(base_arity_v<F> == 1,
                  "WindFlow Compilation Error - unknown signature passed to the Filter_Builder:\n"
                  "  Candidate 1 : bool(tuple_t &)\n  Candidate 2 : bool(tuple_t &, RuntimeContext &)\n"
                  "  Candidate 3 : std::optional<result_t>(const tuple_t &)\n"
                  "  Candidate 4 : std::optional<result_t>(const tuple_t &, RuntimeContext &)\n"
                  "  Candidate 5 : std::optional<result_t *>(const tuple_t &)\n"
                  "  Candidate 6 : std::optional<result_t *>(const tuple_t &, RuntimeContext &)");
    using tuple_t = arg_bare_t<F, 0>;
    using R = ret_t<F>;
    static constexpr bool in_place = std::is_same_v<R, bool>;
    static_assert(in_place || is_optional<R>::value, "WindFlow Compilation Error - Filter function must return bool or std::optional");
    template <typename X, bool B>
    struct res {
        using type = tuple_t;
        static constexpr int mode = 0;
    };
    template <typename X>
    struct res<X, false> {
        using inner = typename is_optional<X>::inner;
        static constexpr int mode = std::is_pointer_v<inner> ? 2 : 1;
        using type = std::remove_pointer_t<inner>;
    };
    using result_t = typename res<R, in_place>::type;
    static constexpr int mode = res<R, in_place>::mode;  // 0 in-place bool, 1 optional<R>, 2 optional<R*>
};

// ---- Map: void(T&) | void(const T&, R&) -----------------------------------
template <typename F>
struct map_sig {
    static_assert((base_arity_v<F> == 1 || base_arity_v<F> == 2) && std::is_void_v<ret_t<F>>,
                  "WindFlow Compilation Error - unknown signature passed to the Map_Builder:\n"
                  "  Candidate 1 : void(tuple_t &)\n  Candidate 2 : void(tuple_t &, RuntimeContext &)\n"
                  "  Candidate 3 : void(const tuple_t &, result_t &)\n"
                  "  Candidate 4 : void(const tuple_t &, result_t &, RuntimeContext &)");
    using tuple_t = arg_bare_t<F, 0>;
    static constexpr bool in_place = base_arity_v<F> == 1;
    template <bool B, typename X>
    struct res {
        using type = tuple_t;
    };
    template <typename X>
    struct res<false, X> {
        using type = arg_bare_t<X, 1>;
    };
    using result_t = typename res<in_place, F>::type;
};

// ---- FlatMap: void(const T&, Shipper<R>&) ---------------------------------
template <typename F>
struct flatmap_sig {
    static_assert(base_arity_v<F> == 2 && std::is_void_v<ret_t<F>> && is_shipper<arg_bare_t<F, 1>>::value,
                  "WindFlow Compilation Error - unknown signature passed to the FlatMap_Builder:\n"
                  "  Candidate 1 : void(const tuple_t &, Shipper<result_t> &)\n"
                  "  Candidate 2 : void(const tuple_t &, Shipper<result_t> &, RuntimeContext &)");
    using tuple_t = arg_bare_t<F, 0>;
    using result_t = typename is_shipper<arg_bare_t<F, 1>>::inner;
};

// ---- Accumulator: void(const T&, R&) --------------------------------------
template <typename F>
struct acc_sig {
    static_assert(base_arity_v<F> == 2 && std::is_void_v<ret_t<F>>,
                  "WindFlow Compilation Error - unknown signature passed to the Accumulator_Builder:\n"
                  "  Candidate 1 : void(const tuple_t &, result_t &)\n"
                  "  Candidate 2 : void(const tuple_t &, result_t &, RuntimeContext &)");
    using tuple_t = arg_bare_t<F, 0>;
    using result_t = arg_bare_t<F, 1>;
};

// ---- Sink: void(optional<T>&) | void(optional<reference_wrapper<T>>) ------
template <typename F>
struct sink_sig {
    static_assert(base_arity_v<F> == 1 && std::is_void_v<ret_t<F>> && is_optional<arg_bare_t<F, 0>>::value,
                  "WindFlow Compilation Error - unknown signature passed to the Sink_Builder:\n"
                  "  Candidate 1 : void(std::optional<tuple_t> &)\n"
                  "  Candidate 2 : void(std::optional<tuple_t> &, RuntimeContext &)\n"
                  "  Candidate 3 : void(std::optional<std::reference_wrapper<tuple_t>>)\n"
                  "  Candidate 4 : void(std::optional<std::reference_wrapper<tuple_t>>, RuntimeContext &)");
    using inner = typename is_optional<arg_bare_t<F, 0>>::inner;
    static constexpr bool by_ref = is_refwrap<inner>::value;
    template <bool B, typename X>
    struct pick {
        using type = X;
    };
    template <typename X>
    struct pick<true, X> {
        using type = typename is_refwrap<X>::inner;
    };
    using tuple_t = typename pick<by_ref, inner>::type;
};

// ---- Window functions ------------------------------------------------------
//   non-incremental: void(uint64_t, const Iterable<T>&, R&) [+ctx]
//   incremental:     void(uint64_t, const T&, R&)           [+ctx]
template <typename F>
struct win_sig {
    static_assert(base_arity_v<F> == 3 && std::is_void_v<ret_t<F>>,
                  "WindFlow Compilation Error - unknown signature passed to a window-based operator builder:\n"
                  "  Candidate 1 : void(uint64_t, const Iterable<tuple_t> &, result_t &)\n"
                  "  Candidate 2 : void(uint64_t, const Iterable<tuple_t> &, result_t &, RuntimeContext &)\n"
                  "  Candidate 3 : void(uint64_t, const tuple_t &, result_t &)\n"
                  "  Candidate 4 : void(uint64_t, const tuple_t &, result_t &, RuntimeContext &)");
    using a1 = arg_bare_t<F, 1>;
    static constexpr bool is_nic = is_iterable<a1>::value;
    template <bool B, typename X>
    struct pick {
        using type = X;
    };
    template <typename X>
    struct pick<true, X> {
        using type = typename is_iterable<X>::inner;
    };
    using tuple_t = typename pick<is_nic, a1>::type;
    using result_t = arg_bare_t<F, 2>;
};

// ---- FFAT lift / combine ---------------------------------------------------
template <typename F>
struct lift_sig {
    static_assert(base_arity_v<F> == 2 && std::is_void_v<ret_t<F>>,
                  "WindFlow Compilation Error - unknown signature passed as lift function:\n"
                  "  Candidate 1 : void(const tuple_t &, result_t &)\n"
                  "  Candidate 2 : void(const tuple_t &, result_t &, RuntimeContext &)");
    using tuple_t = arg_bare_t<F, 0>;
    using result_t = arg_bare_t<F, 1>;
};
template <typename F>
struct comb_sig {
    static_assert(base_arity_v<F> == 3 && std::is_void_v<ret_t<F>>,
                  "WindFlow Compilation Error - unknown signature passed as combine function:\n"
                  "  Candidate 1 : void(const result_t &, const result_t &, result_t &)\n"
                  "  Candidate 2 : void(const result_t &, const result_t &, result_t &, RuntimeContext &)");
    using result_t = arg_bare_t<F, 2>;
};

// ---- split function: integral(T&) | vector<integral>(T&) -------------------
template <typename F>
struct split_sig {
    static_assert(arity_v<F> == 1,
                  "WindFlow Compilation Error - unknown signature passed to split():\n"
                  "  Candidate 1 : integral_t(const tuple_t &)\n  Candidate 2 : std::vector<integral_t>(const tuple_t &)\n"
                  "  Candidate 3 : integral_t(tuple_t &)\n  Candidate 4 : std::vector<integral_t>(tuple_t &)");
    using tuple_t = arg_bare_t<F, 0>;
    using R = ret_t<F>;
    static constexpr bool multi = is_vector<R>::value;
    static_assert(multi || std::is_integral_v<R>, "WindFlow Compilation Error - split function must return an integral or a vector of integrals");
};

// ---- GPU window functor: (uint64_t, const T*, size_t, R*, char*, size_t) ---
template <typename F>
struct win_gpu_sig {
    static_assert(arity_v<F> == 6,
                  "WindFlow Compilation Error - unknown signature passed to a GPU window operator builder:\n"
                  "  Candidate : __host__ __device__ void(uint64_t, const tuple_t *, size_t, result_t *, char *, size_t)");
    using tuple_t = std::remove_cv_t<std::remove_pointer_t<arg_bare_t<F, 1>>>;
    using result_t = std::remove_pointer_t<arg_bare_t<F, 3>>;
};
// detect whether a callable has the 6-argument device-window shape
template <typename F>
constexpr bool is_gpu_win_func() {
    if constexpr (arity_v<F> == 6)
        return std::is_pointer_v<arg_bare_t<F, 1>> && std::is_pointer_v<arg_bare_t<F, 3>>;
    else
        return false;
}

// closing function: void(RuntimeContext&)
using closing_func_t = std::function<void(RuntimeContext&)>;

}  // namespace meta
}  // namespace wf
