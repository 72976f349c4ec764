// Basic definitions shared by the whole library: enums, defaults, time helpers,
// window-operator configuration.  Public names follow the reference's API
// (/root/reference/wf/basic.hpp:46-274) so user code keeps compiling; the
// implementation is new.
#pragma once
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <atomic>
#include <thread>
#include <string>
#include <tuple>
#include <type_traits>

#include "rt/tuple_pool.hpp"

// Tuple types used by GPU operators must expose their control-field accessors to
// device code: mark them WF_HOST_DEVICE (expands to nothing for a host compiler).
#if defined(__HIPCC__)
#define WF_HOST_DEVICE __host__ __device__
#else
#define WF_HOST_DEVICE
#endif

namespace wf {

// ---- time helpers (reference: basic.hpp:55-74) ----------------------------
inline uint64_t current_time_usecs() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(
               std::chrono::steady_clock::now().time_since_epoch())
        .count();
}
inline uint64_t current_time_nsecs() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch())
        .count();
}

// ---- defaults (reference: basic.hpp:77-83) --------------------------------
constexpr size_t DEFAULT_VECTOR_CAPACITY = 500;
constexpr size_t DEFAULT_BATCH_SIZE_TB = 1000;
// wave64 hardware: 256 threads = 4 waves, one per SIMD of a CU
constexpr size_t DEFAULT_GPU_THREADS_PER_BLOCK = 256;

// ---- enums (reference: basic.hpp:86-132) ----------------------------------
enum class Mode { DEFAULT, DETERMINISTIC };
enum win_type_t { CB, TB };
enum opt_level_t { LEVEL0, LEVEL1, LEVEL2 };
enum routing_modes_t { NONE, FORWARD, KEYBY, COMPLEX };
enum pattern_t { SEQ_CPU, SEQ_GPU, KF_CPU, KF_GPU, KFF_CPU, KFF_GPU, WF_CPU, WF_GPU, PF_CPU, PF_GPU, WMR_CPU, WMR_GPU };
enum win_event_t { OLD, IN, DELAYED, FIRED, BATCHED };
enum ordering_mode_t { ID, TS, TS_RENUMBERING };
enum role_t { SEQ, PLQ, WLQ, MAP, REDUCE };

// ---- ANSI colours (reference: basic.hpp:135-151) --------------------------
#define WF_DEFAULT_COLOR "\033[0m"
#define WF_RED "\033[1;31m"
#define WF_GREEN "\033[1;32m"
#define WF_YELLOW "\033[1;33m"
#define WF_BLUE "\033[1;34m"

// configuration error: the reference prints and exit()s; we do
// the same by default but go through one function so tests can intercept it.
[[noreturn]] inline void fatal_error(const std::string& msg) {
    // Several runtime threads can hit the same condition at once (every replica of an operator whose key table is full):
    // the first one reports and ends the process WITHOUT running static destructors under the feet of the others
    // (exit() from two threads is undefined behaviour and used to end in a segmentation fault); the others wait for it.
    static std::atomic<bool> once{false};
    if (once.exchange(true)) {
        for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));
    }
    std::cerr << WF_RED << "WindFlow Error: " << msg << WF_DEFAULT_COLOR << std::endl;
    std::fflush(nullptr);
    std::_Exit(EXIT_FAILURE);
}

// Position of a window replica inside (possibly nested) window-parallel
// operators; drives the window-assignment arithmetic of SURVEY.md A.4
// (reference: basic.hpp:154-184).
struct WinOperatorConfig {
    size_t id_outer = 0;
    size_t n_outer = 1;
    uint64_t slide_outer = 0;
    size_t id_inner = 0;
    size_t n_inner = 1;
    uint64_t slide_inner = 0;
    WinOperatorConfig() = default;
    WinOperatorConfig(size_t io, size_t no, uint64_t so, size_t ii, size_t ni, uint64_t si)
        : id_outer(io), n_outer(no), slide_outer(so), id_inner(ii), n_inner(ni), slide_inner(si) {}
};

// ---- tuple contract helpers -----------------------------------------------
// User tuples expose tuple<key_t, uint64_t id, uint64_t ts> getControlFields()
// and setControlFields(key, id, ts) (SURVEY.md §2.1 "Tuple contract").
template <typename T>
using key_of_t = typename std::remove_cv<typename std::remove_reference<decltype(std::get<0>(
    std::declval<const T&>().getControlFields()))>::type>::type;

template <typename T>
inline key_of_t<T> key_of(const T& t) {
    return std::get<0>(t.getControlFields());
}
template <typename T>
inline uint64_t id_of(const T& t) {
    return std::get<1>(t.getControlFields());
}
template <typename T>
inline uint64_t ts_of(const T& t) {
    return std::get<2>(t.getControlFields());
}
template <typename T>
inline size_t hash_of(const T& t) {
    return std::hash<key_of_t<T>>()(key_of(t));
}

// ---- fixed-size keys ------------------------------------------------------------------------------------------------------
// Keys of tuples processed by the GPU operators must be trivially copyable: an integer, or a fixed-size value type with
//     WF_HOST_DEVICE uint64_t hash() const;   WF_HOST_DEVICE bool operator==(const K&) const;
// of at most 32 bytes (a multiple of 4).  FixedKey<N> is the ready-made one for what applications usually key by when it is
// not a number - names, UUIDs (the Yahoo benchmark's campaign ids): N bytes, zero-padded, built from a string.  std::hash is
// specialised to the same hash(), so CPU and GPU operators agree on everything that is derived from a key's hash (routing, the
// window a replica of a Win_Farm starts at).  (The reference's GPU programs key by std::string,
// src/mp_test_gpu/mp_common_string.hpp:44-84, and keep the keys on the host; std::string cannot live in HBM.)
WF_HOST_DEVICE inline uint64_t mix64_key(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
template <size_t N>
struct FixedKey {
    static_assert(N % 8 == 0 && N >= 8 && N <= 32, "FixedKey: 8, 16, 24 or 32 bytes");
    uint64_t w[N / 8];
    WF_HOST_DEVICE FixedKey() {
        for (size_t i = 0; i < N / 8; i++) w[i] = 0;
    }
    FixedKey(const char* s) : FixedKey(std::string(s)) {}
    FixedKey(const std::string& s) {
        char bytes[N] = {0};
        for (size_t i = 0; i < N && i < s.size(); i++) bytes[i] = s[i];
        std::memcpy(w, bytes, N);
    }
    std::string str() const {
        char bytes[N + 1];
        std::memcpy(bytes, w, N);
        bytes[N] = 0;
        return std::string(bytes);
    }
    WF_HOST_DEVICE bool operator==(const FixedKey& o) const {
        bool eq = true;
        for (size_t i = 0; i < N / 8; i++) eq = eq && w[i] == o.w[i];
        return eq;
    }
    WF_HOST_DEVICE bool operator!=(const FixedKey& o) const { return !(*this == o); }
    WF_HOST_DEVICE bool operator<(const FixedKey& o) const {
        for (size_t i = 0; i < N / 8; i++)
            if (w[i] != o.w[i]) return w[i] < o.w[i];
        return false;
    }
    WF_HOST_DEVICE uint64_t hash() const {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (size_t i = 0; i < N / 8; i++) h = mix64_key(h ^ w[i]);
        return h;
    }
};

// ---- message tagging -------------------------------------------------------
// Messages between nodes are heap pointers to tuples owned by the receiver.
// Window operators additionally need "end-of-stream markers": a copy of the last
// tuple of a key that only advances window state.  The reference
// wraps such tuples in a ref-counted envelope (meta.hpp:769-872); we tag bit 0 of
// the pointer instead (operator new returns >= 8-byte aligned storage).
template <typename T>
inline void* tag_marker(T* p) {
    return reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(p) | uintptr_t(1));
}
inline bool is_marker(const void* p) { return (reinterpret_cast<uintptr_t>(p) & uintptr_t(1)) != 0; }
template <typename T>
inline T* untag(void* p) {
    return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(1));
}

}  // namespace wf

namespace std {
template <size_t N>
struct hash<wf::FixedKey<N>> {
    size_t operator()(const wf::FixedKey<N>& k) const { return (size_t)k.hash(); }
};
}  // namespace std
