// Keyed (stateful) Map_GPU and Filter_GPU: the user functor receives, besides the
// tuple, a per-KEY scratchpad that persists across batches
//     map    in place : void(tuple_t&, char* state, size_t size)
//     map    out-of-pl: void(const tuple_t&, result_t&, char* state, size_t size)
//     filter          : void(const tuple_t&, bool& keep, char* state, size_t size)
// and the tuples of a key are processed in arrival order.
//
// A stream with few keys leaves such a function little parallelism: it is sequential per key by contract.  When the state
// is a FOLD of the key's history (running sum / count / max / last value ...), keyed_scan(lift, combine, apply) says so:
//     lift    : void(const tuple_t&, state_t&)                       the state of a history of one tuple
//     combine : void(const state_t& older, const state_t& newer, state_t& out)   associative
//     apply   : void(tuple_t&, const state_t& history)               history = every earlier tuple of the key and this one
// and the tuples of a key are then processed as a parallel prefix scan: a wave folds 64 of them per step instead of one
// (MODE 3 below).  No counterpart in the reference.
//
// Capability parity with the keyed variants of the fork's MapGPU_Node /
// FilterGPU_Node (/root/reference/wf/map_gpu_node.hpp:89-125, :372-399, :492-496;
// wf/filter_gpu_node.hpp:61-76) and with the stateful benchmarks
// (src/GPU_Tests/new_tests/benchmarks/gpu_map_stateful.cpp:410-431,
// gpu_filter_stateful.cpp).  The reference kernels make EVERY thread scan the whole
// batch (O(N*threads), K7/K8/K10); the benchmarks instead build per-key linked
// lists on the host while filling the batch and allocate each key's state with its
// own cudaMalloc.  Here everything is on the device: keys map to slots through a
// hash table, a stable radix sort of (slot, index) gives each key a contiguous
// run in arrival order, one thread walks each run, and the states are one slab
// indexed by slot (zero-initialised when a key first appears).
#pragma once
#include "batch_ops.hpp"
#include "../dist/exchange_nodes.hpp"
#include "keyby_gpu.hpp"
#include "kernels_basic.hpp"
#include "primitives.hpp"

namespace wf {
namespace gpu {

// runs of at least this many tuples of one key go to the wave-per-run kernel
constexpr uint32_t KEYED_LONG_RUN = 32;

// the associative form of a keyed map (see the top of this file)
template <typename L, typename C, typename A>
struct KeyedScan {
    using tuple_t = meta::arg_bare_t<L, 0>;
    using state_t = meta::arg_bare_t<L, 1>;
    static_assert(std::is_trivially_copyable_v<state_t> && sizeof(state_t) % 4 == 0, "keyed_scan: the state travels through wave shuffles (32-bit words)");
    L lift;
    C combine;
    A apply;
    // per-key record in the state slab: the state, then a word that says whether the key has a history yet
    static constexpr size_t flag_offset = (sizeof(state_t) + 7) & ~size_t(7);
    static constexpr size_t record_size = flag_offset + 8;
};
template <typename X>
struct is_keyed_scan : std::false_type {};
template <typename L, typename C, typename A>
struct is_keyed_scan<KeyedScan<L, C, A>> : std::true_type {};

// mode 0: in-place map, 1: out-of-place map, 2: filter flags
template <int MODE, typename T, typename R, typename F>
__global__ void __launch_bounds__(BLOCK_THREADS)
    keyed_apply_kernel(const BatchHeader* hdr, T* data, R* out, uint8_t* flags, const uint32_t* sslot, const uint32_t* sidx, char* states,
                       size_t state_size, uint32_t* table_full, uint32_t* long_heads, uint32_t* n_long, F f) {
    const uint32_t n = hdr->count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = sslot[i];
        if (i > 0 && sslot[i - 1] == slot) continue;
        if (slot == 0xffffffffu) {
            // tuples without a slot (table full / reserved key) are not processed: the operator fails at its next check;
            // until then their outputs are defined (zeroed results, tuples dropped by a filter) rather than garbage
            uint32_t cnt = 0;
            for (uint32_t j = i; j < n && sslot[j] == slot; j++, cnt++) {
                if constexpr (MODE == 1) out[sidx[j]] = R{};
                if constexpr (MODE == 2) flags[sidx[j]] = 0;
            }
            atomicAdd(table_full, cnt);
            continue;
        }
        if (i + KEYED_LONG_RUN - 1 < n && sslot[i + KEYED_LONG_RUN - 1] == slot) {  // left to the wave-per-run kernel
            long_heads[atomicAdd(n_long, 1u)] = i;
            continue;
        }
        char* st = states + (size_t)slot * state_size;
        if constexpr (MODE == 3) {
            using S = typename F::state_t;
            S acc{};
            bool has = *reinterpret_cast<const uint32_t*>(st + F::flag_offset) != 0;
            if (has) acc = *reinterpret_cast<const S*>(st);
            for (uint32_t j = i; j < n && sslot[j] == slot; j++) {
                const uint32_t p = sidx[j];
                T t = data[p];
                S x{};
                f.lift(static_cast<const T&>(t), x);
                if (has) {
                    S tmp{};
                    f.combine(static_cast<const S&>(acc), static_cast<const S&>(x), tmp);
                    acc = tmp;
                } else {
                    acc = x;
                    has = true;
                }
                f.apply(t, static_cast<const S&>(acc));
                data[p] = t;
            }
            *reinterpret_cast<S*>(st) = acc;
            *reinterpret_cast<uint32_t*>(st + F::flag_offset) = 1u;
            continue;
        }
        for (uint32_t j = i; j < n && sslot[j] == slot; j++) {
            const uint32_t p = sidx[j];
            if constexpr (MODE == 3) {
            } else if constexpr (MODE == 0) {
                T t = data[p];
                f(t, st, state_size);
                data[p] = t;
            } else if constexpr (MODE == 1) {
                const T t = data[p];
                R r{};
                f(t, r, st, state_size);
                out[p] = r;
            } else {
                const T t = data[p];
                bool keep = false;
                f(t, keep, st, state_size);
                flags[p] = keep ? 1 : 0;
            }
        }
    }
}

// lane 0 of a wave applies the user function to the `cnt` staged tuples of a chunk, in order and in place
// (the function reads and writes the staged tuple / result directly: no private copies)
template <int MODE, typename T, typename R, typename F>
__device__ __forceinline__ void keyed_apply_chunk(T* stage, R* rstage, uint8_t* fstage, uint32_t cnt, char* st, size_t state_size, F& f) {
    for (uint32_t u = 0; u < cnt; u++) {
        if constexpr (MODE == 0) {
            f(stage[u], st, state_size);
        } else if constexpr (MODE == 1) {
            rstage[u] = R{};
            f(static_cast<const T&>(stage[u]), rstage[u], st, state_size);
        } else {
            bool keep = false;
            f(static_cast<const T&>(stage[u]), keep, st, state_size);
            fstage[u] = keep ? 1 : 0;
        }
    }
}

// Long runs (streams with few keys): one wave64 per run.  The 64 lanes gather the next 64 tuples of
// the run together (and the chunk after that while lane 0 works), the chunk is staged in LDS and, when
// it fits, so is the key's state: the user function is sequential per key by contract, but its
// loads and stores then hit LDS instead of paying an HBM/L2 round trip per tuple.  Results are written
// back by all lanes.  Dynamic LDS per wave: 64 tuples (+ 64 results, MODE 1) + `lds_state` bytes.
template <int MODE, typename T, typename R, typename F>
__global__ void __launch_bounds__(BLOCK_THREADS)
    keyed_apply_long_kernel(const BatchHeader* hdr, T* data, R* out, uint8_t* flags, const uint32_t* sslot, const uint32_t* sidx, char* states,
                            size_t state_size, const uint32_t* long_heads, const uint32_t* n_long, uint32_t lds_state, size_t wave_bytes, F f) {
    extern __shared__ __align__(16) char keyed_smem[];
    const uint32_t n = hdr->count;
    const uint32_t runs = *n_long;
    const int lane = lane_id();
    char* my = keyed_smem + (size_t)wave_id() * wave_bytes;
    T* stage = reinterpret_cast<T*>(my);
    size_t off = (WAVE * sizeof(T) + 15) & ~size_t(15);
    R* rstage = reinterpret_cast<R*>(my + off);
    if (MODE == 1) off += (WAVE * sizeof(R) + 15) & ~size_t(15);
    uint8_t* fstage = reinterpret_cast<uint8_t*>(my + off);
    if (MODE == 2) off += WAVE;
    off = (off + 15) & ~size_t(15);
    char* lstate = my + off;
    const uint32_t waves_total = gridDim.x * (BLOCK_THREADS / WAVE);
    for (uint32_t r = blockIdx.x * (BLOCK_THREADS / WAVE) + wave_id(); r < runs; r += waves_total) {
        const uint32_t head = long_heads[r];
        const uint32_t slot = sslot[head];
        char* gstate = states + (size_t)slot * state_size;
        if (lds_state != 0) {
            // state_size is a multiple of 4 when staged (checked on the host)
            for (uint32_t k = lane; k < state_size / 4; k += WAVE) reinterpret_cast<uint32_t*>(lstate)[k] = reinterpret_cast<const uint32_t*>(gstate)[k];
        }
        uint32_t j = head + lane;
        bool ok = (j < n) && (sslot[j] == slot);
        uint32_t p = ok ? sidx[j] : 0;
        TupleWords<T> reg;
        if (ok) reg.load(data + p);
        while (true) {
            const uint32_t cnt = (uint32_t)__popcll(__ballot(ok));  // valid lanes are a prefix: the arrays are sorted
            if (ok) reg.store(stage + lane);
            const uint32_t my_p = p;
            const bool my_ok = ok;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // next chunk's loads go out now and complete while lane 0 works
            ok = false;
            if (cnt == WAVE) {
                j += WAVE;
                ok = (j < n) && (sslot[j] == slot);
                p = ok ? sidx[j] : 0;
                if (ok) reg.load(data + p);
            }
            if (lane == 0) {
                // two copies of the loop so that the compiler knows the address space of the state in each
                // (LDS instructions instead of flat ones when the state is staged)
                if (lds_state != 0)
                    keyed_apply_chunk<MODE, T, R, F>(stage, rstage, fstage, cnt, lstate, state_size, f);
                else
                    keyed_apply_chunk<MODE, T, R, F>(stage, rstage, fstage, cnt, gstate, state_size, f);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (my_ok) {
                if constexpr (MODE == 0)
                    data[my_p] = stage[lane];
                else if constexpr (MODE == 1)
                    out[my_p] = rstage[lane];
                else
                    flags[my_p] = fstage[lane];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (cnt < WAVE) break;
        }
        if (lds_state != 0) {
            for (uint32_t k = lane; k < state_size / 4; k += WAVE) reinterpret_cast<uint32_t*>(gstate)[k] = reinterpret_cast<const uint32_t*>(lstate)[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// the value of `v` held by lane `src` / by lane (lane - d): any trivially copyable struct made of 32-bit words
template <typename X>
__device__ __forceinline__ X keyed_shfl(const X& v, int src) {
    int w[sizeof(X) / 4];
    __builtin_memcpy(w, &v, sizeof(X));
#pragma unroll
    for (size_t k = 0; k < sizeof(X) / 4; k++) w[k] = __shfl(w[k], src, WAVE);
    X o;
    __builtin_memcpy(&o, w, sizeof(X));
    return o;
}
template <typename X>
__device__ __forceinline__ X keyed_shfl_up(const X& v, int d) {
    int w[sizeof(X) / 4];
    __builtin_memcpy(w, &v, sizeof(X));
#pragma unroll
    for (size_t k = 0; k < sizeof(X) / 4; k++) w[k] = __shfl_up(w[k], d, WAVE);
    X o;
    __builtin_memcpy(&o, w, sizeof(X));
    return o;
}

// Long runs of the associative form: one wave64 per run, 64 tuples per step.  The lanes lift their tuples, an ordered
// inclusive wave scan (6 shuffle + combine steps) gives every lane the fold of the chunk up to its tuple, the history of
// the key before the chunk is combined in front, every lane applies its own tuple.
template <typename T, typename F>
__global__ void __launch_bounds__(BLOCK_THREADS)
    keyed_scan_long_kernel(const BatchHeader* hdr, T* data, const uint32_t* sslot, const uint32_t* sidx, char* states, size_t state_size,
                           const uint32_t* long_heads, const uint32_t* n_long, F f) {
    using S = typename F::state_t;
    const uint32_t n = hdr->count;
    const uint32_t runs = *n_long;
    const int lane = lane_id();
    const uint32_t waves_total = gridDim.x * (BLOCK_THREADS / WAVE);
    for (uint32_t r = blockIdx.x * (BLOCK_THREADS / WAVE) + wave_id(); r < runs; r += waves_total) {
        const uint32_t head = long_heads[r];
        const uint32_t slot = sslot[head];
        char* st = states + (size_t)slot * state_size;
        S carry{};
        bool has = *reinterpret_cast<const uint32_t*>(st + F::flag_offset) != 0;  // identical in every lane
        if (has) carry = *reinterpret_cast<const S*>(st);
        uint32_t j = head + (uint32_t)lane;
        bool ok = (j < n) && (sslot[j] == slot);
        uint32_t p = ok ? sidx[j] : 0;
        T t{};
        if (ok) t = data[p];
        while (true) {
            const uint32_t cnt = (uint32_t)__popcll(__ballot(ok));  // the valid lanes are a prefix: the arrays are sorted
            if (cnt == 0) break;  // the run ended exactly on the previous chunk
            S val{};
            if (ok) f.lift(static_cast<const T&>(t), val);
            // the next chunk's loads go out now and complete while this one is scanned
            T nt{};
            bool nok = false;
            uint32_t np = 0;
            if (cnt == WAVE) {
                j += WAVE;
                nok = (j < n) && (sslot[j] == slot);
                np = nok ? sidx[j] : 0;
                if (nok) nt = data[np];
            }
            for (int d = 1; d < WAVE; d <<= 1) {
                const S other = keyed_shfl_up(val, d);
                if (ok && lane >= d) {
                    S tmp{};
                    f.combine(other, static_cast<const S&>(val), tmp);
                    val = tmp;
                }
            }
            if (ok) {
                if (has) {
                    S tmp{};
                    f.combine(static_cast<const S&>(carry), static_cast<const S&>(val), tmp);
                    val = tmp;
                }
                f.apply(t, static_cast<const S&>(val));
                data[p] = t;
            }
            carry = keyed_shfl(val, (int)cnt - 1);
            has = true;
            if (cnt < WAVE) break;
            t = nt;
            ok = nok;
            p = np;
        }
        if (lane == 0) {
            *reinterpret_cast<S*>(st) = carry;
            *reinterpret_cast<uint32_t*>(st + F::flag_offset) = 1u;
        }
    }
}

// device-resident keyed context: key table + sorter + state slab
template <typename K>
class KeyedContext {
public:
    KeyedContext(size_t max_keys, size_t state_size, size_t batch_capacity, int device) : state_size_(state_size ? state_size : 1) {
        WF_HIP_CHECK(hipSetDevice(device));
        lds_budget_ = max_lds_per_block(device) - 4096;  // 160 KB per workgroup on gfx950, less some slack
        // never the legacy (null) stream: its implicit dependencies would break stream captures running in other threads
        Stream init;
        table_cap_ = 2;
        while (table_cap_ < 2 * max_keys) table_cap_ <<= 1;
        while ((1ull << slot_bits_) < table_cap_) slot_bits_++;
        table_.init(table_cap_, init.get());
        states_.resize(table_cap_ * state_size_);
        WF_HIP_CHECK(hipMemsetAsync(states_.data(), 0, states_.bytes(), init.get()));
        slot_.resize(batch_capacity);
        idx_.resize(batch_capacity);
        full_.resize(1);
        WF_HIP_CHECK(hipMemsetAsync(full_.data(), 0, 4, init.get()));
        long_heads_.resize(batch_capacity / KEYED_LONG_RUN + 2);
        n_long_.resize(1);
        WF_HIP_CHECK(hipMemsetAsync(n_long_.data(), 0, 4, init.get()));
        // the operator's kernels run on another stream: wait for the initialisation here
        init.sync();
        sorter_.reserve(batch_capacity);
    }
    template <typename T>
    RadixSorter::Sorted group(Batch* in, hipStream_t s, uint64_t& kernels) {
        const size_t upper = std::max<size_t>(1, in->count_on_host ? in->host_hdr.count : in->capacity);
        hipLaunchKernelGGL((assign_slots_kernel<T, K>), dim3(grid_for(upper)), dim3(BLOCK_THREADS), 0, s, in->header(),
                           in->template data<T>(), table_.view(), slot_.data(), idx_.data(), full_.data(),
                           (BatchHeader*)nullptr, n_long_.data());  // also resets the long-run counter of the batch
        kernels += 1 + sorter_.kernels_per_sort(slot_bits_ + 1);
        return sorter_.sort(slot_.data(), idx_.data(), in->device_count_ptr(), upper, slot_bits_ + 1, s);
    }
    char* states() const { return states_.data(); }
    uint32_t* long_heads() const { return long_heads_.data(); }
    uint32_t* n_long() const { return n_long_.data(); }
    // dynamic LDS of the long-run kernel for tuples of `tuple_bytes` (+ `extra` staged bytes per tuple):
    // per wave {bytes, staged state bytes (0 = the state stays in global memory)}
    std::pair<size_t, uint32_t> long_layout(size_t tuple_bytes, size_t extra_per_tuple) const {
        size_t base = ((WAVE * tuple_bytes + 15) & ~size_t(15)) + ((WAVE * extra_per_tuple + 15) & ~size_t(15));
        const size_t budget = lds_budget_ / (BLOCK_THREADS / WAVE);
        uint32_t st = 0;
        if (state_size_ % 4 == 0 && base + state_size_ + 16 <= budget) st = (uint32_t)state_size_;
        return {base + ((st + 15) & ~size_t(15)), st};
    }
    size_t state_size() const { return state_size_; }
    size_t lds_budget() const { return lds_budget_; }
    uint32_t* table_full_counter() const { return full_.data(); }
    uint32_t read_table_full(hipStream_t s) {
        uint32_t v = 0;
        WF_HIP_CHECK(hipMemcpyAsync(&v, full_.data(), 4, hipMemcpyDeviceToHost, s));
        WF_HIP_CHECK(hipStreamSynchronize(s));
        return v;
    }

private:
    size_t state_size_, table_cap_ = 0, lds_budget_ = 60 * 1024;
    int slot_bits_ = 0;
    KeyTable<K> table_;
    DeviceBuffer<char> states_;
    DeviceBuffer<uint32_t> slot_, idx_, full_, long_heads_, n_long_;
    RadixSorter sorter_;
};

}  // namespace gpu

struct KeyedGPUConfig {
    size_t max_keys = 1 << 16;     // distinct keys per replica
    size_t state_size = 64;        // bytes of per-key scratchpad (reference default: wf/map_gpu.hpp:141)
    bool lanes_are_key_disjoint = false;  // batch input: lane i already holds only replica i's keys
};

// MODE 0/1: map (in place / out of place), MODE 2: filter
template <int MODE, typename tuple_t, typename result_t, typename F>
class KeyedGPU_Node : public GPU_Replica<tuple_t> {
public:
    KeyedGPU_Node(F func, KeyedGPUConfig kcfg, GPUConfig cfg, edge_kind_t in_kind, std::string name, size_t par, size_t idx,
                  meta::closing_func_t closing, std::shared_ptr<StatsHub> hub)
        : GPU_Replica<tuple_t>(cfg, in_kind, std::move(name), par, idx, std::move(closing), std::move(hub)), func_(func), kcfg_(kcfg) {}

protected:
    void process_device(gpu::Batch* d) override {
        hipStream_t s = this->stream();
        if (!ctx_) {
            ctx_.reset(new gpu::KeyedContext<key_of_t<tuple_t>>(kcfg_.max_keys, kcfg_.state_size, d->capacity, this->cfg_.gpu_id));
            batch_cap_ = d->capacity;
        }
        if (d->capacity > batch_cap_) gpu::failwith("keyed GPU operator: a batch larger than the first one received");
        const size_t upper = std::max<size_t>(1, d->count_on_host ? d->host_hdr.count : d->capacity);
        const auto sorted = ctx_->template group<tuple_t>(d, s, this->num_kernels_);
        const unsigned grid = gpu::grid_for(upper);
        if (++since_check_ >= CHECK_EVERY) {
            since_check_ = 0;
            check_table();
        }
        if constexpr (MODE == 0 || MODE == 3) {
            launch_apply<tuple_t>(d, (tuple_t*)nullptr, (uint8_t*)nullptr, sorted, upper, grid, s);
            this->emit_device(d);
        } else if constexpr (MODE == 1) {
            if (!out_pool_)
                out_pool_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, this->cfg_.gpu_id, sizeof(result_t), d->capacity, this->cfg_.pool_slabs));
            gpu::Batch* o = out_pool_->acquire(s);
            WF_HIP_CHECK(hipMemcpyAsync(o->header(), d->header(), sizeof(gpu::BatchHeader), hipMemcpyDeviceToDevice, s));
            launch_apply<result_t>(d, o->template data<result_t>(), (uint8_t*)nullptr, sorted, upper, grid, s);
            o->host_hdr = d->host_hdr;
            o->count_on_host = d->count_on_host;
            if (!d->count_on_host)
                WF_HIP_CHECK(hipMemcpyAsync((void*)o->count_mirror, d->device_count_ptr(), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            d->release(s);
            this->emit_device(o);
        } else {
            if (!out_pool_)
                out_pool_.reset(new gpu::BatchPool(gpu::MemKind::DEVICE, this->cfg_.gpu_id, sizeof(tuple_t), d->capacity, this->cfg_.pool_slabs));
            flags_.resize(d->capacity);
            ws_.resize(gpu::filter_workspace_words(d->capacity));
            launch_apply<tuple_t>(d, (tuple_t*)nullptr, flags_.data(), sorted, upper, grid, s);
            WF_HIP_CHECK(hipMemsetAsync(ws_.data(), 0, gpu::filter_workspace_words(upper) * sizeof(unsigned long long), s));
            gpu::Batch* o = out_pool_->acquire(s);
            const unsigned cgrid = (unsigned)std::max<size_t>(1, (upper + gpu::FILTER_TILE - 1) / gpu::FILTER_TILE);
            hipLaunchKernelGGL((gpu::filter_compact_kernel<tuple_t, gpu::FlagPredicate<tuple_t>>), dim3(cgrid), dim3(gpu::BLOCK_THREADS), 0, s,
                               d->header(), d->template data<tuple_t>(), o->header(), o->template data<tuple_t>(), o->count_mirror, ws_.data(),
                               gpu::FlagPredicate<tuple_t>{flags_.data()});
            this->num_kernels_ += 1;
            o->host_hdr = d->host_hdr;
            o->count_on_host = false;
            d->release(s);
            this->emit_device(o);
        }
    }
    void gpu_flush() override { check_table(); }
    // an unbounded stream never reaches gpu_flush(): the counter is also read every CHECK_EVERY batches
    void check_table() {
        if (ctx_ && ctx_->read_table_full(this->stream()) > 0)
            fatal_error("keyed GPU operator: tuples without a state slot - the device key table is full (raise withMaxKeys()) or a tuple "
                        "carries the reserved key 2^64-1");
    }
    static constexpr uint32_t CHECK_EVERY = 256;
    uint32_t since_check_ = 0;

private:
    // short runs: a thread per run; long runs: a wave per run (recorded by the first kernel)
    template <typename RR>
    void launch_apply(gpu::Batch* d, RR* out, uint8_t* flags, const gpu::RadixSorter::Sorted& sorted, size_t upper, unsigned grid, hipStream_t s) {
        hipLaunchKernelGGL((gpu::keyed_apply_kernel<MODE, tuple_t, RR, F>), dim3(grid), dim3(gpu::BLOCK_THREADS), 0, s, d->header(),
                           d->template data<tuple_t>(), out, flags, sorted.keys, sorted.vals, ctx_->states(), ctx_->state_size(),
                           ctx_->table_full_counter(), ctx_->long_heads(), ctx_->n_long(), func_);
        if constexpr (MODE == 3) {
            const unsigned sgrid =
                (unsigned)std::min<size_t>(512, std::max<size_t>(1, upper / gpu::KEYED_LONG_RUN / (gpu::BLOCK_THREADS / gpu::WAVE) + 1));
            hipLaunchKernelGGL((gpu::keyed_scan_long_kernel<tuple_t, F>), dim3(sgrid), dim3(gpu::BLOCK_THREADS), 0, s, d->header(),
                               d->template data<tuple_t>(), sorted.keys, sorted.vals, ctx_->states(), ctx_->state_size(), ctx_->long_heads(),
                               ctx_->n_long(), func_);
        } else {
            const auto lay = ctx_->long_layout(sizeof(tuple_t), MODE == 1 ? sizeof(RR) : (MODE == 2 ? 1 : 0));
            const size_t lds = (gpu::BLOCK_THREADS / gpu::WAVE) * lay.first;
            if (lds > ctx_->lds_budget()) gpu::failwith("keyed GPU operator: tuple type too large for the LDS staging buffers");
            gpu::allow_dynamic_lds(&gpu::keyed_apply_long_kernel<MODE, tuple_t, RR, F>, lds);
            const unsigned lgrid =
                (unsigned)std::min<size_t>(512, std::max<size_t>(1, upper / gpu::KEYED_LONG_RUN / (gpu::BLOCK_THREADS / gpu::WAVE) + 1));
            hipLaunchKernelGGL((gpu::keyed_apply_long_kernel<MODE, tuple_t, RR, F>), dim3(lgrid), dim3(gpu::BLOCK_THREADS), lds, s, d->header(),
                               d->template data<tuple_t>(), out, flags, sorted.keys, sorted.vals, ctx_->states(), ctx_->state_size(),
                               ctx_->long_heads(), ctx_->n_long(), lay.second, lay.first, func_);
        }
        this->num_kernels_ += 2;
    }

    F func_;
    KeyedGPUConfig kcfg_;
    size_t batch_cap_ = 0;
    std::unique_ptr<gpu::KeyedContext<key_of_t<tuple_t>>> ctx_;
    std::unique_ptr<gpu::BatchPool> out_pool_;
    gpu::DeviceBuffer<uint8_t> flags_;
    gpu::DeviceBuffer<unsigned long long> ws_;
};

// Keyed operator: tuple input is routed by key on the host; batch input goes through
// the device KeyBy emitter unless the lanes are declared key-disjoint.
template <int MODE, typename tuple_t, typename result_t>
class Keyed_GPU_Operator : public Basic_Operator {
public:
    template <typename F>
    Keyed_GPU_Operator(F func, size_t parallelism, std::string name, KeyedGPUConfig kcfg, GPUConfig cfg, meta::closing_func_t closing,
                       routing_func_t routing)
        : name_(std::move(name)), par_(parallelism), kcfg_(kcfg), cfg_(cfg), routing_(std::move(routing)),
          kind_(std::make_shared<edge_kind_t>(EDGE_TUPLE)) {
        if (par_ == 0) fatal_error("operator " + name_ + " has parallelism zero");
        auto hub = hub_;
        const std::string n = name_;
        const size_t N = par_;
        factory_ = [=](rt::Graph& g, edge_kind_t node_kind) {
            std::vector<rt::Node*> reps;
            for (size_t i = 0; i < N; i++)
                reps.push_back(g.adopt(new KeyedGPU_Node<MODE, tuple_t, result_t, F>(func, kcfg, cfg, node_kind, n, N, i, closing, hub)));
            return reps;
        };
    }
    std::string getName() const override { return name_; }
    size_t getParallelism() const override { return par_; }
    routing_modes_t getRoutingMode() const override { return KEYBY; }
    bool isGPUOperator() const override { return true; }
    const std::type_info& input_type() const override { return typeid(tuple_t); }
    const std::type_info& output_type() const override { return typeid(result_t); }
    edge_kind_t output_kind() const override { return EDGE_DEVICE_BATCH; }
    bool accepts_input_kind(edge_kind_t) const override { return true; }
    void set_input_kind(edge_kind_t k) override { *kind_ = k; }
    std::function<rt::Node*(edge_kind_t)> output_adapter_factory() const override { return batch_adapter_factory<result_t>(cfg_.pool_slabs); }

    bool merges_batch_lanes_in_order() const override { return !kcfg_.lanes_are_key_disjoint; }
    bool takes_batch_lanes_one_to_one() const override { return kcfg_.lanes_are_key_disjoint; }
    // tuple producers fill batches themselves; the device KeyBy then replaces the host routing (a custom routing function only
    // applies to the tuple path: WF_TUPLE_PATH=1)
    std::function<rt::Node*(size_t, size_t)> input_batcher_factory(Mode) const override {
        return kcfg_.lanes_are_key_disjoint ? nullptr : batcher_factory<tuple_t>(cfg_);
    }
    std::vector<StageSpec> build_stages(rt::Graph& g, Mode mode) override {
        StageSpec s;
        s.name = name_;
        const size_t N = par_;
        const edge_kind_t kind = *kind_;
        if (kind == EDGE_TUPLE) {
            s.routing = KEYBY;
            s.replicas = factory_(g, EDGE_TUPLE);
            routing_func_t rf = routing_;
            s.make_emitter = [rf, N](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<tuple_t>(rf, N), N); };
            s.make_ordering = [](Mode m) -> rt::Node* { return m == Mode::DETERMINISTIC ? new Ordering_Node<tuple_t>(TS) : nullptr; };
        } else if (kcfg_.lanes_are_key_disjoint) {
            s.routing = FORWARD;
            s.replicas = factory_(g, kind);
            s.make_emitter = [N](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<tuple_t>(N), N); };
        } else {
            // device KeyBy (dist/exchange_nodes.hpp): producers pack their batches by replica, one endpoint merges them per replica
            std::vector<StageSpec> stages = device_keyby_stages<tuple_t>(g, name_, cfg_, N, kind, hub_, mode == Mode::DETERMINISTIC);
            s.routing = FORWARD;
            s.replicas = factory_(g, EDGE_DEVICE_BATCH);
            s.make_emitter = [N](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<tuple_t>(N), N); };
            stages.push_back(s);
            return stages;
        }
        return {s};
    }

private:
    std::string name_;
    size_t par_;
    KeyedGPUConfig kcfg_;
    GPUConfig cfg_;
    routing_func_t routing_;
    std::shared_ptr<edge_kind_t> kind_;
    std::function<std::vector<rt::Node*>(rt::Graph&, edge_kind_t)> factory_;
};

}  // namespace wf
