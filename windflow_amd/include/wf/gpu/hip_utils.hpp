// HIP substrate: error checking that aborts, RAII device / pinned buffers,
// streams, events, and the stream-ordered batch pools used for host<->device
// batch shipping.
//
// Capability parity with /root/reference/wf/gpu_utils.hpp:43-312 (GPUBuffer,
// PinnedCPUBuffer, GPUStream, failwith) and with the double-buffered pinned
// shipping + device-buffer recycling of the reference benchmarks
// (src/GPU_Tests/new_tests/benchmarks/gpu_map_stateful.cpp:150-349).  Design is
// ours: batches are [64-byte header | payload] slabs so one hipMemcpyAsync ships
// count and tuples together; a slab returns to its pool with an event recorded
// on the last stream that touched it, and the next user waits on that event
// (stream-ordered recycling: the host thread never blocks in steady state).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace wf {
namespace gpu {

// the reference's gpuErrChk prints and keeps going (basic.hpp:104-118); we abort
#define WF_HIP_CHECK(expr)                                                                                        \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess) {                                                                                   \
            std::fprintf(stderr, "WindFlow HIP error: %s (%d) at %s:%d in %s\n", hipGetErrorString(_e), (int)_e, \
                         __FILE__, __LINE__, #expr);                                                              \
            std::abort();                                                                                         \
        }                                                                                                         \
    } while (0)

[[noreturn]] inline void failwith(const std::string& msg) {
    std::fprintf(stderr, "WindFlow GPU error: %s\n", msg.c_str());
    std::abort();
}

// LDS a workgroup may allocate on `device`: asked of the device, never assumed (gfx950: 160 KB per CU; earlier CDNA: 64 KB).
// WF_LDS_LIMIT (bytes) lowers it, e.g. to reproduce the 64 KB tilings of round 2.
inline size_t max_lds_per_block(int device) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || v <= 0) v = 64 * 1024;
    size_t lim = (size_t)v;
    if (const char* e = std::getenv("WF_LDS_LIMIT")) {
        const size_t x = (size_t)std::strtoull(e, nullptr, 10);
        if (x >= 16 * 1024 && x < lim) lim = x;
    }
    return lim;
}
// kernels that ask for more dynamic LDS than the 64 KB of older parts say so once (a no-op where the runtime does not need it)
template <typename K>
inline void allow_dynamic_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

inline int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- RAII device buffer (grow-only resize, like the reference's GPUBuffer) ----
template <typename T>
class DeviceBuffer {
public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t n) { resize(n); }
    ~DeviceBuffer() { release(); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    DeviceBuffer(DeviceBuffer&& o) noexcept : ptr_(o.ptr_), n_(o.n_) {
        o.ptr_ = nullptr;
        o.n_ = 0;
    }
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
        if (this != &o) {
            release();
            ptr_ = o.ptr_;
            n_ = o.n_;
            o.ptr_ = nullptr;
            o.n_ = 0;
        }
        return *this;
    }
    // grow to at least n elements (contents are NOT preserved)
    void resize(size_t n) {
        if (n <= n_) return;
        release();
        WF_HIP_CHECK(hipMalloc((void**)&ptr_, n * sizeof(T)));
        n_ = n;
    }
    void zero(hipStream_t s = nullptr) {
        if (ptr_) WF_HIP_CHECK(hipMemsetAsync(ptr_, 0, n_ * sizeof(T), s));
    }
    T* data() const { return ptr_; }
    size_t size() const { return n_; }
    size_t bytes() const { return n_ * sizeof(T); }

private:
    void release() {
        if (ptr_) (void)hipFree(ptr_);
        ptr_ = nullptr;
        n_ = 0;
    }
    T* ptr_ = nullptr;
    size_t n_ = 0;
};

// ---- RAII pinned host buffer ----------------------------------------------
template <typename T>
class PinnedBuffer {
public:
    PinnedBuffer() = default;
    explicit PinnedBuffer(size_t n) { resize(n); }
    ~PinnedBuffer() { release(); }
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    void resize(size_t n) {
        if (n <= n_) return;
        release();
        WF_HIP_CHECK(hipHostMalloc((void**)&ptr_, n * sizeof(T), hipHostMallocDefault));
        n_ = n;
    }
    T* data() const { return ptr_; }
    T& operator[](size_t i) { return ptr_[i]; }
    const T& operator[](size_t i) const { return ptr_[i]; }
    size_t size() const { return n_; }

private:
    void release() {
        if (ptr_) (void)hipHostFree(ptr_);
        ptr_ = nullptr;
        n_ = 0;
    }
    T* ptr_ = nullptr;
    size_t n_ = 0;
};

// ---- stream / event ---------------------------------------------------------
class Stream {
public:
    Stream() { WF_HIP_CHECK(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking)); }
    ~Stream() {
        if (s_) (void)hipStreamDestroy(s_);
    }
    Stream(const Stream&) = delete;
    Stream& operator=(const Stream&) = delete;
    hipStream_t get() const { return s_; }
    operator hipStream_t() const { return s_; }
    void sync() const { WF_HIP_CHECK(hipStreamSynchronize(s_)); }

private:
    hipStream_t s_ = nullptr;
};

// HIP loads a binary's device code lazily, at the first launch, and its own copy / fill kernels at the first
// hipMemcpy / hipMemset.  The lanes of a pipeline start together, so those first uses would race each other - which
// is harmless for HIP itself but crashes inside rocprofv3's code-object tracking (seen with ROCm 7.2: SIGSEGV in
// librocprofiler-sdk under the first hipLaunchKernel of several threads).  Every GPU node therefore calls
// preload_module(device) from its constructor (the graph is built by one thread): one probe launch, one fill and
// one copy per device, serialised by a mutex, before any lane thread exists.
__global__ void wf_probe_kernel(uint32_t* p) {
    if (p != nullptr && threadIdx.x == 0) *p += 1u;
}

inline void preload_module(int device) {
    static std::mutex m;
    static std::vector<char> done;
    std::lock_guard<std::mutex> l(m);
    int n_dev = 0;
    if (device < 0 || hipGetDeviceCount(&n_dev) != hipSuccess || device >= n_dev) return;  // reported where the device is first used
    if ((size_t)device >= done.size()) done.resize((size_t)device + 1, 0);
    if (done[(size_t)device]) return;
    (void)hipGetLastError();  // errors of earlier, unrelated calls are not ours to report
    int prev = 0;
    WF_HIP_CHECK(hipGetDevice(&prev));
    WF_HIP_CHECK(hipSetDevice(device));
    hipStream_t s = nullptr;
    WF_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t* buf = nullptr;
    uint32_t* host = nullptr;
    WF_HIP_CHECK(hipMalloc((void**)&buf, 2 * 256));
    WF_HIP_CHECK(hipHostMalloc((void**)&host, 256, hipHostMallocDefault));
    WF_HIP_CHECK(hipMemsetAsync(buf, 0, 2 * 256, s));
    hipLaunchKernelGGL(wf_probe_kernel, dim3(1), dim3(64), 0, s, buf);
    WF_HIP_CHECK(hipGetLastError());
    WF_HIP_CHECK(hipMemcpyAsync(buf + 64, buf, 256, hipMemcpyDeviceToDevice, s));
    WF_HIP_CHECK(hipMemcpyAsync(host, buf, 256, hipMemcpyDeviceToHost, s));
    WF_HIP_CHECK(hipMemcpyAsync(buf, host, 256, hipMemcpyHostToDevice, s));
    WF_HIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(buf);
    (void)hipHostFree(host);
    (void)hipStreamDestroy(s);
    WF_HIP_CHECK(hipSetDevice(prev));
    done[(size_t)device] = 1;
}

// Streams and hardware queues.  The streams of a process are spread over a few hardware queues (4 by default,
// GPU_MAX_HW_QUEUES); streams that share a queue run their kernels one after the other, and a stream that waits
// ON THE DEVICE for an event of another stream parks a barrier packet that holds up every other stream of its
// queue until the event fires.  The runtime therefore waits on the HOST wherever a lane depends on another
// stream (a recycled slab's last use, the producer of a batch to download): the calling thread has nothing
// else to enqueue anyway, and the queues only ever hold work that can run.  Measured on MI355X
// (apps/bench/ab_sync.sh, profiles/r1_sync_policy): stateful Map_GPU at 1024 keys 0.59-0.81 -> 0.90-0.97 G
// tuples/s, flagship at 64 keys 0.57 -> 0.70 G, at 1 Mi keys 1.14-1.21 -> 1.20-1.26 G.
//
// Optional copy streams shared by all lanes of a device (slot 0: device->host adapters with WF_D2H_STREAMS=shared,
// slots 1..: uploads with WF_UPLOAD_STREAMS=n): one copy at a time in FIFO order instead of one per lane.  They
// measured SLOWER than per-lane streams on this hardware and are off by default.  Never destroyed: nodes of any
// thread may hold them until the process ends.
inline hipStream_t shared_copy_stream(int device, int slot = 0) {
    constexpr int SLOTS = 8;
    static std::mutex m;
    static std::vector<hipStream_t> streams;
    std::lock_guard<std::mutex> l(m);
    const size_t at = (size_t)device * SLOTS + (size_t)(slot % SLOTS);
    if (at >= streams.size()) streams.resize((size_t)(device + 1) * SLOTS, nullptr);
    if (!streams[at]) {
        WF_HIP_CHECK(hipSetDevice(device));
        WF_HIP_CHECK(hipStreamCreateWithFlags(&streams[at], hipStreamNonBlocking));
    }
    return streams[at];
}

// WF_UPLOAD_STREAMS=n (1..7): uploads go through n shared streams per device and the kernels are enqueued when
// the copy is known to be complete; 0: uploads go through each lane's compute stream.  Unset: by the operator's parallelism -
// with many replicas the lanes' streams overlap each other's copies and kernels best (the flagship loses 10-20 % with upload
// streams); with one or two replicas nothing else would overlap a lane's copy with its own kernels (single-key windows:
// 0.54 -> 0.78 G tuples/s, profiles/r2_win_seq)
inline size_t upload_streams(size_t parallelism) {
    static const long env = [] {
        const char* e = std::getenv("WF_UPLOAD_STREAMS");
        const long v = e ? std::atol(e) : -1;
        return v > 7 ? 7 : v;
    }();
    if (env >= 0) return (size_t)env;
    return parallelism <= 2 ? parallelism : 0;
}
// WF_POOL_WAIT=device: a recycled slab's last use is waited for on the device (stream-ordered) instead of the host
inline bool pool_wait_on_host() {
    static const bool h = [] {
        const char* e = std::getenv("WF_POOL_WAIT");
        return !(e && std::string(e) == "device");
    }();
    return h;
}
// WF_COALESCE_TUPLES=n: a GPU replica that finds several small host batches queued uploads them into ONE device batch of at
// most n tuples (default 131072, times the number of its producers up to 4; 0: never) - the launches of a batch cost the same for 16 Ki tuples as for 128 Ki, so a
// replica that has fallen behind catches up with fewer, larger batches, and one that keeps up sees single batches
inline size_t coalesce_tuples() {
    static const size_t v = [] {
        const char* e = std::getenv("WF_COALESCE_TUPLES");
        return e ? (size_t)std::atoll(e) : (size_t)131072;
    }();
    return v;
}
// WF_COALESCE_SHARE=d: a replica with ONE producer groups at most 1/d of the slabs that producer keeps in flight (default 2;
// 1: all of them) unless the group is small anyway (32 Ki tuples): a group that holds every slab of its producer leaves it
// nothing to fill meanwhile, and the tail latency pays for the burst (16 Ki batches, 4 in flight: p99 2.6 -> 1.4 ms at the
// same 1.02 G tuples/s, profiles/r3_small_batch/coalescing_share.txt)
inline size_t coalesce_share() {
    static const size_t v = [] {
        const char* e = std::getenv("WF_COALESCE_SHARE");
        return e ? std::max<size_t>(1, (size_t)std::atoll(e)) : (size_t)2;
    }();
    return v;
}
constexpr size_t COALESCE_SMALL_GROUP = 32768;
// WF_INHERIT_STREAMS=0: a GPU replica fused onto its producer keeps a stream of its own instead of continuing on the producer's
inline bool inherit_streams() {
    static const bool h = [] {
        const char* e = std::getenv("WF_INHERIT_STREAMS");
        return !(e && std::atoi(e) == 0);
    }();
    return h;
}
// WF_EDGE_WAIT=host|device: how a GPU operator waits for the producer of a DEVICE batch that arrives from another
// operator's stream
inline bool edge_wait_on_host() {
    static const bool h = [] {
        const char* e = std::getenv("WF_EDGE_WAIT");
        return e && std::string(e) == "host";
    }();
    return h;
}
// HIP keeps host-side bookkeeping for the commands of a stream until the host SYNCHRONISES the stream: waiting on
// events or querying is not enough (apps/bench/probe/hip_leak_probe.hip: ~1 KB per hipEventRecord, ~170 B per async
// copy, linear and unbounded; zero with a hipStreamSynchronize now and then).  A lane only ever waits on events, so
// its stream would accumulate ~3 KB of host memory per batch (24 MB/s on the flagship).  Every node therefore
// synchronises its stream once per `stream_reap_interval()` batches (WF_STREAM_REAP, default 128; 0 = never): a
// bubble of the few batches in flight, once in 128.
inline uint32_t stream_reap_interval() {
    static const uint32_t n = [] {
        const char* e = std::getenv("WF_STREAM_REAP");
        const long v = e ? std::atol(e) : 128;
        return (uint32_t)(v < 0 ? 0 : v);
    }();
    return n;
}
struct StreamReaper {
    uint32_t count = 0;
    void tick(hipStream_t s) {
        const uint32_t every = stream_reap_interval();
        if (every != 0 && ++count >= every) {
            count = 0;
            WF_HIP_CHECK(hipStreamSynchronize(s));
        }
    }
};
// WF_D2H_STREAMS=shared: the device->host adapters of a device share one stream instead of owning one each
inline bool d2h_shared_stream() {
    static const bool h = [] {
        const char* e = std::getenv("WF_D2H_STREAMS");
        return e && std::string(e) == "shared";
    }();
    return h;
}

// Host waits on events sleep in the driver instead of spinning: the waiting threads (one per lane) leave their
// cores to the sources and sinks, at no measurable cost in throughput or latency (profiles/r1_sync_policy).
// WF_EVENT_BLOCKING=0 selects spinning waits.
inline bool event_blocking_sync() {
    static const bool b = [] {
        const char* e = std::getenv("WF_EVENT_BLOCKING");
        return !(e && std::atoi(e) == 0);
    }();
    return b;
}

class Event {
public:
    Event() { WF_HIP_CHECK(hipEventCreateWithFlags(&e_, hipEventDisableTiming | (event_blocking_sync() ? hipEventBlockingSync : 0))); }
    ~Event() {
        if (e_) (void)hipEventDestroy(e_);
    }
    Event(const Event&) = delete;
    Event& operator=(const Event&) = delete;
    hipEvent_t get() const { return e_; }
    void record(hipStream_t s) {
        WF_HIP_CHECK(hipEventRecord(e_, s));
        recorded_ = true;
        stream_ = s;
    }
    // the stream the event was last recorded on (nullptr: never recorded)
    hipStream_t recorded_on() const { return recorded_ ? stream_ : nullptr; }
    // make stream s wait for the event (no-op if never recorded, recorded on s itself - stream order already holds - or
    // already complete: a wait parks a barrier packet in the stream's hardware queue, which other streams share)
    void wait_on(hipStream_t s) const {
        if (recorded_ && stream_ == s) return;
        if (!done()) WF_HIP_CHECK(hipStreamWaitEvent(s, e_, 0));
    }
    void host_sync() const {
        if (recorded_) WF_HIP_CHECK(hipEventSynchronize(e_));
    }
    // wait for an event recorded by ANOTHER operator's stream before enqueueing on s (policy: WF_EDGE_WAIT)
    inline void edge_wait(hipStream_t s) const;
    bool done() const {
        if (!recorded_) return true;
        hipError_t r = hipEventQuery(e_);
        if (r == hipSuccess) return true;
        if (r != hipErrorNotReady) WF_HIP_CHECK(r);
        return false;
    }

private:
    hipEvent_t e_ = nullptr;
    bool recorded_ = false;
    hipStream_t stream_ = nullptr;
};

inline void Event::edge_wait(hipStream_t s) const {
    if (recorded_ && stream_ == s) return;
    if (edge_wait_on_host())
        host_sync();
    else
        wait_on(s);
}

// ---- batches ----------------------------------------------------------------
// 64-byte header in front of the payload; lives in the same slab as the tuples
// so it travels with them in one copy.
struct alignas(64) BatchHeader {
    uint32_t count;       // number of valid tuples in the payload
    uint32_t source_id;   // producer replica
    uint64_t watermark;   // greatest timestamp contained (low-watermark propagation)
    uint64_t first_seq;   // sequence number of the first tuple (per producer)
    uint64_t created_ns;  // host clock when the batch was opened (latency accounting)
    uint64_t reserved[4];  // [0] step punctuation (Batch_Source), [1] marker batch flag, [2] upper bound of the timestamps when the
                           // batch holds more than its watermark claims (deterministic merge, dist/exchange_nodes.hpp)
};
static_assert(sizeof(BatchHeader) == 64, "BatchHeader must be 64 bytes");

enum class MemKind { PINNED_HOST, DEVICE };

class BatchPool;

// A slab [header | payload] in pinned host memory or in device memory.
struct Batch {
    MemKind kind;
    int device = 0;
    size_t elem_size = 0;
    size_t capacity = 0;       // tuples
    char* slab = nullptr;      // header at slab, payload at slab + 64
    BatchHeader host_hdr{};    // host-side view of the header (authoritative when count_on_host)
    bool count_on_host = true; // false: only the device header / count_mirror know the count
    volatile uint32_t* count_mirror = nullptr;  // pinned word a kernel writes the final count to
    Event ready;               // producer's completion / last-use event
    BatchPool* pool = nullptr;

    BatchHeader* header() const { return reinterpret_cast<BatchHeader*>(slab); }
    template <typename T>
    T* data() const {
        return reinterpret_cast<T*>(slab + sizeof(BatchHeader));
    }
    size_t bytes_used() const { return sizeof(BatchHeader) + (size_t)host_hdr.count * elem_size; }
    uint32_t* device_count_ptr() const { return &header()->count; }
    inline void release(hipStream_t last_use_stream);
    inline void release_host();
};

// Bounded pool of equally sized slabs.  acquire() blocks when `max_slabs` are in
// flight: this is the back-pressure between host producers and the device.
class BatchPool {
public:
    BatchPool(MemKind kind, int device, size_t elem_size, size_t capacity, size_t max_slabs)
        : kind_(kind), device_(device), elem_size_(elem_size), capacity_(capacity), max_slabs_(max_slabs) {}
    ~BatchPool() {
        for (Batch* b : all_) {
            if (b->slab) {
                if (kind_ == MemKind::DEVICE)
                    (void)hipFree(b->slab);
                else
                    (void)hipHostFree(b->slab);
            }
            if (b->count_mirror) (void)hipHostFree((void*)b->count_mirror);
            delete b;
        }
    }
    BatchPool(const BatchPool&) = delete;
    BatchPool& operator=(const BatchPool&) = delete;

    // acquire for device work.  The slab's last use (usually a copy enqueued by a downstream adapter) is waited
    // for on the HOST: a device-side wait would park a barrier packet in the caller's hardware queue and stall
    // the other lanes' streams that share it, while the calling thread has nothing else to enqueue anyway.
    Batch* acquire(hipStream_t stream) {
        Batch* b = pop();
        if (pool_wait_on_host())
            b->ready.host_sync();
        else
            b->ready.wait_on(stream);
        reset(b);
        return b;
    }
    // acquire for CPU writes: waits until the last device use completed
    Batch* acquire_host() {
        Batch* b = pop();
        b->ready.host_sync();
        reset(b);
        return b;
    }
    void put(Batch* b) {
        {
            std::lock_guard<std::mutex> l(m_);
            free_.push_back(b);
        }
        cv_.notify_one();
    }
    size_t capacity() const { return capacity_; }
    size_t elem_size() const { return elem_size_; }
    size_t allocated() const { return all_.size(); }
    size_t max_slabs() const { return max_slabs_; }

private:
    Batch* pop() {
        std::unique_lock<std::mutex> l(m_);
        if (free_.empty() && all_.size() < max_slabs_) {
            l.unlock();
            Batch* nb = allocate();
            l.lock();
            all_.push_back(nb);
            return nb;
        }
        cv_.wait(l, [this] { return !free_.empty(); });
        // oldest release first: its last use is the most likely to be complete already
        Batch* b = free_.front();
        free_.pop_front();
        return b;
    }
    static void reset(Batch* b) {
        b->host_hdr = BatchHeader{};
        b->count_on_host = true;
    }
    Batch* allocate() {
        Batch* b = new Batch();
        b->kind = kind_;
        b->device = device_;
        b->elem_size = elem_size_;
        b->capacity = capacity_;
        b->pool = this;
        const size_t bytes = sizeof(BatchHeader) + capacity_ * elem_size_;
        if (kind_ == MemKind::DEVICE) {
            WF_HIP_CHECK(hipMalloc((void**)&b->slab, bytes));
            WF_HIP_CHECK(hipHostMalloc((void**)&b->count_mirror, 64, hipHostMallocDefault));
            *b->count_mirror = 0;
        } else {
            WF_HIP_CHECK(hipHostMalloc((void**)&b->slab, bytes, hipHostMallocDefault));
        }
        return b;
    }

    MemKind kind_;
    int device_;
    size_t elem_size_, capacity_, max_slabs_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Batch*> free_;
    std::vector<Batch*> all_;
};

// WF_DEBUG_EVENTS=1: every release checks, before it records the slab's event, that the stream is alive, belongs to the
// device the slab (and its event) were created on, and that this device is current in the calling thread - the three ways a
// release could hand hipEventRecord a bad pair.  (Added to settle the origin of a fault seen under rocprofv3 with 8 lanes of
// small batches, profiles/r2_small_batch: with the checks on, the run records no violation.)
inline bool debug_events() {
    static const bool on = [] {
        const char* e = std::getenv("WF_DEBUG_EVENTS");
        return e && std::atoi(e) != 0;
    }();
    return on;
}
inline void Batch::release(hipStream_t last_use_stream) {
    if (debug_events()) {
        int cur = -1;
        hipDevice_t sdev = -1;
        const hipError_t q = hipStreamQuery(last_use_stream);
        if (q != hipSuccess && q != hipErrorNotReady) failwith(std::string("Batch::release: dead stream (") + hipGetErrorString(q) + ")");
        WF_HIP_CHECK(hipGetDevice(&cur));
        WF_HIP_CHECK(hipStreamGetDevice(last_use_stream, &sdev));
        if ((int)sdev != device || cur != device)
            failwith("Batch::release: slab of device " + std::to_string(device) + " released on a stream of device " + std::to_string((int)sdev) +
                     " with device " + std::to_string(cur) + " current");
    }
    ready.record(last_use_stream);
    pool->put(this);
}
inline void Batch::release_host() { pool->put(this); }

// A slab's `ready` event belongs to the device that was current when its pool created it; HIP refuses to record an
// event on a stream of another device (hipErrorInvalidHandle).  A consumer on GPU B that has enqueued its last use of
// a slab owned by GPU A (a peer copy, an upload from a pinned slab allocated under A's context) therefore cannot call
// Batch::release(stream of B).  It parks the slab here instead: an event of its OWN device is recorded behind the
// last use, and the slab goes back to its pool (whose next user then has nothing to wait for) once that event has
// completed - polled whenever the consumer releases the next slab, waited for when too many are parked.
class ForeignReleaser {
public:
    explicit ForeignReleaser(size_t max_parked = 2) : max_parked_(max_parked) {}
    ~ForeignReleaser() { drain(); }
    // `s` must be a stream of the calling thread's current device
    void release(Batch* b, hipStream_t s) {
        std::unique_ptr<Event> e(new Event());
        e->record(s);
        parked_.emplace_back(std::move(e), b);
        poll();
        while (parked_.size() > max_parked_) pop_front(true);
    }
    void poll() {
        while (!parked_.empty() && parked_.front().first->done()) pop_front(false);
    }
    void drain() {
        while (!parked_.empty()) pop_front(true);
    }
    size_t parked() const { return parked_.size(); }

private:
    void pop_front(bool wait) {
        if (wait) parked_.front().first->host_sync();
        parked_.front().second->release_host();
        parked_.pop_front();
    }
    size_t max_parked_;
    std::deque<std::pair<std::unique_ptr<Event>, Batch*>> parked_;
};

// release `b` after the work enqueued on `s` (a stream of device `stream_device`, current in the calling thread)
inline void release_after(Batch* b, hipStream_t s, int stream_device, ForeignReleaser& foreign) {
    if (b->device == stream_device)
        b->release(s);
    else
        foreign.release(b, s);
}

}  // namespace gpu
}  // namespace wf
