// Optional rocTX ranges around batch shipping, kernel enqueue and exchange rounds
// (the reference has no NVTX/CUPTI integration; rocprofv3
// --marker-trace shows these ranges next to the kernels).  Enabled at run time
// with WF_ROCTX=1; the marker library is resolved with dlopen so that programs
// that never trace do not link or load it.
#pragma once
#include <dlfcn.h>

#include <cstdlib>

namespace wf {
namespace gpu {

class Roctx {
public:
    static Roctx& get() {
        static Roctx r;
        return r;
    }
    bool enabled() const { return push_ != nullptr; }
    void push(const char* name) const {
        if (push_) push_(name);
    }
    void pop() const {
        if (pop_) pop_();
    }

private:
    Roctx() {
        const char* v = std::getenv("WF_ROCTX");
        if (v == nullptr || std::atoi(v) == 0) return;
        // the rocprofiler-sdk flavour is what rocprofv3 intercepts; the roctracer one is the fallback
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (h == nullptr) continue;
            push_ = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop_ = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push_ && pop_) return;
            push_ = nullptr;
            pop_ = nullptr;
        }
    }
    int (*push_)(const char*) = nullptr;
    int (*pop_)() = nullptr;
};

// RAII range; a no-op unless WF_ROCTX=1
class TraceRange {
public:
    explicit TraceRange(const char* name) : on_(Roctx::get().enabled()) {
        if (on_) Roctx::get().push(name);
    }
    ~TraceRange() {
        if (on_) Roctx::get().pop();
    }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;

private:
    bool on_;
};

}  // namespace gpu
}  // namespace wf
