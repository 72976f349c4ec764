// Does host memory grow in a plain HIP loop of the calls a lane makes per batch?  (no framework code)
// usage: hip_leak_probe [mode] [iterations]   mode bits: 1 = H2D copy, 2 = kernel, 4 = event record + host wait,
//                                                8 = hipStreamWaitEvent on another stream's event, 16 = D2H copy,
//                                                32 = events without hipEventBlockingSync, 64 = hipStreamSynchronize every 64
//                                                iterations as well, 128 = hipEventQuery polling instead of hipEventSynchronize
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            std::exit(1);                                                             \
        }                                                                             \
    } while (0)

__global__ void touch(unsigned* p, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1u;
}

static long rss_kb() {
    std::ifstream f("/proc/self/status");
    std::string k;
    while (f >> k) {
        if (k == "VmRSS:") {
            long v;
            f >> v;
            return v;
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? std::atoi(argv[1]) : 31;
    const long iters = argc > 2 ? std::atol(argv[2]) : 200000;
    const size_t bytes = 1 << 20;
    hipStream_t s, s2;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned *host, *dev, *host2;
    CHECK(hipHostMalloc((void**)&host, bytes, hipHostMallocDefault));
    CHECK(hipHostMalloc((void**)&host2, bytes, hipHostMallocDefault));
    CHECK(hipMalloc((void**)&dev, bytes));
    hipEvent_t ev[4];
    for (auto& e : ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming | ((mode & 32) ? 0 : hipEventBlockingSync)));
    const long r0 = rss_kb();
    for (long i = 0; i < iters; i++) {
        hipEvent_t e = ev[i & 3];
        if (mode & 1) CHECK(hipMemcpyAsync(dev, host, bytes / 4, hipMemcpyHostToDevice, s));
        if (mode & 2) hipLaunchKernelGGL(touch, dim3(64), dim3(256), 0, s, dev, 16384u);
        if (mode & 4) CHECK(hipEventRecord(e, s));
        if (mode & 8) CHECK(hipStreamWaitEvent(s2, e, 0));
        if (mode & 16) CHECK(hipMemcpyAsync(host2, dev, bytes / 16, hipMemcpyDeviceToHost, (mode & 8) ? s2 : s));
        if (mode & 4) {
            if ((i & 3) == 3) {
                if (mode & 128) {
                    while (hipEventQuery(ev[(i - 2) & 3]) == hipErrorNotReady) {
                    }
                } else {
                    CHECK(hipEventSynchronize(ev[(i - 2) & 3]));
                }
            }
            if ((mode & 64) && (i & 63) == 63) CHECK(hipStreamSynchronize(s));
        } else if ((i & 63) == 63) {
            CHECK(hipStreamSynchronize(s));
        }
        if (i % (iters / 5) == 0) std::printf("mode %d iter %ld rss_kb %ld (+%ld)\n", mode, i, rss_kb(), rss_kb() - r0);
    }
    CHECK(hipStreamSynchronize(s));
    CHECK(hipStreamSynchronize(s2));
    std::printf("mode %d done: rss grew by %ld KB over %ld iterations = %.1f bytes per iteration\n", mode, rss_kb() - r0, iters,
                (rss_kb() - r0) * 1024.0 / iters);
    return 0;
}
