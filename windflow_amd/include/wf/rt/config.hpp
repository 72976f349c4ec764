// Runtime configuration of the native host runtime.
//
// The reference configures its FastFlow substrate through compile-time macros
// (FF_BOUNDED_BUFFER, DEFAULT_BUFFER_CAPACITY, BLOCKING_MODE, NO_DEFAULT_MAPPING,
// TRACE_WINDFLOW, LOG_DIR: /root/reference/wf/pipegraph.hpp:1033-1047 and
// SURVEY.md §5.6).  Here the same knobs are a run-time struct that can be
// overridden from the environment, so one binary covers every combination.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <string>

namespace wf {
namespace rt {

struct RuntimeConfig {
    size_t queue_capacity = 2048;  // slots per SPSC ring (rounded up to a power of two)
    bool blocking = false;         // true: a thread with nothing to do sleeps on a futex until a ring it waits for changes
    bool pin_threads = false;      // pin threads to cores (of `pin_cpus` when given, else of the process affinity mask)
    std::string pin_cpus;          // WF_PIN_CPUS: cpu list ("0-15,32-47") the threads are pinned to round-robin, e.g. the
                                   // cores of the NUMA node the process's GPU hangs off
    unsigned park_timeout_us = 2000;  // blocking mode: upper bound of one sleep (safety net + on_idle polling period)
    bool trace = false;            // dump per-replica statistics at the end of the run
    std::string log_dir = "log";   // destination of the statistics dump
    unsigned spin_iters = 64;      // busy-wait iterations before yielding
    unsigned yield_iters = 16;     // sched_yield iterations before sleeping
    unsigned sleep_us = 50;        // sleep quantum once the back-off is exhausted

    static RuntimeConfig from_env() {
        RuntimeConfig c;
        if (const char* v = std::getenv("WF_QUEUE_CAPACITY")) c.queue_capacity = std::strtoull(v, nullptr, 10);
        if (const char* v = std::getenv("WF_BLOCKING")) c.blocking = std::atoi(v) != 0;
        if (const char* v = std::getenv("WF_PIN_THREADS")) c.pin_threads = std::atoi(v) != 0;
        if (const char* v = std::getenv("WF_PIN_CPUS")) {
            c.pin_cpus = v;
            c.pin_threads = true;
        }
        if (const char* v = std::getenv("WF_PARK_TIMEOUT_US")) c.park_timeout_us = (unsigned)std::atoi(v);
        if (const char* v = std::getenv("WF_TRACE")) c.trace = std::atoi(v) != 0;
        if (const char* v = std::getenv("WF_LOG_DIR")) c.log_dir = v;
        if (const char* v = std::getenv("WF_SPIN_ITERS")) c.spin_iters = std::atoi(v);
        if (const char* v = std::getenv("WF_SLEEP_US")) c.sleep_us = std::atoi(v);
        if (c.queue_capacity < 2) c.queue_capacity = 2;
        return c;
    }
};

}  // namespace rt
}  // namespace wf
