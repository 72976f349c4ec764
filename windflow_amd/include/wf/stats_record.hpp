// Stats_Record: per-replica statistics and their dump.
// Capability parity with /root/reference/wf/stats_record.hpp:50-156 (counters,
// service time, GPU-side kernel/byte counters, dump to log/<pid>_<op>_<replica>.log)
// but always compiled in: the counters are cheap and the dump is driven by the
// run-time RuntimeConfig::trace flag instead of -DTRACE_WINDFLOW.
#pragma once
#include <cstdint>
#include <fstream>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <unistd.h>

namespace wf {

struct Stats_Record {
    std::string nameOP;
    std::string nameReplica;
    bool isGPUReplica = false;
    bool isWinOperator = false;
    uint64_t start_time_us = 0;
    uint64_t end_time_us = 0;
    uint64_t inputs_received = 0;
    uint64_t inputs_ignored = 0;
    uint64_t bytes_received = 0;
    uint64_t outputs_sent = 0;
    uint64_t bytes_sent = 0;
    double service_time_us = 0.0;      // average per-input service time: time inside svc() (fused successors included) / inputs
    double eff_service_time_us = 0.0;  // average inter-departure time: running time / outputs
    bool service_time_measured = false;  // false: the replica runs fused in another node's thread, whose record holds the time
    uint64_t busy_time_us = 0;         // time the replica's thread spent processing (not waiting for input)
    uint64_t parks = 0;                // blocking mode: times the thread slept waiting for input
    // GPU replicas
    uint64_t num_kernels = 0;
    uint64_t bytes_copied_hd = 0;
    uint64_t bytes_copied_dh = 0;

    Stats_Record() = default;
    Stats_Record(std::string op, std::string replica, bool win = false, bool gpu = false)
        : nameOP(std::move(op)), nameReplica(std::move(replica)), isGPUReplica(gpu), isWinOperator(win) {}

    std::string to_json() const {
        std::ostringstream o;
        o << "{\"operator\":\"" << nameOP << "\",\"replica\":\"" << nameReplica << "\",\"gpu\":" << (isGPUReplica ? "true" : "false")
          << ",\"window\":" << (isWinOperator ? "true" : "false") << ",\"inputs_received\":" << inputs_received
          << ",\"inputs_ignored\":" << inputs_ignored << ",\"bytes_received\":" << bytes_received << ",\"outputs_sent\":" << outputs_sent
          << ",\"bytes_sent\":" << bytes_sent << ",\"service_time_us\":" << service_time_us
          << ",\"service_time_measured\":" << (service_time_measured ? "true" : "false") << ",\"eff_service_time_us\":" << eff_service_time_us
          << ",\"busy_time_us\":" << busy_time_us << ",\"parks\":" << parks << ",\"running_time_us\":" << (end_time_us - start_time_us);
        if (isGPUReplica)
            o << ",\"num_kernels\":" << num_kernels << ",\"bytes_copied_hd\":" << bytes_copied_hd << ",\"bytes_copied_dh\":" << bytes_copied_dh;
        o << "}";
        return o.str();
    }

    // write <dir>/<pid>_<op>_<replica>.log
    void dump_toFile(const std::string& dir = "log") const {
        ::mkdir(dir.c_str(), 0777);
        std::ofstream f(dir + "/" + std::to_string(::getpid()) + "_" + nameOP + "_" + nameReplica + ".log");
        f << to_json() << std::endl;
    }
};

}  // namespace wf
