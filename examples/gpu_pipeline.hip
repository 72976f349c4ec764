// GPU example: sensors -> Filter_GPU -> Map_GPU -> Key_FFAT_GPU (sliding max per sensor) -> Sink
// Batches stay in HBM between the three GPU operators; only window results come back to the host.
// build:  hipcc --offload-arch=gfx950 -O2 -std=c++17 -I windflow_amd/include examples/gpu_pipeline.hip -o gpu_pipeline -lpthread
#include <iostream>
#include <optional>
#include <tuple>

#include <wf/windflow_gpu.hpp>

using namespace wf;

struct Reading {
    uint64_t sensor = 0, id = 0, ts = 0;
    double celsius = 0.0;
    WF_HOST_DEVICE Reading() {}
    WF_HOST_DEVICE std::tuple<uint64_t, uint64_t, uint64_t> getControlFields() const { return {sensor, id, ts}; }
    WF_HOST_DEVICE void setControlFields(uint64_t k, uint64_t i, uint64_t t) {
        sensor = k;
        id = i;
        ts = t;
    }
};
struct Peak {
    uint64_t sensor = 0, wid = 0, ts = 0;
    double fahrenheit = -1e300;
    WF_HOST_DEVICE Peak() {}
    WF_HOST_DEVICE std::tuple<uint64_t, uint64_t, uint64_t> getControlFields() const { return {sensor, wid, ts}; }
    WF_HOST_DEVICE void setControlFields(uint64_t k, uint64_t i, uint64_t t) {
        sensor = k;
        wid = i;
        ts = t;
    }
};

int main() {
    if (gpu::device_count() < 1) {
        std::cerr << "no GPU" << std::endl;
        return 3;
    }
    const uint64_t n_sensors = 1000, n_readings = 4000000;
    PipeGraph graph("gpu_pipeline", Mode::DEFAULT);
    uint64_t n = 0;
    auto src = BatchSource_Builder([=](Reading& r) mutable {
                   r.setControlFields(n % n_sensors, n / n_sensors, n);  // 1 reading per microsecond
                   r.celsius = (double)((n * 2654435761ull >> 8) % 600) / 10.0 - 10.0;
                   return ++n < n_readings;
               }).withName("sensors").withBatch(1 << 16).build_unique();
    MultiPipe& mp = graph.add_source(*src);
    auto valid = FilterGPU_Builder([] __host__ __device__(const Reading& r) { return r.celsius > -5.0; }).withName("valid").build_unique();
    mp.add(*valid);
    auto to_f = MapGPU_Builder([] __host__ __device__(Reading & r) { r.celsius = r.celsius * 9.0 / 5.0 + 32.0; }).withName("to_fahrenheit").build_unique();
    mp.add(*to_f);
    auto peaks = KeyFFATGPU_Builder([] __host__ __device__(const Reading& r, Peak& p) { p.fahrenheit = r.celsius; },
                                    [] __host__ __device__(const Peak& a, const Peak& b, Peak& out) {
                                        out.fahrenheit = a.fahrenheit > b.fahrenheit ? a.fahrenheit : b.fahrenheit;
                                    })
                     .withTBWindows(std::chrono::milliseconds(100), std::chrono::milliseconds(25))
                     .withName("peaks").withMaxKeys(n_sensors).withOutputCapacity(1 << 17).build_unique();
    mp.add(*peaks);
    uint64_t windows = 0;
    double hottest = -1e300;
    Sink sink = Sink_Builder([&](std::optional<Peak>& p) {
                    if (p) {
                        windows++;
                        if (p->fahrenheit > hottest) hottest = p->fahrenheit;
                    } else {
                        std::cout << windows << " windows, hottest reading " << hottest << " F" << std::endl;
                    }
                }).withName("report").build();
    mp.add_sink(sink);
    graph.run();
    return 0;
}
