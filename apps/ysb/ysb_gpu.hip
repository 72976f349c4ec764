// YSB with the whole query on the MI355X (ysb_gpu_common.hpp): events are written straight into pinned slabs, filtered
// and joined on the device, partitioned by campaign on the device (KeyBy) and aggregated by Key_FFAT_GPU in 10 s tumbling
// time windows whose state never leaves HBM; only window results come back to the host.
//   ysb_gpu (-l seconds | -e events per source [-p usec between events]) -n sources -m aggregation replicas
//           [-b batch length] [-d triggering delay, usec] [-g GPUs to spread the replicas over] [-w farm|ffat]
// -w farm uses the non-incremental Key_Farm_GPU (an opaque window function over the archived events of the window).
#include "ysb_gpu_common.hpp"

using namespace wf;

int main(int argc, char** argv) {
    // options beyond the common ones: -d <usec>, -w <farm|ffat>
    uint64_t delay_us = 1000000;
    std::string engine = "ffat";
    std::vector<char*> rest{argv[0]};
    for (int i = 1; i < argc; i++) {
        if (std::string(argv[i]) == "-d" && i + 1 < argc)
            delay_us = std::strtoull(argv[++i], nullptr, 10);
        else if (std::string(argv[i]) == "-w" && i + 1 < argc)
            engine = argv[++i];
        else
            rest.push_back(argv[i]);
    }
    ysb::Options o = ysb::parse((int)rest.size(), rest.data(), " [-b <batch>] [-d <triggering delay usec>] [-g <GPUs>] [-w farm|ffat]");
    if (!o.ok) return 0;
    if (gpu::device_count() < 1 || (size_t)gpu::device_count() < o.gpus) {
        std::cerr << "ysb_gpu: not enough GPUs" << std::endl;
        return 3;
    }
    // replica i of every GPU operator runs on GPU i % gpus (one process driving several GPUs)
    std::vector<int> placement;
    for (size_t i = 0; i < std::max<size_t>(1, o.gpus); i++) placement.push_back((int)i);
    ysb::Campaigns campaigns;
    ysb::DeviceCampaignTables tables(campaigns);  // hash table + record table of the join, one copy per GPU in use
    ysb::Metrics metrics;
    const auto us = [](uint64_t v) { return std::chrono::microseconds(v); };

    // DEFAULT mode: the source lanes are merged by the device KeyBy in arrival order; events of a lane that lags behind
    // are still counted as long as they arrive within the triggering delay of their window
    PipeGraph graph("ysb_gpu", Mode::DEFAULT);
    auto source = BatchSource_Builder(ysb::BulkEventSource(&campaigns, &metrics, o.seconds, o.events, o.period_us)).withName("ysb_source")
                      .withParallelism(o.par1).withBatch(o.batch).build_unique();
    MultiPipe& pipe = graph.add_source(*source);
    auto filter = FilterGPU_Builder(ysb::ViewFilterGPU()).withName("ysb_filter_gpu").withParallelism(o.par1).withGPUPlacement(placement).build_unique();
    pipe.add(*filter);
    auto join = MapGPU_Builder(ysb::JoinGPU(tables)).withName("ysb_join_gpu").withParallelism(o.par1).withGPUPlacement(placement).build_unique();
    pipe.chain(*join);
    std::unique_ptr<Basic_Operator> agg;
    if (engine == "farm") {
        // every view of an open window stays archived on the device until the window fires
        const size_t per_window = (size_t)(ysb::WINDOW_US / std::max<uint64_t>(1, o.period_us)) * o.par1 / 100 + 4096;
        agg = KeyFarmGPU_Builder(ysb::CountWindowGPU()).withTBWindows(us(ysb::WINDOW_US), us(ysb::WINDOW_US), us(delay_us)).withName("ysb_kf_gpu")
                  .withParallelism(o.par2).withGPUPlacement(placement).withMaxKeys(256).withArchiveCapacity(2 * per_window).build_unique();
    } else {
        agg = KeyFFATGPU_Builder(ysb::CountLift(), ysb::CountCombine()).withTBWindows(us(ysb::WINDOW_US), us(ysb::WINDOW_US), us(delay_us))
                  .withName("ysb_kff_gpu").withParallelism(o.par2).withGPUPlacement(placement).withMaxKeys(256).withOutputCapacity(8192).build_unique();
    }
    pipe.add(*agg);
    auto sink = BatchSink_Builder(ysb::BatchResultSink(&metrics)).withName("ysb_sink").withParallelism(o.par2).build_unique();
    pipe.chain_sink(*sink);
    metrics.start_us = current_time_usecs();
    graph.run();
    metrics.report((current_time_usecs() - metrics.start_us) / 1e6, "ysb_gpu");
    return 0;
}
