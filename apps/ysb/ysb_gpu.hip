// YSB with the stateless stages and the window aggregation on the MI355X:
//   Batch_Source (events written straight into pinned slabs)
//     -> Filter_GPU (event_type == 0, fused predicate + stable compaction)
//     -> Map_GPU    (stream-table join through a dense ad -> campaign table in HBM)
//     -> Key_Farm_GPU (10 s tumbling windows per campaign, windows batched across keys)
//     -> Sink
// The MultiPipe keeps batches on the device between Filter_GPU and Map_GPU and
// inserts the device->host adapter in front of the window operator by itself.
#include <wf/windflow_gpu.hpp>

#include "ysb_common.hpp"

using namespace wf;

struct ViewFilterGPU {
    __host__ __device__ bool operator()(const ysb::Event& e) const { return e.event_type == 0; }
};
struct JoinGPU {
    uint64_t n_ads, per_campaign, inv_a, b;  // inverse of the ad-id permutation (ysb::Campaigns)
    __host__ __device__ void operator()(const ysb::Event& e, ysb::Joined& j) const {
        j.ts = e.ts;
        j.seq = e.seq;
        j.ad_id = e.ad_id;
        const uint64_t pos = ((e.ad_id % n_ads + n_ads - b % n_ads) % n_ads) * inv_a % n_ads;
        j.cmp_id = e.ad_id < n_ads ? pos / per_campaign : 0xffffffffull;
    }
};
struct CountWindowGPU {
    __host__ __device__ void operator()(uint64_t, const ysb::Joined* data, size_t n, ysb::WinResult* res, char*, size_t) const {
        uint64_t last = 0;
        for (size_t i = 0; i < n; i++) last = data[i].ts > last ? data[i].ts : last;
        res->count = n;
        res->last_update = last;
    }
};

int main(int argc, char** argv) {
    ysb::Options o = ysb::parse(argc, argv, " [-b <device batch length>] [-g <GPUs to spread the replicas over>]");
    if (!o.ok) return 0;
    if (gpu::device_count() < 1 || (size_t)gpu::device_count() < o.gpus) {
        std::cerr << "ysb_gpu: not enough GPUs" << std::endl;
        return 3;
    }
    // replica i of every GPU operator runs on GPU i % gpus (one process driving several GPUs: BASELINE config 5)
    std::vector<int> placement;
    for (size_t i = 0; i < std::max<size_t>(1, o.gpus); i++) placement.push_back((int)i);
    ysb::Campaigns campaigns;
    ysb::Metrics metrics;

    // DETERMINISTIC: the batch lanes keep their order and the window stage merges them by timestamp, so no
    // event is dropped as late when several sources run at different speeds
    PipeGraph graph("ysb_gpu", Mode::DETERMINISTIC);
    auto source = BatchSource_Builder(ysb::EventSource(&campaigns, &metrics, o.seconds, o.events, o.period_us)).withName("ysb_source")
                      .withParallelism(o.par1).withBatch(o.batch).build_unique();
    MultiPipe& pipe = graph.add_source(*source);
    auto filter = FilterGPU_Builder(ViewFilterGPU()).withName("ysb_filter_gpu").withParallelism(o.par1).withGPUPlacement(placement).build_unique();
    pipe.add(*filter);
    auto join = MapGPU_Builder(JoinGPU{campaigns.n_ads(), campaigns.ads_per_campaign(), campaigns.inverse_multiplier(), ysb::Campaigns::B})
                    .withName("ysb_join_gpu").withParallelism(o.par1).withGPUPlacement(placement).build_unique();
    pipe.add(*join);
    auto agg = KeyFarmGPU_Builder(CountWindowGPU())
                   .withTBWindows(std::chrono::microseconds(ysb::WINDOW_US), std::chrono::microseconds(ysb::WINDOW_US))
                   .withName("ysb_kf_gpu").withParallelism(o.par2).withBatch(64).withGPUPlacement(placement).build_unique();
    pipe.add(*agg);
    Sink sink = Sink_Builder(ysb::ResultSink(&metrics)).withName("ysb_sink").withParallelism(o.par2).build();
    pipe.chain_sink(sink);
    metrics.start_us = current_time_usecs();
    graph.run();
    metrics.report((current_time_usecs() - metrics.start_us) / 1e6, "ysb_gpu");
    return 0;
}
