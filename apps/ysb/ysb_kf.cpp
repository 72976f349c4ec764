// YSB with the aggregation parallelised by key: Source -> Filter -> FlatMap(join)
// chained in one thread per replica, Key_Farm of incremental tumbling windows,
// Sink chained to the Key_Farm replicas (reference: test_ysb_kf.cpp:88-129).
#include "ysb_common.hpp"

using namespace wf;

int main(int argc, char** argv) {
    ysb::Options o = ysb::parse(argc, argv);
    if (!o.ok) return 0;
    ysb::Campaigns campaigns;
    ysb::Metrics metrics;
    PipeGraph graph("ysb_kf", Mode::DETERMINISTIC);
    Source source = Source_Builder(ysb::EventSource(&campaigns, &metrics, o.seconds, o.events, o.period_us)).withName("ysb_source")
                        .withParallelism(o.par1).build();
    MultiPipe& pipe = graph.add_source(source);
    Filter filter = Filter_Builder(ysb::ViewFilter()).withName("ysb_filter").withParallelism(o.par1).build();
    pipe.chain(filter);
    FlatMap join = FlatMap_Builder(ysb::CampaignJoin(&campaigns)).withName("ysb_join").withParallelism(o.par1).build();
    pipe.chain(join);
    Key_Farm agg = KeyFarm_Builder(ysb::aggregate_inc)
                       .withTBWindows(std::chrono::microseconds(ysb::WINDOW_US), std::chrono::microseconds(ysb::WINDOW_US))
                       .withName("ysb_kf").withParallelism(o.par2).build();
    pipe.add(agg);
    Sink sink = Sink_Builder(ysb::ResultSink(&metrics)).withName("ysb_sink").withParallelism(o.par2).build();
    pipe.chain_sink(sink);
    metrics.start_us = current_time_usecs();
    graph.run();
    metrics.report((current_time_usecs() - metrics.start_us) / 1e6, "ysb_kf");
    return 0;
}
