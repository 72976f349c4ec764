// YSB with every window split across workers: Win_MapReduce with an incremental
// MAP stage (partial counts) and a non-incremental REDUCE stage
// (reference: test_ysb_wmr.cpp:88-129).
#include "ysb_common.hpp"

using namespace wf;

int main(int argc, char** argv) {
    ysb::Options o = ysb::parse(argc, argv);
    if (!o.ok) return 0;
    if (o.par2 < 2) o.par2 = 2;  // Win_MapReduce needs at least two MAP replicas
    ysb::Campaigns campaigns;
    ysb::Metrics metrics;
    PipeGraph graph("ysb_wmr", Mode::DETERMINISTIC);
    Source source = Source_Builder(ysb::EventSource(&campaigns, &metrics, o.seconds, o.events, o.period_us)).withName("ysb_source")
                        .withParallelism(o.par1).build();
    MultiPipe& pipe = graph.add_source(source);
    Filter filter = Filter_Builder(ysb::ViewFilter()).withName("ysb_filter").withParallelism(o.par1).build();
    pipe.chain(filter);
    FlatMap join = FlatMap_Builder(ysb::CampaignJoin(&campaigns)).withName("ysb_join").withParallelism(o.par1).build();
    pipe.chain(join);
    Win_MapReduce agg = WinMapReduce_Builder(ysb::aggregate_inc, ysb::reduce_nic)
                            .withTBWindows(std::chrono::microseconds(ysb::WINDOW_US), std::chrono::microseconds(ysb::WINDOW_US))
                            .withName("ysb_wmr").withParallelism(o.par2, 1).build();
    pipe.add(agg);
    Sink sink = Sink_Builder(ysb::ResultSink(&metrics)).withName("ysb_sink").withParallelism(1).build();
    pipe.chain_sink(sink);
    metrics.start_us = current_time_usecs();
    graph.run();
    metrics.report((current_time_usecs() - metrics.start_us) / 1e6, "ysb_wmr");
    return 0;
}
