// YSB across GPUs: one rank per GPU, every rank ingests its own events, filters and joins them on its device, and the
// joined views are re-partitioned by campaign ACROSS the ranks by Exchange_GPU (all-to-all over xGMI), so that a campaign's
// 10 s tumbling count lives on exactly one GPU (and one aggregation lane of it).  BASELINE.json config 5; the reference
// harness is single-process (/root/reference/src/yahoo_test_cpu/test_ysb_kf.cpp:113-135).
//
//   ysb_gpu_dist (-l seconds | -e events per source [-p usec between events]) -n sources -m aggregation lanes
//                [-b batch length] [-d triggering delay, usec] [-x 1: split/merge variant]
//                [-t rccl|local] [-r ranks, local transport] [-i file that carries the RCCL id]
//
//   -t rccl  (default) one PROCESS per GPU: RANK / WORLD_SIZE / LOCAL_RANK come from the launcher's environment
//            (torchrun, mpirun wrappers, a shell loop), rank 0 hands the RCCL unique id to the others through the file
//            given with -i (default /tmp/wf_ysb_rccl_id.<MASTER_PORT>; the launcher's pid is appended, so a file a
//            crashed job left behind is never read by the next one).
//   -t local all ranks are threads of THIS process coupled by LocalTransport (device copies; peer copies when the
//            ranks sit on different GPUs): the way to run the distributed query on one GPU, and the test vehicle.
// With -x 1 the events are split on the device by ad type into two filter+join branches that are merged again in
// front of the exchange (PipeGraph split/merge on device-batch edges).
#include <cstdlib>
#include <thread>

#include "ysb_gpu_common.hpp"

using namespace wf;

struct DistOptions {
    ysb::Options o;
    uint64_t delay_us = 1000000;
    bool split_merge = false;
    std::string transport = "rccl", id_file;
    int ranks = 2;
};

static void run_rank(const DistOptions& d, std::shared_ptr<dist::Transport> tr, int device, ysb::Metrics& metrics, ExchangeStats* xstats) {
    const ysb::Options& o = d.o;
    ysb::Campaigns campaigns;
    ysb::DeviceCampaignTables tables(campaigns);  // hash table + record table of the join on this rank's GPU
    const auto us = [](uint64_t v) { return std::chrono::microseconds(v); };
    PipeGraph graph("ysb_gpu_dist_rank" + std::to_string(tr->rank()), Mode::DEFAULT);
    auto source = BatchSource_Builder(ysb::BulkEventSource(&campaigns, &metrics, o.seconds, o.events, o.period_us)).withName("ysb_source")
                      .withParallelism(o.par1).withBatch(o.batch).withGPUConfiguration(device).build_unique();
    MultiPipe* pipe = &graph.add_source(*source);
    const size_t branches = d.split_merge ? 2 : 1;
    std::vector<std::unique_ptr<Filter_GPU<ysb::Event>>> filters;
    std::vector<std::unique_ptr<Map_GPU<ysb::Event, ysb::Joined>>> joins;
    for (size_t b = 0; b < branches; b++) {
        filters.push_back(FilterGPU_Builder(ysb::ViewFilterGPU()).withName("ysb_filter_gpu").withParallelism(o.par1).withGPUConfiguration(device).build_unique());
        joins.push_back(MapGPU_Builder(ysb::JoinGPU(tables)).withName("ysb_join_gpu").withParallelism(o.par1).withGPUConfiguration(device).build_unique());
    }
    if (d.split_merge) {
        GPUConfig scfg;
        scfg.gpu_id = device;
        split_gpu(*pipe, ysb::AdTypeSplit(), 2, scfg);
        MultiPipe& b0 = pipe->select(0);
        b0.chain(*filters[0]);
        b0.chain(*joins[0]);
        MultiPipe& b1 = pipe->select(1);
        b1.chain(*filters[1]);
        b1.chain(*joins[1]);
        pipe = &b0.merge(b1);
    } else {
        pipe->add(*filters[0]);
        pipe->chain(*joins[0]);
    }
    auto exchange = ExchangeGPU_Builder<ysb::Joined>(tr).withName("exchange").withParallelism(o.par1).withLocalLanes(o.par1 * branches)
                        .withOutputLanes(o.par2).withGPUConfiguration(device).build_unique();
    pipe->add(*exchange);
    auto agg = KeyFFATGPU_Builder(ysb::CountLift(), ysb::CountCombine()).withTBWindows(us(ysb::WINDOW_US), us(ysb::WINDOW_US), us(d.delay_us))
                   .withName("ysb_kff_gpu").withParallelism(o.par2).withGPUConfiguration(device).withMaxKeys(256).withOutputCapacity(8192)
                   .withKeyDisjointLanes().build_unique();
    pipe->add(*agg);
    auto sink = BatchSink_Builder(ysb::BatchResultSink(&metrics)).withName("ysb_sink").withParallelism(o.par2).build_unique();
    pipe->chain_sink(*sink);
    graph.run();
    if (xstats) *xstats = exchange->stats();
}

int main(int argc, char** argv) {
    DistOptions d;
    std::vector<char*> rest{argv[0]};
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-d" && i + 1 < argc)
            d.delay_us = std::strtoull(argv[++i], nullptr, 10);
        else if (a == "-x" && i + 1 < argc)
            d.split_merge = std::atoi(argv[++i]) != 0;
        else if (a == "-t" && i + 1 < argc)
            d.transport = argv[++i];
        else if (a == "-r" && i + 1 < argc)
            d.ranks = std::atoi(argv[++i]);
        else if (a == "-i" && i + 1 < argc)
            d.id_file = argv[++i];
        else
            rest.push_back(argv[i]);
    }
    d.o = ysb::parse((int)rest.size(), rest.data(), " [-b <batch>] [-d <triggering delay usec>] [-x 0|1] [-t rccl|local] [-r <ranks>] [-i <id file>]");
    if (!d.o.ok) return 0;
    const int n_dev = gpu::device_count();
    if (n_dev < 1) {
        std::cerr << "ysb_gpu_dist: no GPU" << std::endl;
        return 3;
    }
    if (d.transport == "local") {
        if (d.ranks < 1 || d.ranks > dist::MAX_WORLD) {
            std::cerr << "ysb_gpu_dist: between 1 and " << dist::MAX_WORLD << " ranks" << std::endl;
            return 2;
        }
        auto fabric = std::make_shared<dist::LocalFabric>(d.ranks);
        std::vector<std::unique_ptr<ysb::Metrics>> metrics;
        std::vector<ExchangeStats> xs((size_t)d.ranks);
        for (int r = 0; r < d.ranks; r++) metrics.emplace_back(new ysb::Metrics());
        const uint64_t start = current_time_usecs();
        std::vector<std::thread> threads;
        for (int r = 0; r < d.ranks; r++) {
            metrics[(size_t)r]->start_us = start;
            threads.emplace_back([&, r] { run_rank(d, std::make_shared<dist::LocalTransport>(fabric, r), r % n_dev, *metrics[(size_t)r], &xs[(size_t)r]); });
        }
        for (auto& t : threads) t.join();
        const double elapsed = (current_time_usecs() - start) / 1e6;
        ysb::Metrics total;
        uint64_t rounds = 0, max_link = 0;
        for (int r = 0; r < d.ranks; r++) {
            ysb::Metrics& m = *metrics[(size_t)r];
            total.generated += m.generated.load();
            total.results += m.results.load();
            total.counted_events += m.counted_events.load();
            total.latencies_us.insert(total.latencies_us.end(), m.latencies_us.begin(), m.latencies_us.end());
            rounds = std::max(rounds, xs[(size_t)r].rounds);
            max_link = std::max(max_link, xs[(size_t)r].max_link_bytes_per_round);
        }
        total.report(elapsed, "ysb_gpu_dist");
        std::cout << "[ysb_gpu_dist] ranks " << d.ranks << " transport local exchange rounds " << rounds << " max link bytes per round " << max_link << std::endl;
        return 0;
    }
    // one process per GPU
    const char* e_rank = std::getenv("RANK");
    const char* e_world = std::getenv("WORLD_SIZE");
    const char* e_local = std::getenv("LOCAL_RANK");
    const int rank = e_rank ? std::atoi(e_rank) : 0, world = e_world ? std::atoi(e_world) : 1;
    const int device = (e_local ? std::atoi(e_local) : rank) % n_dev;
    if (d.id_file.empty()) {
        const char* port = std::getenv("MASTER_PORT");
        d.id_file = std::string("/tmp/wf_ysb_rccl_id.") + (port ? port : "0");
    }
    WF_HIP_CHECK(hipSetDevice(device));
    const std::string id = dist::bootstrap_id_via_file(d.id_file, rank);
    auto tr = std::make_shared<dist::RcclTransport>(rank, world, id, device);
    if (rank == 0) dist::bootstrap_id_cleanup(d.id_file);  // everybody has read it: the constructor ends with a barrier
    ysb::Metrics metrics;
    ExchangeStats xs;
    metrics.start_us = current_time_usecs();
    run_rank(d, tr, device, metrics, &xs);
    const double elapsed = (current_time_usecs() - metrics.start_us) / 1e6;
    // totals of the job on rank 0
    uint64_t mine[3] = {metrics.generated.load(), metrics.results.load(), metrics.counted_events.load()};
    std::vector<uint64_t> all(3 * (size_t)world);
    {
        gpu::Stream s;
        tr->all_gather_host(mine, all.data(), 3, s.get());
    }
    metrics.report(elapsed, ("ysb_gpu_dist rank " + std::to_string(rank)).c_str());
    if (rank == 0) {
        uint64_t g = 0, r = 0, c = 0;
        for (int i = 0; i < world; i++) {
            g += all[3 * (size_t)i];
            r += all[3 * (size_t)i + 1];
            c += all[3 * (size_t)i + 2];
        }
        std::cout << "[ysb_gpu_dist] generated events " << g << "\n[ysb_gpu_dist] received results " << r << "\n[ysb_gpu_dist] events counted in windows " << c
                  << "\n[ysb_gpu_dist] ranks " << world << " transport rccl exchange rounds " << xs.rounds << " max link bytes per round "
                  << xs.max_link_bytes_per_round << "\n[ysb_gpu_dist] throughput events/s " << (elapsed > 0 ? g / elapsed : 0.0) << std::endl;
    }
    return 0;
}
