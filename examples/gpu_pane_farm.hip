// GPU example in the reference's programming model: a CPU Source emits trades one by one, a Pane_Farm_GPU computes the traded
// volume per symbol over sliding count-based windows with the pane stage (PLQ) on the device and the window stage (WLQ) on the
// host, a CPU Sink prints the totals.  The key is not a number: a 16-byte fixed-size symbol (wf::FixedKey<16>).
// The source's tuples become pinned batches in its own thread (Batcher); the device stage is keyed and fed by those batches.
// build:  hipcc --offload-arch=gfx950 -O2 -std=c++17 -I windflow_amd/include examples/gpu_pane_farm.hip -o gpu_pane_farm -lpthread
#include <iostream>
#include <optional>
#include <string>
#include <tuple>

#include <wf/windflow_gpu.hpp>

using namespace wf;
using symbol_t = FixedKey<16>;

struct Trade {
    symbol_t symbol;
    uint64_t id = 0, ts = 0;
    int64_t shares = 0;
    WF_HOST_DEVICE Trade() {}
    WF_HOST_DEVICE std::tuple<symbol_t, uint64_t, uint64_t> getControlFields() const { return {symbol, id, ts}; }
    WF_HOST_DEVICE void setControlFields(symbol_t k, uint64_t i, uint64_t t) {
        symbol = k;
        id = i;
        ts = t;
    }
};
struct Volume {
    symbol_t symbol;
    uint64_t wid = 0, ts = 0;
    int64_t shares = 0;
    WF_HOST_DEVICE Volume() {}
    WF_HOST_DEVICE std::tuple<symbol_t, uint64_t, uint64_t> getControlFields() const { return {symbol, wid, ts}; }
    WF_HOST_DEVICE void setControlFields(symbol_t k, uint64_t i, uint64_t t) {
        symbol = k;
        wid = i;
        ts = t;
    }
};

// PLQ on the device: the volume of one pane (6-argument window functor of the reference's GPU operators)
struct PaneVolume {
    __host__ __device__ void operator()(uint64_t, const Trade* data, size_t n, Volume* res, char*, size_t) const {
        int64_t s = 0;
        for (size_t i = 0; i < n; i++) s += data[i].shares;
        res->shares = s;
    }
};

int main() {
    if (gpu::device_count() < 1) {
        std::cerr << "no GPU" << std::endl;
        return 3;
    }
    const uint64_t n_symbols = 50, n_trades = 2000000, win = 1000, slide = 250;
    PipeGraph graph("gpu_pane_farm", Mode::DETERMINISTIC);  // count-based windows need ordered input
    uint64_t n = 0;
    Source src = Source_Builder([=](Trade& t) mutable {
                     t.setControlFields(symbol_t("SYM_" + std::to_string(n % n_symbols)), n / n_symbols, n);
                     t.shares = (int64_t)(n % 97) + 1;
                     return ++n < n_trades;
                 }).withName("trades").build();
    MultiPipe& mp = graph.add_source(src);
    // WLQ on the host: the volume of a window from its panes (non-incremental signature over an Iterable)
    auto window_volume = [](uint64_t, const Iterable<Volume>& panes, Volume& out) {
        out.shares = 0;
        for (const Volume& p : panes) out.shares += p.shares;
    };
    auto pf = PaneFarmGPU_Builder(PaneVolume(), window_volume).withCBWindows(win, slide).withParallelism(2, 2).withBatch(1 << 14)
                  .withMaxKeys(2 * n_symbols).withName("volume").build_unique();
    mp.add(*pf);
    uint64_t windows = 0;
    int64_t total = 0;
    Sink sink = Sink_Builder([&](std::optional<Volume>& v) {
                    if (v) {
                        windows++;
                        total += v->shares;
                    } else {
                        std::cout << windows << " windows, " << total << " shares counted" << std::endl;
                    }
                }).withName("report").build();
    mp.add_sink(sink);
    graph.run();
    // every symbol sees n_trades / n_symbols trades; a window every `slide` of them (the last ones incomplete)
    const uint64_t per_symbol = n_trades / n_symbols;
    const uint64_t expected = n_symbols * ((per_symbol + slide - 1) / slide);
    if (windows != expected) {
        std::cerr << "expected " << expected << " windows" << std::endl;
        return 1;
    }
    return 0;
}
