"""Flagship streaming applications assembled from the C++ library (PipeGraph + GPU operators).

``WinAgg``      keyed sliding-window aggregation: Batch_Source xS -> [Exchange_GPU] -> Key_FFAT_GPU xS -> Batch_Sink
``WinSeq``      non-incremental windows: Batch_Source xS -> Key_Farm_GPU (opaque window function or wave_reduce) -> Batch_Sink
``Ysb``         Yahoo Streaming Benchmark on the device: filter -> join -> [Exchange_GPU] -> 10 s tumbling count per campaign
``MapFilter``   stateless Map_GPU -> Filter_GPU chain

All are persistent pipelines driven step by step (``step(tuples_per_source)`` returns
when every result of the step reached the sinks), which is what ``bench.py`` times.
"""
from .. import load_ext

__all__ = ["WinAgg", "WinSeq", "Ysb", "MapFilter"]


class _Pipeline:
    def __init__(self, native):
        self._p = native

    def step(self, tuples_per_source):
        self._p.step(int(tuples_per_source))

    def steps(self, k, tuples_per_source):
        """k steps streamed back to back; returns when the results of the last one reached the sinks."""
        self._p.steps(int(k), int(tuples_per_source))

    def stats(self):
        return dict(self._p.stats())

    def reset_latency(self):
        self._p.reset_latency()

    def dot(self):
        return self._p.dot()

    def close(self):
        self._p.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class WinAgg(_Pipeline):
    """Keyed sliding-window sum on one GPU (optionally one rank of a multi-GPU job).

    win/slide are in tuples of a key for count-based windows, in microseconds of the
    synthetic per-source clock for time-based ones.  With ``transport`` (see
    ``windflow_amd.parallel``) tuples are re-partitioned by key across ranks first.
    """

    KEY_DISTS = {"roundrobin": 0, "uniform": 1, "zipf": 2}

    def __init__(self, gpu_id=0, sources=8, batch=256 * 1024, keys_per_source=1 << 17, win=16, slide=4, time_based=False, delay=0,
                 inflight=6, transport=None, agg_replicas=0, key_dist="roundrobin", zipf_alpha=1.3, shared_keys=False, deterministic=False,
                 device_source=False, combiner=False):
        ext = load_ext()
        super().__init__(ext.WinAggPipeline(gpu_id, sources, batch, keys_per_source, win, slide, time_based, delay, inflight, transport,
                                            agg_replicas, self.KEY_DISTS[key_dist], zipf_alpha, shared_keys, deterministic, device_source,
                                            combiner))
        self.sources = sources


class WinSeq(_Pipeline):
    """Non-incremental keyed windows on one GPU (``Key_Farm_GPU`` over the device-resident window engine).

    ``wave_reduce=False`` runs the opaque 6-argument window functor (a sequential sum per window), ``True`` the associative
    form ``wave_reduce(lift, combine)`` evaluated by a wave64 per window.  ``replicas=0``: one replica per source.
    """

    def __init__(self, gpu_id=0, sources=8, batch=128 * 1024, keys=1024, win=1024, slide=128, time_based=False, inflight=4, replicas=0,
                 wave_reduce=False, key_dist="roundrobin", zipf_alpha=1.3, shared_keys=False):
        ext = load_ext()
        super().__init__(ext.WinSeqPipeline(gpu_id, sources, batch, keys, win, slide, time_based, inflight, replicas, wave_reduce,
                                            WinAgg.KEY_DISTS[key_dist], zipf_alpha, shared_keys))
        self.sources = sources


class Ysb(_Pipeline):
    """Yahoo Streaming Benchmark on one GPU (optionally one rank of a multi-GPU job): events are filtered, joined with the
    campaign table (device hash probe) and counted in 10 s tumbling windows per campaign, all on the device."""

    def __init__(self, gpu_id=0, sources=8, batch=128 * 1024, period_us=1, delay_us=10_000_000, inflight=4, agg_replicas=1, transport=None,
                 split_merge=False, deterministic=False):
        ext = load_ext()
        super().__init__(ext.YsbPipeline(gpu_id, sources, batch, period_us, delay_us, inflight, agg_replicas, transport, split_merge, deterministic))
        self.sources = sources


class MapFilter(_Pipeline):
    """Stateless Map_GPU + Filter_GPU chain on one GPU."""

    def __init__(self, gpu_id=0, sources=8, batch=256 * 1024, keys_per_source=1 << 10, inflight=6):
        ext = load_ext()
        super().__init__(ext.MapFilterPipeline(gpu_id, sources, batch, keys_per_source, inflight))
        self.sources = sources
