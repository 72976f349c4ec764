"""Flagship streaming applications assembled from the C++ library (PipeGraph + GPU operators).

``WinAgg``      keyed sliding-window aggregation: Batch_Source xS -> [Exchange_GPU] -> Key_FFAT_GPU xS -> Batch_Sink
``MapFilter``   stateless Map_GPU -> Filter_GPU chain

Both are persistent pipelines driven step by step (``step(tuples_per_source)`` returns
when every result of the step reached the sinks), which is what ``bench.py`` times.
"""
from .. import load_ext

__all__ = ["WinAgg", "MapFilter"]


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

    def __init__(self, gpu_id=0, sources=8, batch=256 * 1024, keys_per_source=1 << 17, win=16, slide=4, time_based=False, delay=0,
                 inflight=6, transport=None):
        ext = load_ext()
        super().__init__(ext.WinAggPipeline(gpu_id, sources, batch, keys_per_source, win, slide, time_based, delay, inflight, transport))
        self.sources = sources


class MapFilter(_Pipeline):
    """Stateless Map_GPU + Filter_GPU chain on one GPU."""

    def __init__(self, gpu_id=0, sources=8, batch=256 * 1024, keys_per_source=1 << 10, inflight=6):
        ext = load_ext()
        super().__init__(ext.MapFilterPipeline(gpu_id, sources, batch, keys_per_source, inflight))
        self.sources = sources
