"""Tensor-level entry points of the gfx950 kernels (torch CUDA tensors in, torch tensors out).

Tuples are rows of 5 x int64 (40 bytes): key, id, ts, value (float64 bits), aux (float64 bits) -
the layout of ``wfbench::tuple_t`` in ``csrc/bench_common.hpp``.  Every function
launches the library's own HIP kernels through the native extension; there is no
PyTorch fallback (a missing extension raises).
"""
import numpy as np
import torch

from .. import load_ext

__all__ = ["make_tuples", "tuple_values", "map_scale_", "filter_compact", "exclusive_scan", "radix_sort_pairs_", "key_ffat"]

TUPLE_WORDS = 5


def _check(t, dtype, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype == dtype):
        raise TypeError("%s must be a contiguous CUDA tensor of dtype %s" % (what, dtype))


def make_tuples(keys, values, ts=None, ids=None, aux=1.0, device="cuda"):
    """Pack key/value columns into the 40-byte tuple layout; returns an (n, 5) int64 tensor."""
    keys = np.asarray(keys, dtype=np.int64)
    n = len(keys)
    t = np.zeros((n, TUPLE_WORDS), dtype=np.int64)
    t[:, 0] = keys
    t[:, 1] = np.arange(n) if ids is None else np.asarray(ids, dtype=np.int64)
    t[:, 2] = np.arange(n) if ts is None else np.asarray(ts, dtype=np.int64)
    t[:, 3] = np.asarray(values, dtype=np.float64).view(np.int64)
    t[:, 4] = np.full(n, aux, dtype=np.float64).view(np.int64)
    out = torch.from_numpy(t)
    return out.to(device) if device else out


def tuple_values(tuples):
    """The float64 value column of a tuple tensor."""
    return tuples[:, 3].contiguous().view(torch.float64)


def map_scale_(tuples, factor):
    """In-place Map_GPU kernel: value = value * factor + aux."""
    _check(tuples, torch.int64, "tuples")
    load_ext().map_scale_inplace(tuples.data_ptr(), tuples.shape[0], float(factor))
    return tuples


def filter_compact(tuples, mode="value_below", threshold=0.0):
    """Fused Filter_GPU predicate + stable compaction; returns the surviving tuples (new tensor)."""
    _check(tuples, torch.int64, "tuples")
    out = torch.empty_like(tuples)
    m = {"value_below": 0, "even_key": 1}[mode]
    n = load_ext().filter_compact(tuples.data_ptr(), tuples.shape[0], out.data_ptr(), m, float(threshold))
    return out[:n]


def exclusive_scan(x):
    """Exclusive prefix sum of an int32 CUDA tensor; returns (scan, total)."""
    _check(x, torch.int32, "x")
    out = torch.empty_like(x)
    total = load_ext().exclusive_scan(x.data_ptr(), out.data_ptr(), x.numel())
    return out, total


def radix_sort_pairs_(keys, vals, bits=32):
    """Stable LSD radix sort of (uint32 key, uint32 value) pairs held as int32 bit patterns, in place."""
    _check(keys, torch.int32, "keys")
    _check(vals, torch.int32, "vals")
    load_ext().radix_sort_pairs(keys.data_ptr(), vals.data_ptr(), keys.numel(), int(bits))
    return keys, vals


def key_ffat(tuples_host, win, slide, time_based=False, delay=0, max_keys=1 << 16, batch_len=65536, out_capacity=0):
    """Run the keyed FlatFAT engine (Key_FFAT_GPU) over a host array of tuples; returns (results, counters)."""
    t = np.ascontiguousarray(tuples_host, dtype=np.int64)
    return load_ext().key_ffat_run(t, int(win), int(slide), int(delay), bool(time_based), int(max_keys), int(batch_len), int(out_capacity))
