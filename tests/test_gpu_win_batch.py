"""WinBatchEngine (device-resident non-incremental windows: Win_Seq_GPU / Key_Farm_GPU / Win_Farm_GPU) driven directly,
against NumPy oracles; and the batch-fed Key_Farm_GPU pipeline end to end (GPU only).

The reference has no numeric test of K1 (src/mp_test_gpu only compares runs with each other); the oracle here is the
definition: window lw of a key holds the tuples whose ordinal (per-key progressive id / timestamp) lies in
[lw*slide, lw*slide + win), every opened window is emitted (partial ones at end of stream), count-based results carry the
timestamp of their last tuple and time-based ones the last instant of the window."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_stream(n, n_keys, seed, ts_step=1, skew=False, per_key_ids=True):
    rng = np.random.default_rng(seed)
    if skew:
        keys = np.minimum(rng.zipf(1.3, n) - 1, n_keys - 1).astype(np.int64)
    else:
        keys = rng.integers(0, n_keys, n).astype(np.int64)
    t = np.zeros((n, 5), dtype=np.int64)
    t[:, 0] = keys * 7919 + 3
    if per_key_ids:  # progressive ids per key, in arrival order (what an ordering node's renumbering produces)
        counts = {}
        ids = np.zeros(n, dtype=np.int64)
        for i, k in enumerate(keys):
            ids[i] = counts.get(k, 0)
            counts[k] = ids[i] + 1
        t[:, 1] = ids
    else:
        t[:, 1] = 1_000_000 - np.arange(n)  # garbage ids: only arrival order counts (renumbering on the device)
    t[:, 2] = np.arange(n) * ts_step
    vals = rng.integers(-50, 50, n).astype(np.float64)
    t[:, 3] = vals.view(np.int64)
    t[:, 4] = np.float64(1.0).view(np.int64)
    return t, vals


def fold(w, mode):
    if mode == 0:
        return float(w.sum())
    if mode == 1:
        return float(w[-1]) if len(w) else -1.0
    if mode == 2:
        return float(w[0]) if len(w) else -1.0
    return float(len(w))


def oracle(t, vals, win, slide, tb, mode, reduce_form=False):
    """(key, lwid) -> (value, ts) for every opened window"""
    out = {}
    for key in np.unique(t[:, 0]):
        m = t[:, 0] == key
        v, ts = vals[m], t[m, 2]
        if tb:
            order = np.argsort(ts, kind="stable")
            v, ts = v[order], ts[order]
            ords = ts
        else:
            ords = np.arange(len(v))
        n_win = int(ords.max()) // slide + 1
        for lw in range(n_win):
            a, b = lw * slide, lw * slide + win
            sel = (ords >= a) & (ords < b)
            w = v[sel]
            if reduce_form and len(w) == 0:
                val = 0.0  # wave_reduce leaves the default result for an empty window
            else:
                val = fold(w, mode)
            stamp = b - 1 if tb else (int(ts[sel][-1]) if sel.any() else 0)
            out[(int(key), lw)] = (val, stamp)
    return out


def unpack(res):
    vals = res[:, 3].copy().view(np.float64)
    return {(int(k), int(i)): (float(v), int(ts)) for k, i, ts, v in zip(res[:, 0], res[:, 1], res[:, 2], vals)}


def check(res, ctr, exp):
    assert ctr["table_full"] == 0 and ctr["carry_overflow"] == 0 and ctr["work_overflow"] == 0
    got = unpack(res)
    assert len(got) == len(res), "duplicate (key, window) results"
    assert len(got) == len(exp)
    for k, (v, ts) in exp.items():
        assert got[k][0] == pytest.approx(v, abs=1e-9), k
        assert got[k][1] == ts, k


@pytest.mark.parametrize("n,n_keys,win,slide,batch,mode", [
    (5000, 7, 10, 3, 512, 0),
    (20000, 1, 64, 16, 4096, 1),       # a single key: one run per batch, thousands of windows
    (50000, 1000, 12, 4, 8192, 0),     # many keys, a few tuples of each per batch
    (30000, 5000, 5, 1, 4096, 2),
    (40000, 50, 100, 7, 65536, 0),     # the whole stream in one batch
    (30000, 3, 1000, 999, 2048, 3),    # long windows that span many batches
    (20000, 4, 7, 7, 1000, 0),         # tumbling
    (20000, 5, 3, 10, 777, 3),         # hopping: 7 of 10 tuples fall in the gaps
    (60000, 3000, 200, 50, 4096, 0),   # one or two tuples of a key per batch, a window every ~35 batches: runs appended in place
    (60000, 3000, 64, 1, 4096, 3),     # ... and a window per tuple once the first one is full
])
def test_count_based_windows_match_oracle(ext, n, n_keys, win, slide, batch, mode):
    t, vals = make_stream(n, n_keys, seed=n + win)
    res, ctr = ext.win_batch_run(t, win, slide, 0, False, max(16, n_keys), batch, 0, 0, mode)
    assert ctr["dropped"] == 0
    check(res, ctr, oracle(t, vals, win, slide, False, mode))


@pytest.mark.parametrize("n,n_keys,win,slide,batch", [(20000, 9, 10, 3, 512), (30000, 1, 256, 32, 4096), (30000, 2000, 6, 2, 8192),
                                                      (60000, 2500, 300, 100, 2048)])
def test_count_based_renumbering_ignores_ids(ext, n, n_keys, win, slide, batch):
    """batch-fed streams: the tuples of a key are numbered by arrival on the device"""
    t, vals = make_stream(n, n_keys, seed=n, per_key_ids=False)
    res, ctr = ext.win_batch_run(t, win, slide, 0, False, max(16, n_keys), batch, 0, 0, 1, False, True)
    check(res, ctr, oracle(t, vals, win, slide, False, 1))


@pytest.mark.parametrize("n,n_keys,win,slide,batch,ts_step", [
    (5000, 5, 100, 25, 512, 1),
    (30000, 300, 4000, 1000, 4096, 1),
    (20000, 40, 900, 600, 8192, 3),
    (20000, 6, 300, 1000, 1500, 1),   # hopping
    (20000, 3, 500, 500, 2048, 2),    # tumbling
    (60000, 2000, 6000, 2000, 2048, 1),  # a tuple of a key per batch, a window every 60 batches or so
])
def test_time_based_windows_match_oracle(ext, n, n_keys, win, slide, batch, ts_step):
    t, vals = make_stream(n, n_keys, seed=n + slide, ts_step=ts_step)
    res, ctr = ext.win_batch_run(t, win, slide, 0, True, max(16, n_keys), batch, 8192, 1 << 16, 0)
    assert ctr["dropped"] == 0
    check(res, ctr, oracle(t, vals, win, slide, True, 0))


def test_time_based_out_of_order_within_delay(ext):
    """arrival order differs from timestamp order by less than the triggering delay: nothing is lost, windows see the tuples in
    timestamp order (newest-value function)"""
    n, n_keys, win, slide, delay = 20000, 4, 60, 20, 16
    t, vals = make_stream(n, n_keys, seed=77)
    rng = np.random.default_rng(5)
    t[:, 2] = np.arange(n) + rng.integers(0, 8, n)
    # distinct timestamps inside a key so that "newest" is well defined
    t[:, 2] = t[:, 2] * 4 + (np.arange(n) % 4)
    res, ctr = ext.win_batch_run(t, 4 * win, 4 * slide, 4 * delay, True, 16, 4096, 4096, 1 << 16, 1)
    assert ctr["dropped"] == 0
    check(res, ctr, oracle(t, vals, 4 * win, 4 * slide, True, 1))


@pytest.mark.parametrize("presort", ["0", "1", "2"])
@pytest.mark.parametrize("n_keys,lanes,lag", [(4, 4, 300), (1, 3, 2000), (50, 8, 100)])
def test_time_based_merged_lanes_out_of_order(ext, n_keys, lanes, lag, presort, monkeypatch):
    """The stream a replica sees behind the KeyBy of a DEFAULT graph: every batch is a concatenation of `lanes` pieces, each in
    timestamp order, that lag each other - every run is out of order and overlaps the tuples carried from earlier batches.
    With the (slot, timestamp) pre-sort (2: always, 1: once the device has reported disorder) runs are merged with the carry;
    without it (0) they are repaired by the sequential fallback.  Same windows either way (newest-value function: the
    windows must see their tuples in timestamp order)."""
    monkeypatch.setenv("WF_TS_PRESORT", presort)
    n, batch, win, slide = 24000, 4096, 400, 100
    rng = np.random.default_rng(11 + n_keys)
    per = batch // lanes
    t = np.zeros((n, 5), dtype=np.int64)
    vals = rng.integers(-50, 50, n).astype(np.float64)
    clock = np.zeros(lanes, dtype=np.int64) + np.arange(lanes) * lag // max(1, lanes - 1)
    i = 0
    while i < n:
        for l in range(lanes):
            m = min(per, n - i)
            if m <= 0:
                break
            t[i:i + m, 2] = (clock[l] + np.arange(m)) * lanes + l  # distinct timestamps, every lane ascending
            clock[l] += m
            i += m
    t[:, 0] = rng.integers(0, n_keys, n) * 7919 + 3
    t[:, 3] = vals.view(np.int64)
    delay = 2 * (batch + lag * lanes)  # covers the disorder: nothing is late, windows fire all along the stream
    res, ctr = ext.win_batch_run(t, win, slide, delay, True, max(16, n_keys), batch, 1 << 16, 1 << 16, 1)
    assert ctr["dropped"] == 0
    check(res, ctr, oracle(t, vals, win, slide, True, 1))


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("n,n_keys,win,slide,batch", [(30000, 3, 300, 40, 4096), (40000, 500, 24, 6, 8192), (20000, 1, 1000, 100, 2048)])
def test_wave_reduce_form_matches_oracle(ext, n, n_keys, win, slide, batch, mode):
    """wave_reduce(lift, combine): a wave64 per window, order-preserving (newest / oldest are not commutative)"""
    t, vals = make_stream(n, n_keys, seed=n + mode)
    res, ctr = ext.win_batch_run(t, win, slide, 0, False, max(16, n_keys), batch, 0, 0, mode, True)
    check(res, ctr, oracle(t, vals, win, slide, False, mode))


def test_output_batch_smaller_than_fired_windows_defers(ext):
    """a burst of windows larger than the output batch (a long silence on time-based windows fires every empty window it
    spans at once): the excess is deferred to the following batches, not lost"""
    n = 40000
    t, vals = make_stream(n, 2, seed=9)
    t[n // 2:, 2] += 50000
    res, ctr = ext.win_batch_run(t, 8, 1, 0, True, 16, 1024, 32768, 4096, 0)
    assert ctr["deferred"] > 0 and ctr["dropped"] == 0
    check(res, ctr, oracle(t, vals, 8, 1, True, 0))


def test_skewed_keys(ext):
    t, vals = make_stream(60000, 2000, seed=5, skew=True)
    res, ctr = ext.win_batch_run(t, 20, 5, 0, False, 4096, 16384, 0, 0, 0)
    check(res, ctr, oracle(t, vals, 20, 5, False, 0))


def _fuzz_cases(n_cases, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n_cases):
        tb = bool(rng.integers(0, 2))
        slide = int(rng.integers(1, 60))
        kind = int(rng.integers(0, 4))
        win = slide + int(rng.integers(1, 300)) if kind < 2 else (slide if kind == 2 else max(1, slide - int(rng.integers(1, slide + 1))))
        n_keys = int(rng.choice([1, 2, 5, 40, 300, 2500]))
        n = int(rng.integers(2000, 30000))
        batch = int(rng.choice([257, 1000, 2048, 4096, 16384]))
        mode = int(rng.choice([0, 1, 2, 3]))
        reduce_form = bool(rng.integers(0, 2)) and mode < 3
        # a time-based key fires a window per slide of TIME up to its last tuple: many keys x a small slide would ask the Python
        # oracle for hundreds of millions of windows (timestamps advance by up to 3 per tuple of the stream)
        while tb and n_keys > 1 and n_keys * (3 * n // slide + 1) > 5_000_000:
            n_keys = max(1, n_keys // 8)
        cases.append((i, tb, n, n_keys, win, slide, batch, mode, reduce_form))
    return cases


import os as _os


@pytest.mark.parametrize("case", _fuzz_cases(int(_os.environ.get("WF_FUZZ_CASES", "30")), seed=int(_os.environ.get("WF_WB_FUZZ_SEED", "20260202"))),
                         ids=lambda c: "case%d" % c[0])
def test_win_batch_fuzz(ext, case):
    i, tb, n, n_keys, win, slide, batch, mode, reduce_form = case
    t, vals = make_stream(n, n_keys, seed=3000 + i, ts_step=int(1 + i % 3))
    res, ctr = ext.win_batch_run(t, win, slide, 0, tb, max(16, n_keys), batch, 1 << 14, 1 << 17, mode, reduce_form)
    check(res, ctr, oracle(t, vals, win, slide, tb, mode, reduce_form))


@pytest.mark.parametrize("shared,replicas,reduce_form", [(False, 0, False), (True, 1, False), (True, 2, True), (False, 0, True)])
def test_win_seq_pipeline_end_to_end(ext, shared, replicas, reduce_form):
    """Batch_Source x2 -> [device KeyBy] -> Key_Farm_GPU -> Batch_Sink: count-based windows, arrivals numbered on the device"""
    n_keys, win, slide = (40, 16, 4)
    pipe = ext.WinSeqPipeline(0, 2, 4096, n_keys, win, slide, False, 4, replicas, reduce_form, 0, 1.3, shared)
    per_source = 40_000
    pipe.step(per_source)
    pipe.step(per_source)
    pipe.close()
    st = pipe.stats()
    assert st["table_full"] == 0 and st["carry_overflow"] == 0
    if not shared:
        # every source owns n_keys keys, 2 * per_source tuples round-robin over them
        per_key = 2 * per_source // n_keys
        assert st["sink_tuples"] == 2 * n_keys * math.ceil(per_key / slide)
        vals = (np.arange(2 * per_source) & 1023).astype(np.float64)
        total = 0.0
        for k in range(n_keys):
            v = vals[k::n_keys]
            c = np.concatenate([[0.0], np.cumsum(v)])
            for lw in range(math.ceil(per_key / slide)):
                total += c[min(per_key, slide * lw + win)] - c[slide * lw]
        assert st["sink_value_sum"] == pytest.approx(2 * total, rel=1e-12)
    else:
        # both sources emit the same n_keys keys: every key receives 2 * (2 * per_source / n_keys) tuples, interleaved in an
        # arrival order the test does not control - the number of windows does not depend on it
        per_key = 2 * (2 * per_source // n_keys)
        assert st["sink_tuples"] == n_keys * math.ceil(per_key / slide)
