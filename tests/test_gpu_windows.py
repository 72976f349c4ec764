"""Device primitives and the keyed FlatFAT engine against PyTorch/NumPy oracles (GPU only)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 2047, 2048, 2049, 100_000, 1_000_003])
def test_exclusive_scan_matches_torch(ext, n):
    g = torch.Generator().manual_seed(n)
    x = torch.randint(0, 1000, (n,), generator=g, dtype=torch.int32)
    d = x.cuda()
    out = torch.empty_like(d)
    total = ext.exclusive_scan(d.data_ptr(), out.data_ptr(), n)
    torch.cuda.synchronize()
    ref = torch.cumsum(x.to(torch.int64), 0) - x
    assert total == int(x.sum())
    assert torch.equal(out.cpu().to(torch.int64), ref)


@pytest.mark.parametrize("n,bits", [(1, 8), (1000, 8), (2049, 11), (20_000, 16), (30_000, 17), (32_768, 22), (32_769, 11), (40_000, 8),
                                    (100_000, 17), (1_000_000, 21), (300_000, 32)])
def test_radix_sort_pairs_is_stable_and_sorted(ext, n, bits):
    g = torch.Generator().manual_seed(bits * 7 + n)
    hi = (1 << bits) if bits < 31 else (1 << 31) - 1
    keys = torch.randint(0, hi, (n,), generator=g, dtype=torch.int64)
    if bits >= 32:
        keys = keys * 2 + torch.randint(0, 2, (n,), generator=g, dtype=torch.int64)
    vals = torch.arange(n, dtype=torch.int64)
    dk = keys.to(torch.int64).cuda().to(torch.uint32) if hasattr(torch, "uint32") else None
    # uint32 tensors: go through int32 bit patterns
    dk = torch.from_numpy(keys.numpy().astype(np.uint32).view(np.int32)).cuda()
    dv = torch.from_numpy(vals.numpy().astype(np.uint32).view(np.int32)).cuda()
    ext.radix_sort_pairs(dk.data_ptr(), dv.data_ptr(), n, bits)
    torch.cuda.synchronize()
    sk = dk.cpu().numpy().view(np.uint32).astype(np.int64)
    sv = dv.cpu().numpy().view(np.uint32).astype(np.int64)
    order = np.argsort(keys.numpy(), kind="stable")
    assert np.array_equal(sk, keys.numpy()[order])
    assert np.array_equal(sv, order)  # stability: equal keys keep arrival order


@pytest.mark.parametrize("n,bits,mode", [(1, 8, 1), (2, 11, 1), (1000, 19, 1), (10_000, 12, 1), (16_384, 19, 1), (1, 8, 2), (1023, 8, 2), (1025, 9, 2),
                                         (10_000, 19, 2), (131_072, 19, 2), (200_001, 24, 2), (50_000, 32, 2), (5000, 19, 3), (100_000, 19, 3)])
def test_one_workgroup_sorts_behind_a_gate(ext, n, bits, mode):
    """The sorts that run behind a SortGate: the bitonic network for small batches (mode 1), the one-workgroup radix sort of the
    rescue path (mode 2) - both only execute when a batch that was expected to hold no key twice does - and the closed gate (3)."""
    g = torch.Generator().manual_seed(bits * 11 + n + mode)
    hi = (1 << bits) if bits < 31 else (1 << 31) - 1
    keys = torch.randint(0, hi, (n,), generator=g, dtype=torch.int64)
    if n > 100:
        keys[::7] = keys[3]  # plenty of equal keys: stability matters
    if bits >= 32:
        keys = keys * 2 + torch.randint(0, 2, (n,), generator=g, dtype=torch.int64)
    # values are NOT positions (the network sorts (key, position) words and gathers the values afterwards)
    vals = torch.arange(n, dtype=torch.int64) * 3 + 5
    dk = torch.from_numpy(keys.numpy().astype(np.uint32).view(np.int32)).cuda()
    dv = torch.from_numpy(vals.numpy().astype(np.uint32).view(np.int32)).cuda()
    ext.radix_sort_pairs(dk.data_ptr(), dv.data_ptr(), n, bits, mode)
    torch.cuda.synchronize()
    sk = dk.cpu().numpy().view(np.uint32).astype(np.int64)
    sv = dv.cpu().numpy().view(np.uint32).astype(np.int64)
    if mode == 3:
        assert np.array_equal(sk, keys.numpy()) and np.array_equal(sv, vals.numpy())
        return
    order = np.argsort(keys.numpy(), kind="stable")
    assert np.array_equal(sk, keys.numpy()[order])
    assert np.array_equal(sv, vals.numpy()[order])


def make_stream(n, n_keys, seed, ts_step=1, skew=False):
    rng = np.random.default_rng(seed)
    if skew:
        keys = np.minimum(rng.zipf(1.3, n) - 1, n_keys - 1).astype(np.int64)
    else:
        keys = rng.integers(0, n_keys, n).astype(np.int64)
    t = np.zeros((n, 5), dtype=np.int64)
    t[:, 0] = keys * 7919 + 3  # sparse key values
    t[:, 1] = np.arange(n)
    t[:, 2] = np.arange(n) * ts_step
    vals = rng.integers(-50, 50, n).astype(np.float64)
    t[:, 3] = vals.view(np.int64)
    t[:, 4] = np.float64(1.0).view(np.int64)
    return t, vals


def oracle_cb(t, vals, win, slide):
    """(key, lwid) -> (sum, last ts); every opened window, partial ones at end of stream."""
    out = {}
    for key in np.unique(t[:, 0]):
        m = t[:, 0] == key
        v, ts = vals[m], t[m, 2]
        n = len(v)
        lw = 0
        while lw * slide < n:
            a, b = lw * slide, min(n, lw * slide + win)
            out[(int(key), lw)] = (float(v[a:b].sum()), int(ts[b - 1]))
            lw += 1
    return out


def oracle_tb(t, vals, win, slide):
    q = math.gcd(win, slide)
    sp = slide // q
    out = {}
    for key in np.unique(t[:, 0]):
        m = t[:, 0] == key
        v, ts = vals[m], t[m, 2]
        n_panes = int(ts.max()) // q + 1
        lw = 0
        while lw * sp < n_panes:
            a, b = lw * slide, lw * slide + win
            sel = (ts >= a) & (ts < b)
            out[(int(key), lw)] = float(v[sel].sum())
            lw += 1
    return out


def unpack(res):
    vals = res[:, 3].copy().view(np.float64)
    return {(int(k), int(i)): (float(v), int(ts)) for k, i, ts, v in zip(res[:, 0], res[:, 1], res[:, 2], vals)}


@pytest.mark.parametrize("n,n_keys,win,slide,batch", [
    (5000, 7, 10, 3, 512),
    (20000, 1, 64, 16, 4096),
    (50000, 1000, 12, 4, 8192),
    (30000, 5000, 5, 1, 4096),
    (40000, 50, 100, 7, 65536),
])
def test_key_ffat_count_based_matches_oracle(ext, n, n_keys, win, slide, batch):
    t, vals = make_stream(n, n_keys, seed=n + win)
    res, ctr = ext.key_ffat_run(t, win, slide, 0, False, max(16, n_keys), batch)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0 and ctr["dropped"] == 0
    got = unpack(res)
    exp = oracle_cb(t, vals, win, slide)
    assert len(res) == len(got) == len(exp)
    for k, (s, ts) in exp.items():
        assert got[k][0] == pytest.approx(s, abs=1e-9)
        assert got[k][1] == ts


@pytest.mark.parametrize("n,n_keys,win,slide,batch,ts_step", [
    (5000, 5, 100, 25, 512, 1),
    (30000, 300, 4000, 1000, 4096, 1),
    (20000, 40, 900, 600, 8192, 3),
])
def test_key_ffat_time_based_matches_oracle(ext, n, n_keys, win, slide, batch, ts_step):
    t, vals = make_stream(n, n_keys, seed=n + slide, ts_step=ts_step)
    res, ctr = ext.key_ffat_run(t, win, slide, 0, True, max(16, n_keys), batch, 4 * batch)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0 and ctr["dropped"] == 0 and ctr["skipped_windows"] == 0
    got = unpack(res)
    exp = oracle_tb(t, vals, win, slide)
    assert len(res) == len(got) == len(exp)
    for k, s in exp.items():
        assert got[k][0] == pytest.approx(s, abs=1e-9)


@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("n,n_keys,win,slide,batch", [
    (30000, 100, 10, 10, 4096),    # tumbling
    (30000, 3, 64, 64, 2048),      # tumbling, long runs (wave-per-run path)
    (40000, 700, 4, 12, 8192),     # hopping: pane 4, a window every 3 panes
    (30000, 2, 6, 10, 1000),       # hopping, pane 2, long runs
    (20000, 5, 1, 1, 512),         # every tuple its own window
])
def test_key_ffat_tumbling_and_hopping(ext, tb, n, n_keys, win, slide, batch):
    """slide >= win (the reference's Key_FFAT_GPU refuses them, key_ffat.hpp:132-135): pane = gcd, the tree is one pane wide"""
    if tb:
        n_keys = min(n_keys, 70)  # timestamps are global: with many keys every tuple would span hundreds of (empty) windows of its key
    t, vals = make_stream(n, n_keys, seed=n + win + slide)
    res, ctr = ext.key_ffat_run(t, win, slide, 0, tb, max(16, n_keys), batch, 1 << 17)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0 and ctr["dropped"] == 0
    got = unpack(res)
    assert len(got) == len(res)
    n_exp = 0
    for key in np.unique(t[:, 0]):
        m = t[:, 0] == key
        v, ts = vals[m], t[m, 2]
        ords = ts if tb else np.arange(len(v))
        lw = 0
        while lw * slide <= int(ords.max()):
            sel = (ords >= lw * slide) & (ords < lw * slide + win)
            assert got[(int(key), lw)][0] == pytest.approx(float(v[sel].sum()), abs=1e-9), (key, lw)
            lw += 1
            n_exp += 1
    assert len(got) == n_exp


def test_key_ffat_skewed_keys(ext):
    t, vals = make_stream(60000, 2000, seed=5, skew=True)
    res, ctr = ext.key_ffat_run(t, 20, 5, 0, False, 4096, 16384)
    got = unpack(res)
    exp = oracle_cb(t, vals, 20, 5)
    assert len(got) == len(exp)
    for k, (s, _) in exp.items():
        assert got[k][0] == pytest.approx(s, abs=1e-9)


def test_win_agg_pipeline_end_to_end(ext):
    # 2 sources x 50 keys each, count-based window 8 slide 2 over values seq & 1023
    pipe = ext.WinAggPipeline(0, 2, 4096, 50, 8, 2, False, 0, 4)
    per_source = 40_000
    pipe.step(per_source)
    pipe.step(per_source)
    st = pipe.stats()
    pipe.close()
    st2 = pipe.stats()
    n = 2 * per_source  # tuples per source, round-robin over 50 keys -> 1600 per key
    per_key = n // 50
    fired = (per_key - 8) // 2 + 1  # complete windows per key
    assert st["sink_tuples"] == 2 * 50 * fired
    assert st2["sink_tuples"] == 2 * 50 * math.ceil(per_key / 2)  # + partial windows flushed at end of stream
    # exact sum: window lw of key k (source-local index) covers its elements [2lw, 2lw+8)
    seq = np.arange(n)
    vals = (seq & 1023).astype(np.float64)
    total = 0.0
    for k in range(50):
        v = vals[k::50]
        c = np.concatenate([[0.0], np.cumsum(v)])
        for lw in range(math.ceil(per_key / 2)):
            total += c[min(per_key, 2 * lw + 8)] - c[2 * lw]
    assert st2["sink_value_sum"] == pytest.approx(2 * total, rel=1e-12)
    assert st2["overflow"] == 0 and st2["table_full"] == 0


@pytest.mark.parametrize("keys", [64, 700])
def test_win_agg_pipeline_shared_keys_replica_counts_agree(ext, keys):
    """Every source emits every key (time-based windows, a delay that covers the lag the sources are allowed): the default plan -
    ONE replica without a device KeyBy for key spaces of at most 512 keys, four behind the KeyBy above - fires the same windows
    with the same sums as an explicit two-replica plan.  (bench.py --shared-keys; profiles/r3_shared_keys/replicas.md)"""
    n_sources, per_source, win, slide, delay = 3, 120_000, 4096, 1024, 40 * 8192

    def run(replicas):
        pipe = ext.WinAggPipeline(0, n_sources, 8192, keys, win, slide, True, delay, 4, None, replicas, 0, 1.3, True)
        pipe.step(per_source)
        pipe.step(per_source)
        pipe.close()
        return pipe.stats()

    default, two = run(0), run(2)
    for st in (default, two):
        assert st["dropped"] == 0 and st["overflow"] == 0 and st["table_full"] == 0
    assert default["sink_tuples"] == two["sink_tuples"] > 0
    assert default["sink_value_sum"] == two["sink_value_sum"]  # (integers in doubles: exact)


def test_win_agg_pipeline_shared_keys_single_replica_in_a_deterministic_graph(ext):
    """DETERMINISTIC graph, count-based windows over shared keys: the lanes are merged in timestamp order in front of the
    replica(s), so one replica and two fire the same windows with the same sums (and a step ends on ONE merged punctuation)"""

    def run(replicas):
        pipe = ext.WinAggPipeline(0, 3, 8192, 64, 16, 4, False, 0, 4, None, replicas, 0, 1.3, True, True)
        pipe.step(60_000)
        pipe.step(60_000)
        pipe.close()
        return pipe.stats()

    one, two = run(1), run(2)
    for st in (one, two):
        assert st["dropped"] == 0 and st["overflow"] == 0 and st["table_full"] == 0
    assert one["sink_tuples"] == two["sink_tuples"] > 0
    assert one["sink_value_sum"] == two["sink_value_sum"]


def test_key_ffat_with_hip_graph_replay(ext, monkeypatch):
    """WF_HIP_GRAPHS=1: the per-batch kernel sequence is captured once per slab pair and replayed."""
    monkeypatch.setenv("WF_HIP_GRAPHS", "1")
    t, vals = make_stream(60000, 500, seed=11)
    res, ctr = ext.key_ffat_run(t, 12, 4, 0, False, 1024, 4096)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0
    got = unpack(res)
    exp = oracle_cb(t, vals, 12, 4)
    assert len(got) == len(exp)
    for k, (s, ts) in exp.items():
        assert got[k][0] == pytest.approx(s, abs=1e-9) and got[k][1] == ts


# ---- the wave-per-run paths (few keys: long runs of one key inside a batch) --------------------------------------

@pytest.mark.parametrize("n,n_keys,win,slide,batch,mode", [
    (40000, 1, 12, 4, 4096, 1),     # pane = 4 tuples, bulk path, newest value wins
    (40000, 3, 9, 3, 2048, 2),      # oldest value wins
    (30000, 2, 7, 1, 1000, 1),      # pane = 1 tuple: 64 panes per chunk; batch not a multiple of the wave
    (50000, 5, 192, 3, 8192, 1),    # 64 panes per window: the largest window the bulk path takes
    (30000, 2, 300, 4, 4096, 2),    # 75 panes per window: staged tree, sequential per segment
    (20000, 4, 1000, 999, 4096, 0), # pane = 1 tuple, 1000 panes per window
])
def test_key_ffat_long_runs_keep_order(ext, n, n_keys, win, slide, batch, mode):
    t, vals = make_stream(n, n_keys, seed=n + win + mode)
    res, ctr = ext.key_ffat_run(t, win, slide, 0, False, 16, batch, 4 * batch, mode)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0
    got = unpack(res)
    for key in np.unique(t[:, 0]):
        v = vals[t[:, 0] == key]
        lw = 0
        while lw * slide < len(v):
            w = v[lw * slide:lw * slide + win]
            exp = w.sum() if mode == 0 else (w[-1] if mode == 1 else w[0])
            assert got[(int(key), lw)][0] == pytest.approx(exp, abs=1e-9), (key, lw)
            lw += 1
    assert len(got) == sum(math.ceil((t[:, 0] == k).sum() / slide) for k in np.unique(t[:, 0]))


def bursty_stream(n, n_keys, seed, gap_every, gap_len, disorder):
    """Per-key timestamps that advance by 1, jump by gap_len every gap_every tuples, with bounded local disorder."""
    rng = np.random.default_rng(seed + 7919)  # (not the seed of make_stream: the disorder would be a function of the key)
    t, vals = make_stream(n, n_keys, seed)
    ts = np.zeros(n, dtype=np.int64)
    clock = {}
    count = {}
    for i in range(n):
        k = t[i, 0]
        c = count.get(k, 0)
        clock[k] = clock.get(k, 0) + (gap_len if c and c % gap_every == 0 else 1)
        count[k] = c + 1
        ts[i] = clock[k]
    if disorder:
        ts = ts + rng.integers(0, disorder + 1, n)  # arrival order no longer matches timestamp order
    t[:, 2] = ts
    return t, vals


@pytest.mark.parametrize("n,n_keys,win,slide,batch,gap_every,gap_len,disorder,delay", [
    (30000, 2, 40, 10, 4096, 500, 35, 0, 0),       # short silences: empty panes inside bulk chunks
    (30000, 3, 40, 10, 2048, 700, 5000, 0, 0),     # long silences: more than 64 empty panes, sequential fallback
    (20000, 2, 60, 20, 4096, 10**9, 1, 7, 16),     # out-of-order arrivals within the triggering delay
    (20000, 1, 8, 2, 512, 300, 90, 0, 0),
])
def test_key_ffat_time_based_long_runs(ext, n, n_keys, win, slide, batch, gap_every, gap_len, disorder, delay):
    t, vals = bursty_stream(n, n_keys, seed=n + win, gap_every=gap_every, gap_len=gap_len, disorder=disorder)
    res, ctr = ext.key_ffat_run(t, win, slide, delay, True, 16, batch, 1 << 18)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0 and ctr["dropped"] == 0 and ctr["skipped_windows"] == 0
    got = unpack(res)
    exp = oracle_tb(t, vals, win, slide)
    assert len(res) == len(got) == len(exp)
    for k, s in exp.items():
        assert got[k][0] == pytest.approx(s, abs=1e-9), k


@pytest.mark.parametrize("mode", [0, 1, 2])  # sum / newest arrival wins / oldest arrival wins
@pytest.mark.parametrize("n_keys,disorder,delay,gap_every,gap_len", [
    (2, 30, 50, 10**9, 1),     # disorder inside the triggering delay: nothing is dropped, several quanta stay open
    (2, 30, 10, 10**9, 1),     # ... larger than the delay: late tuples are dropped, by the rules of the tuple-at-a-time engine
    (3, 40, 0, 10**9, 1),      # no delay at all
    (40, 25, 30, 300, 200),    # mid-length runs, silences of 20 panes
    (1, 0, 64, 10**9, 1),      # in order, but windows fire 64 time units late
    (5, 12, 1000, 700, 90),    # a delay of 100 panes: a large ring of open quanta
    (3, 900, 20000, 10**9, 1), # ... of 2000 panes: the ring does not fit the LDS and is updated in place
    (2, 900, 500, 10**9, 1),   # ... disorder of 90 panes, most of it beyond the delay
])
def test_key_ffat_time_based_bulk_path_follows_the_sequential_rules(ext, n_keys, disorder, delay, gap_every, gap_len, mode, monkeypatch):
    """Long runs are processed 64 tuples at a time by a wave (prefix maxima of the watermark decide which tuples are late,
    ordered folds per quantum); runs shorter than 12 tuples take the tuple-at-a-time rules.  Same stream, same results,
    same number of dropped tuples - also for combines that depend on the arrival order.  (The (key, pane) pre-sort of
    disordered batches changes what "arrival order" means inside a batch and has a test of its own below.)"""
    monkeypatch.setenv("WF_TS_PRESORT", "0")
    n = 24000
    t, vals = bursty_stream(n, n_keys, seed=77 + n_keys + delay, gap_every=gap_every, gap_len=gap_len, disorder=disorder)
    res_a, ctr_a = ext.key_ffat_run(t, 80, 10, delay, True, 64, 8192, 1 << 18, mode)
    res_b, ctr_b = ext.key_ffat_run(t, 80, 10, delay, True, 64, 8 * n_keys, 1 << 18, mode)
    assert ctr_a["overflow"] == 0 and ctr_b["overflow"] == 0
    # the late tuples by the rules of the tuple-at-a-time engine (win_seqffat.hpp TB path): a quantum is complete once a
    # tuple `delay` later than its end has been seen, and the ring of open quanta holds delay / pane + 2 of them
    pane, late = math.gcd(80, 10), 0
    ring = 2
    while ring < delay // pane + 2:
        ring *= 2
    for key in np.unique(t[:, 0]):
        oldest_open = 0
        for ts in t[t[:, 0] == key, 2]:
            q = int(ts) // pane
            if q < oldest_open:
                late += 1
                continue
            done = (int(ts) - delay) // pane if ts > delay else 0
            oldest_open = max(oldest_open, done, q - ring + 1)
    assert ctr_a["dropped"] == ctr_b["dropped"] == late
    a, b = unpack(res_a), unpack(res_b)
    assert len(a) == len(res_a) and a.keys() == b.keys()
    for k in a:
        assert a[k][0] == pytest.approx(b[k][0], abs=1e-9), k
        assert a[k][1] == b[k][1], k


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("n_keys,disorder,delay", [(2, 30, 50), (2, 30, 10), (3, 40, 0), (1, 900, 500), (6, 200, 100)])
def test_key_ffat_presort_treats_a_batch_as_arrived_in_pane_order(ext, n_keys, disorder, delay, mode, monkeypatch):
    """Batches whose long runs are out of timestamp order are ordered by (key, pane) on the device before the engine sees them
    (primitives.hpp TsPresorter; here forced on): same results and same late tuples as the tuple-at-a-time rules applied to
    the stream in which every batch has been stably sorted by pane - arrival order inside a pane is kept, which the
    order-sensitive combines (newest / oldest arrival wins) check."""
    n, batch, win, slide = 24000, 4096, 80, 10
    pane = math.gcd(win, slide)
    t, vals = bursty_stream(n, n_keys, seed=501 + n_keys + delay, gap_every=10**9, gap_len=1, disorder=disorder)
    monkeypatch.setenv("WF_TS_PRESORT", "2")
    res_a, ctr_a = ext.key_ffat_run(t, win, slide, delay, True, 64, batch, 1 << 18, mode)
    # the reference stream: each batch sorted by pane (stable), fed tuple by tuple without any pre-sort
    t_ref = t.copy()
    for off in range(0, n, batch):
        blk = t_ref[off:off + batch]
        t_ref[off:off + batch] = blk[np.argsort(blk[:, 2] // pane, kind="stable")]
    monkeypatch.setenv("WF_TS_PRESORT", "0")
    res_b, ctr_b = ext.key_ffat_run(t_ref, win, slide, delay, True, 64, 8 * n_keys, 1 << 18, mode)
    assert ctr_a["overflow"] == 0 and ctr_b["overflow"] == 0
    assert ctr_a["dropped"] == ctr_b["dropped"]
    a, b = unpack(res_a), unpack(res_b)
    assert len(a) == len(res_a) and a.keys() == b.keys()
    for k in a:
        assert a[k][0] == pytest.approx(b[k][0], abs=1e-9), k
        assert a[k][1] == b[k][1], k


@pytest.mark.parametrize("n_keys,batch", [(3, 2048), (400, 4096)])
def test_key_ffat_skips_long_empty_stretches_when_asked(ext, n_keys, batch):
    """skip_gaps: a key silent for much longer than a window does not emit one empty result per slide."""
    win, slide, gap = 40, 10, 100_000
    t, vals = bursty_stream(12000, n_keys, seed=3, gap_every=1500 if n_keys < 10 else 10, gap_len=gap, disorder=0)
    res, ctr = ext.key_ffat_run(t, win, slide, 0, True, 1024, batch, 1 << 18, 0, True)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0 and ctr["dropped"] == 0
    assert ctr["skipped_windows"] > 0
    got = unpack(res)
    # sparse oracle: only the windows that hold data are enumerated (the stream spans millions of windows)
    q = math.gcd(win, slide)
    with_data, total_windows = {}, 0
    for key in np.unique(t[:, 0]):
        m = t[:, 0] == key
        for ts, v in zip(t[m, 2], vals[m]):
            for lw in range(max(0, -(-(int(ts) - win + 1) // slide)), int(ts) // slide + 1):
                with_data[(int(key), lw)] = with_data.get((int(key), lw), 0.0) + v
        n_panes = int(t[m, 2].max()) // q + 1
        total_windows += -(-n_panes // (slide // q))
    # every window that holds data is there with the right value; what was skipped was empty
    for k, s in with_data.items():
        assert k in got and got[k][0] == pytest.approx(s, abs=1e-9), k
    assert len(got) + ctr["skipped_windows"] == total_windows
    assert len(got) < total_windows // 10


# ---- randomised cross-check of every engine path against the oracles ----------------------------------------------

def _fuzz_cases(n_cases, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n_cases):
        tb = bool(rng.integers(0, 2))
        slide = int(rng.integers(1, 40))
        win = slide + int(rng.integers(1, 200)) if rng.integers(0, 3) else slide * int(rng.integers(2, 9))
        n_keys = int(rng.choice([1, 2, 3, 9, 40, 300, 2500]))
        n = int(rng.integers(2000, 40000))
        batch = int(rng.choice([257, 1000, 2048, 4096, 16384, 32768]))
        if n_keys <= 3 and rng.integers(0, 2):
            n = int(rng.integers(40000, 120000))  # runs of tens of thousands of tuples: spread over the chip
        gap_every = int(rng.choice([10**9, 50, 400, 3000]))
        gap_len = int(rng.choice([3, 60, 2500]))
        disorder = int(rng.choice([0, 0, 5, 40])) if tb else 0
        delay = 0 if disorder == 0 else disorder + int(rng.integers(1, 30))
        if tb and disorder == 0 and rng.integers(0, 5) < 2:
            delay = int(rng.integers(1, 400))  # in order, but the windows fire late: several quanta stay open
        mode = int(rng.choice([0, 1, 2])) if not tb else 0
        cases.append((i, tb, n, n_keys, win, slide, batch, gap_every, gap_len, disorder, delay, mode))
    return cases


import os as _os


@pytest.mark.parametrize("case", _fuzz_cases(int(_os.environ.get("WF_FUZZ_CASES", "40")), seed=int(_os.environ.get("WF_FUZZ_SEED", "20260101"))),
                         ids=lambda c: "case%d" % c[0])
def test_key_ffat_fuzz(ext, case):
    i, tb, n, n_keys, win, slide, batch, gap_every, gap_len, disorder, delay, mode = case
    if tb:
        t, vals = bursty_stream(n, n_keys, seed=1000 + i, gap_every=gap_every, gap_len=gap_len, disorder=disorder)
    else:
        t, vals = make_stream(n, n_keys, seed=1000 + i)
    # results per output batch: a time-based key fires a window per slide of TIME, tuples or not (a small slide and long silences
    # make millions of them; in the worst case one batch fires them all)
    out_cap = 1 << 19
    if tb:
        out_cap = max(out_cap, int(sum(int(t[t[:, 0] == k, 2].max()) // slide + 2 for k in np.unique(t[:, 0]))))
    res, ctr = ext.key_ffat_run(t, win, slide, delay, tb, max(16, n_keys), batch, out_cap, mode)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0 and ctr["dropped"] == 0 and ctr["skipped_windows"] == 0
    got = unpack(res)
    assert len(got) == len(res), "duplicate (key, window) results"
    if tb:
        exp = oracle_tb(t, vals, win, slide)
        assert len(got) == len(exp)
        for k, s in exp.items():
            assert got[k][0] == pytest.approx(s, abs=1e-9), (case, k)
    else:
        n_exp = 0
        for key in np.unique(t[:, 0]):
            v = vals[t[:, 0] == key]
            lw = 0
            while lw * slide < len(v):
                w = v[lw * slide:lw * slide + win]
                e = w.sum() if mode == 0 else (w[-1] if mode == 1 else w[0])
                assert got[(int(key), lw)][0] == pytest.approx(e, abs=1e-9), (case, key, lw)
                lw += 1
                n_exp += 1
        assert len(got) == n_exp


# ---- huge runs: a key that owns thousands of tuples of a batch is spread over the chip (ffat_huge_gpu.hpp) -----------------

def hot_stream(n, n_cold, hot_share, seed, ts_step=1):
    """one hot key receives `hot_share` of the tuples, the rest goes to n_cold other keys"""
    rng = np.random.default_rng(seed)
    t, vals = make_stream(n, max(n_cold, 1), seed, ts_step=ts_step)
    hot = rng.random(n) < hot_share
    t[hot, 0] = 999_983
    return t, vals


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("n,n_cold,share,win,slide,batch", [
    (120_000, 3000, 0.4, 16, 4, 16384),     # the flagship geometry, zipf-like head
    (150_000, 1, 0.5, 12, 4, 32768),        # two keys, both huge
    (100_000, 200, 0.9, 7, 1, 8192),        # one-tuple panes: a window fires per tuple
    (100_000, 50, 0.6, 1024, 128, 16384),   # long panes (128 tuples): few leaves per chunk
    (90_000, 10, 0.7, 100, 100, 16384),     # tumbling
    (90_000, 10, 0.7, 30, 90, 16384),       # hopping
    (60_000, 5, 0.8, 12, 1, 4096),          # 12 panes per window: the widest the direct fold takes... 
    (60_000, 5, 0.8, 13, 1, 4096),          # ... and beyond it: block prefix / suffix folds, one combine per window
    (60_000, 5, 0.8, 63, 1, 4096),
    (60_000, 5, 0.8, 64, 1, 4096),          # a block of exactly one wave
    (60_000, 5, 0.8, 200, 1, 4096),         # blocks of several wave steps
    (90_000, 2, 0.9, 1000, 1, 16384),       # the long-window / small-slide query FlatFAT exists for
    (90_000, 2, 0.9, 999, 37, 16384),       # pane = 1 tuple, slide of 37 panes: windows start anywhere in a block
    (70_000, 3, 0.8, 4096, 8, 8192),        # 512 panes of 8 tuples; the first batches are shorter than a window
])
def test_key_ffat_huge_runs_count_based(ext, n, n_cold, share, win, slide, batch, mode):
    t, vals = hot_stream(n, n_cold, share, seed=n + win + mode)
    res, ctr = ext.key_ffat_run(t, win, slide, 0, False, max(16, n_cold + 1), batch, 4 * batch, mode)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0 and ctr["dropped"] == 0
    assert ctr["huge_runs"] > 0, "the hot key's runs were not spread over the chip"
    got = unpack(res)
    assert len(got) == len(res), "duplicate (key, window) results"
    n_exp = 0
    for key in np.unique(t[:, 0]):
        m = t[:, 0] == key
        v, ts = vals[m], t[m, 2]
        lw = 0
        while lw * slide < len(v):
            w = v[lw * slide:lw * slide + win]
            if len(w):
                exp = w.sum() if mode == 0 else (w[-1] if mode == 1 else w[0])
                assert got[(int(key), lw)][0] == pytest.approx(exp, abs=1e-9), (key, lw)
                if mode == 0:
                    assert got[(int(key), lw)][1] == int(ts[min(len(v), lw * slide + win) - 1]), (key, lw)
            lw += 1
            n_exp += 1
    assert len(got) == n_exp


@pytest.mark.parametrize("n,n_cold,share,win,slide,batch,ts_step", [
    (120_000, 2000, 0.4, 400, 100, 16384, 1),        # many panes per batch
    (120_000, 100, 0.5, 400_000, 100_000, 16384, 1), # less than one pane per batch: the run is one big reduction
    (100_000, 3, 0.6, 64, 16, 8192, 5),              # time advances faster than the tuples: quanta nobody hits
    (100_000, 3, 0.6, 5000, 5000, 8192, 1),          # tumbling
    (100_000, 3, 0.6, 3000, 10, 8192, 1),            # 300 panes per window: block prefix / suffix folds
    (100_000, 3, 0.6, 2000, 30, 8192, 3),            # ... 200 panes of 10 us over 3-spaced timestamps of 4 keys: panes nobody hits
])
def test_key_ffat_huge_runs_time_based(ext, n, n_cold, share, win, slide, batch, ts_step):
    t, vals = hot_stream(n, n_cold, share, seed=n + slide, ts_step=ts_step)
    res, ctr = ext.key_ffat_run(t, win, slide, 0, True, max(16, n_cold + 1), batch, 1 << 19)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0 and ctr["dropped"] == 0 and ctr["skipped_windows"] == 0
    assert ctr["huge_runs"] > 0
    got = unpack(res)
    exp = oracle_tb(t, vals, win, slide)
    assert len(res) == len(got) == len(exp)
    for k, s in exp.items():
        assert got[k][0] == pytest.approx(s, abs=1e-9), k


@pytest.mark.parametrize("win,slide,delay,ts_step,batch", [
    (64, 16, 16, 1, 8192),        # one pane of delay: two quanta open
    (64, 16, 200, 1, 8192),       # 12 panes of delay
    (400, 100, 250, 2, 16384),
    (3000, 10, 95, 1, 8192),      # long windows (block prefix / suffix folds), 10 quanta open
    (5000, 5000, 20000, 1, 8192), # tumbling, delay of four windows
])
def test_key_ffat_huge_runs_time_based_with_triggering_delay(ext, win, slide, delay, ts_step, batch):
    """in-order hot keys with a triggering delay: the newest quanta of a run stay open in the key's ring"""
    t, vals = hot_stream(100_000, 3, 0.6, seed=win + delay, ts_step=ts_step)
    res, ctr = ext.key_ffat_run(t, win, slide, delay, True, 16, batch, 1 << 19)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0 and ctr["dropped"] == 0 and ctr["skipped_windows"] == 0
    assert ctr["huge_runs"] > 0
    got = unpack(res)
    exp = oracle_tb(t, vals, win, slide)
    assert len(res) == len(got) == len(exp)
    for k, s in exp.items():
        assert got[k][0] == pytest.approx(s, abs=1e-9), k


def lagging_lanes_stream(n_keys, lanes, piece, rounds, lags, seed):
    """Batches as the device KeyBy of a DEFAULT graph hands them to a replica: one piece per lane and round, every lane in
    timestamp order (every key ticks once per time unit in every lane), lane l `lags[l]` rounds behind in arrival - the time of
    a key jumps forth and back by lag * piece / n_keys between the pieces of a batch."""
    rng = np.random.default_rng(seed)
    rows = []
    for r in range(rounds + max(lags)):
        for l in range(lanes):
            p = r - lags[l]
            if p < 0 or p >= rounds:
                continue
            j = np.arange(p * piece, (p + 1) * piece)
            blk = np.zeros((piece, 5), dtype=np.int64)
            blk[:, 0] = (j % n_keys) * 7919 + 3
            blk[:, 2] = j // n_keys
            rows.append(blk)
    t = np.concatenate(rows)
    t[:, 1] = np.arange(len(t))
    vals = rng.integers(-50, 50, len(t)).astype(np.float64)
    t[:, 3] = vals.view(np.int64)
    t[:, 4] = np.float64(1.0).view(np.int64)
    return t, vals


@pytest.mark.parametrize("presort", ["0", "1", "2"])
@pytest.mark.parametrize("n_keys,lanes,piece,rounds,lags", [
    (4, 4, 4096, 10, (0, 3, 1, 6)),     # runs of 4096 tuples of a key per batch (huge-run path when in order), jumps of 300-600 quanta
    (64, 4, 4096, 8, (0, 2, 5, 1)),     # runs of 256 (wave-per-run bulk path), jumps of 13-32 quanta ... and of more than 64:
    (16, 8, 2048, 8, (0, 7, 2, 5, 1, 6, 3, 4)),
])
def test_key_ffat_lanes_that_lag_behind_each_other(ext, n_keys, lanes, piece, rounds, lags, presort, monkeypatch):
    """A triggering delay that covers the lag between the lanes keeps thousands of quanta open; the pieces of a batch make a
    key's time jump by hundreds of quanta.  Nothing is late, so the plain window sums are the oracle - with the (key, pane)
    pre-sort off (arrival order: the run is out of timestamp order), adaptive, and always on (the run is in order and jumps)."""
    monkeypatch.setenv("WF_TS_PRESORT", presort)
    win, slide = 80, 10
    t, vals = lagging_lanes_stream(n_keys, lanes, piece, rounds, lags, seed=n_keys + lanes)
    delay = (max(lags) + 2) * piece // n_keys
    res, ctr = ext.key_ffat_run(t, win, slide, delay, True, 2 * n_keys, lanes * piece, 1 << 20)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0 and ctr["dropped"] == 0 and ctr["skipped_windows"] == 0
    got = unpack(res)
    exp = oracle_tb(t, vals, win, slide)
    assert len(res) == len(got) == len(exp)
    for k, s in exp.items():
        assert got[k][0] == pytest.approx(s, abs=1e-9), k


def test_key_ffat_huge_runs_with_disorder_fall_back(ext, monkeypatch):
    """a hot key whose tuples are not in timestamp order (delay 0: the late ones are dropped, by the same rules as a short run;
    without the (key, pane) pre-sort, which would make the tuples of a batch arrive in order)"""
    monkeypatch.setenv("WF_TS_PRESORT", "0")
    t, vals = hot_stream(80_000, 20, 0.7, seed=17)
    rng = np.random.default_rng(3)
    t[:, 2] = np.maximum(0, t[:, 2] + rng.integers(-400, 1, len(t)))
    res_a, ctr_a = ext.key_ffat_run(t, 1000, 250, 0, True, 64, 16384, 1 << 18)
    # same stream in batches too small for a huge run: the sequential rules decide alone
    res_b, ctr_b = ext.key_ffat_run(t, 1000, 250, 0, True, 64, 1024, 1 << 18)
    assert ctr_a["dropped"] == ctr_b["dropped"] and ctr_a["dropped"] > 0
    assert unpack(res_a) == unpack(res_b)


# ---- FlatFAT_GPU: one whole tree processed by many threads (reference: wf/flatfat_gpu.hpp) ------------------------------

def _fold(w, mode):
    return float(np.sum(w)) if mode == 0 else (float(w[-1]) if mode == 1 else float(w[0]))


@pytest.mark.parametrize("mode", [0, 1, 2])  # sum / newest wins / oldest wins: the last two are not commutative
@pytest.mark.parametrize("cap,first,step,rounds,win,slide", [
    (100, 60, 30, 12, 20, 5),        # a tree smaller than one 512-leaf tile; the circular buffer wraps several times
    (512, 512, 128, 9, 64, 16),      # exactly one tile, always full
    (3000, 1000, 700, 11, 300, 50),  # 4096 leaves: 8 tiles + the top of the tree, dirty ranges that wrap around
    (5000, 4000, 1, 40, 1000, 250),  # single-leaf updates of a large tree
])
def test_flatfat_gpu_matches_a_plain_fold(ext, mode, cap, first, step, rounds, win, slide):
    rng = np.random.default_rng(cap + mode)
    fat = ext.FlatFatGPU(cap, 4096, mode)
    assert fat.capacity >= cap and fat.capacity & (fat.capacity - 1) == 0
    live = list(rng.integers(-100, 100, first).astype(float))
    ts0 = 1  # timestamps are the arrival numbers of the leaves
    fat.build(live)

    def check():
        n_win = 0 if len(live) < win else (len(live) - win) // slide + 1
        got = fat.results(n_win, win, slide)
        assert len(got) == n_win
        for w, (val, ts) in enumerate(got):
            seg = live[w * slide:w * slide + win]
            assert val == pytest.approx(_fold(seg, mode), abs=1e-9), (w, len(live))
            assert ts == ts0 + w * slide + win - 1  # the newest leaf of the window

    check()
    for _ in range(rounds):
        new = list(rng.integers(-100, 100, step).astype(float))
        remove = max(0, len(live) + step - fat.capacity) if len(live) + step > fat.capacity else int(rng.integers(0, min(len(live), step) + 1))
        fat.update(new, remove)
        live = live[remove:] + new
        ts0 += remove
        assert fat.size == len(live)
        check()
    assert fat.kernels > 0
