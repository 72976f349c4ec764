"""Distributed layer on one GPU: in-process multi-rank exchange (LocalTransport), RCCL with world size 1."""
import json
import math
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,n,keys,batch,sources,lanes,per_round", [
    (2, 40_000, 101, 2048, 1, 1, 0),
    (3, 60_000, 1000, 4096, 1, 1, 0),
    (4, 30_000, 7, 1024, 1, 1, 0),
    (1, 50_000, 300, 2048, 3, 4, 0),     # one rank: the exchange is the device KeyBy, merging the lanes' batches
    (1, 20_000, 50, 1024, 2, 1, 0),      # one rank, one lane: batches pass through
    (2, 60_000, 500, 2048, 4, 3, 0),     # several batches per round, several lanes behind the exchange
    (3, 40_000, 64, 1024, 2, 4, 0),
    (2, 40_000, 500, 2048, 4, 2, 1),     # one batch per round
    (4, 30_000, 3, 512, 4, 2, 0),        # three keys per source: most (rank, lane) destinations stay empty, a few are hot
    (8, 16_000, 1, 512, 2, 4, 0),        # eight ranks, one key per source: whole rounds land on a few lanes (more lane batches per
                                         # round than the lane pool owns: they must be handed downstream as they fill)
])
def test_exchange_conserves_and_routes(ext, world, n, keys, batch, sources, lanes, per_round):
    res = ext.exchange_local_run(0, world, n, keys, batch, sources, lanes, per_round)
    sent = [sources * max(1, n * (r + 1) // world) for r in range(world)]
    assert sum(r[0] for r in res) == sum(sent)
    exp_sum = sum(sources * float(np.sum(np.arange(s // sources) % 97)) for s in sent)
    assert sum(r[1] for r in res) == pytest.approx(exp_sum)
    assert all(r[2] == 0 for r in res), "tuples delivered to a (rank, lane) that does not own their key"
    assert all(r[3] == 0 for r in res), "per-key FIFO order broken by the exchange"
    if keys >= 100:
        assert all(r[0] > 0 for r in res)
    if per_round == 1 and world > 1:
        assert all(r[5] <= r[4] for r in res), "more input batches than rounds although a round may carry only one"


def test_rccl_world1_collectives(ext):
    uid = ext.rccl_create_id()
    tr = ext.rccl_transport(0, 1, uid, 0)
    assert (tr.rank, tr.world, tr.name) == (0, 1, "rccl")
    x = torch.arange(1000, dtype=torch.float64, device="cuda")
    tr.all_reduce_sum(x.data_ptr(), x.numel())
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float64))
    assert tr.all_gather([7, 8, 9]) == [7, 8, 9]
    tr.barrier()


def _expected_single_source(n, n_keys, win, slide):
    per_key = n // n_keys
    vals = (np.arange(n) & 1023).astype(np.float64)
    total, windows = 0.0, 0
    for k in range(n_keys):
        v = vals[k::n_keys]
        c = np.concatenate([[0.0], np.cumsum(v)])
        for lw in range(math.ceil(per_key / slide)):
            total += c[min(per_key, slide * lw + win)] - c[slide * lw]
            windows += 1
    return total, windows


def _run_ranks(ext, world, make_transport, n_sources=2, n_keys=50, per_source=40_000):
    stats = [None] * world
    errors = []

    def rank_main(r):
        try:
            pipe = ext.WinAggPipeline(0, n_sources, 4096, n_keys, 8, 2, False, 0, 4, make_transport(r))
            pipe.step(per_source)
            pipe.step(per_source)
            pipe.close()
            stats[r] = pipe.stats()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=100)
    assert not errors, errors
    assert all(s is not None for s in stats)
    return stats


@pytest.mark.parametrize("world", [2, 3])
def test_win_agg_with_cross_rank_exchange(ext, world):
    fabric = ext.LocalFabric(world)
    stats = _run_ranks(ext, world, lambda r: ext.local_transport(fabric, r))
    total, windows = _expected_single_source(2 * 40_000, 50, 8, 2)
    assert sum(s["sink_tuples"] for s in stats) == world * 2 * windows
    assert sum(s["sink_value_sum"] for s in stats) == pytest.approx(world * 2 * total, rel=1e-12)
    assert all(s["overflow"] == 0 and s["table_full"] == 0 for s in stats)
    assert all(s["sink_tuples"] > 0 for s in stats)


def test_win_agg_with_rccl_world1(ext):
    uid = ext.rccl_create_id()
    stats = _run_ranks(ext, 1, lambda r: ext.rccl_transport(0, 1, uid, 0))
    total, windows = _expected_single_source(2 * 40_000, 50, 8, 2)
    assert stats[0]["sink_tuples"] == 2 * windows
    assert stats[0]["sink_value_sum"] == pytest.approx(2 * total, rel=1e-12)


def _ysb_expected(events, sources, period_us=10):
    views = sum(1 for v in range(events) if (v % 100000) % 3 == 0)
    windows = (events * period_us + 10_000_000 - 1) // 10_000_000
    return views * sources, windows * 100


@pytest.mark.parametrize("world,sources,lanes,split_merge", [(1, 2, 2, False), (2, 2, 1, False), (3, 2, 2, False), (2, 3, 2, True), (1, 2, 1, True)])
def test_ysb_over_exchange(ext, world, sources, lanes, split_merge):
    """Yahoo Streaming Benchmark with the campaigns partitioned across ranks (and lanes) by Exchange_GPU: every rank ingests
    its own events, filters and joins them on its GPU, the joined views travel to the rank that owns their campaign, which
    counts them in 10 s tumbling windows.  Closed form: every view is counted exactly once, every (campaign, window) is
    reported by exactly one rank."""
    per_source, period = 600_000, 10
    fabric = ext.LocalFabric(world)
    stats = [None] * world
    errors = []

    def rank_main(r):
        try:
            pipe = ext.YsbPipeline(0, sources, 16384, period, 30_000_000, 4, lanes, ext.local_transport(fabric, r), split_merge)
            pipe.step(per_source)
            pipe.step(per_source)
            pipe.close()
            stats[r] = pipe.stats()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=100)
    assert not errors, errors
    assert all(s is not None for s in stats)
    views, results = _ysb_expected(2 * per_source, sources * world, period)
    assert sum(s["counted_views"] for s in stats) == views
    # every rank's sources emit the same timestamps: a (campaign, window) pair exists once in the job
    assert sum(s["sink_tuples"] for s in stats) == results
    assert all(s["dropped"] == 0 and s["overflow"] == 0 and s["table_full"] == 0 for s in stats)
    if world > 1:
        assert all(s["sink_tuples"] > 0 for s in stats), "some rank owns no campaign"
        assert all(s["exchange_tuples_sent"] > 0 for s in stats)


@pytest.mark.parametrize("world,sources,lanes,split_merge", [(1, 2, 2, False), (2, 2, 1, False), (3, 2, 2, False), (2, 3, 2, True)])
def test_ysb_over_exchange_deterministic_needs_no_delay(ext, world, sources, lanes, split_merge):
    """DETERMINISTIC graph: the exchange releases every output lane in timestamp order under the low-watermark rule over all the
    source lanes of all the ranks, so the windows need NO triggering delay to tolerate the lag between sources and ranks
    (the DEFAULT graph above runs with 30 s) - and still no view is late."""
    per_source, period = 400_000, 10
    fabric = ext.LocalFabric(world)
    stats = [None] * world
    errors = []

    def rank_main(r):
        try:
            pipe = ext.YsbPipeline(0, sources, 16384, period, 0, 4, lanes, ext.local_transport(fabric, r), split_merge, True)
            pipe.step(per_source)
            pipe.step(per_source)
            pipe.close()
            stats[r] = pipe.stats()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=100)
    assert not errors, errors
    assert all(s is not None for s in stats)
    views, results = _ysb_expected(2 * per_source, sources * world, period)
    assert all(s["dropped"] == 0 and s["overflow"] == 0 and s["table_full"] == 0 for s in stats)
    assert sum(s["counted_views"] for s in stats) == views
    assert sum(s["sink_tuples"] for s in stats) == results


@pytest.mark.parametrize("world", [2, 3])
def test_win_agg_deterministic_across_ranks(ext, world):
    """count-based windows over keys that EVERY source of EVERY rank emits: only a total order of the job's tuples makes the
    windows well defined.  The deterministic exchange provides it (timestamp, then source rank, then arrival), and the result is
    the same whatever the number of ranks: the stream is the union of world x 2 sources that each emit key k at times k, k + K, ..."""
    n_sources, n_keys, per_source, win, slide = 2, 40, 30_000, 8, 2
    fabric = ext.LocalFabric(world)
    stats = [None] * world
    errors = []

    def rank_main(r):
        try:
            # shared keys, DETERMINISTIC, count-based windows
            pipe = ext.WinAggPipeline(0, n_sources, 2048, n_keys, win, slide, False, 0, 4, ext.local_transport(fabric, r), 2, 0, 1.3, True, True)
            pipe.step(per_source)
            pipe.close()
            stats[r] = pipe.stats()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=100)
    assert not errors, errors
    # every source emits the same (key, ts, value) sequence: at every timestamp a key receives S = world * n_sources tuples
    # with the same value, so in the total order the key's stream is v0 x S, v1 x S, ... and the windows follow in closed form
    S = world * n_sources
    seq = np.arange(per_source)
    total, windows = 0.0, 0
    for k in range(n_keys):
        v = np.repeat((seq[k::n_keys] & 1023).astype(np.float64), S)
        c = np.concatenate([[0.0], np.cumsum(v)])
        for lw in range(math.ceil(len(v) / slide)):
            total += c[min(len(v), slide * lw + win)] - c[slide * lw]
            windows += 1
    assert sum(s["sink_tuples"] for s in stats) == windows
    assert sum(s["sink_value_sum"] for s in stats) == pytest.approx(total, rel=1e-12)
    assert all(s["overflow"] == 0 and s["table_full"] == 0 for s in stats)


@pytest.mark.parametrize("world,dist,keys", [(2, 2, 5000), (3, 0, 64), (2, 1, 300)])
def test_pane_combiner_in_front_of_the_exchange(ext, world, dist, keys):
    """Pane_Combiner_GPU folds every batch per (key, pane) on the GPU that produced it; the exchange carries partial results and the
    owner's Key_FFAT_GPU combines them: same windows, same sums as shipping the tuples - with fewer bytes on the links (zipf keys:
    far fewer on the busiest one)."""
    n_sources, per_source, win, slide, delay = 2, 200_000, 4096, 1024, 40 * 8192  # (the delay covers the lag the sources are allowed: 8 batches)

    def run(combiner):
        fabric = ext.LocalFabric(world)
        stats, errors = [None] * world, []

        def rank_main(r):
            try:
                pipe = ext.WinAggPipeline(0, n_sources, 8192, keys, win, slide, True, delay, 4, ext.local_transport(fabric, r), 2, dist, 1.3, True, False, False,
                                          combiner)
                pipe.step(per_source)
                pipe.step(per_source)
                pipe.close()
                stats[r] = pipe.stats()
            except Exception as e:  # pragma: no cover
                errors.append(e)

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=100)
        assert not errors, errors
        assert all(s is not None for s in stats)
        return stats

    plain, comb = run(False), run(True)
    for stats in (plain, comb):
        assert all(s["dropped"] == 0 and s["overflow"] == 0 and s["table_full"] == 0 for s in stats)
    assert sum(s["sink_tuples"] for s in plain) == sum(s["sink_tuples"] for s in comb) > 0
    assert sum(s["sink_value_sum"] for s in plain) == sum(s["sink_value_sum"] for s in comb)  # (integers in doubles: exact)
    assert sum(s["combiner_tuples_in"] for s in comb) == world * n_sources * 2 * per_source
    sent_plain, sent_comb = sum(s["exchange_tuples_sent"] for s in plain), sum(s["exchange_tuples_sent"] for s in comb)
    assert sent_plain == world * n_sources * 2 * per_source and sent_comb < sent_plain
    if dist == 2:  # zipf(1.3): the hot keys collapse to a partial per pane and batch
        assert max(s["exchange_busiest_link_bytes"] for s in plain) > 3 * max(s["exchange_busiest_link_bytes"] for s in comb)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch_ranks(world, kind, pipeline, lanes, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), kind, pipeline, str(lanes)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=110, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("world,lanes", [(2, 1), (3, 2)])
def test_win_agg_across_processes_ipc(world, lanes):
    """one PROCESS per rank (torch.distributed.run), control board in POSIX shared memory, data pulled through hipIpc
    handles; the ranks share the GPU of the box when it has fewer GPUs than ranks"""
    out = _launch_ranks(world, "ipc", "win_agg", lanes, 29611 + world)
    assert out["transport"] == "ipc" and out["world"] == world
    stats = out["ranks"]
    total, windows = _expected_single_source(2 * 40_000, 50, 8, 2)
    assert sum(s["sink_tuples"] for s in stats) == world * 2 * windows
    assert sum(s["sink_value_sum"] for s in stats) == pytest.approx(world * 2 * total, rel=1e-12)
    assert all(s["overflow"] == 0 and s["table_full"] == 0 for s in stats)
    assert all(s["exchange_tuples_sent"] > 0 and s["sink_tuples"] > 0 for s in stats)


def test_ysb_across_processes_ipc():
    out = _launch_ranks(2, "ipc", "ysb", 2, 29621)
    stats = out["ranks"]
    views, results = _ysb_expected(2 * 600_000, 2 * 2, 10)
    assert sum(s["counted_views"] for s in stats) == views
    assert sum(s["sink_tuples"] for s in stats) == results
    assert all(s["dropped"] == 0 and s["overflow"] == 0 for s in stats)


def test_win_agg_across_gpus_rccl():
    """RCCL between distinct devices (grouped ncclSend / ncclRecv over xGMI): runs whenever the box has 2+ GPUs"""
    n_gpus = torch.cuda.device_count()
    if n_gpus < 2:
        pytest.skip("needs 2 GPUs (RCCL refuses two ranks on one device)")
    world = min(n_gpus, 4)
    out = _launch_ranks(world, "rccl", "win_agg", 2, 29631)
    assert out["transport"] == "rccl"
    stats = out["ranks"]
    total, windows = _expected_single_source(2 * 40_000, 50, 8, 2)
    assert sum(s["sink_tuples"] for s in stats) == world * 2 * windows
    assert sum(s["sink_value_sum"] for s in stats) == pytest.approx(world * 2 * total, rel=1e-12)
    assert all(s["overflow"] == 0 and s["table_full"] == 0 for s in stats)
