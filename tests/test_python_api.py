"""Python wrappers: windflow_amd.ops (tensor-level kernels), windflow_amd.models (pipelines), windflow_amd.utils."""
import math
import os

import numpy as np
import pytest
import torch


def test_utils_runtime_env_roundtrip():
    from windflow_amd import utils

    assert "WF_QUEUE_CAPACITY" in utils.RUNTIME_ENV
    before = os.environ.get("WF_TRACE")
    with utils.runtime_env(WF_TRACE=1, WF_QUEUE_CAPACITY=64):
        assert os.environ["WF_TRACE"] == "1" and os.environ["WF_QUEUE_CAPACITY"] == "64"
    assert os.environ.get("WF_TRACE") == before
    with pytest.raises(KeyError):
        with utils.runtime_env(WF_NOT_A_KNOB=1):
            pass
    assert utils.percentile([5, 1, 3], 50) == 3
    assert utils.percentile([], 99) == 0.0


def test_parallel_env_single_process():
    from windflow_amd import parallel

    env = parallel.DistEnv()
    assert env.world >= 1 and 0 <= env.rank < env.world
    assert parallel.all_reduce_max(3.5, env) == 3.5


@pytest.mark.gpu
def test_ops_map_filter_scan_sort():
    from windflow_amd import ops

    rng = np.random.default_rng(1)
    n = 100_000
    keys = rng.integers(0, 1000, n)
    vals = rng.normal(size=n)
    t = ops.make_tuples(keys, vals, aux=2.0)
    ref = torch.from_numpy(vals * 3.0 + 2.0)
    ops.map_scale_(t, 3.0)
    torch.cuda.synchronize()
    assert torch.allclose(ops.tuple_values(t).cpu(), ref, rtol=1e-14, atol=1e-13)  # the device contracts a*b+c into an fma
    dev_vals = ops.tuple_values(t).cpu()
    kept = ops.filter_compact(t, "value_below", 0.5)
    exp = dev_vals[dev_vals < 0.5]
    assert kept.shape[0] == exp.numel()
    assert torch.equal(ops.tuple_values(kept).cpu(), exp)  # stable compaction: same order
    x = torch.randint(0, 50, (70_001,), dtype=torch.int32, device="cuda")
    scan, total = ops.exclusive_scan(x)
    assert total == int(x.sum()) and torch.equal(scan.cpu().long(), (torch.cumsum(x.cpu().long(), 0) - x.cpu().long()))
    k = torch.randint(0, 1 << 20, (50_000,), dtype=torch.int32, device="cuda")
    v = torch.arange(50_000, dtype=torch.int32, device="cuda")
    k0 = k.cpu().numpy().copy()
    ops.radix_sort_pairs_(k, v, bits=20)
    torch.cuda.synchronize()
    order = np.argsort(k0, kind="stable")
    assert np.array_equal(k.cpu().numpy(), k0[order]) and np.array_equal(v.cpu().numpy(), order)


@pytest.mark.gpu
def test_ops_key_ffat_and_model_pipeline():
    from windflow_amd import models, ops

    n, n_keys, win, slide = 12_000, 13, 9, 3
    keys = np.arange(n) % n_keys
    vals = (np.arange(n) % 7).astype(np.float64)
    res, ctr = ops.key_ffat(ops.make_tuples(keys, vals, device=None).numpy(), win, slide, max_keys=64, batch_len=2048)
    assert ctr["overflow"] == 0 and ctr["table_full"] == 0
    per_key = math.ceil(n / n_keys)
    assert len(res) == sum(math.ceil(len(vals[k::n_keys]) / slide) for k in range(n_keys))
    got = {(int(r[0]), int(r[1])): float(np.int64(r[3]).view(np.float64)) for r in res}
    for k in range(n_keys):
        v = vals[k::n_keys]
        for lw in range(math.ceil(len(v) / slide)):
            assert got[(k, lw)] == pytest.approx(v[lw * slide:lw * slide + win].sum())
    assert per_key > win
    with models.WinAgg(gpu_id=0, sources=2, batch=4096, keys_per_source=32, win=8, slide=2, inflight=4) as p:
        p.step(20_000)
        assert "digraph" in p.dot()
    st = p.stats()
    assert st["sink_tuples"] == 2 * 32 * math.ceil((20_000 // 32) / 2) and st["overflow"] == 0
    with models.MapFilter(gpu_id=0, sources=2, batch=4096, keys_per_source=16, inflight=4) as m:
        m.step(10_000)
        st = m.stats()
    assert st["sink_tuples"] == sum(1 for s in range(2) for q in range(10_000) if (q & 1023) * 2 + 1 < 1024)
    # k steps streamed back to back produce what k drained steps produce
    with models.WinAgg(gpu_id=0, sources=2, batch=4096, keys_per_source=32, win=8, slide=2, inflight=4) as a:
        for _ in range(3):
            a.step(20_000)
    with models.WinAgg(gpu_id=0, sources=2, batch=4096, keys_per_source=32, win=8, slide=2, inflight=4) as b:
        b.steps(3, 20_000)
    sa, sb = a.stats(), b.stats()
    assert sa["sink_tuples"] == sb["sink_tuples"] > 0 and sa["sink_value_sum"] == pytest.approx(sb["sink_value_sum"])


def _gloo_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from windflow_amd import parallel

    env = parallel.init_distributed("gloo")
    q.put((rank, env.world, parallel.all_reduce_max(10.0 + rank, env)))
    parallel.shutdown()


def test_parallel_control_plane_two_processes_gloo():
    """The control plane of a multi-rank job (rendezvous + max-reduction of the timings) with 2 CPU processes."""
    import socket
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, 2, 11.0), (1, 2, 11.0)]
