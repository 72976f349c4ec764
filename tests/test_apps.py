"""Applications built on the library: Yahoo! Streaming Benchmark (CPU Key_Farm / Win_MapReduce versions, GPU version).

Counted mode (-e) makes the stream deterministic: every source emits N events 10 us apart, one third are views,
all ads are known, windows are 10 s tumbling per campaign (100 campaigns)."""
import os
import re
import subprocess

import pytest

from windflow_amd import build as wfbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ext():
    import windflow_amd

    return windflow_amd.load_ext()


def run_app(name, args, hip=False, timeout=110, extra=()):
    src = os.path.join(ROOT, "apps", "ysb", name + (".hip" if hip else ".cpp"))
    exe = wfbuild.build_cpp(src, os.path.join(ROOT, "apps", "bin", name), hip=hip, extra=["-I" + os.path.dirname(src), *extra])
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, WF_QUIET="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = {}
    for key in ("generated events", "received results", "events counted in windows"):
        out[key] = int(re.search(r"\[" + name + r"\] " + key + r" (\d+)", r.stdout).group(1))
    return out


def expected(events, sources, period_us=10):
    views = sum(1 for v in range(events) if (v % 100000) % 3 == 0)
    windows = (events * period_us + 10_000_000 - 1) // 10_000_000
    return {"generated events": events * sources, "events counted in windows": views * sources, "received results": windows * 100}


@pytest.mark.parametrize("app,n,m", [("ysb_kf", 1, 1), ("ysb_kf", 2, 3), ("ysb_wmr", 2, 2), ("ysb_wmr", 1, 3)])
def test_ysb_cpu(app, n, m):
    got = run_app(app, ["-e", 2_500_000, "-p", 10, "-n", n, "-m", m])
    assert got == expected(2_500_000, n)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,batch,engine", [(1, 1, 4096, "ffat"), (2, 2, 10000, "ffat"), (3, 1, 65536, "ffat"), (2, 2, 8192, "farm")])
def test_ysb_gpu(n, m, batch, engine):
    """the whole query on the device; a triggering delay longer than the stream makes the result independent of how the
    source lanes interleave (the windows fire at end of stream)"""
    got = run_app("ysb_gpu", ["-e", 2_500_000, "-p", 10, "-n", n, "-m", m, "-b", batch, "-d", 30_000_000, "-w", engine], hip=True)
    assert got == expected(2_500_000, n)


@pytest.mark.gpu
@pytest.mark.parametrize("sources,replicas,split_merge", [(2, 1, False), (3, 2, False), (2, 2, True), (3, 1, True)])
def test_ysb_pipeline_steps(ext, sources, replicas, split_merge):
    """the pipeline bench.py drives: gated steps, closed-form oracle.  split_merge: the events are split on the device by
    ad type into two filter+join branches that are merged again (device batches all the way)"""
    per_source, period = 1_200_000, 10
    pipe = ext.YsbPipeline(0, sources, 16384, period, 30_000_000, 4, replicas, None, split_merge)
    pipe.step(per_source)
    pipe.step(per_source)
    pipe.close()
    st = pipe.stats()
    exp = expected(2 * per_source, sources, period)
    assert st["counted_views"] == exp["events counted in windows"]
    assert st["sink_tuples"] == exp["received results"]
    assert st["dropped"] == 0 and st["overflow"] == 0 and st["table_full"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,n,m,split", [(1, 2, 2, 0), (2, 2, 1, 0), (3, 1, 2, 1)])
def test_ysb_gpu_dist_local_ranks(ranks, n, m, split):
    """the distributed application with its ranks as threads of one process (LocalTransport): campaigns partitioned across
    ranks by Exchange_GPU, totals of the job against the closed form"""
    got = run_app("ysb_gpu_dist", ["-e", 2_500_000, "-p", 10, "-n", n, "-m", m, "-b", 8192, "-d", 30_000_000, "-x", split, "-t", "local", "-r", ranks],
                  hip=True, extra=wfbuild.RCCL_FLAGS)
    exp = expected(2_500_000, n * ranks)
    exp["received results"] = expected(2_500_000, 1)["received results"]  # a (campaign, window) pair exists once in the job
    assert got == exp


@pytest.mark.gpu
def test_ysb_gpu_dist_rccl_single_rank():
    """one process, RCCL transport with world size 1 (the id travels through the bootstrap file)"""
    got = run_app("ysb_gpu_dist", ["-e", 1_000_000, "-p", 10, "-n", 2, "-m", 2, "-b", 8192, "-d", 30_000_000, "-t", "rccl", "-i", "/tmp/wf_test_rccl_id"],
                  hip=True, extra=wfbuild.RCCL_FLAGS)
    assert got == expected(1_000_000, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("name,expect", [("gpu_pipeline", r"\d+ windows, hottest reading [\d.]+ F"),
                                         ("gpu_pane_farm", r"8000 windows, 388322956 shares counted")])  # (NumPy: sum over all windows)
def test_gpu_examples_run(name, expect):
    """examples/: the device pipeline (Filter_GPU -> Map_GPU -> Key_FFAT_GPU) and the reference-style program (CPU Source ->
    Pane_Farm_GPU with the PLQ on the device, the WLQ on the host and 16-byte symbol keys -> CPU Sink; it checks its own
    window count)"""
    src = os.path.join(ROOT, "examples", name + ".hip")
    exe = wfbuild.build_cpp(src, os.path.join(ROOT, "apps", "bin", "example_" + name), hip=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=110, env=dict(os.environ, WF_QUIET="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert re.search(expect, r.stdout), r.stdout[-2000:]
