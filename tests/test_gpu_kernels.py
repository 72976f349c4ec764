"""Numerics of the HIP kernels against plain PyTorch references (GPU only)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_tuples(n, n_keys=97, seed=0):
    """[n,5] int64 view of 40-byte tuples: key, id, ts, value(f64), aux(f64)."""
    g = torch.Generator().manual_seed(seed)
    t = torch.empty((n, 5), dtype=torch.int64)
    t[:, 0] = torch.randint(0, n_keys, (n,), generator=g)
    t[:, 1] = torch.arange(n)
    t[:, 2] = torch.arange(n) * 3
    val = torch.rand(n, generator=g, dtype=torch.float64) * 2048.0
    aux = torch.rand(n, generator=g, dtype=torch.float64)
    t[:, 3] = val.view(torch.int64)
    t[:, 4] = aux.view(torch.int64)
    return t


@pytest.mark.parametrize("n", [1, 63, 64, 1000, 1024, 1025, 100_000, 1_000_003])
def test_map_inplace_matches_torch(ext, n):
    t = make_tuples(n)
    d = t.cuda()
    ext.map_scale_inplace(d.data_ptr(), n, 2.5)
    torch.cuda.synchronize()
    out = d.cpu()
    ref_val = t[:, 3].view(torch.float64) * 2.5 + t[:, 4].view(torch.float64)
    assert torch.equal(out[:, :3], t[:, :3])
    # the device contracts a*b+c into one fma: allow 1 ulp
    assert torch.allclose(out[:, 3].view(torch.float64), ref_val, rtol=1e-15, atol=0)
    assert torch.equal(out[:, 4], t[:, 4])


@pytest.mark.parametrize("n", [1, 64, 1000, 1025, 100_000, 1_000_003])
def test_map_out_of_place_matches_torch(ext, n):
    """K6 (gpu::map_kernel): tuple -> result of another type, batch header carried over"""
    t = make_tuples(n)
    d = t.cuda()
    out = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    assert ext.map_scale_out_of_place(d.data_ptr(), n, out.data_ptr(), 1.75) == n
    torch.cuda.synchronize()
    res = out.cpu()
    assert torch.equal(res[:, :3], t[:, :3])
    ref = t[:, 3].view(torch.float64) * 1.75 + t[:, 4].view(torch.float64)
    assert torch.allclose(res[:, 3].view(torch.float64), ref, rtol=1e-15, atol=0)
    assert torch.equal(d.cpu(), t), "the input batch must not change"


@pytest.mark.parametrize("n", [0, 1, 64, 1023, 1024, 1025, 4096, 65_537, 1_000_003])
@pytest.mark.parametrize("mode", [0, 1])
def test_filter_compact_matches_torch(ext, n, mode):
    t = make_tuples(max(n, 1))[:n]
    d = t.cuda() if n > 0 else torch.empty((1, 5), dtype=torch.int64).cuda()
    out = torch.zeros((max(n, 1), 5), dtype=torch.int64).cuda()
    cnt = ext.filter_compact(d.data_ptr(), n, out.data_ptr(), mode, 1024.0)
    torch.cuda.synchronize()
    if mode == 0:
        mask = t[:, 3].view(torch.float64) < 1024.0
    else:
        mask = (t[:, 0] & 1) == 0
    ref = t[mask]
    assert cnt == ref.shape[0]
    # stable compaction: same tuples in the same order
    assert torch.equal(out[:cnt].cpu(), ref)


def test_filter_all_and_none(ext):
    n = 10_000
    t = make_tuples(n)
    d = t.cuda()
    out = torch.zeros_like(d)
    assert ext.filter_compact(d.data_ptr(), n, out.data_ptr(), 0, 1e9) == n
    assert torch.equal(out.cpu(), t)
    assert ext.filter_compact(d.data_ptr(), n, out.data_ptr(), 0, -1.0) == 0


def test_map_filter_pipeline_end_to_end(ext):
    pipe = ext.MapFilterPipeline(0, 3, 8192, 50, 4)
    per_source = 100_000
    for _ in range(3):
        pipe.step(per_source)
    st = pipe.stats()
    pipe.close()
    # per source: value = seq & 1023 -> map 2v+1 -> keep < 1024
    seq = torch.arange(3 * per_source)
    v = (seq & 1023).double() * 2 + 1
    keep = v < 1024
    assert st["sink_tuples"] == 3 * int(keep.sum())
    assert abs(st["sink_value_sum"] - 3 * float(v[keep].sum())) < 1e-6 * max(1.0, st["sink_value_sum"])
    assert st["latency_p99_us"] > 0
