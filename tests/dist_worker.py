"""One rank of a multi-process exchange test; launched by torch.distributed.run from tests/test_gpu_dist.py.

    python -m torch.distributed.run --nproc-per-node W tests/dist_worker.py <rccl|ipc> <pipeline: win_agg|ysb> <lanes>

Every rank runs the pipeline on GPU LOCAL_RANK % device_count (so the ranks share the card on a one-GPU box - the ipc
transport allows that, RCCL does not), rank 0 prints the gathered per-rank statistics as one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WF_QUIET", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from windflow_amd import load_ext, parallel as wfpar  # noqa: E402


def main():
    kind, pipeline, lanes = sys.argv[1], sys.argv[2], int(sys.argv[3])
    saved = os.dup(1)
    os.dup2(2, 1)  # gloo / RCCL banners must not reach stdout
    try:
        env = wfpar.init_distributed()
        device = env.local_rank % torch.cuda.device_count()
        transport = (wfpar.ipc_transport if kind == "ipc" else wfpar.rccl_transport)(env, device)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    ext = load_ext()
    if pipeline == "ysb":
        pipe = ext.YsbPipeline(device, 2, 16384, 10, 30_000_000, 4, lanes, transport, False)
        per_source = 600_000
    else:
        pipe = ext.WinAggPipeline(device, 2, 4096, 50, 8, 2, False, 0, 4, transport, lanes)
        per_source = 40_000
    pipe.step(per_source)
    pipe.step(per_source)
    pipe.close()
    st = pipe.stats()
    keep = {k: st[k] for k in st if isinstance(st[k], (int, float))}
    box = [None] * env.world
    if env.world > 1:
        dist.all_gather_object(box, keep)
    else:
        box = [keep]
    if env.rank == 0:
        print(json.dumps({"transport": transport.name, "world": env.world, "ranks": box}), flush=True)
    wfpar.shutdown()


if __name__ == "__main__":
    main()
