"""Multi-GPU bootstrap: one process per GPU, RCCL over xGMI between them.

``torch.distributed`` (gloo) is only the out-of-band channel that carries the RCCL
unique id and the benchmark barriers; the data path is the library's own
``wf::dist::RcclTransport`` (grouped ncclSend/ncclRecv all-to-all, see
``include/wf/dist/transport.hpp``) used by the ``Exchange_GPU`` operator.
"""
import os
import sys

import torch
import torch.distributed as dist

from .. import load_ext

__all__ = ["DistEnv", "init_distributed", "rccl_transport", "ipc_transport", "local_fabric", "local_transport", "shutdown"]


class DistEnv:
    """Rank layout read from the launcher's environment (torchrun / torch.distributed.run)."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    def __repr__(self):
        return "DistEnv(rank=%d, world=%d, local_rank=%d)" % (self.rank, self.world, self.local_rank)


def init_distributed(backend="gloo"):
    """Join the job's process group (no-op for a single process) and return the DistEnv."""
    env = DistEnv()
    if env.world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=env.rank, world_size=env.world)
    return env


def rccl_transport(env=None, device=None):
    """Create this rank's RCCL transport; rank 0 mints the unique id, the process group carries it."""
    ext = load_ext()
    env = env or DistEnv()
    device = env.local_rank if device is None else device
    box = [ext.rccl_create_id() if env.rank == 0 else None]
    if env.world > 1:
        dist.broadcast_object_list(box, src=0)
    # RCCL prints a version banner on stdout when the first communicator comes up; programs that
    # print machine-readable results on stdout (bench.py) must not see it there
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        return ext.rccl_transport(env.rank, env.world, box[0], device)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def ipc_transport(env=None, device=None):
    """Create this rank's IPC transport (one process per rank, peers pull from each other's device buffers through
    hipIpc memory handles: works across GPUs over xGMI and between ranks that share a GPU); rank 0 mints the job token."""
    ext = load_ext()
    env = env or DistEnv()
    device = env.local_rank if device is None else device
    box = ["%d-%s" % (os.getpid(), os.urandom(8).hex()) if env.rank == 0 else None]
    if env.world > 1:
        dist.broadcast_object_list(box, src=0)
    return ext.ipc_transport(env.rank, env.world, box[0], device)


def local_fabric(world):
    """In-process fabric: `world` ranks as threads of this process (tests, single-process multi-GPU)."""
    return load_ext().LocalFabric(world)


def local_transport(fabric, rank):
    return load_ext().local_transport(fabric, rank)


def all_reduce_max(value, env=None):
    """Max of a Python float over the ranks (control plane, gloo)."""
    env = env or DistEnv()
    if env.world == 1 or not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
