"""Runtime knobs (environment variables read by the C++ runtime, ``include/wf/rt/config.hpp``) and small helpers."""
import contextlib
import json
import os

__all__ = ["RUNTIME_ENV", "runtime_env", "read_stats_dump", "percentile"]

# name -> meaning (defaults in parentheses)
RUNTIME_ENV = {
    # host runtime (include/wf/rt/config.hpp)
    "WF_QUEUE_CAPACITY": "slots of every inter-thread SPSC ring (2048)",
    "WF_BLOCKING": "1: a thread whose input rings are empty (or whose output ring is full) parks on a futex and is woken by its "
                   "peer; 0: spin / yield / sleep back-off (0)",
    "WF_PARK_TIMEOUT_US": "blocking mode: longest a parked thread sleeps before it looks again (2000)",
    "WF_SPIN_ITERS": "back-off mode: polls before a thread starts yielding (64)",
    "WF_SLEEP_US": "back-off mode: sleep between polls once yielding did not help (50)",
    "WF_PIN_THREADS": "1: pin the runtime threads, one per CPU of the process's affinity mask (or of WF_PIN_CPUS) (0)",
    "WF_PIN_CPUS": "CPU list to pin to, e.g. 0-15,64-79 (default: the process's affinity mask - bench.py gives every rank the "
                   "CPUs of its GPU's NUMA node)",
    "WF_TRACE": "1: collect per-replica statistics and dump JSON + DOT at the end of the run (0)",
    "WF_LOG_DIR": "directory of the statistics dump (./log)",
    "WF_QUIET": "1: no banner on stdout (0)",
    # device side (include/wf/gpu/hip_utils.hpp, the engines)
    "WF_UPLOAD_STREAMS": "n (1..7): host batches are uploaded through n shared streams per device, 0: through each lane's compute stream (unset: upload streams for operators with one or two replicas, compute streams otherwise)",
    "WF_D2H_STREAMS": "own: every DeviceToHost adapter copies on its own stream; shared: on the producer's",
    "WF_POOL_WAIT": "host | device: where a re-acquired slab's last use is waited for (host)",
    "WF_EDGE_WAIT": "host | device: where a consumer waits for the event of a batch produced by another stream (device)",
    "WF_WIN_QUIET_RUNS": "0: the non-incremental GPU window engine gathers every run of a batch into the work buffer, also the short ones that fire no window (1: those are appended to the key's carry area in place)",
    "WF_COALESCE_TUPLES": "a GPU replica that finds several small host batches queued uploads them into one device batch of at most n tuples; 0: never (131072)",
    "WF_COALESCE_SHARE": "a GPU replica with one producer groups at most 1/d of the slabs that producer keeps in flight, unless the group is under 32 Ki tuples; 1: all of them (2)",
    "WF_FFAT_LONG_RUN": "Key_FFAT_GPU: runs of at least n tuples of a key in a batch are folded by a wave instead of a thread (12; profiles/r3_flagship/long_run_threshold.txt)",
    "WF_DEBUG_EVENTS": "1: every batch release checks that its stream is alive and that slab, event and calling thread belong to the same device (0)",
    "WF_LDS_LIMIT": "bytes of LDS per workgroup the kernels may plan with (what the device reports: 160 KB on gfx950)",
    "WF_SORT_SKIP": "0: the keyed engines always sort their batches, even when no key repeats (1)",
    "WF_SORT_RESCUE": "0: a stream trusted to hold no repeats keeps the six gated sort launches instead of the single rescue launch (1)",
    "WF_SMALL_SORT": "0: batches of at most 16 Ki pairs behind a gate use the multi-launch radix sort instead of the one-workgroup network (1)",
    "WF_TS_PRESORT": "(key, timestamp) pre-sort of batches whose runs are out of timestamp order: 0 never, 1 when the device reports disorder, 2 always (1)",
    "WF_TUPLE_PATH": "1: GPU operators behind CPU producers are fed tuple by tuple (no Batcher), as in the reference (0)",
    "WF_RCCL_VERBOSE": "set: RcclTransport prints the RCCL version and the path of the loaded library even under WF_QUIET=1 (it always does otherwise)",
    "WF_ALLOW_STALE_BINARIES": "1: build_cpp falls back to an existing binary when its rebuild fails instead of raising (0)",
    "WF_INHERIT_STREAMS": "0: a GPU replica fused onto its producer keeps its own stream instead of continuing on the producer's (1)",
    "WF_EVENT_BLOCKING": "1: host waits on events sleep instead of spinning",
    "WF_STREAM_REAP": "how often finished stream work is reaped",
    "WF_HIP_GRAPHS": "1: Key_FFAT_GPU replays its per-batch launch sequence from a hipGraph (0: measured not to be faster)",
    "WF_FFAT_HUGE": "0: never spread huge runs of a key over the chip (gpu/ffat_huge_gpu.hpp) (1)",
    "WF_WIN_TILE": "0: non-incremental window engine without the LDS-tiled evaluation kernel (1)",
    "WF_ROCTX": "1: roctx ranges around the operators' work (for rocprofv3 --marker-trace)",
    "WF_SEGV_TRACE": "1: print a backtrace on SIGSEGV (debugging aid of the benchmark programs)",
}


@contextlib.contextmanager
def runtime_env(**kv):
    """Temporarily set runtime knobs, e.g. ``with runtime_env(WF_TRACE=1): ...``."""
    old = {}
    for k, v in kv.items():
        if k not in RUNTIME_ENV:
            raise KeyError("unknown runtime knob %s" % k)
        old[k] = os.environ.get(k)
        os.environ[k] = str(v)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def read_stats_dump(path):
    """Load a statistics dump written by PipeGraph with WF_TRACE=1."""
    with open(path) as f:
        return json.load(f)


def percentile(samples, p):
    s = sorted(samples)
    if not s:
        return 0.0
    k = min(len(s) - 1, max(0, int(round(p / 100.0 * (len(s) - 1)))))
    return s[k]
