"""Runtime knobs (environment variables read by the C++ runtime, ``include/wf/rt/config.hpp``) and small helpers."""
import contextlib
import json
import os

__all__ = ["RUNTIME_ENV", "runtime_env", "read_stats_dump", "percentile"]

# name -> meaning (defaults in parentheses)
RUNTIME_ENV = {
    "WF_QUEUE_CAPACITY": "slots of every inter-thread SPSC ring (2048)",
    "WF_BLOCKING": "1: consumers sleep on a condition variable instead of spin/yield back-off (0)",
    "WF_PIN_THREADS": "1: pin runtime threads to cores round-robin (0)",
    "WF_TRACE": "1: collect per-replica statistics and dump JSON + DOT at the end of the run (0)",
    "WF_LOG_DIR": "directory of the statistics dump (./log)",
    "WF_QUIET": "1: no banner on stdout (0)",
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
