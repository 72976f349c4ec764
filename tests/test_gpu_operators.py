"""GPU window operators inside full MultiPipes (C++ program built with hipcc), checked
against the sequential oracle; mirrors the reference's src/mp_test_gpu suite."""
import os
import subprocess

import pytest

from windflow_amd import build as wfbuild

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "cpp", "bin")


def run_gpu_cpp(args, timeout=110, fixed_key=False):
    exe = wfbuild.build_cpp(os.path.join(HERE, "cpp", "test_mp_windows_gpu.hip"),
                            os.path.join(BIN, "test_mp_windows_gpu_fixedkey" if fixed_key else "test_mp_windows_gpu"), hip=True,
                            extra=["-I" + os.path.join(HERE, "cpp")] + (["-DWF_TEST_FIXED_KEY"] if fixed_key else []))
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, WF_QUIET="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


# one stage of the two-stage operators on the CPU (the reference's matrix, incl. all four nestings), then both on the device
GPU_OPS = ["seq", "kf", "wf", "pf_plq", "pf_wlq", "wmr_map", "wmr_red", "kff", "seqffat", "kf+pf", "wf+wmr", "kf+wmr", "wf+pf",
           "pf_gg", "wmr_gg", "kf+pf_gg", "wf+pf_gg", "kf+wmr_gg", "wf+wmr_gg"]


@pytest.mark.parametrize("op", GPU_OPS)
@pytest.mark.parametrize("mode", ["cb", "tb"])
def test_gpu_window_operators(op, mode):
    win, slide = ("14", "4") if mode == "cb" else ("1000", "300")
    out = run_gpu_cpp(["-t", op, "-m", mode, "-r", 2, "-l", 1200, "-k", 5, "-w", win, "-s", slide])
    assert out.count("  OK") == 2 and "FAILED" not in out


# keys that are not numbers: the reference's four string-keyed GPU programs (src/mp_test_gpu/test_mp_gpu_{kf,wf,pf,wmr}_cb_string.cpp)
# with 16-byte fixed-size keys "key_<n>" (wf::FixedKey<16>: std::string cannot live in HBM), plus the incremental operator, the
# all-device two-stage operators and a nesting; time-based for the two engines as well
@pytest.mark.parametrize("op,mode", [("kf", "cb"), ("wf", "cb"), ("pf_plq", "cb"), ("wmr_map", "cb"), ("kff", "cb"), ("pf_gg", "cb"), ("wmr_gg", "cb"),
                                     ("kf+pf", "cb"), ("kf", "tb"), ("kff", "tb"), ("kff", "tbd"), ("kf", "tbd")])
def test_gpu_window_operators_fixed_size_keys(op, mode):
    win, slide = ("14", "4") if mode == "cb" else ("1000", "300")
    out = run_gpu_cpp(["-t", op, "-m", mode, "-r", 2, "-l", 1200, "-k", 5, "-w", win, "-s", slide], fixed_key=True)
    assert out.count("  OK") == 2 and "FAILED" not in out


@pytest.mark.parametrize("op", ["kf", "wf", "kff", "pf_plq", "pf_wlq", "wmr_map", "wmr_red", "pf_gg", "wmr_gg", "kf+pf_gg", "wf+wmr_gg"])
def test_gpu_window_operators_default_mode(op):
    out = run_gpu_cpp(["-t", op, "-m", "tbd", "-r", 2, "-l", 1000, "-k", 4, "-w", "900", "-s", "300"])
    assert out.count("  OK") == 2 and "FAILED" not in out


@pytest.mark.parametrize("op", ["kf", "wf", "pf_plq", "wmr_red", "kff", "pf_gg", "wmr_gg", "kf+pf_gg"])
@pytest.mark.parametrize("mode", ["cb", "tb"])
def test_gpu_window_operators_chaining(op, mode):
    """reference: src/mp_test_gpu/test_mp_gpu_*_chaining.cpp - equal parallelism, every operator chained"""
    win, slide = ("14", "4") if mode == "cb" else ("1000", "300")
    out = run_gpu_cpp(["-t", op, "-m", mode, "-r", 2, "-l", 1200, "-k", 5, "-w", win, "-s", slide, "-c"])
    assert out.count("  OK") == 2 and "FAILED" not in out and "chaining" in out


def test_gpu_window_operators_tuple_path_still_works():
    """WF_TUPLE_PATH=1: GPU operators behind CPU producers are fed tuple by tuple (no batchers), as in the reference"""
    for op in ("kf", "wf", "kff", "pf_gg"):
        exe = wfbuild.build_cpp(os.path.join(HERE, "cpp", "test_mp_windows_gpu.hip"), os.path.join(BIN, "test_mp_windows_gpu"), hip=True,
                                extra=["-I" + os.path.join(HERE, "cpp")])
        r = subprocess.run([exe, "-t", op, "-m", "cb", "-r", "1", "-l", "800", "-k", "5", "-w", "14", "-s", "4"], capture_output=True, text=True, timeout=110,
                           env=dict(os.environ, WF_QUIET="1", WF_TUPLE_PATH="1"))
        assert r.returncode == 0 and "  OK" in r.stdout, op + r.stdout[-2000:] + r.stderr[-2000:]


def run_keyed_cpp(args, timeout=110, env=None):
    exe = wfbuild.build_cpp(os.path.join(HERE, "cpp", "test_gpu_keyed.hip"), os.path.join(BIN, "test_gpu_keyed"), hip=True,
                            extra=["-I" + os.path.join(HERE, "cpp")])
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, WF_QUIET="1", **(env or {})))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("n,keys", [(60000, 37), (20000, 1), (50000, 3000)])
def test_keyed_gpu_operators_and_device_keyby(n, keys):
    """Keyed Map_GPU / Filter_GPU (per-key scratchpad state) and Key_FFAT_GPU behind the device KeyBy emitter."""
    out = run_keyed_cpp([n, keys, 2])
    # scenarios A, H (keyed_scan: tuples, batches), B, C, D, G (device split/merge), F x 2 runs (+ E on a box with 2+ GPUs)
    assert out.count("  OK") >= 26 and "FAILED" not in out
    assert out.count("H keyed_scan") == 4 and out.count("G device split/merge") == 2 and out.count("I deterministic") == 8
    assert out.count("J deterministic across ranks") == 2  # world 2 and 3 over LocalTransport


# the copy / synchronisation policies that are not the default (wf/gpu/hip_utils.hpp) must stay correct
SYNC_POLICIES = [
    {"WF_POOL_WAIT": "device"},
    {"WF_D2H_STREAMS": "shared", "WF_UPLOAD_STREAMS": "2"},
    {"WF_UPLOAD_STREAMS": "1", "WF_POOL_WAIT": "device", "WF_EDGE_WAIT": "host"},
    {"WF_INHERIT_STREAMS": "0", "WF_UPLOAD_STREAMS": "0"},  # every replica on a stream of its own, uploads in the compute streams
]


@pytest.mark.parametrize("policy", SYNC_POLICIES, ids=lambda p: ",".join("%s=%s" % kv for kv in sorted(p.items())))
def test_keyed_gpu_operators_alternative_sync_policies(policy):
    out = run_keyed_cpp([40000, 37, 1], env=policy)
    assert out.count("  OK") >= 10 and "FAILED" not in out
