"""CPU-only tests of the native runtime and of the basic operators: each test
compiles a C++ program against the library headers and checks its output."""
import os
import subprocess

import pytest

from windflow_amd import build as wfbuild

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "cpp", "bin")


def run_cpp(name, args=(), timeout=300, env=None):
    exe = wfbuild.build_cpp(os.path.join(HERE, "cpp", name + ".cpp"), os.path.join(BIN, name))
    e = dict(os.environ, WF_QUIET="1")
    if env:
        e.update(env)
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_basic_ops_smoke():
    out = run_cpp("smoke_basic_ops")
    sums = [int(l.split("=")[1]) for l in out.splitlines() if l.startswith("sum=")]
    # 1 source -> 50000 even values x3 ; 2 sources -> twice as many
    assert sums == [150000, 300000, 150000, 300000, 150000]


OPS = ["seq", "ffat", "wf", "kf", "kff", "pf", "wmr", "kf+pf", "kf+wmr", "wf+pf", "wf+wmr"]


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("mode", ["cb", "tb", "tbd"])
def test_mp_windows_random_parallelism(op, mode):
    """Reference suite src/mp_test_cpu/test_mp_*_{cb,tb,tb_2}: results must not depend on
    the degree of parallelism - and must match the sequential oracle."""
    win, slide = ("14", "4") if mode == "cb" else ("1000", "300")
    out = run_cpp("test_mp_windows", ["-t", op, "-m", mode, "-r", 3, "-l", 1500, "-k", 5, "-w", win, "-s", slide])
    assert out.count("  OK") == 3 and "FAILED" not in out


@pytest.mark.parametrize("op", ["seq", "wf", "kf", "pf", "wmr"])
@pytest.mark.parametrize("mode", ["cb", "tb"])
def test_mp_windows_incremental_functions(op, mode):
    win, slide = ("12", "5") if mode == "cb" else ("900", "250")
    out = run_cpp("test_mp_windows", ["-t", op, "-m", mode, "-r", 2, "-l", 1000, "-k", 3, "-w", win, "-s", slide, "-i"])
    assert out.count("  OK") == 2 and "FAILED" not in out


@pytest.mark.parametrize("op", ["wf", "kf", "pf", "wmr"])
def test_mp_windows_chaining(op):
    """Reference *_chaining tests: equal parallelism so that chain() really fuses."""
    out = run_cpp("test_mp_windows", ["-t", op, "-m", "tb", "-r", 2, "-l", 1000, "-k", 4, "-w", "800", "-s", "200", "-c"])
    assert out.count("  OK") == 2 and "FAILED" not in out


@pytest.mark.parametrize("win,slide", [("7", "7"), ("1", "1"), ("20", "1")])
def test_mp_windows_tumbling_and_dense_sliding(win, slide):
    for op in ["wf", "kf"]:
        out = run_cpp("test_mp_windows", ["-t", op, "-m", "cb", "-r", 2, "-l", 600, "-k", 3, "-w", win, "-s", slide])
        assert out.count("  OK") == 2 and "FAILED" not in out
