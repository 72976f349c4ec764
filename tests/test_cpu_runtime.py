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
