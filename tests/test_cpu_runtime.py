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


@pytest.mark.parametrize("op", ["wf", "kf", "kff", "pf", "wmr", "kf+pf"])
def test_mp_windows_fixed_size_keys(op):
    """The suite keyed by 16-byte fixed-size keys "key_<n>" (wf::FixedKey<16>, -DWF_TEST_FIXED_KEY): the CPU twin of the reference's
    string-keyed programs (src/mp_test_cpu/mp_common_string.hpp); the same key type runs on the GPU operators (test_gpu_operators.py)."""
    exe = wfbuild.build_cpp(os.path.join(HERE, "cpp", "test_mp_windows.cpp"), os.path.join(BIN, "test_mp_windows_fixedkey"),
                            extra=["-DWF_TEST_FIXED_KEY"])
    r = subprocess.run([exe, "-t", op, "-m", "cb", "-r", "2", "-l", "1200", "-k", "5", "-w", "14", "-s", "4"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, WF_QUIET="1"))
    assert r.returncode == 0 and r.stdout.count("  OK") == 2 and "FAILED" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


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


@pytest.mark.parametrize("op", ["seq", "kf", "wf", "wmr", "kf+wmr", "wf+wmr"])
@pytest.mark.parametrize("mode", ["cb", "tb"])
def test_mp_windows_hopping(op, mode):
    """Hopping windows (win < slide, reference win_seq.hpp:352-365): tuples in the gaps belong to no window but still
    close the window before them, so every operator emits the same windows as the sequential oracle."""
    win, slide = ("3", "10") if mode == "cb" else ("300", "1000")
    out = run_cpp("test_mp_windows", ["-t", op, "-m", mode, "-r", 2, "-l", 1500, "-k", 5, "-w", win, "-s", slide])
    assert out.count("  OK") == 2 and "FAILED" not in out


@pytest.mark.parametrize("op,mode", [("kf", "tb"), ("wf", "cb"), ("pf", "tb")])
def test_mp_windows_blocking_mode(op, mode):
    """WF_BLOCKING=1: idle threads sleep on a futex and are woken by the producer (rt/spsc_queue.hpp Parker)."""
    win, slide = ("14", "4") if mode == "cb" else ("1000", "300")
    out = run_cpp("test_mp_windows", ["-t", op, "-m", mode, "-r", 2, "-l", 1500, "-k", 5, "-w", win, "-s", slide],
                  env={"WF_BLOCKING": "1", "WF_QUEUE_CAPACITY": "64"})
    assert out.count("  OK") == 2 and "FAILED" not in out


GRAPH_SCENARIOS = ["merge_1", "merge_2", "merge_3", "merge_4", "split_1", "split_2", "split_3", "split_4", "split_5",
                   "graph_1", "graph_2", "graph_3", "graph_4", "graph_5", "graph_6", "graph_7", "graph_8", "graph_9"]


@pytest.mark.parametrize("scenario", GRAPH_SCENARIOS)
def test_split_merge_graphs(scenario):
    """Split / merge / DAG topologies under random parallelism against sequential oracles (reference: merge_test,
    split_test, graph_test)."""
    out = run_cpp("test_graphs", ["-t", scenario, "-r", 3, "-n", 12000])
    assert out.count("  OK") == 3 and "FAILED" not in out


def test_runtime_units():
    """SPSC ring, graph EOS/back-pressure, string keys, FlatFAT vs brute force, statistics + DOT."""
    assert "ALL OK" in run_cpp("test_runtime")


@pytest.mark.parametrize("case", range(8))
def test_illegal_topologies_are_refused(case):
    exe = wfbuild.build_cpp(os.path.join(HERE, "cpp", "test_runtime.cpp"), os.path.join(BIN, "test_runtime"))
    r = subprocess.run([exe, "illegal", str(case)], capture_output=True, text=True, timeout=60, env=dict(os.environ, WF_QUIET="1"))
    assert r.returncode == 1 and "WindFlow Error" in r.stderr


def test_runtime_tsan():
    """The runtime unit tests under ThreadSanitizer (host code only)."""
    src = os.path.join(HERE, "cpp", "test_runtime.cpp")
    exe = os.path.join(BIN, "test_runtime_tsan")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I" + wfbuild.INCLUDE, src, "-o", exe, "-pthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, WF_QUIET="1"))
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]


@pytest.mark.parametrize("prog,args", [("test_runtime", []), ("test_api_variants", []), ("test_mp_windows", ["-r", "1", "-l", "300", "-k", "3", "-w", "9", "-s", "3"]),
                                       ("test_graphs", ["-r", "1"])])
def test_address_and_undefined_behaviour_sanitizers(prog, args):
    """The CPU regression programs under AddressSanitizer + UndefinedBehaviorSanitizer (host code only; `make asan` runs the same)."""
    src = os.path.join(HERE, "cpp", prog + ".cpp")
    exe = os.path.join(BIN, prog + "_asan")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        os.makedirs(BIN, exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                               "-fno-omit-frame-pointer", "-I" + wfbuild.INCLUDE, "-I" + os.path.join(HERE, "cpp"), src, "-o", exe, "-pthread"])
    env = dict(os.environ, WF_QUIET="1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)  # the sanitizer runtime must come first in the library list
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.stdout[-1500:] + r.stderr[-3000:])


def test_api_variants():
    """Every functor signature accepted by the builders (plain / rich, Source loop mode, Filter modes, Sink forms,
    window function flavours, closing functions, LocalStorage, custom KeyBy routing)."""
    out = run_cpp("test_api_variants")
    assert "ALL OK" in out and "FAILED" not in out
