"""In-tree build of the native extension and of the C++ test/benchmark programs.

Everything is compiled for gfx950 only (hipcc cross-compiles without a GPU).
The built artefacts stay inside the repository tree so they travel to the GPU
box with the snapshot.
"""
import os
import subprocess
import sys
import sysconfig

ROOT = os.path.dirname(os.path.abspath(__file__))
INCLUDE = os.path.join(ROOT, "include")
CSRC = os.path.join(ROOT, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# Device code generation: loops whose trip count is only known at run time (the window loop of a user's functor, the per-run
# loops of the keyed kernels) are unrolled with a remainder loop, so that their loads are issued ahead of their uses instead of
# one load + full wait per iteration (measured on the opaque window sum: global_load x8 in flight instead of 1).
DEVICE_FLAGS = ["-mllvm", "-unroll-runtime"]


# RCCL: compiled against the headers of the ROCm tree and linked with an RPATH to the same tree, so a native program always
# runs with the library its headers describe.  A Python process that imported torch first already holds torch's bundled
# librccl (same SONAME): the transports print which library answered and its version, and refuse a major-version mismatch
# (dist/transport.hpp rccl_identity()).
RCCL_FLAGS = ["-I/opt/rocm/include/rccl", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-ldl"]


def ext_path():
    return os.path.join(ROOT, "_C" + sysconfig.get_config_var("EXT_SUFFIX"))


def _newest_mtime(paths):
    newest = 0.0
    for p in paths:
        if os.path.isdir(p):
            for d, _, files in os.walk(p):
                for f in files:
                    newest = max(newest, os.path.getmtime(os.path.join(d, f)))
        elif os.path.exists(p):
            newest = max(newest, os.path.getmtime(p))
    return newest


def _pybind_includes():
    import pybind11

    return ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"]]


def build_extension(force=False, verbose=True):
    """Compile windflow_amd/_C*.so (pybind11 + HIP kernels) if missing or stale."""
    out = ext_path()
    srcs = [os.path.join(CSRC, "wf_ext.hip")]
    deps = [INCLUDE, CSRC, os.path.join(os.path.dirname(ROOT), "apps", "ysb")]
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest_mtime(deps):
        return out
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-I" + INCLUDE, "-I" + CSRC,
           "-I" + os.path.join(os.path.dirname(ROOT), "apps", "ysb")]
    cmd += DEVICE_FLAGS + _pybind_includes() + srcs + ["-o", out, "-lpthread"]
    rccl_inc = "/opt/rocm/include/rccl"
    if os.path.isdir(rccl_inc):
        cmd += list(RCCL_FLAGS)
    if verbose:
        print("[windflow_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_cpp(src, out, hip=False, extra=()):
    """Compile one C++ (g++) or HIP (hipcc) program against the library headers."""
    if os.path.exists(out) and os.path.getmtime(out) >= max(_newest_mtime([INCLUDE]), os.path.getmtime(src)):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if hip:
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-O2", "-std=c++17", *DEVICE_FLAGS, "-I" + INCLUDE, "-I" + CSRC, src, "-o", out, "-lpthread"]
    else:
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I" + INCLUDE, src, "-o", out, "-pthread"]
    cmd += list(extra)
    try:
        subprocess.check_call(cmd)
    except (subprocess.CalledProcessError, OSError):
        # A failed rebuild must never leave a suite green on yesterday's binary.  Only a box that has no toolchain at all may
        # opt in to running what travelled with the tree: WF_ALLOW_STALE_BINARIES=1.
        if os.path.exists(out) and os.environ.get("WF_ALLOW_STALE_BINARIES", "0") not in ("", "0"):
            print("[windflow_amd.build] could not rebuild %s, WF_ALLOW_STALE_BINARIES is set: using the existing binary" % out,
                  file=sys.stderr, flush=True)
            return out
        raise
    return out


REPO = os.path.dirname(ROOT)



# (source, binary, hip?[, extra flags]) of every native program in the tree: regression programs and applications
PROGRAMS = [
    ("tests/cpp/smoke_basic_ops.cpp", "tests/cpp/bin/smoke_basic_ops", False),
    ("tests/cpp/test_mp_windows.cpp", "tests/cpp/bin/test_mp_windows", False),
    ("tests/cpp/test_graphs.cpp", "tests/cpp/bin/test_graphs", False),
    ("tests/cpp/test_runtime.cpp", "tests/cpp/bin/test_runtime", False),
    ("tests/cpp/test_api_variants.cpp", "tests/cpp/bin/test_api_variants", False),
    ("tests/cpp/test_mp_windows_gpu.hip", "tests/cpp/bin/test_mp_windows_gpu", True),
    ("tests/cpp/test_gpu_keyed.hip", "tests/cpp/bin/test_gpu_keyed", True),
    # the same MultiPipe suite keyed by 16-byte fixed-size keys (the reference's *_string programs)
    ("tests/cpp/test_mp_windows_gpu.hip", "tests/cpp/bin/test_mp_windows_gpu_fixedkey", True, ["-DWF_TEST_FIXED_KEY"]),
    ("tests/cpp/test_mp_windows.cpp", "tests/cpp/bin/test_mp_windows_fixedkey", False, ["-DWF_TEST_FIXED_KEY"]),
    ("apps/ysb/ysb_kf.cpp", "apps/bin/ysb_kf", False),
    ("apps/ysb/ysb_wmr.cpp", "apps/bin/ysb_wmr", False),
    ("apps/ysb/ysb_gpu.hip", "apps/bin/ysb_gpu", True),
    ("apps/ysb/ysb_gpu_dist.hip", "apps/bin/ysb_gpu_dist", True, RCCL_FLAGS),
    ("apps/bench/bench_ops_gpu.hip", "apps/bin/bench_ops_gpu", True),
    ("apps/bench/bench_win_gpu.hip", "apps/bin/bench_win_gpu", True),
    ("apps/bench/probe/hip_leak_probe.hip", "apps/bin/hip_leak_probe", True),
    ("apps/bench/bench_cpu.cpp", "apps/bin/bench_cpu", False),
    ("examples/wordcount_windows.cpp", "apps/bin/example_wordcount", False),
    ("examples/gpu_pipeline.hip", "apps/bin/example_gpu_pipeline", True),
    ("examples/gpu_pane_farm.hip", "apps/bin/example_gpu_pane_farm", True),
]


def build_programs(verbose=True, jobs=None):
    """Compile every native program that exists in the tree (stale ones only), a few at a time."""
    from concurrent.futures import ThreadPoolExecutor

    jobs = jobs or int(os.environ.get("MAX_JOBS", "0")) or min(6, os.cpu_count() or 1)

    def one(prog):
        src, out, hip = prog[:3]
        s, o = os.path.join(REPO, src), os.path.join(REPO, out)
        if not os.path.exists(s):
            return None
        if verbose:
            print("[windflow_amd.build] program", src, flush=True)
        return build_cpp(s, o, hip=hip, extra=["-I" + os.path.dirname(s)] + list(prog[3] if len(prog) > 3 else []))

    with ThreadPoolExecutor(max_workers=jobs) as pool:
        return [b for b in pool.map(one, PROGRAMS) if b]


if __name__ == "__main__":
    build_extension(force="--force" in sys.argv)
    if "--all" in sys.argv:
        build_programs()
