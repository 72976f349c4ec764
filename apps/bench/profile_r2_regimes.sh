#!/bin/bash
# rocprofv3 kernel traces of the engine paths behind profiles/r2_regimes (one MI355X):
#   bash apps/bench/profile_r2_regimes.sh [out dir, default gpurun_out/r2/prof_regimes]
set -o pipefail
OUT=${1:-gpurun_out/r2/prof_regimes}
ROOT=$(pwd)
mkdir -p "$OUT" || exit 1
export TMPDIR=/tmp WF_QUIET=1
cd /tmp || exit 1
prof() {  # prof <name> <bench args...>
    local d=$1; shift
    echo "== $d: $*" >&2
    mkdir -p "$ROOT/$OUT/$d"
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$d/trace" -o t -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --tuples-per-step 67108864 "$@" > "$ROOT/$OUT/$d/trace.log" 2>&1 || { echo "FAILED $d" >&2; exit 1; }
    cp "$ROOT/$OUT/$d/trace/"*kernel_stats.csv "$ROOT/$OUT/$d.kernel_stats.csv"
}
prof long_windows_1ki_keys --keys 1024 --count-based --win 1024 --slide 1
prof long_windows_8_keys --keys 8 --count-based --win 4000 --slide 4
prof delay_8_keys --keys 8 --delay 8
prof delay_1ki_keys --keys 1024 --delay 8
prof shared_keys_deterministic --keys 1024 --shared-keys --deterministic
prof win_seq_64ki_keys --pipeline win_seq --keys 65536
echo done >&2
