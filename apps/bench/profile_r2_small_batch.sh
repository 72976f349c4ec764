#!/bin/bash
# rocprofv3 kernel trace of the small-batch regime (16 Ki-tuple batches) of the flagship, one MI355X.  Round 1 had no trace of
# it (the profiler died on the host with 8 lanes of small batches, profiles/r1_sync_policy/README.md): one lane first, then the
# 8-lane run once, each a single attempt.
#   bash apps/bench/profile_r2_small_batch.sh [out dir, default gpurun_out/r2/prof_small_batch]
set -o pipefail
OUT=${1:-gpurun_out/r2/prof_small_batch}
ROOT=$(pwd)
mkdir -p "$OUT" || exit 1
export TMPDIR=/tmp WF_QUIET=1
cd /tmp || exit 1
prof() {  # prof <name> <bench args...>
    local d=$1; shift
    echo "== $d: $*" >&2
    mkdir -p "$ROOT/$OUT/$d"
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$d/trace" -o t -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 "$@" > "$ROOT/$OUT/$d/trace.log" 2>&1 || { echo "FAILED $d" >&2; exit 1; }
    cp "$ROOT/$OUT/$d/trace/"*kernel_stats.csv "$ROOT/$OUT/$d.kernel_stats.csv"
    grep -a '^{' "$ROOT/$OUT/$d/trace.log" > "$ROOT/$OUT/$d.json"
}
prof batch16k_1lane --batch 16384 --sources 1 --keys 131072 --tuples-per-step 8388608
prof batch16k_8lanes --batch 16384 --tuples-per-step 33554432
echo done >&2
