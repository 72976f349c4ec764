#!/bin/bash
# Round-2 measurements on one MI355X: headline benches and rocprofv3 passes.  Run from the repository root on the GPU box:
#   bash apps/bench/profile_r2.sh [out dir, default gpurun_out/r2/profiles]
# Every GPU step has its own time limit; the script stops at the first failure.
set -o pipefail
OUT=${1:-gpurun_out/r2/profiles}
ROOT=$(pwd)
mkdir -p "$OUT"/{flagship,keyby,win_seq_1key,win_seq_1ki,ysb} || exit 1
export TMPDIR=/tmp WF_QUIET=1
run() { echo "== $*" >&2; "$@" || { echo "FAILED ($?): $*" >&2; exit 1; }; }

# --- headline numbers (no profiler); SKIP_BENCH=1 keeps the ones already there ---
if [ -z "$SKIP_BENCH" ]; then
run timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 > "$OUT/flagship/bench_line.json" 2> "$OUT/flagship/bench.err"
run timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --parallelism keyby > "$OUT/keyby/bench_line.json" 2> "$OUT/keyby/bench.err"
run timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --pipeline win_seq --keys 1 > "$OUT/win_seq_1key/bench_line.json" 2> "$OUT/win_seq_1key/bench.err"
run timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --pipeline win_seq --keys 1024 > "$OUT/win_seq_1ki/bench_line.json" 2> "$OUT/win_seq_1ki/bench.err"
run timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --pipeline win_seq --keys 1 --wave-reduce > "$OUT/win_seq_1key/bench_line_wave_reduce.json" 2>> "$OUT/win_seq_1key/bench.err"
run timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --pipeline win_seq --keys 1024 --wave-reduce > "$OUT/win_seq_1ki/bench_line_wave_reduce.json" 2>> "$OUT/win_seq_1ki/bench.err"
run timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --pipeline ysb > "$OUT/ysb/bench_line.json" 2> "$OUT/ysb/bench.err"
run timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --pipeline ysb --parallelism keyby --agg-replicas 2 > "$OUT/ysb/bench_line_keyby.json" 2>> "$OUT/ysb/bench.err"
fi

# --- kernel traces (one lane where the small-batch regime is profiled; see profiles/r1_sync_policy) ---
cd /tmp || exit 1
prof() {  # prof <dir> <bench args...>
    local d=$1; shift
    run timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$d/trace" -o t -- python3 "$ROOT/bench.py" "$@" > "$ROOT/$OUT/$d/trace.log" 2>&1
}
pmc() {  # pmc <dir> <name> <counters...> -- <bench args...>
    local d=$1 n=$2; shift 2
    local ctr=()
    while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
    shift
    run timeout -s KILL 240 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d "$ROOT/$OUT/$d/pmc_$n" -o p -- python3 "$ROOT/bench.py" "$@" > "$ROOT/$OUT/$d/pmc_$n.log" 2>&1
}
prof win_seq_1key --steps 2 --warmup 1 --pipeline win_seq --keys 1 --tuples-per-step 67108864
prof win_seq_1ki --steps 2 --warmup 1 --pipeline win_seq --keys 1024 --tuples-per-step 67108864
prof flagship --steps 2 --warmup 1 --tuples-per-step 67108864
prof keyby --steps 2 --warmup 1 --tuples-per-step 67108864 --parallelism keyby
pmc win_seq_1key lds SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -- --steps 1 --warmup 1 --pipeline win_seq --keys 1 --tuples-per-step 33554432
pmc win_seq_1key fetch FETCH_SIZE -- --steps 1 --warmup 1 --pipeline win_seq --keys 1 --tuples-per-step 33554432
pmc win_seq_1key write WRITE_SIZE -- --steps 1 --warmup 1 --pipeline win_seq --keys 1 --tuples-per-step 33554432
pmc win_seq_1ki fetch FETCH_SIZE -- --steps 1 --warmup 1 --pipeline win_seq --keys 1024 --tuples-per-step 33554432
pmc win_seq_1ki write WRITE_SIZE -- --steps 1 --warmup 1 --pipeline win_seq --keys 1024 --tuples-per-step 33554432
# the profiler has crashed inside librocprofiler-sdk on this pipeline (profiles/r1_sync_policy): last, and with one source lane
prof ysb --steps 2 --warmup 1 --tuples-per-step 33554432 --pipeline ysb --sources 1
echo done >&2
