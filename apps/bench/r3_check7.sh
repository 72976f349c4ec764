#!/bin/bash
# round 3: K1 retuned for gfx950 (LDS tile from the device attribute, 16-wave workgroups, address-space-specialised calls,
# run-time unrolling) - GPU suite, the win_seq rows, counters of the single-key run
set -o pipefail
O=${1:-gpurun_out/r3/check7}
ROOT=$(pwd)
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
if [ -z "$SKIP_PYTEST" ]; then
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; echo "pytest failed"; exit 1; }
tail -3 $O/pytest.log
fi
: > $O/rows.txt
: > $O/rows.jsonl
row() {  # row <label> <bench.py args...>
    local label=$1; shift
    local line
    line=$(timeout -k 10 150 python3 bench.py --steps 5 --warmup 1 --tuples-per-step 134217728 --device-probe-steps 0 "$@" 2>> "$O/bench.err" | grep -a '^{') || { echo "FAILED $label" >&2; tail -5 $O/bench.err; exit 1; }
    echo "$line" >> "$O/rows.jsonl"
    printf '%-58s ' "$label" >> "$O/rows.txt"
    echo "$line" | python3 apps/bench/jline.py >> "$O/rows.txt"
}
row "win_seq opaque keys=1"  --pipeline win_seq --keys 1
row "win_seq opaque keys=1 win=64 slide=1" --pipeline win_seq --keys 1 --win 64 --slide 1
row "win_seq opaque keys=1 tumbling 4096" --pipeline win_seq --keys 1 --win 4096 --slide 4096
row "win_seq opaque keys=1024" --pipeline win_seq --keys 1024
row "win_seq opaque keys=65536" --pipeline win_seq --keys 65536
row "win_seq wave_reduce keys=1" --pipeline win_seq --keys 1 --wave-reduce
WF_LDS_LIMIT=65536 row "win_seq opaque keys=1 WF_LDS_LIMIT=64K" --pipeline win_seq --keys 1
row "flagship host-fed"
row "flagship device source 128Ki" --source device
row "win_agg TB keys=8 device" --source device --keys 8
row "win_agg CB keys=1024 win=1024 slide=16" --keys 1024 --count-based --win 1024 --slide 16
row "ysb" --pipeline ysb
cat $O/rows.txt
for a in "-o map_stateful -s 14 -n 14 -k 1000 -b 10000" "-o map_stateful -s 8 -n 8 -k 64 -b 131072" "-o map_stateless -s 8 -n 8 -b 131072" "-o filter_stateful -s 8 -n 8 -k 1024 -b 131072"; do
  echo "== bench_ops_gpu $a" >> $O/ops.txt
  timeout -k 10 90 apps/bin/bench_ops_gpu $a -t 3 >> $O/ops.txt 2>&1 || { tail -5 $O/ops.txt; echo "bench_ops_gpu $a failed"; exit 1; }
done
cat $O/ops.txt
cd /tmp || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/trace_1key" -o t -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --tuples-per-step 67108864 --pipeline win_seq --keys 1 --device-probe-steps 0 > "$ROOT/$O/trace_1key.log" 2>&1 || { tail -5 "$ROOT/$O/trace_1key.log"; echo "trace failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$ROOT/$O/pmc_1key_lds" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --tuples-per-step 33554432 --pipeline win_seq --keys 1 --device-probe-steps 0 > "$ROOT/$O/pmc_1key_lds.log" 2>&1 || { tail -5 "$ROOT/$O/pmc_1key_lds.log"; echo "pmc failed"; exit 1; }
cd "$ROOT"
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +20M -delete
ls -la $O/pmc_1key_lds | head
