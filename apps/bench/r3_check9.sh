#!/bin/bash
# round 3: small-batch path (one-workgroup sort, finalize folded into the long-run kernel) - GPU suite, small-batch rows,
# hardware-queue probe, the 8-lane small-batch kernel trace with WF_DEBUG_EVENTS=1
set -o pipefail
O=${1:-gpurun_out/r3/check9}
ROOT=$(pwd)
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
if [ -z "$SKIP_PYTEST" ]; then
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; echo "pytest failed"; exit 1; }
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
row "flagship host 16Ki" --batch 16384
row "flagship device 16Ki" --batch 16384 --source device
row "flagship host 4Ki" --batch 4096 --tuples-per-step 33554432
row "flagship host-fed 128Ki"
row "flagship device 128Ki" --source device
row "win_agg uniform keys device 16Ki" --source device --key-dist uniform --batch 16384
row "win_agg TB keys=1024 16Ki" --keys 1024 --batch 16384
row "win_seq opaque keys=1024 16Ki" --pipeline win_seq --keys 1024 --batch 16384
WF_SMALL_SORT=0 row "flagship host 16Ki WF_SMALL_SORT=0" --batch 16384
GPU_MAX_HW_QUEUES=8 row "flagship host 16Ki GPU_MAX_HW_QUEUES=8" --batch 16384
GPU_MAX_HW_QUEUES=8 row "flagship host 128Ki GPU_MAX_HW_QUEUES=8"
GPU_MAX_HW_QUEUES=8 row "flagship device 128Ki GPU_MAX_HW_QUEUES=8" --source device
GPU_MAX_HW_QUEUES=16 row "flagship device 128Ki GPU_MAX_HW_QUEUES=16" --source device
GPU_MAX_HW_QUEUES=16 row "flagship device 16Ki GPU_MAX_HW_QUEUES=16" --source device --batch 16384
cat $O/rows.txt
for q in 4 16; do
for a in "-o map_stateful -s 14 -n 14 -k 1000 -b 10000" "-o map_stateful -s 8 -n 8 -k 64 -b 131072" "-o filter_stateful -s 14 -n 14 -k 1000 -b 10000"; do
  echo "== GPU_MAX_HW_QUEUES=$q bench_ops_gpu $a" >> $O/ops.txt
  GPU_MAX_HW_QUEUES=$q timeout -k 10 90 apps/bin/bench_ops_gpu $a -t 3 >> $O/ops.txt 2>&1 || { tail -5 $O/ops.txt; echo "bench_ops_gpu $a failed"; exit 1; }
done
done
cat $O/ops.txt
cd /tmp || exit 1
# the run that killed the profiler in round 2 (profiles/r2_small_batch): once, with the release checks on
WF_DEBUG_EVENTS=1 timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/trace_16k_8lanes" -o t -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --batch 16384 --tuples-per-step 33554432 --device-probe-steps 0 > "$ROOT/$O/trace_16k_8lanes.log" 2>&1
echo "profiler exit status: $?" | tee -a "$ROOT/$O/trace_16k_8lanes.log"
tail -5 "$ROOT/$O/trace_16k_8lanes.log" | cut -c1-300
cd "$ROOT"
find $O -name "*kernel_trace.csv" -size +8M -delete
