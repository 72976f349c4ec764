#!/bin/bash
# round 3: how the device-fed flagship scales with the number of lanes (sources) and the batch size
set -o pipefail
O=${1:-gpurun_out/r3/lanes}
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
: > $O/rows.txt
: > $O/rows.jsonl
row() {  # row <label> <bench.py args...>
    local label=$1; shift
    local line
    line=$(timeout -k 10 150 python3 bench.py --steps 4 --warmup 1 --tuples-per-step 134217728 --device-probe-steps 0 "$@" 2>> "$O/bench.err" | grep -a '^{') || { echo "FAILED $label" >&2; tail -5 $O/bench.err; exit 1; }
    echo "$line" >> "$O/rows.jsonl"
    printf '%-58s ' "$label" >> "$O/rows.txt"
    echo "$line" | python3 apps/bench/jline.py >> "$O/rows.txt"
}
for s in 1 2 4 8 16; do
  row "device sources=$s batch=128Ki" --source device --sources $s
done
for s in 1 4 16; do
  row "device sources=$s batch=1Mi" --source device --sources $s --batch 1048576
done
row "device sources=16 batch=128Ki inflight 2" --source device --sources 16 --inflight 2
GPU_MAX_HW_QUEUES=8 row "device sources=16 batch=128Ki 8 hw queues" --source device --sources 16
cat $O/rows.txt
