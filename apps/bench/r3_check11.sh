#!/bin/bash
# round 3: A/B of the ring-line prefetch in the short-run kernel (device-fed flagship, alternating, 3 repetitions)
set -o pipefail
O=${1:-gpurun_out/r3/check11}
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
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
for rep in 1 2 3; do
WF_FFAT_PREFETCH=1 row "device 128Ki prefetch on (rep $rep)" --source device
WF_FFAT_PREFETCH=0 row "device 128Ki prefetch off (rep $rep)" --source device
done
WF_FFAT_PREFETCH=1 row "device 512Ki prefetch on" --source device --batch 524288
WF_FFAT_PREFETCH=0 row "device 512Ki prefetch off" --source device --batch 524288
WF_FFAT_PREFETCH=1 row "device uniform 128Ki prefetch on" --source device --key-dist uniform
WF_FFAT_PREFETCH=0 row "device uniform 128Ki prefetch off" --source device --key-dist uniform
WF_SORT_SKIP=0 row "device 128Ki sort always" --source device
cat $O/rows.txt
