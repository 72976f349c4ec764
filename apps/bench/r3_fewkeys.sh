#!/bin/bash
# round 3: few-key regimes of the device-fed flagship (long-run / huge-run kernels)
set -o pipefail
O=${1:-gpurun_out/r3/fewkeys}
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
for rep in 1 2; do
for k in 8 64 1024 16384 131072; do
  row "device TB keys=$k (rep $rep)" --source device --keys $k
done
done
for k in 8 1024 16384; do
  row "device CB keys=$k" --source device --keys $k --count-based
  row "host TB keys=$k" --keys $k
done
cat $O/rows.txt
