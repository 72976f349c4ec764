#!/bin/bash
# round 3: the thread-per-run / wave-per-run threshold of Key_FFAT_GPU in the middle key counts (device source)
set -o pipefail
O=${1:-gpurun_out/r3/check36}
mkdir -p $O
export WF_QUIET=1
: > $O/rows.txt
row() {
    local label=$1; shift
    local line
    line=$(timeout -k 10 150 python3 bench.py --steps 5 --warmup 1 --tuples-per-step 134217728 --device-probe-steps 0 --source device "$@" 2>> "$O/bench.err" | grep -a '^{') || { echo "FAILED $label" >&2; tail -5 $O/bench.err; exit 1; }
    printf '%-46s ' "$label" >> "$O/rows.txt"
    echo "$line" | python3 apps/bench/jline.py >> "$O/rows.txt"
}
for lr in 12 4 6 24 48 96; do
  export WF_FFAT_LONG_RUN=$lr
  row "long_run=$lr keys=131072 (runs of 8)" --keys 131072
  row "long_run=$lr keys=32768 (runs of 32)" --keys 32768
  row "long_run=$lr keys=16384 (runs of 64)" --keys 16384
  row "long_run=$lr uniform 1Mi" --key-dist uniform
done
cat $O/rows.txt
