#!/bin/bash
# round 3: 128 / 96 / 64 Ki default batch on whatever box this lands on (the CPU share differs between boxes); first a CPU probe
set -o pipefail
O=${1:-gpurun_out/r3/check30}
mkdir -p $O
export WF_QUIET=1
timeout -k 10 60 apps/bin/bench_win_gpu -o kff_gpu -s 4 -n 4 -k 64 -b 65536 -t 2 2>&1 | tail -1
: > $O/rows.txt
row() {
    local label=$1; shift
    local line
    line=$(timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --device-probe-steps 0 "$@" 2>> "$O/bench.err" | grep -a '^{') || { echo "FAILED $label" >&2; tail -5 $O/bench.err; exit 1; }
    printf '%-40s ' "$label" >> "$O/rows.txt"
    echo "$line" | python3 apps/bench/jline.py >> "$O/rows.txt"
}
for rep in 1 2 3; do
row "128Ki x4" --batch 131072
row "96Ki x4" --batch 98304
row "64Ki x4" --batch 65536
done
cat $O/rows.txt
