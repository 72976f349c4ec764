#!/bin/bash
# round 3: headline batch length with coalescing on - throughput and latency at 32 / 64 / 128 Ki batches, driver-style run
set -o pipefail
O=${1:-gpurun_out/r3/check28}
mkdir -p $O
export WF_QUIET=1
: > $O/rows.txt
row() {
    local label=$1; shift
    local line
    line=$(timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --device-probe-steps 0 "$@" 2>> "$O/bench.err" | grep -a '^{') || { echo "FAILED $label" >&2; tail -5 $O/bench.err; exit 1; }
    printf '%-40s ' "$label" >> "$O/rows.txt"
    echo "$line" | python3 apps/bench/jline.py >> "$O/rows.txt"
}
for rep in 1 2; do
row "64Ki x3" --batch 65536 --inflight 3
row "64Ki x2" --batch 65536 --inflight 2
row "96Ki x3" --batch 98304 --inflight 3
row "128Ki x3" --inflight 3
row "128Ki x2" --inflight 2
row "48Ki x4" --batch 49152
done
cat $O/rows.txt
