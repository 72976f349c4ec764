#!/bin/bash
# round 3: 64 Ki vs 128 Ki batches for the other win_agg modes (exchange on one GPU, shared keys)
set -o pipefail
O=${1:-gpurun_out/r3/check29}
mkdir -p $O
export WF_QUIET=1
: > $O/rows.txt
row() {
    local label=$1; shift
    local line
    line=$(timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --device-probe-steps 0 "$@" 2>> "$O/bench.err" | grep -a '^{') || { echo "FAILED $label" >&2; tail -5 $O/bench.err; exit 1; }
    printf '%-44s ' "$label" >> "$O/rows.txt"
    echo "$line" | python3 apps/bench/jline.py >> "$O/rows.txt"
}
for b in 65536 131072; do
row "keyby batch=$b" --parallelism keyby --batch $b
row "shared 64 keys DEFAULT batch=$b" --keys 64 --shared-keys --batch $b
row "shared 1Ki keys DEFAULT batch=$b" --keys 1024 --shared-keys --batch $b
row "shared 1Ki keys DETERMINISTIC batch=$b" --keys 1024 --shared-keys --deterministic --batch $b
row "shared 1Mi keys CB batch=$b" --shared-keys --count-based --batch $b
row "zipf batch=$b" --key-dist zipf --batch $b
done
cat $O/rows.txt
