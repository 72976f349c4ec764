#!/bin/bash
# Where the engines stand away from the headline configuration, on one MI355X: window length, triggering delay, sources that
# share keys, key count - for the incremental engine (Key_FFAT_GPU, bench.py win_agg) and the non-incremental one (Key_Farm_GPU,
# bench.py win_seq).  Every line is a bench.py run of 5 timed steps of 128 Mi tuples.
#   bash apps/bench/regimes_r2.sh [out dir]
set -o pipefail
OUT=${1:-gpurun_out/r2/regimes}
mkdir -p "$OUT" || exit 1
export WF_QUIET=1
: > "$OUT/regimes.jsonl"
: > "$OUT/regimes.txt"
row() {  # row <label> <bench.py args...>
    local label=$1; shift
    echo "== $label: bench.py $*" >&2
    local line
    line=$(timeout -k 10 150 python3 bench.py --steps 5 --warmup 1 --tuples-per-step 134217728 "$@" 2>> "$OUT/bench.err" | grep -a '^{') || { echo "FAILED $label" >&2; exit 1; }
    echo "$line" >> "$OUT/regimes.jsonl"
    printf '%-58s ' "$label" >> "$OUT/regimes.txt"
    echo "$line" | python3 apps/bench/jline.py >> "$OUT/regimes.txt"
}
# count-based windows of growing length (panes per window = win / gcd(win, slide)); slide 1 = one result per tuple
for k in 8 1024 65536 1048576; do
    row "win_agg CB keys=$k win=16 slide=4" --keys $k --count-based
    row "win_agg CB keys=$k win=1024 slide=16" --keys $k --count-based --win 1024 --slide 16
    row "win_agg CB keys=$k win=1024 slide=1" --keys $k --count-based --win 1024 --slide 1
done
row "win_agg CB keys=8 win=4000 slide=4" --keys 8 --count-based --win 4000 --slide 4
# time-based windows: triggering delay (in tuples of a key), keys owned by one source
for k in 8 64 1024 1048576; do
    row "win_agg TB keys=$k delay=0" --keys $k
    row "win_agg TB keys=$k delay=8" --keys $k --delay 8
done
# every source emits every key: DEFAULT graph with the default delay (16 batches of time; late tuples are dropped and counted),
# and DETERMINISTIC graph (the device KeyBy merges its lanes in timestamp order: nothing is late)
for k in 64 1024 1048576; do
    row "win_agg TB shared keys=$k DEFAULT" --keys $k --shared-keys
    row "win_agg TB shared keys=$k DETERMINISTIC" --keys $k --shared-keys --deterministic
done
# non-incremental windows
for k in 1 8 1024 65536; do
    row "win_seq CB keys=$k win=1024 slide=128" --pipeline win_seq --keys $k
    row "win_seq CB keys=$k win=64 slide=1" --pipeline win_seq --keys $k --win 64 --slide 1
    row "win_seq CB keys=$k win=4096 slide=4096" --pipeline win_seq --keys $k --win 4096 --slide 4096
done
cat "$OUT/regimes.txt"
