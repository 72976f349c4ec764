#!/bin/bash
# round 3: a larger coalescing limit for pipelines whose single replica is the bound (single-key non-incremental windows)
set -o pipefail
O=${1:-gpurun_out/r3/check38}
mkdir -p $O
export WF_QUIET=1
: > $O/rows.txt
row() {
    local label=$1; shift
    local line
    line=$(timeout -k 10 150 python3 bench.py --steps 8 --warmup 2 --device-probe-steps 0 "$@" 2>> "$O/bench.err" | grep -a '^{') || { echo "FAILED $label" >&2; tail -5 $O/bench.err; exit 1; }
    printf '%-56s ' "$label" >> "$O/rows.txt"
    echo "$line" | python3 apps/bench/jline.py >> "$O/rows.txt"
}
row "win_seq 1 key" --pipeline win_seq --keys 1
row "win_seq 1 key wave_reduce" --pipeline win_seq --keys 1 --wave-reduce
row "win_seq 1 key tumbling 4096" --pipeline win_seq --keys 1 --win 4096 --slide 4096
row "win_seq 1 key win 64 slide 1" --pipeline win_seq --keys 1 --win 64 --slide 1
row "win_seq 8 keys" --pipeline win_seq --keys 8
row "win_seq 1Ki keys" --pipeline win_seq --keys 1024
row "win_seq 64Ki keys" --pipeline win_seq --keys 65536
row "flagship"
row "flagship 1 key 1 source" --keys 1 --sources 1
row "flagship 8 keys" --keys 8
row "ysb" --pipeline ysb
row "map_filter" --pipeline map_filter
timeout -k 10 60 apps/bin/bench_win_gpu -o wmr_gpu -s 4 -n 4 -k 64 -b 65536 -t 3 -D 2>&1 | tail -1 >> $O/rows.txt
timeout -k 10 60 apps/bin/bench_win_gpu -o kf_gpu -s 4 -n 1 -k 64 -b 65536 -t 3 -D 2>&1 | tail -1 >> $O/rows.txt
timeout -k 10 60 apps/bin/bench_ops_gpu -o map_stateful -s 8 -n 2 -k 1024 -t 3 2>&1 | tail -1 >> $O/rows.txt
cat $O/rows.txt
