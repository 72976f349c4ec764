#!/bin/bash
# round 3: A/B of the finalize fold and of the ring-line prefetch, small sort only behind a closed gate
set -o pipefail
O=${1:-gpurun_out/r3/check10}
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
for rep in 1 2; do
row "flagship device 128Ki (rep $rep)" --source device
WF_FOLD_FINALIZE=0 row "flagship device 128Ki WF_FOLD_FINALIZE=0 (rep $rep)" --source device
done
row "flagship device 512Ki" --source device --batch 524288
row "flagship device 16Ki" --source device --batch 16384
row "flagship host 16Ki" --batch 16384
WF_FOLD_FINALIZE=0 row "flagship host 16Ki WF_FOLD_FINALIZE=0" --batch 16384
row "win_agg uniform keys device 16Ki" --source device --key-dist uniform --batch 16384
row "win_agg TB keys=1024 16Ki" --keys 1024 --batch 16384
row "win_seq opaque keys=1024 16Ki" --pipeline win_seq --keys 1024 --batch 16384
row "win_agg uniform keys device 128Ki" --source device --key-dist uniform
cat $O/rows.txt
for a in "-o map_stateful -s 14 -n 14 -k 1000 -b 10000" "-o filter_stateful -s 14 -n 14 -k 1000 -b 10000" "-o win_agg -s 8 -n 8 -k 131072 -b 131072 -d"; do
  echo "== bench_ops_gpu $a" >> $O/ops.txt
  timeout -k 10 90 apps/bin/bench_ops_gpu $a -t 3 >> $O/ops.txt 2>&1 || { tail -5 $O/ops.txt; echo "bench_ops_gpu $a failed"; exit 1; }
done
cat $O/ops.txt
