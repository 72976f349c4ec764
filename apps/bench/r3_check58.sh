#!/bin/bash
# YSB split/merge graph: one or two aggregation replicas, four runs each
set -o pipefail
O=${1:-gpurun_out/r3/check58}
mkdir -p $O
export TMPDIR=/tmp
: > $O/ysb_split_replicas.txt
for n in 1 2; do
for i in 1 2 3 4; do
  timeout -k 10 150 python bench.py --steps 10 --warmup 2 --pipeline ysb --ysb-split-merge --agg-replicas $n --device-probe-steps 0 > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo failed; exit 1; }
  printf "ysb split/merge agg-replicas=%s  " $n >> $O/ysb_split_replicas.txt; python3 apps/bench/jline.py < $O/j.json >> $O/ysb_split_replicas.txt; tail -1 $O/ysb_split_replicas.txt
done
done
