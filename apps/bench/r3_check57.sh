#!/bin/bash
# run-to-run spread of the YSB split/merge rows and of the keyby flagship
set -o pipefail
O=${1:-gpurun_out/r3/check57}
mkdir -p $O
export TMPDIR=/tmp
: > $O/spread.txt
for i in 1 2 3 4 5; do
  timeout -k 10 150 python bench.py --steps 10 --warmup 2 --pipeline ysb --ysb-split-merge --agg-replicas 2 --device-probe-steps 0 > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo failed; exit 1; }
  printf "ysb split/merge   " >> $O/spread.txt; python3 apps/bench/jline.py < $O/j.json >> $O/spread.txt; tail -1 $O/spread.txt
done
for i in 1 2 3 4; do
  timeout -k 10 60 apps/bin/ysb_gpu_dist -l 5 -n 4 -m 2 -b 65536 -t local -r 2 -x 1 2>&1 | grep throughput | sed 's/^/ysb_gpu_dist local2 split /' | tee -a $O/spread.txt || { echo failed; exit 1; }
done
for i in 1 2 3; do
  timeout -k 10 150 python bench.py --steps 10 --warmup 3 --parallelism keyby --device-probe-steps 0 > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo failed; exit 1; }
  printf "flagship keyby    " >> $O/spread.txt; python3 apps/bench/jline.py < $O/j.json >> $O/spread.txt; tail -1 $O/spread.txt
done
