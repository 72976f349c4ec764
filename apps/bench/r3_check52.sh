#!/bin/bash
# 64 shared keys, DEFAULT graph: the number of Key_FFAT_GPU replicas behind the device KeyBy, five runs each
set -o pipefail
O=${1:-gpurun_out/r3/check52}
mkdir -p $O
export TMPDIR=/tmp
: > $O/agg_replicas.txt
for keys in 64 1024; do
for n in 2 3 4 5 6 8; do
  for i in 1 2 3 4 5; do
    timeout -k 10 150 python bench.py --steps 5 --warmup 1 --keys $keys --shared-keys --tuples-per-step 134217728 --device-probe-steps 0 --agg-replicas $n > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo "failed: $n"; exit 1; }
    printf "keys=%-5s agg-replicas=%s " $keys $n >> $O/agg_replicas.txt
    python3 apps/bench/jline.py < $O/j.json >> $O/agg_replicas.txt
    tail -1 $O/agg_replicas.txt
  done
done
done
