#!/bin/bash
# few shared keys: one or two Key_FFAT_GPU replicas (a single replica needs no device KeyBy in front)
set -o pipefail
O=${1:-gpurun_out/r3/check54}
mkdir -p $O
export TMPDIR=/tmp
: > $O/few_replicas.txt
for keys in 64 8 256 1024; do
for n in 1 2; do
  for i in 1 2 3 4; do
    timeout -k 10 150 python bench.py --steps 5 --warmup 1 --keys $keys --shared-keys --tuples-per-step 134217728 --device-probe-steps 0 --agg-replicas $n > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo "failed: $n"; exit 1; }
    printf "keys=%-5s agg-replicas=%s " $keys $n >> $O/few_replicas.txt
    python3 apps/bench/jline.py < $O/j.json >> $O/few_replicas.txt
    tail -1 $O/few_replicas.txt
  done
done
done
