#!/bin/bash
# 16 Ki batches: the coalescing limit against the batches in flight (throughput vs tail latency)
set -o pipefail
O=${1:-gpurun_out/r3/check39}
mkdir -p $O
export TMPDIR=/tmp
: > $O/sweep16k.txt
for lim in 32768 49152 65536 131072; do
  for inf in 2 3 4; do
    WF_COALESCE_TUPLES=$lim timeout -k 10 120 python bench.py --batch 16384 --inflight $inf --steps 10 --warmup 3 --device-probe-steps 0 > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo "failed at limit=$lim inflight=$inf"; exit 1; }
    printf "limit=%-7s inflight=%s " $lim $inf >> $O/sweep16k.txt
    python3 apps/bench/jline.py < $O/j.json >> $O/sweep16k.txt
    tail -1 $O/sweep16k.txt
  done
done
