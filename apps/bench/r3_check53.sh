#!/bin/bash
# 64 shared keys: do the two states come from which streams share a hardware queue?  (every stream a queue of its own with 24)
set -o pipefail
O=${1:-gpurun_out/r3/check53}
mkdir -p $O
export TMPDIR=/tmp
: > $O/hw_queues.txt
for q in 24 8 2; do
  for i in 1 2 3 4 5 6; do
    GPU_MAX_HW_QUEUES=$q timeout -k 10 150 python bench.py --steps 5 --warmup 1 --keys 64 --shared-keys --tuples-per-step 134217728 --device-probe-steps 0 > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo "failed: $q"; exit 1; }
    printf "GPU_MAX_HW_QUEUES=%-3s " $q >> $O/hw_queues.txt
    python3 apps/bench/jline.py < $O/j.json >> $O/hw_queues.txt
    tail -1 $O/hw_queues.txt
  done
done
