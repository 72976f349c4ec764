#!/bin/bash
# the two states of the 64-shared-keys row: per-replica statistics of several runs (WF_TRACE=1)
set -o pipefail
O=${1:-gpurun_out/r3/check48}
mkdir -p $O
export TMPDIR=/tmp
: > $O/runs.txt
for i in 1 2 3 4 5 6; do
  WF_TRACE=1 WF_LOG_DIR=$O/run$i timeout -k 10 150 python bench.py --steps 5 --warmup 1 --keys 64 --shared-keys --tuples-per-step 134217728 --device-probe-steps 0 > $O/j$i.json 2> $O/j$i.err || { tail -5 $O/j$i.err; echo failed; exit 1; }
  printf "run %s " $i >> $O/runs.txt
  python3 apps/bench/jline.py < $O/j$i.json >> $O/runs.txt
  tail -1 $O/runs.txt
done
