#!/bin/bash
# run-to-run spread of the 64-shared-keys row (DEFAULT graph): short runs (the summary's 5 steps after 1) against longer ones
set -o pipefail
O=${1:-gpurun_out/r3/check47}
mkdir -p $O
export TMPDIR=/tmp
: > $O/shared64_spread.txt
for i in 1 2 3 4; do
  for sw in "5 1" "12 4"; do
    set -- $sw
    timeout -k 10 150 python bench.py --steps $1 --warmup $2 --keys 64 --shared-keys --tuples-per-step 134217728 --device-probe-steps 0 > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo failed; exit 1; }
    printf "run %s steps=%-2s warmup=%s " $i $1 $2 >> $O/shared64_spread.txt
    python3 apps/bench/jline.py < $O/j.json >> $O/shared64_spread.txt
    tail -1 $O/shared64_spread.txt
  done
done
