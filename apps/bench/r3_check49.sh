#!/bin/bash
# WF_STREAM_PRIORITY: operator compute streams in a hardware-queue pool of their own (HIP keeps one pool per stream priority)
set -o pipefail
O=${1:-gpurun_out/r3/check49}
mkdir -p $O
export TMPDIR=/tmp
: > $O/stream_priority.txt
row() {  # label, runs, bench.py args...
  local label=$1 runs=$2; shift; shift
  for pr in 0 1; do
    for i in $(seq 1 $runs); do
      WF_STREAM_PRIORITY=$pr timeout -k 10 150 python bench.py "$@" --device-probe-steps 0 > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo "failed: $label priority=$pr"; exit 1; }
      printf "priority=%s %-22s " $pr "$label" >> $O/stream_priority.txt
      python3 apps/bench/jline.py < $O/j.json >> $O/stream_priority.txt
      tail -1 $O/stream_priority.txt
    done
  done
}
row "shared64" 6 --steps 5 --warmup 1 --keys 64 --shared-keys --tuples-per-step 134217728
row "flagship" 2 --steps 10 --warmup 3
row "flagship 16Ki" 2 --steps 10 --warmup 3 --batch 16384
row "shared1ki" 2 --steps 5 --warmup 1 --keys 1024 --shared-keys --tuples-per-step 134217728
row "keyby" 1 --steps 10 --warmup 3 --parallelism keyby
row "winseq 1 key" 1 --steps 10 --warmup 2 --pipeline win_seq --keys 1
row "ysb" 1 --steps 10 --warmup 2 --pipeline ysb
row "map_filter" 1 --steps 10 --warmup 2 --pipeline map_filter
row "device" 1 --steps 10 --warmup 2 --source device
