#!/bin/bash
# the slow state of the 64-shared-keys row and the 16 CPUs of the box: 17 threads that spin while they wait
set -o pipefail
O=${1:-gpurun_out/r3/check51}
mkdir -p $O
export TMPDIR=/tmp
: > $O/threads.txt
row() {  # label, env, args
  local label=$1 e=$2; shift; shift
  for i in 1 2 3 4 5; do
    env $e timeout -k 10 150 python bench.py --steps 5 --warmup 1 --keys 64 --shared-keys --tuples-per-step 134217728 --device-probe-steps 0 "$@" > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo "failed: $label"; exit 1; }
    printf "%-24s " "$label" >> $O/threads.txt
    python3 apps/bench/jline.py < $O/j.json >> $O/threads.txt
    tail -1 $O/threads.txt
  done
}
row "WF_BLOCKING=1" WF_BLOCKING=1
row "WF_EVENT_BLOCKING=1" WF_EVENT_BLOCKING=1
row "both" "WF_BLOCKING=1 WF_EVENT_BLOCKING=1"
row "6 sources" WF_X=0 --sources 6
row "agg replicas 3" WF_X=0 --agg-replicas 3
