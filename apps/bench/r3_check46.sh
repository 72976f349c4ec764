#!/bin/bash
# (needs apps/bin/bench_ops_gpu_timing: bench_ops_gpu built with -DWF_SHIP_TIMING from a tree that timed GPU_Replica::ship and DeviceToHost::pump;
# the instrumentation was removed again after the measurement - see profiles/r3_keyed_map_14lanes/README.md)
# DeviceToHost: one copy in flight at a time (WF_D2H_PIPELINE=0, before) against every copy enqueued as soon as its batch is done
set -o pipefail
O=${1:-gpurun_out/r3/check46}
mkdir -p $O
export TMPDIR=/tmp
: > $O/d2h_pipeline_ops.txt
for op in map_stateful map_stateless filter_stateful filter_stateless; do
  for pl in 0 1; do
    printf "WF_D2H_PIPELINE=%s " $pl >> $O/d2h_pipeline_ops.txt
    WF_D2H_PIPELINE=$pl timeout -k 10 60 apps/bin/bench_ops_gpu_timing -o $op -s 14 -n 14 -k 1000 -b 10000 -t 3 2> $O/t.err | tail -1 >> $O/d2h_pipeline_ops.txt || { tail -5 $O/t.err; exit 1; }
    tail -1 $O/d2h_pipeline_ops.txt
    grep "ship timing" $O/t.err | head -2
  done
done
for pl in 0 1; do
  printf "WF_D2H_PIPELINE=%s " $pl >> $O/d2h_pipeline_ops.txt
  WF_D2H_PIPELINE=$pl timeout -k 10 60 apps/bin/bench_ops_gpu_timing -o map_stateful -s 8 -n 8 -k 512 -b 131072 -t 3 2> $O/t.err | tail -1 >> $O/d2h_pipeline_ops.txt || { tail -5 $O/t.err; exit 1; }
  tail -1 $O/d2h_pipeline_ops.txt
done
