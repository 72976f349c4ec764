#!/bin/bash
# (needs apps/bin/bench_ops_gpu_timing: bench_ops_gpu built with -DWF_SHIP_TIMING from a tree that timed GPU_Replica::ship and DeviceToHost::pump;
# the instrumentation was removed again after the measurement - see profiles/r3_keyed_map_14lanes/README.md)
# host time of a keyed-map replica per coalesced group (instrumented build, -DWF_SHIP_TIMING)
set -o pipefail
O=${1:-gpurun_out/r3/check44}
mkdir -p $O
export TMPDIR=/tmp
for cfg in "-s 4 -n 4 -k 284" "-s 8 -n 8 -k 568" "-s 14 -n 14 -k 1000"; do
  timeout -k 10 60 apps/bin/bench_ops_gpu_timing -o map_stateful $cfg -b 10000 -t 2 2> $O/t.err | tail -1 || { tail -5 $O/t.err; exit 1; }
  grep "ship timing" $O/t.err | head -2
  grep "d2h timing" $O/t.err | head -2
done
timeout -k 10 60 apps/bin/bench_ops_gpu_timing -o map_stateless -s 8 -n 8 -k 568 -b 10000 -t 2 2> $O/t.err | tail -1 || { tail -5 $O/t.err; exit 1; }
grep "ship timing" $O/t.err | head -2
grep "d2h timing" $O/t.err | head -2
