#!/bin/bash
# the 14-lane keyed map waits for its D2H copies (apps/bench/r3_check44.sh): copy stream policy and hardware queues
set -o pipefail
O=${1:-gpurun_out/r3/check45}
mkdir -p $O
export TMPDIR=/tmp
: > $O/d2h_policy.txt
run() {  # label, env assignments..., -- args
  local label=$1; shift
  printf "%-44s " "$label" >> $O/d2h_policy.txt
  env "$@" timeout -k 10 60 apps/bin/bench_ops_gpu -o $OP -s 14 -n 14 -k 1000 -b 10000 -t 3 2> $O/o.err | tail -1 >> $O/d2h_policy.txt || { tail -5 $O/o.err; echo "failed: $label"; exit 1; }
  tail -1 $O/d2h_policy.txt
}
for OP in map_stateful map_stateless filter_stateful; do
  run "default" WF_X=0
  run "WF_D2H_STREAMS=shared" WF_D2H_STREAMS=shared
  run "GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8
  run "GPU_MAX_HW_QUEUES=8 WF_D2H_STREAMS=shared" GPU_MAX_HW_QUEUES=8 WF_D2H_STREAMS=shared
  run "GPU_MAX_HW_QUEUES=16" GPU_MAX_HW_QUEUES=16
done
