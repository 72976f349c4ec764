#!/bin/bash
# after WF_COALESCE_SHARE=2 became the default: the reference's 14-lane operator configurations with the benchmark's default ring depth
set -o pipefail
O=${1:-gpurun_out/r3/check41}
mkdir -p $O
export TMPDIR=/tmp
: > $O/ops14.txt
for op in map_stateful map_stateless filter_stateful filter_stateless; do
  timeout -k 10 60 apps/bin/bench_ops_gpu -o $op -s 14 -n 14 -k 1000 -b 10000 -t 3 2> $O/o.err | tail -1 >> $O/ops14.txt || { tail -5 $O/o.err; echo "failed: $op"; exit 1; }
  tail -1 $O/ops14.txt
done
