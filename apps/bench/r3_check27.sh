#!/bin/bash
# round 3: the reference's keyed-map configuration with more batches in flight per source (deeper queues: larger coalesced batches)
set -o pipefail
O=${1:-gpurun_out/r3/check27}
mkdir -p $O
export WF_QUIET=1
: > $O/ops.txt
for i in 6 12 16 24; do
  timeout -k 10 60 apps/bin/bench_ops_gpu -o map_stateful -s 14 -n 14 -k 1000 -b 10000 -t 3 -i $i 2>&1 | tail -1 | sed "s/^/inflight=$i /" >> $O/ops.txt || exit 1
done
for i in 6 16; do
  timeout -k 10 60 apps/bin/bench_ops_gpu -o map_stateless -s 14 -n 14 -k 1000 -b 10000 -t 3 -i $i 2>&1 | tail -1 | sed "s/^/inflight=$i /" >> $O/ops.txt || exit 1
  timeout -k 10 60 apps/bin/bench_ops_gpu -o filter_stateful -s 14 -n 14 -k 1000 -b 10000 -t 3 -i $i 2>&1 | tail -1 | sed "s/^/inflight=$i /" >> $O/ops.txt || exit 1
  timeout -k 10 60 apps/bin/bench_ops_gpu -o filter_stateless -s 14 -n 14 -k 1000 -b 10000 -t 3 -i $i 2>&1 | tail -1 | sed "s/^/inflight=$i /" >> $O/ops.txt || exit 1
done
env WF_COALESCE_TUPLES=262144 timeout -k 10 60 apps/bin/bench_ops_gpu -o map_stateful -s 14 -n 14 -k 1000 -b 10000 -t 3 -i 32 2>&1 | tail -1 | sed "s/^/inflight=32 coalesce=262144 /" >> $O/ops.txt || exit 1
env WF_COALESCE_TUPLES=65536 timeout -k 10 60 apps/bin/bench_ops_gpu -o map_stateful -s 14 -n 14 -k 1000 -b 10000 -t 3 -i 12 2>&1 | tail -1 | sed "s/^/inflight=12 coalesce=65536 /" >> $O/ops.txt || exit 1
cat $O/ops.txt
