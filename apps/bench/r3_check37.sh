#!/bin/bash
# round 3: the hybrid two-stage GPU operators fed with BATCHES (Batch_Source -> PLQ / MAP on the device -> WLQ / REDUCE on the host)
set -o pipefail
export WF_QUIET=1
B=apps/bin/bench_win_gpu
for op in pf_hy wmr_hy pf_gpu wmr_gpu; do
  timeout -k 10 60 $B -o $op -s 4 -n 4 -k 256 -b 131072 -B -t 3 2>&1 | tail -2 || exit 1
done
timeout -k 10 60 $B -o pf_hy -s 4 -n 2 -k 256 -b 65536 -B -t 3 2>&1 | tail -2 || exit 1
timeout -k 10 60 $B -o wmr_hy -s 4 -n 2 -k 256 -b 16384 -B -t 3 2>&1 | tail -2 || exit 1
