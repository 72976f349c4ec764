#!/bin/bash
# A/B of the copy / synchronisation policies of the GPU runtime (3 repetitions per point, medians matter:
# single 2-second runs vary by +-20 % with the way streams happen to share hardware queues).
# usage: apps/bench/ab_sync.sh "VAR=value ..." ["VAR=value ..." ...]
cd "$(dirname "$0")/../.."
export WF_QUIET=1
for variant in "$@"; do
  echo "== $variant"
  for cfg in "map_stateful 1024" "filter_stateful 1024" "map_stateless 16384"; do
    set -- $cfg
    for rep in 1 2 3; do
      env $variant timeout -k 10 60 apps/bin/bench_ops_gpu -o $1 -s 8 -k $2 -b 262144 -t 2 | sed 's/sources=8 replicas=8 //; s/ (.*//' || exit 1
    done
  done
  for k in 64 1048576; do
    for rep in 1 2 3; do
      env $variant timeout -k 10 100 python bench.py --steps 20 --warmup 3 --keys $k 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('flagship keys=%d : %.1f M tuples/s p50 %.0f p99 %.0f us' % (d['config']['keys_per_gpu'], d['value'] / 1e6, d['p50_latency_us'], d['p99_latency_us']))" || exit 1
    done
  done
done
