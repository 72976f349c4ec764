#!/bin/bash
# throughput / latency trade-off of the flagship: batch length x batches in flight
# usage: apps/bench/latency_sweep.sh  -> one line per point
cd "$(dirname "$0")/../.."
export WF_QUIET=1
for cfg in "16384 3" "32768 3" "65536 4" "131072 4" "262144 6" "524288 6"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --steps 8 --warmup 2 --batch $1 --inflight $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('batch=%7d inflight=%d : %8.1f M tuples/s  p50 %8.0f us  p99 %8.0f us' % ($1, $2, d['value'] / 1e6, d['p50_latency_us'], d['p99_latency_us']))" || exit 1
done
