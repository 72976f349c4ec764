#!/bin/bash
# few shared keys after the default became one replica: DEFAULT rows, and the DETERMINISTIC graph with 1 and 4 replicas
set -o pipefail
O=${1:-gpurun_out/r3/check55}
mkdir -p $O
export TMPDIR=/tmp
: > $O/one_replica.txt
row() {
  local label=$1 runs=$2; shift; shift
  for i in $(seq 1 $runs); do
    timeout -k 10 150 python bench.py --steps 5 --warmup 1 --shared-keys --tuples-per-step 134217728 --device-probe-steps 0 "$@" > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo "failed: $label"; exit 1; }
    printf "%-34s " "$label" >> $O/one_replica.txt
    python3 apps/bench/jline.py < $O/j.json >> $O/one_replica.txt
    tail -1 $O/one_replica.txt
  done
}
row "64 keys (default: 1 replica)" 5 --keys 64
row "256 keys (default: 1 replica)" 3 --keys 256
row "512 keys (default: 4 replicas)" 3 --keys 512
row "512 keys, 1 replica" 3 --keys 512 --agg-replicas 1
row "64 keys deterministic, 4 replicas" 3 --keys 64 --deterministic
row "64 keys deterministic, 1 replica" 3 --keys 64 --deterministic --agg-replicas 1
row "64 keys count-based" 2 --keys 64 --count-based
row "64 keys zipf" 2 --keys 64 --key-dist zipf
