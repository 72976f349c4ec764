#!/bin/bash
# few shared keys, one replica by default: the remaining rows (count-based, zipf, 8 keys, the guard of the DETERMINISTIC graph)
set -o pipefail
O=${1:-gpurun_out/r3/check56}
mkdir -p $O
export TMPDIR=/tmp
: > $O/one_replica_more.txt
row() {
  local label=$1 runs=$2; shift; shift
  for i in $(seq 1 $runs); do
    timeout -k 10 100 python bench.py --steps 5 --warmup 1 --shared-keys --tuples-per-step 134217728 --device-probe-steps 0 "$@" > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo "failed: $label"; exit 1; }
    printf "%-34s " "$label" >> $O/one_replica_more.txt
    python3 apps/bench/jline.py < $O/j.json >> $O/one_replica_more.txt
    tail -1 $O/one_replica_more.txt
  done
}
row "64 keys" 3 --keys 64
row "512 keys" 3 --keys 512
row "64 keys zipf" 2 --keys 64 --key-dist zipf
row "64 keys, 64 Ki batches" 2 --keys 64 --batch 65536
row "64 keys deterministic" 2 --keys 64 --deterministic
row "1024 keys" 2 --keys 1024
row "64 keys deterministic, 1 replica" 1 --keys 64 --deterministic --agg-replicas 1
