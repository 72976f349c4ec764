#!/bin/bash
# WF_COALESCE_SHARE: a lone producer's slabs are grouped all at once (1) or at most half of them (2, the default)
set -o pipefail
O=${1:-gpurun_out/r3/check40}
mkdir -p $O
export TMPDIR=/tmp
: > $O/coalescing_share.txt
row() {  # label, bench.py args...
  local label=$1; shift
  for sh in 1 2; do
    WF_COALESCE_SHARE=$sh timeout -k 10 150 python bench.py "$@" --steps 10 --warmup 3 --device-probe-steps 0 > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo "failed: $label share=$sh"; exit 1; }
    printf "share=%s %-28s " $sh "$label" >> $O/coalescing_share.txt
    python3 apps/bench/jline.py < $O/j.json >> $O/coalescing_share.txt
    tail -1 $O/coalescing_share.txt
  done
}
row "16Ki x4" --batch 16384 --inflight 4
row "16Ki x8" --batch 16384 --inflight 8
row "4Ki x4" --batch 4096 --inflight 4
row "4Ki x16" --batch 4096 --inflight 16
row "32Ki x4" --batch 32768 --inflight 4
row "flagship (64Ki x4)"
row "map_filter" --pipeline map_filter
row "ysb" --pipeline ysb
for sh in 1 2; do
  for op in "map_stateful -i 6" "map_stateful -i 12" "map_stateless -i 6" "filter_stateful -i 6"; do
    printf "share=%s inflight/op=%-22s " $sh "$op" >> $O/coalescing_share.txt
    WF_COALESCE_SHARE=$sh timeout -k 10 60 apps/bin/bench_ops_gpu -o $op -s 14 -n 14 -k 1000 -b 10000 -t 3 2> $O/o.err | tail -1 >> $O/coalescing_share.txt || { tail -5 $O/o.err; echo "failed: $op share=$sh"; exit 1; }
    tail -1 $O/coalescing_share.txt
  done
done
