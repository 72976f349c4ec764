#!/bin/bash
# round 3: the numbers the kernel work of the round is measured against (one MI355X).
#   bash apps/bench/r3_base.sh [out dir] ; SKIP_PYTEST=1 skips the GPU test suite
set -o pipefail
O=${1:-gpurun_out/r3/base}
mkdir -p $O
export WF_QUIET=1
if [ -z "$SKIP_PYTEST" ]; then
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; echo "pytest failed"; exit 1; }
tail -3 $O/pytest.log
fi
: > $O/rows.txt
row() {  # row <label> <bench.py args...>
    local label=$1; shift
    local line
    line=$(timeout -k 10 150 python3 bench.py --steps 5 --warmup 1 --tuples-per-step 134217728 --device-probe-steps 0 "$@" 2>> "$O/bench.err" | grep -a '^{') || { echo "FAILED $label" >&2; tail -5 $O/bench.err; exit 1; }
    echo "$line" >> "$O/rows.jsonl"
    printf '%-58s ' "$label" >> "$O/rows.txt"
    echo "$line" | python3 apps/bench/jline.py >> "$O/rows.txt"
}
: > $O/rows.jsonl
row "flagship host-fed"
row "flagship device source 128Ki" --source device
row "flagship device source 512Ki" --source device --batch 524288
row "flagship device source 16Ki" --source device --batch 16384
row "flagship host 16Ki" --batch 16384
row "win_agg CB keys=1Mi device" --source device --count-based
row "win_agg uniform keys device" --source device --key-dist uniform
row "win_seq opaque keys=1" --pipeline win_seq --keys 1
row "win_seq opaque keys=1 win=64 slide=1" --pipeline win_seq --keys 1 --win 64 --slide 1
row "win_seq opaque keys=1 tumbling 4096" --pipeline win_seq --keys 1 --win 4096 --slide 4096
row "win_seq opaque keys=1024" --pipeline win_seq --keys 1024
row "ysb" --pipeline ysb
cat $O/rows.txt
for a in "-o win_agg -s 8 -n 8 -k 131072 -b 131072 -d" "-o map_stateful -s 14 -n 14 -k 1000 -b 10000" "-o map_stateful -s 8 -n 8 -k 8 -b 131072" "-o map_stateful -s 8 -n 8 -k 64 -b 131072"; do
  echo "== bench_ops_gpu $a" >> $O/ops.txt
  timeout -k 10 90 apps/bin/bench_ops_gpu $a -t 3 >> $O/ops.txt 2>&1 || { tail -5 $O/ops.txt; echo "bench_ops_gpu $a failed"; exit 1; }
done
cat $O/ops.txt
