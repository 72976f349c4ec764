#!/bin/bash
# round 3: fixed-size keys on the device, YSB hash join, balanced K1 groups - GPU suite + rows + host-load probe of map_stateful
set -o pipefail
O=${1:-gpurun_out/r3/check8}
ROOT=$(pwd)
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
if [ -z "$SKIP_PYTEST" ]; then
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; echo "pytest failed"; exit 1; }
tail -3 $O/pytest.log
fi
: > $O/rows.txt
: > $O/rows.jsonl
row() {  # row <label> <bench.py args...>
    local label=$1; shift
    local line
    line=$(timeout -k 10 150 python3 bench.py --steps 5 --warmup 1 --tuples-per-step 134217728 --device-probe-steps 0 "$@" 2>> "$O/bench.err" | grep -a '^{') || { echo "FAILED $label" >&2; tail -5 $O/bench.err; exit 1; }
    echo "$line" >> "$O/rows.jsonl"
    printf '%-58s ' "$label" >> "$O/rows.txt"
    echo "$line" | python3 apps/bench/jline.py >> "$O/rows.txt"
}
row "win_seq opaque keys=1"  --pipeline win_seq --keys 1
row "win_seq opaque keys=1 win=64 slide=1" --pipeline win_seq --keys 1 --win 64 --slide 1
row "win_seq opaque keys=1 tumbling 4096" --pipeline win_seq --keys 1 --win 4096 --slide 4096
row "win_seq opaque keys=1024" --pipeline win_seq --keys 1024
row "win_seq opaque keys=65536" --pipeline win_seq --keys 65536
row "ysb" --pipeline ysb
row "ysb keyby 2 agg replicas" --pipeline ysb --parallelism keyby --agg-replicas 2
row "flagship host-fed"
cat $O/rows.txt
for a in "-o map_stateful -s 14 -n 14 -k 1000 -b 10000" "-o map_stateful -s 7 -n 7 -k 1000 -b 10000" "-o map_stateful -s 4 -n 4 -k 1000 -b 10000" "-o map_stateful -s 14 -n 14 -k 1000 -b 10000 -d" "-o map_stateful -s 4 -n 4 -k 1000 -b 10000 -d"; do
  echo "== bench_ops_gpu $a" >> $O/ops.txt
  timeout -k 10 90 apps/bin/bench_ops_gpu $a -t 3 >> $O/ops.txt 2>&1 || { tail -5 $O/ops.txt; echo "bench_ops_gpu $a failed"; exit 1; }
done
cat $O/ops.txt
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
