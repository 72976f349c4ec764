#!/bin/bash
# round 3: small host batches coalesced by a replica that has fallen behind (svc_bulk): GPU suite, small-batch rows with and without
set -o pipefail
O=${1:-gpurun_out/r3/check25}
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
if [ -z "$SKIP_PYTEST" ]; then
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; echo "pytest failed"; exit 1; }
tail -3 $O/pytest.log
fi
: > $O/rows.txt
row() {
    local label=$1; shift
    local line
    line=$(timeout -k 10 150 python3 bench.py --device-probe-steps 0 "$@" 2>> "$O/bench.err" | grep -a '^{') || { echo "FAILED $label" >&2; tail -5 $O/bench.err; exit 1; }
    printf '%-58s ' "$label" >> "$O/rows.txt"
    echo "$line" | python3 apps/bench/jline.py >> "$O/rows.txt"
}
row "flagship 16Ki" --steps 10 --warmup 2 --batch 16384 --tuples-per-step 134217728
WF_COALESCE_TUPLES=0 row "flagship 16Ki WF_COALESCE_TUPLES=0" --steps 10 --warmup 2 --batch 16384 --tuples-per-step 134217728
row "flagship 16Ki inflight 8" --steps 10 --warmup 2 --batch 16384 --tuples-per-step 134217728 --inflight 8
row "flagship 4Ki" --steps 10 --warmup 2 --batch 4096 --tuples-per-step 33554432
row "flagship 4Ki inflight 16" --steps 10 --warmup 2 --batch 4096 --tuples-per-step 33554432 --inflight 16
WF_COALESCE_TUPLES=0 row "flagship 4Ki WF_COALESCE_TUPLES=0" --steps 10 --warmup 2 --batch 4096 --tuples-per-step 33554432
row "flagship 128Ki" --steps 10 --warmup 2
row "flagship 64Ki" --steps 10 --warmup 2 --batch 65536
row "win_seq 1Ki keys 16Ki batches" --steps 10 --warmup 2 --pipeline win_seq --keys 1024 --batch 16384 --tuples-per-step 134217728
row "ysb 16Ki batches" --steps 10 --warmup 2 --pipeline ysb --batch 16384
row "map_filter 16Ki batches" --steps 10 --warmup 2 --pipeline map_filter --batch 16384
cat $O/rows.txt
for c in "" 0; do
  echo "== WF_COALESCE_TUPLES=$c" >> $O/ops.txt
  env WF_COALESCE_TUPLES=${c:-131072} timeout -k 10 60 apps/bin/bench_ops_gpu -o map_stateful -s 14 -n 14 -k 1000 -b 10000 -t 3 2>&1 | tail -1 >> $O/ops.txt || exit 1
  env WF_COALESCE_TUPLES=${c:-131072} timeout -k 10 60 apps/bin/bench_ops_gpu -o map_stateless -s 14 -n 14 -k 1000 -b 10000 -t 3 2>&1 | tail -1 >> $O/ops.txt || exit 1
  env WF_COALESCE_TUPLES=${c:-131072} timeout -k 10 60 apps/bin/bench_win_gpu -o kff_gpu -s 4 -n 4 -k 4096 -b 4096 -B -t 3 2>&1 | tail -1 >> $O/ops.txt || exit 1
  env WF_COALESCE_TUPLES=${c:-131072} timeout -k 10 60 apps/bin/bench_win_gpu -o kf_gpu -s 4 -n 4 -k 4096 -b 4096 -B -t 3 2>&1 | tail -1 >> $O/ops.txt || exit 1
done
cat $O/ops.txt
