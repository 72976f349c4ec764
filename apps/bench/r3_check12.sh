#!/bin/bash
# round 3: rescue sort (one launch behind a confident gate) - GPU suite + rows
set -o pipefail
O=${1:-gpurun_out/r3/check12}
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
for rep in 1 2; do
row "device 128Ki (rep $rep)" --source device
WF_SORT_RESCUE=0 row "device 128Ki WF_SORT_RESCUE=0 (rep $rep)" --source device
done
row "device 512Ki" --source device --batch 524288
WF_SORT_RESCUE=0 row "device 512Ki WF_SORT_RESCUE=0" --source device --batch 524288
row "device 16Ki" --source device --batch 16384
row "host 128Ki" 
row "host 16Ki" --batch 16384
row "device CB 128Ki" --source device --count-based
row "device uniform 128Ki" --source device --key-dist uniform
row "device keys=1024 128Ki" --source device --keys 1024
row "win_seq opaque keys=65536" --pipeline win_seq --keys 65536
row "win_seq opaque keys=1Mi win=16 slide=4" --pipeline win_seq --keys 1048576 --win 16 --slide 4
WF_SORT_RESCUE=0 row "win_seq opaque keys=1Mi win=16 slide=4 WF_SORT_RESCUE=0" --pipeline win_seq --keys 1048576 --win 16 --slide 4
cat $O/rows.txt
cd /tmp || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/trace_flagship" -o t -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --tuples-per-step 67108864 --source device --device-probe-steps 0 > "$ROOT/$O/trace_flagship.log" 2>&1 || { tail -5 "$ROOT/$O/trace_flagship.log"; echo "trace failed"; exit 1; }
cd "$ROOT"
find $O -name "*kernel_trace.csv" -size +8M -delete
