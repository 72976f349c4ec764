#!/bin/bash
# round 3: kernel traces of the remaining slow rows (win 64 / slide 1 single key, YSB split/merge, the reference's keyed-map configuration)
set -o pipefail
O=${1:-gpurun_out/r3/check23}
ROOT=$(pwd)
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
cd /tmp || exit 1
stats() {
python3 - "$1" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:10]:
    print("%-90s calls %7s total %10.1f ms avg %9.1f us %5s%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
}
trp() {  # trp <label> <bench.py args...>
local label=$1; shift
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/trace_$label" -o t -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --tuples-per-step 67108864 --device-probe-steps 0 "$@" > "$ROOT/$O/trace_$label.log" 2>&1 || { tail -5 "$ROOT/$O/trace_$label.log"; echo "trace failed"; exit 1; }
echo "== $label"
grep -a '^{' "$ROOT/$O/trace_$label.log" | python3 "$ROOT/apps/bench/jline.py"
stats "$ROOT/$O/trace_$label"
}
trp win64s1 --pipeline win_seq --keys 1 --win 64 --slide 1
trp ysb_split --pipeline ysb --ysb-split-merge --agg-replicas 2
echo "== map_stateful 14x1000x10000"
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/trace_map14" -o t -- "$ROOT/apps/bin/bench_ops_gpu" -o map_stateful -s 14 -n 14 -k 1000 -b 10000 -t 2 > "$ROOT/$O/trace_map14.log" 2>&1 || { tail -5 "$ROOT/$O/trace_map14.log"; echo "trace failed"; exit 1; }
grep -a 'tuples/s' "$ROOT/$O/trace_map14.log"
stats "$ROOT/$O/trace_map14"
cd "$ROOT"
find $O -name "*kernel_trace.csv" -size +8M -delete
