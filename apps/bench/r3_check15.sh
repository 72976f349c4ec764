#!/bin/bash
# round 3: where the time goes when few keys are shared by unsynchronised lanes (DEFAULT graph, time-based windows)
set -o pipefail
O=${1:-gpurun_out/r3/check15}
ROOT=$(pwd)
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
cd /tmp || exit 1
for k in 64 1024; do
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/trace_shared$k" -o t -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --tuples-per-step 67108864 --keys $k --shared-keys --device-probe-steps 0 > "$ROOT/$O/trace_shared$k.log" 2>&1 || { tail -5 "$ROOT/$O/trace_shared$k.log"; echo "trace failed"; exit 1; }
grep -a '^{' "$ROOT/$O/trace_shared$k.log" | python3 "$ROOT/apps/bench/jline.py"
python3 - "$ROOT/$O/trace_shared$k" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-90s calls %7s total %10.1f ms avg %9.1f us %5s%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
cd "$ROOT"
find $O -name "*kernel_trace.csv" -size +8M -delete
