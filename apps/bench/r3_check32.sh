#!/bin/bash
# round 3: kernel trace of the non-incremental engine with 64 Ki keys (win 1024 / slide 128)
set -o pipefail
O=${1:-gpurun_out/r3/check32}
ROOT=$(pwd)
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
cd /tmp || exit 1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/trace" -o t -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --tuples-per-step 67108864 --device-probe-steps 0 --pipeline win_seq --keys 65536 > "$ROOT/$O/trace.log" 2>&1 || { tail -5 "$ROOT/$O/trace.log"; echo "trace failed"; exit 1; }
grep -a '^{' "$ROOT/$O/trace.log" | python3 "$ROOT/apps/bench/jline.py"
python3 - "$ROOT/$O/trace" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-90s calls %7s total %10.1f ms avg %9.1f us %5s%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cd "$ROOT"
find $O -name "*kernel_trace.csv" -size +8M -delete
