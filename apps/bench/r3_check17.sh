#!/bin/bash
# round 3: kernel traces of the middle key counts (device source) - where the time goes between 16 Ki and 128 Ki keys
set -o pipefail
O=${1:-gpurun_out/r3/check17}
ROOT=$(pwd)
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
cd /tmp || exit 1
tr() {  # tr <label> <bench.py args...>
local label=$1; shift
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/trace_$label" -o t -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --tuples-per-step 67108864 --device-probe-steps 0 "$@" > "$ROOT/$O/trace_$label.log" 2>&1 || { tail -5 "$ROOT/$O/trace_$label.log"; echo "trace failed"; exit 1; }
echo "== $label"
grep -a '^{' "$ROOT/$O/trace_$label.log" | python3 "$ROOT/apps/bench/jline.py"
python3 - "$ROOT/$O/trace_$label" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:9]:
    print("%-90s calls %7s total %10.1f ms avg %9.1f us %5s%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
}
tr keys128k --source device --keys 131072
tr keys16k --source device --keys 16384
tr uniform1m --source device --key-dist uniform
tr keys1m --source device
cd "$ROOT"
find $O -name "*kernel_trace.csv" -size +8M -delete
