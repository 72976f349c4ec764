#!/bin/bash
# round 3: kernel traces of the tuple-fed DEFAULT/TB window operators
set -o pipefail
O=${1:-gpurun_out/r3/check20}
ROOT=$(pwd)
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
cd /tmp || exit 1
tr() {  # tr <label> <bench_win_gpu args...>
local label=$1; shift
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/trace_$label" -o t -- "$ROOT/apps/bin/bench_win_gpu" "$@" > "$ROOT/$O/trace_$label.log" 2>&1 || { tail -5 "$ROOT/$O/trace_$label.log"; echo "trace failed"; exit 1; }
echo "== $label"
grep -a 'tuples/s' "$ROOT/$O/trace_$label.log"
python3 - "$ROOT/$O/trace_$label" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print("%-90s calls %7s total %10.1f ms avg %9.1f us %5s%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
}
tr kf_tb -o kf_gpu -s 4 -n 4 -k 64 -b 65536 -t 2 -D
tr wmr_tb -o wmr_gpu -s 4 -n 4 -k 64 -b 65536 -t 2 -D
tr kf_cb -o kf_gpu -s 4 -n 4 -k 64 -b 65536 -t 2
cd "$ROOT"
find $O -name "*kernel_trace.csv" -size +8M -delete
