#!/bin/bash
# round 3: why 64 shared keys in a DEFAULT graph are three times slower with 64 Ki batches than with 128 Ki
set -o pipefail
O=${1:-gpurun_out/r3/check35}
ROOT=$(pwd)
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
cd /tmp || exit 1
for b in 65536 131072; do
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/trace_$b" -o t -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --tuples-per-step 67108864 --keys 64 --shared-keys --batch $b --device-probe-steps 0 > "$ROOT/$O/trace_$b.log" 2>&1 || { tail -5 "$ROOT/$O/trace_$b.log"; echo "trace failed"; exit 1; }
echo "== batch $b"
grep -a '^{' "$ROOT/$O/trace_$b.log" | python3 "$ROOT/apps/bench/jline.py"
python3 - "$ROOT/$O/trace_$b" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:8]:
    print("%-90s calls %7s total %10.1f ms avg %9.1f us %5s%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
cd "$ROOT"
find $O -name "*kernel_trace.csv" -size +8M -delete
