#!/bin/bash
# kernel statistics of the 64-shared-keys row with the one-replica plan
set -o pipefail
O=${1:-gpurun_out/r3/check59}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o sk -- python3 bench.py --shared-keys --keys 64 --steps 3 --warmup 1 --tuples-per-step 134217728 --device-probe-steps 0 > $R/$O/prof.log 2>&1 || { grep -c SIGSEGV $R/$O/prof.log; tail -3 $R/$O/prof.log; echo "rocprofv3 failed"; exit 1; }
grep -a '^{' $R/$O/prof.log | python3 apps/bench/jline.py
find $R/$O/prof -name "*kernel_trace.csv" -delete
find $R/$O/prof -name "*kernel_stats.csv" | head -2
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for i,row in enumerate(csv.reader(open(sys.argv[1]))):
    if i<14: print(",".join(c[:60] for c in row))
PY
