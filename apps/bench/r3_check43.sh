#!/bin/bash
# kernel durations of the keyed map at the reference's batch length, 4 lanes (71 keys per replica)
set -o pipefail
O=${1:-gpurun_out/r3/check43}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 60 $R/apps/bin/bench_ops_gpu -o map_stateful -s 4 -n 4 -k 284 -b 10000 -t 2 2> $R/$O/plain.err | tail -1 > $R/$O/plain.txt || { tail -5 $R/$O/plain.err; exit 1; }
cat $R/$O/plain.txt
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o ms -- $R/apps/bin/bench_ops_gpu -o map_stateful -s 4 -n 4 -k 284 -b 10000 -t 2 > $R/$O/prof.log 2>&1 || { grep -c SIGSEGV $R/$O/prof.log; echo "rocprofv3 failed"; exit 1; }
tail -1 $R/$O/prof.log
python3 - <<PY
import csv, glob
f = glob.glob("$R/$O/prof/**/ms_kernel_stats.csv", recursive=True)
for p in f:
    for i, row in enumerate(csv.reader(open(p))):
        if i < 14: print(",".join(c[:70] for c in row))
PY
# keep only the stats (the trace is large)
find $R/$O/prof -name "*kernel_trace.csv" -size +20M -delete
