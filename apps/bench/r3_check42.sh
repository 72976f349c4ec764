#!/bin/bash
# where the time of the keyed map at the reference's batch length goes: kernel trace with 8 lanes (71 keys per replica)
set -o pipefail
O=${1:-gpurun_out/r3/check42}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 60 $R/apps/bin/bench_ops_gpu -o map_stateful -s 8 -n 8 -k 568 -b 10000 -t 2 2> $R/$O/plain.err | tail -1 > $R/$O/plain.txt || { tail -5 $R/$O/plain.err; exit 1; }
cat $R/$O/plain.txt
WF_TRACE=1 WF_LOG_DIR=$R/$O/trace timeout -k 10 60 $R/apps/bin/bench_ops_gpu -o map_stateful -s 8 -n 8 -k 568 -b 10000 -t 2 2> $R/$O/trace.err | tail -1 || { tail -5 $R/$O/trace.err; exit 1; }
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o ms -- $R/apps/bin/bench_ops_gpu -o map_stateful -s 8 -n 8 -k 568 -b 10000 -t 2 > $R/$O/prof.log 2>&1 || { tail -5 $R/$O/prof.log; echo "rocprofv3 failed"; exit 1; }
tail -2 $R/$O/prof.log
ls -la $R/$O/prof/*/ 2>/dev/null | head; ls -la $R/$O/prof | head
