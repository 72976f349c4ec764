#!/bin/bash
# round 3: the tuple-fed count-based run under the profiler (a host segmentation fault was seen once): keep the whole log
set -o pipefail
O=${1:-gpurun_out/r3/check22}
ROOT=$(pwd)
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp WF_DEBUG_EVENTS=1
cd /tmp || exit 1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/trace_kf_cb" -o t -- "$ROOT/apps/bin/bench_win_gpu" -o kf_gpu -s 4 -n 4 -k 64 -b 65536 -t 2 > "$ROOT/$O/trace_kf_cb.log" 2>&1
rc=$?
cd "$ROOT"
find $O -name "*kernel_trace.csv" -delete
echo "rc $rc"
tail -60 $O/trace_kf_cb.log
exit $rc
