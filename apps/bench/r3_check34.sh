#!/bin/bash
# round 3: per-node statistics of the hybrid Win_MapReduce_GPU in a DEFAULT graph (host memory grows: where do tuples pile up?)
set -o pipefail
O=${1:-gpurun_out/r3/check34}
mkdir -p $O/trace
export WF_QUIET=1 WF_TRACE=1 WF_LOG_DIR=$O/trace
timeout -k 10 60 apps/bin/bench_win_gpu -o wmr_gpu -s 4 -n 4 -k 64 -b 65536 -t 8 -D > $O/out.txt 2>&1 || { tail $O/out.txt; exit 1; }
tail -1 $O/out.txt
python3 - $O <<'PY'
import json, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "trace", "graph_*.json"))[0]
j = json.load(open(f))
for o in j['operators']:
    print("  %-26s %-10s in %10d ignored %8d out %10d busy %8d us" % (o['operator'], o['replica'], o['inputs_received'], o['inputs_ignored'], o['outputs_sent'], o['busy_time_us']))
PY
