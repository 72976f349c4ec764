#!/bin/bash
# round 3: who is busy behind the exchange on one GPU (runtime statistics per thread head)
set -o pipefail
O=${1:-gpurun_out/r3/check31}
mkdir -p $O/trace_keyby
export WF_QUIET=1 WF_TRACE=1
WF_LOG_DIR=$O/trace_keyby timeout -k 10 150 python3 bench.py --steps 8 --warmup 2 --parallelism keyby --device-probe-steps 0 > $O/keyby.json 2> $O/keyby.err || { tail $O/keyby.err; exit 1; }
cat $O/keyby.json | python3 apps/bench/jline.py
python3 - $O <<'PY'
import json, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "trace_keyby", "graph_*.json"))[0]
j = json.load(open(f))
print("elapsed", j['elapsed_us'], "threads", j['threads'])
seen = set()
for o in j['operators']:
    k = o['operator']
    if (k, o['replica']) in seen: continue
    seen.add((k, o['replica']))
    if o['replica'] in ('replica_0', 'replica_1'):
        print("  %-22s %-10s in %7d out %7d service %8.1f us busy %8d us running %8d kernels %s" % (k, o['replica'], o['inputs_received'], o['outputs_sent'], o['service_time_us'], o['busy_time_us'], o['running_time_us'], o.get('num_kernels')))
PY
