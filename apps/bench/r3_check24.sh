#!/bin/bash
# round 3: who is busy at 16 Ki batches (per-replica statistics of the runtime: busy time of every thread head)
set -o pipefail
O=${1:-gpurun_out/r3/check24}
mkdir -p $O/trace16k $O/trace128k
export WF_QUIET=1 WF_TRACE=1
WF_LOG_DIR=$O/trace16k timeout -k 10 150 python3 bench.py --steps 10 --warmup 2 --batch 16384 --tuples-per-step 134217728 --device-probe-steps 0 > $O/b16k.json 2> $O/b16k.err || { tail $O/b16k.err; exit 1; }
WF_LOG_DIR=$O/trace128k timeout -k 10 150 python3 bench.py --steps 5 --warmup 2 --device-probe-steps 0 > $O/b128k.json 2> $O/b128k.err || { tail $O/b128k.err; exit 1; }
ls -la $O/trace16k | head
python3 - $O <<'PY'
import json, glob, sys, os
for d in ("trace16k", "trace128k"):
    print("==", d)
    for f in sorted(glob.glob(os.path.join(sys.argv[1], d, "*.json"))):
        j = json.load(open(f))
        print(os.path.basename(f), str(j)[:1500])
PY
