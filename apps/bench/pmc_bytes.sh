#!/bin/bash
# HBM bytes per input tuple (FETCH_SIZE / WRITE_SIZE, one rocprofv3 --pmc pass each) of any bench.py configuration:
#   bash apps/bench/pmc_bytes.sh <out dir> <name> <bench.py args...>
set -o pipefail
ROOT=$(pwd)
OUT=$1; NAME=$2; shift 2
mkdir -p "$OUT" || exit 1
export TMPDIR=/tmp WF_QUIET=1
cd /tmp || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "== $NAME $ctr" >&2
    timeout -s KILL 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$ROOT/$OUT/${NAME}_$ctr" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --tuples-per-step 33554432 "$@" > "$ROOT/$OUT/${NAME}_$ctr.log" 2>&1 || { echo "FAILED $NAME $ctr" >&2; exit 1; }
done
cd "$ROOT" && python3 - "$OUT" "$NAME" <<'PY'
import collections, csv, sys
out, name = sys.argv[1], sys.argv[2]
tuples = 2 * 33554432  # warm-up step + timed step
per_kernel = collections.defaultdict(lambda: [0.0, 0.0])
tot = [0.0, 0.0]
for i, ctr in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for r in csv.DictReader(open("%s/%s_%s/p_counter_collection.csv" % (out, name, ctr))):
        v = float(r["Counter_Value"]) * 1024.0  # the counters are in KiB
        tot[i] += v
        per_kernel[r["Kernel_Name"].split("(")[0].replace("void ", "")][i] += v
print("%s: %.1f B fetched, %.1f B written per input tuple (all kernels; a tuple is 40 bytes)" % (name, tot[0] / tuples, tot[1] / tuples))
for k, (f, w) in sorted(per_kernel.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:6]:
    print("    %-70s %8.1f %8.1f" % (k[:70], f / tuples, w / tuples))
PY
