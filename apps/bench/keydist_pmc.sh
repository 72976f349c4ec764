#!/bin/bash
# HBM traffic per tuple of the flagship under the three key distributions (1 Mi keys): FETCH_SIZE and WRITE_SIZE, one
# rocprofv3 --pmc pass each.   bash apps/bench/keydist_pmc.sh [out dir]
set -o pipefail
ROOT=$(pwd)
OUT=${1:-gpurun_out/r2/key_dist_pmc}
mkdir -p "$OUT" || exit 1
export TMPDIR=/tmp WF_QUIET=1
cd /tmp || exit 1
for dist in roundrobin uniform zipf; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "== $dist $ctr" >&2
    timeout -s KILL 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$ROOT/$OUT/${dist}_$ctr" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --keys 1048576 --key-dist $dist --count-based --tuples-per-step 33554432 > "$ROOT/$OUT/${dist}_$ctr.log" 2>&1 || { echo "FAILED $dist $ctr" >&2; exit 1; }
  done
done
cd "$ROOT" && python3 - "$OUT" <<'PY'
import collections, csv, sys
out = sys.argv[1]
tuples = 2 * 33554432  # warm-up step + timed step
print("%-11s %14s %14s   (bytes per input tuple, all kernels of the run; a tuple is 40 bytes)" % ("dist", "fetched", "written"))
for dist in ("roundrobin", "uniform", "zipf"):
    tot = {}
    per_kernel = collections.defaultdict(lambda: [0.0, 0.0])
    for i, ctr in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        s = 0.0
        for r in csv.DictReader(open("%s/%s_%s/p_counter_collection.csv" % (out, dist, ctr))):
            v = float(r["Counter_Value"]) * 1024.0  # the counters are in KiB
            s += v
            per_kernel[r["Kernel_Name"].split("(")[0].replace("void ", "")][i] += v
        tot[ctr] = s
    print("%-11s %14.1f %14.1f" % (dist, tot["FETCH_SIZE"] / tuples, tot["WRITE_SIZE"] / tuples))
    for k, (f, w) in sorted(per_kernel.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:6]:
        print("    %-70s %8.1f %8.1f" % (k[:70], f / tuples, w / tuples))
PY
