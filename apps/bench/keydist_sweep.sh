#!/bin/bash
# Key-distribution study of the flagship (Key_FFAT_GPU sliding-window sum): round-robin / uniform random / zipf keys,
# 1 Ki and 1 Mi keys, sources with disjoint key sets and with a shared key space (device KeyBy in the path).
#   bash apps/bench/keydist_sweep.sh [out file]
set -o pipefail
OUT=${1:-gpurun_out/r2/key_dist.jsonl}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
export WF_QUIET=1
for keys in 1024 1048576; do
  for dist in roundrobin uniform zipf; do
    for shared in "" "--shared-keys"; do
      echo "== keys $keys dist $dist $shared" >&2
      timeout -k 10 150 python3 bench.py --steps 8 --warmup 2 --keys $keys --key-dist $dist $shared --count-based 2>> "$OUT.err" | grep -a '^{' >> "$OUT" || { echo "FAILED keys $keys dist $dist $shared" >&2; exit 1; }
    done
  done
done
python3 - "$OUT" <<'PY'
import json, sys
print("%-10s %-12s %-8s %10s %9s %9s %8s" % ("keys", "dist", "shared", "M t/s", "p50 ms", "p99 ms", "dropped"))
for l in open(sys.argv[1]):
    d = json.loads(l)
    c = d["config"]
    print("%-10d %-12s %-8s %10.1f %9.2f %9.2f %8d" % (c["keys_per_gpu"], c["key_distribution"], c["shared_keys"], d["value"] / 1e6,
                                                  d["p50_latency_us"] / 1e3, d["p99_latency_us"] / 1e3, c["dropped"]))
PY
