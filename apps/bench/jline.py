#!/usr/bin/env python3
"""Print the essentials of a bench.py JSON line read from stdin (sweep scripts)."""
import json
import sys

for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        c = d["config"]
        print("%8.1f M/s  p50 %7.2f ms  p99 %7.2f ms  dropped %d  results %d  %s" % (
            d["value"] / 1e6, d["p50_latency_us"] / 1e3, d["p99_latency_us"] / 1e3, c.get("dropped", 0), c["sink_results_rank0"], c["window"]))
        break
else:
    print("no result line")
