#!/usr/bin/env python3
"""Key skew across ranks, with and without the pre-exchange combiner (Pane_Combiner_GPU).

`world` ranks (threads of this process coupled by LocalTransport, all on GPU 0: the rehearsal of an N-GPU job on a one-GPU
box) run the flagship keyed sliding-window aggregation behind Exchange_GPU over zipf-distributed keys that every source of
every rank emits.  Without the combiner every tuple travels to the rank that owns its key: the link into the owner of the hot
keys carries most of the stream.  With it every lane folds its batches per (key, pane) first and ships partial results.
Prints one JSON line per configuration: sink results (must be identical), bytes on the busiest link, throughput.

    python apps/bench/skew_rehearsal.py [--world 4] [--alpha 1.3] [--keys 1048576] [--steps 3]
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("WF_QUIET", "1")
import windflow_amd  # noqa: E402


def run(ext, world, args, combiner):
    fabric = ext.LocalFabric(world)
    stats = [None] * world
    errors = []
    t_run = [0.0] * world

    def rank_main(r):
        try:
            pipe = ext.WinAggPipeline(0, args.sources, args.batch, args.keys, args.win, args.slide, True, args.delay, 4, ext.local_transport(fabric, r),
                                      args.lanes, 2, args.alpha, True, False, args.device_source, combiner)
            pipe.step(args.per_source)  # warm-up
            t0 = time.perf_counter()
            pipe.steps(args.steps, args.per_source)
            t_run[r] = time.perf_counter() - t0
            pipe.close()
            stats[r] = pipe.stats()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise SystemExit("rank failed: %s" % errors)
    total = world * args.sources * args.per_source * args.steps
    return {
        "combiner": combiner,
        "world": world,
        "zipf_alpha": args.alpha,
        "keys": args.keys,
        "tuples_per_s": total / max(t_run),
        "sink_tuples": sum(s["sink_tuples"] for s in stats),
        "sink_value_sum": sum(s["sink_value_sum"] for s in stats),
        "dropped": sum(s["dropped"] for s in stats),
        "busiest_link_bytes": max(s["exchange_busiest_link_bytes"] for s in stats),
        "link_bytes_total": sum(s["exchange_link_bytes_total"] for s in stats),
        "max_link_bytes_per_round": max(s["exchange_max_link_bytes_per_round"] for s in stats),
        "exchange_tuples_sent": sum(s["exchange_tuples_sent"] for s in stats),
        "rounds": max(s["exchange_rounds"] for s in stats),
        "p99_latency_us": max(s["latency_p99_us"] for s in stats),
    }


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--world", type=int, default=4)
    p.add_argument("--sources", type=int, default=2)
    p.add_argument("--lanes", type=int, default=2)
    p.add_argument("--batch", type=int, default=65536)
    p.add_argument("--keys", type=int, default=1 << 20)
    p.add_argument("--alpha", type=float, default=1.3)
    p.add_argument("--win", type=int, default=262144, help="window, microseconds of a source's clock (a source emits a tuple per microsecond)")
    p.add_argument("--slide", type=int, default=65536)
    p.add_argument("--delay", type=int, default=64 * 65536, help="triggering delay: the lag between sources and ranks that is tolerated")
    p.add_argument("--per-source", type=int, default=4 * 1024 * 1024)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--device-source", action="store_true")
    args = p.parse_args()
    ext = windflow_amd.load_ext()
    out = [run(ext, args.world, args, c) for c in (False, True)]
    for o in out:
        print(json.dumps(o), flush=True)
    a, b = out
    print(json.dumps({"identical_results": a["sink_tuples"] == b["sink_tuples"] and a["sink_value_sum"] == b["sink_value_sum"] and a["dropped"] == b["dropped"] == 0,
                      "busiest_link_bytes_ratio": a["busiest_link_bytes"] / max(1, b["busiest_link_bytes"]),
                      "total_link_bytes_ratio": a["link_bytes_total"] / max(1, b["link_bytes_total"])}), flush=True)


if __name__ == "__main__":
    main()
