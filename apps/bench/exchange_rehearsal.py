#!/usr/bin/env python3
"""Rehearsal of the keyby (Exchange_GPU) flagship pipeline with R ranks on ONE GPU.

Every rank is a thread of this process with its own native pipeline; the ranks are coupled by LocalTransport (device-to-device
copies inside one process) instead of RCCL, so the control plane (rounds, board, regroup into lanes) is the one a real
multi-GPU run uses while the data plane stays on the card.  Prints one JSON line: aggregate tuples/s, exchange rounds/s and
the largest per-link payload of a round.

    python apps/bench/exchange_rehearsal.py --ranks 2 --steps 5
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--ranks", type=int, default=2)
    p.add_argument("--sources", type=int, default=4)
    p.add_argument("--lanes", type=int, default=0)
    p.add_argument("--batch", type=int, default=128 * 1024)
    p.add_argument("--inflight", type=int, default=4)
    p.add_argument("--keys", type=int, default=1 << 18, help="keys per rank")
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--tuples-per-step", type=int, default=64 * 1024 * 1024, help="per rank")
    p.add_argument("--pipeline", default="win_agg", choices=["win_agg", "ysb"])
    args = p.parse_args()

    import windflow_amd

    ext = windflow_amd.load_ext()
    fabric = ext.LocalFabric(args.ranks)
    per_source = args.tuples_per_step // args.sources
    keys_per_source = max(1, args.keys // args.sources)
    pipes = [None] * args.ranks
    errors = []
    start = threading.Barrier(args.ranks + 1)
    done = threading.Barrier(args.ranks + 1)

    def rank_main(r):
        try:
            tr = ext.local_transport(fabric, r)
            if args.pipeline == "ysb":
                pipe = ext.YsbPipeline(0, args.sources, args.batch, 1, 1000000, args.inflight, args.lanes or 1, tr)
            else:
                pipe = ext.WinAggPipeline(0, args.sources, args.batch, keys_per_source, 16 * keys_per_source, 4 * keys_per_source, True, 0,
                                          args.inflight, tr, args.lanes)
            pipes[r] = pipe
            for _ in range(args.warmup):
                pipe.step(per_source)
            pipe.reset_latency()
            start.wait()
            pipe.steps(args.steps, per_source)
            done.wait()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
            start.abort()
            done.abort()

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(args.ranks)]
    for t in threads:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    done.wait()
    elapsed = time.perf_counter() - t0
    stats = [pp.stats() for pp in pipes]
    for pp in pipes:
        threading.Thread(target=pp.close).start()
    for t in threads:
        t.join(timeout=60)
    total = per_source * args.sources * args.ranks * args.steps
    print(json.dumps({
        "pipeline": args.pipeline, "ranks": args.ranks, "sources_per_rank": args.sources, "batch": args.batch,
        "tuples_per_s": total / elapsed, "elapsed_s": elapsed,
        "p99_latency_us": max(s["latency_p99_us"] for s in stats),
        "exchange_rounds": [s.get("exchange_rounds") for s in stats],
        "exchange_rounds_per_s": [round(s.get("exchange_rounds_per_s", 0.0), 1) for s in stats],
        "exchange_empty_rounds": [s.get("exchange_empty_rounds") for s in stats],
        "max_link_bytes_per_round": max(s.get("exchange_max_link_bytes_per_round", 0) for s in stats),
        "lanes": stats[0].get("exchange_lanes"),
        "errors": errors,
    }), flush=True)


if __name__ == "__main__":
    main()
