#!/usr/bin/env python3
"""Headline benchmark: input tuples/sec of the flagship streaming pipeline.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches one rank per GPU with torch.distributed.run; every
rank runs its own native pipeline (C++ runtime + HIP kernels) on its GPU and the
ranks are coupled by RCCL when the pipeline repartitions by key.  A "step" feeds
a fixed number of synthetic tuples per GPU (weak scaling) and ends when every
result of that step has reached the sinks.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# published numbers of the reference: tuples/s per GPU-equipped node
BASELINES = {
    "map_filter": 147.5e6,  # stateless Map_GPU peak, results.org:144
    "win_agg": 118.6e6,     # keyed/stateful Map_GPU peak, results.org:126 (no windowed number is published)
}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--pipeline", default=os.environ.get("WF_BENCH_PIPELINE", "win_agg"),
                   help="win_agg: keyed sliding-window aggregation (flagship); map_filter: stateless Map_GPU+Filter_GPU")
    p.add_argument("--sources", type=int, default=int(os.environ.get("WF_BENCH_SOURCES", "8")))
    p.add_argument("--batch", type=int, default=int(os.environ.get("WF_BENCH_BATCH", str(256 * 1024))))
    p.add_argument("--keys", type=int, default=int(os.environ.get("WF_BENCH_KEYS", str(1 << 20))),
                   help="distinct keys per GPU (all resident in HBM)")
    p.add_argument("--win", type=int, default=16, help="window length, in tuples of one key")
    p.add_argument("--slide", type=int, default=4, help="window slide, in tuples of one key")
    p.add_argument("--count-based", action="store_true", help="count-based instead of time-based windows")
    p.add_argument("--tuples-per-step", type=int, default=int(os.environ.get("WF_BENCH_TUPLES", str(32 * 1024 * 1024))),
                   help="tuples fed per GPU per step")
    p.add_argument("--inflight", type=int, default=6)
    return p.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)

    import windflow_amd

    ext = windflow_amd.load_ext()
    per_source = max(1, args.tuples_per_step // args.sources)
    tuples_per_step_gpu = per_source * args.sources
    keys_per_source = max(1, args.keys // args.sources)
    if args.pipeline == "map_filter":
        pipe = ext.MapFilterPipeline(local_rank, args.sources, args.batch, keys_per_source, args.inflight)
    elif args.pipeline == "win_agg":
        if args.count_based:
            win, slide = args.win, args.slide
        else:
            # one tuple per microsecond per source, keys round-robin: a key ticks every keys_per_source us
            win, slide = args.win * keys_per_source, args.slide * keys_per_source
        pipe = ext.WinAggPipeline(local_rank, args.sources, args.batch, keys_per_source, win, slide, not args.count_based, 0, args.inflight)
    else:
        raise SystemExit("unknown pipeline %r" % args.pipeline)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        pipe.step(per_source)
    pipe.reset_latency()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.step(per_source)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    stats = pipe.stats()
    pipe.close()

    t = torch.tensor([elapsed], dtype=torch.float64)
    lat = torch.tensor([stats["latency_p99_us"]], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(lat, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        total_tuples = tuples_per_step_gpu * world * args.steps
        value = total_tuples / elapsed
        base = BASELINES.get(args.pipeline)
        print(json.dumps({
            "metric": "input tuples/sec (whole node)",
            "value": value,
            "unit": "tuples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": (value / base) if base else None,
            "dtype": "fp64 values / u64 keys (40-byte tuples)",
            "data": "synthetic tuple stream generated on the host (random-free, deterministic)",
            "p99_latency_us": float(lat.item()),
            "p50_latency_us": stats["latency_p50_us"],
            "config": {
                "model": "Key_FFAT_GPU keyed sliding-window sum" if args.pipeline == "win_agg" else args.pipeline,
                "global_batch": tuples_per_step_gpu * world,
                "seq_len": args.batch,
                "parallelism": "dp%d" % world,
                "sources_per_gpu": args.sources,
                "keys_per_gpu": keys_per_source * args.sources,
                "window": "%s win=%d slide=%d (tuples of a key)" % ("CB" if args.count_based else "TB", args.win, args.slide),
                "sink_results_rank0": stats["sink_tuples"],
                "overflow": stats.get("overflow", 0),
                "dropped": stats.get("dropped", 0),
            },
        }), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
