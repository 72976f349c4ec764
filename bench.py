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
os.environ.setdefault("WF_QUIET", "1")  # stdout carries the JSON line only (the runtime's banner goes there otherwise)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# published numbers of the reference: tuples/s per GPU-equipped node
BASELINES = {
    "map_filter": 147.5e6,  # stateless Map_GPU peak, results.org:144
    "win_agg": 118.6e6,     # keyed/stateful Map_GPU peak, results.org:126 (no windowed number is published)
    "win_seq": 118.6e6,     # same: the reference publishes nothing for Win_Seq_GPU / Win_Farm_GPU
    "ysb": None,            # Yahoo Streaming Benchmark: harness only, no published number
}
KEY_DISTS = {"roundrobin": 0, "uniform": 1, "zipf": 2}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=25)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--drain-every-step", action="store_true",
                   help="wait for every step's results before feeding the next one (default: the K timed steps are streamed "
                        "back to back and the timed region ends when the last one has drained)")
    p.add_argument("--pipeline", default=os.environ.get("WF_BENCH_PIPELINE", "win_agg"),
                   help="win_agg: keyed sliding-window aggregation, Key_FFAT_GPU (flagship); win_seq: non-incremental windows, "
                        "Key_Farm_GPU with an opaque window function (count-based win 1024 / slide 128 unless --win/--slide; --keys 1 = the "
                        "single-key configuration); ysb: Yahoo Streaming Benchmark on the device (filter, join, 10 s tumbling count per "
                        "campaign; with --parallelism keyby the campaigns are partitioned across the GPUs by Exchange_GPU); map_filter: "
                        "stateless Map_GPU+Filter_GPU")
    p.add_argument("--ysb-period-us", type=int, default=1, help="ysb: event-time distance between two events of a source")
    p.add_argument("--ysb-delay-us", type=int, default=10000000,
                   help="ysb: triggering delay of the windows = tolerated lag between the event sources, of this GPU and of the others "
                        "(default: one window; the sources of a process advance at most 4 batches apart, the processes are only coupled by "
                        "the exchange).  Late events are counted in config.dropped")
    p.add_argument("--ysb-split-merge", action="store_true",
                   help="ysb: split the events on the device by ad type into two filter+join branches and merge them again in front of the "
                        "aggregation (the split/merge PipeGraph of BASELINE config 5)")
    p.add_argument("--sources", type=int, default=int(os.environ.get("WF_BENCH_SOURCES", "8")))
    p.add_argument("--batch", type=int, default=int(os.environ.get("WF_BENCH_BATCH", "0")),
                   help="tuples per batch; default 128 Ki with 4 batches in flight per lane (dp: the throughput plateau at half "
                        "the latency of 256 Ki x 6), 256 Ki x 6 for keyby (one fat aggregation lane wants large kernels)")
    p.add_argument("--keys", type=int, default=int(os.environ.get("WF_BENCH_KEYS", str(1 << 20))),
                   help="distinct keys per GPU (all resident in HBM)")
    p.add_argument("--win", type=int, default=0, help="window length, in tuples of one key (default 16; win_seq: 1024)")
    p.add_argument("--slide", type=int, default=0, help="window slide, in tuples of one key (default 4; win_seq: 128)")
    p.add_argument("--count-based", action="store_true", help="count-based instead of time-based windows")
    p.add_argument("--delay", type=int, default=-1,
                   help="win_agg, time-based: triggering delay, in tuples of one key (a window fires when a tuple that much later than its "
                        "end has arrived: the lag between the sources of a key that is tolerated without dropping tuples).  Default: 0 when "
                        "the sources own disjoint key sets (every key's tuples arrive in order), the time span of 16 batches with "
                        "--shared-keys (the sources of a key run up to a few batches apart; late tuples are counted in config.dropped)")
    p.add_argument("--tuples-per-step", type=int, default=int(os.environ.get("WF_BENCH_TUPLES", str(256 * 1024 * 1024))),
                   help="tuples fed per GPU per step (default 256 Mi: a step lasts ~0.2 s, so the driver's 20 steps time 4-5 s and ~40 000 batches)")
    p.add_argument("--inflight", type=int, default=0)
    p.add_argument("--agg-replicas", type=int, default=int(os.environ.get("WF_BENCH_AGG_REPLICAS", "0")),
                   help="keyby mode: Key_FFAT_GPU replicas behind the exchange (0: default)")
    p.add_argument("--key-dist", default=os.environ.get("WF_BENCH_KEY_DIST", "roundrobin"), choices=sorted(KEY_DISTS),
                   help="how the sources draw keys: roundrobin (ascending, one tuple per key per round: the friendliest stream), uniform "
                        "random, or zipf (reference: src/GPU_Tests/zipf.hpp)")
    p.add_argument("--zipf-alpha", type=float, default=1.3)
    p.add_argument("--shared-keys", action="store_true",
                   help="every source emits every key (default: sources own disjoint key sets), which puts the device KeyBy in front of "
                        "the keyed operator")
    p.add_argument("--deterministic", action="store_true",
                   help="win_agg: DETERMINISTIC PipeGraph - the device KeyBy merges the lanes of its producers in timestamp order (low-watermark "
                        "rule), so a key that several sources emit reaches its window in order and no triggering delay is needed")
    p.add_argument("--replicas", type=int, default=0, help="win_seq: Key_Farm_GPU replicas (0: one per source)")
    p.add_argument("--wave-reduce", action="store_true", help="win_seq: give the operator wave_reduce(lift, combine) instead of the opaque functor")
    p.add_argument("--no-cpu-partition", action="store_true", help="N>1: do not bind each rank to its own slice of the CPUs")
    p.add_argument("--parallelism", default=os.environ.get("WF_BENCH_PARALLELISM", "auto"), choices=["auto", "dp", "keyby"],
                   help="keyby: tuples are re-partitioned by key across the GPUs with Exchange_GPU (RCCL all-to-all over xGMI) - what "
                        "a keyed operator needs when the sources cannot pre-partition; dp: keys are partitioned at ingestion, every GPU "
                        "aggregates its own sources and no byte crosses xGMI; auto: keyby when N > 1 (win_agg, ysb), dp on one GPU")
    p.add_argument("--transport", default=os.environ.get("WF_BENCH_TRANSPORT", "auto"), choices=["auto", "rccl", "ipc"],
                   help="keyby data plane: rccl = grouped ncclSend/ncclRecv (RCCL over xGMI); ipc = peers pull each other's device buffers "
                        "through hipIpc memory handles (no collective library; also works for ranks that share a GPU); auto (N > 1) = rccl "
                        "if a short self-test of it in child processes passes on every rank, else ipc, else the run falls back to dp and says so")
    p.add_argument("--preflight", default="", help=argparse.SUPPRESS)  # internal: run the transport self-test and exit
    p.add_argument("--dp-probe-steps", type=int, default=int(os.environ.get("WF_BENCH_DP_PROBE", "3")),
                   help="N>1, keyby mode: after the timed run, also time this many steps of the dp pipeline (no exchange) in child "
                        "processes and report them as an extra field (0 disables)")
    return p.parse_args()


def parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(index):
    """NUMA node of GPU `index` from sysfs (None if unknown)."""
    try:
        import windflow_amd

        bdf = windflow_amd.load_ext().device_pci_bus_id(index).lower()
        if not bdf:
            return None
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except Exception:  # noqa: BLE001
        return None


def rank_cpus(local_rank, world):
    """CPUs for this rank: an equal share of the CPUs of its GPU's NUMA node (shared with the other ranks whose
    GPU sits on the same node), or an equal contiguous slice of all allowed CPUs when the topology is unknown."""
    allowed = sorted(os.sched_getaffinity(0))
    nodes = [gpu_numa_node(i) for i in range(world)]
    if not all(n is not None for n in nodes):
        # sysfs does not tell (virtualised PCI topology): on the usual 8-GPU boards the GPUs are split evenly over the
        # sockets in index order, so guess that when the host has several NUMA nodes with CPUs
        try:
            host_nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
            host_nodes = [n for n in host_nodes if open("/sys/devices/system/node/node%d/cpulist" % n).read().strip()]
            if len(host_nodes) > 1 and world >= len(host_nodes):
                nodes = [host_nodes[i * len(host_nodes) // world] for i in range(world)]
        except OSError:
            pass
    if all(n is not None for n in nodes):
        try:
            with open("/sys/devices/system/node/node%d/cpulist" % nodes[local_rank]) as f:
                node_cpus = [c for c in parse_cpulist(f.read()) if c in set(allowed)]
            peers = [i for i in range(world) if nodes[i] == nodes[local_rank]]
            k = peers.index(local_rank)
            # the list usually holds two runs (cores, then their SMT siblings): take the k-th share of each run, so a
            # rank gets whole cores instead of sharing them with the rank that holds the siblings
            runs, mine = [], []
            for c in node_cpus:
                if runs and c == runs[-1][-1] + 1:
                    runs[-1].append(c)
                else:
                    runs.append([c])
            for run in runs:
                per = len(run) // len(peers)
                mine.extend(run[k * per:(k + 1) * per])
            if len(mine) >= 8:
                return mine
        except OSError:
            pass
    per = len(allowed) // world
    return allowed[local_rank * per:(local_rank + 1) * per] if per >= 8 else allowed


def timed_steps(pipe, per_source, steps, barrier, drain_every_step=False):
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if drain_every_step:
        for _ in range(steps):
            pipe.step(per_source)
    else:
        # a stream has no step boundaries: the K steps are fed back to back and the call returns when the results of
        # the last one have reached the sinks (one pipeline drain inside the timed region instead of K)
        pipe.steps(steps, per_source)
    torch.cuda.synchronize()
    barrier()
    return time.perf_counter() - t0


def run_dp_probe(args, world, limit_s=90):
    """Time a few steps of the dp pipeline (keys partitioned at ingestion, no exchange) in a child process; returns a small
    dict for the JSON line.  It is the upper bound the keyby number is compared with."""
    import subprocess

    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 17)
    env.pop("WF_BENCH_FORCE_PROBE", None)
    # the children rendezvous among themselves: rank 0's child hosts the store on the new port (under torchrun the
    # workers would otherwise look for the launcher's store there)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--steps", str(args.dp_probe_steps), "--warmup", "1",
           "--parallelism", "dp", "--dp-probe-steps", "0", "--pipeline", args.pipeline, "--sources", str(args.sources),
           "--keys", str(args.keys), "--win", str(args.win), "--slide", str(args.slide), "--delay", str(args.delay),
           "--tuples-per-step", str(args.tuples_per_step), "--no-cpu-partition"]
    if args.count_based:
        cmd.append("--count-based")
    print("[bench] timed run done; timing the dp pipeline (no exchange) in a child process (limit %d s)" % limit_s, file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t_end = time.time() + limit_s
    while True:
        try:
            out, err = child.communicate(timeout=30)
            break
        except subprocess.TimeoutExpired:
            if time.time() >= t_end:
                child.kill()
                out, err = child.communicate()
                return {"status": "timeout", "stderr_tail": (err or "")[-300:]}
            print("[bench] dp probe still running", file=sys.stderr, flush=True)
    if child.returncode != 0:
        return {"status": "failed (exit %d)" % child.returncode, "stderr_tail": (err or "")[-300:]}
    for line in reversed(out.splitlines()):
        if line.startswith("{"):
            d = json.loads(line)
            return {"status": "ok", "tuples_per_s": d["value"], "steps": d["steps"], "p99_latency_us": d["p99_latency_us"]}
    return {"status": "ok (this rank printed no result)"}


def make_pipeline(args, ext, local_rank):
    """The native pipeline the arguments describe (its constructor starts the graph's threads)."""
    keys_per_source = max(1, args.keys // args.sources)
    if args.pipeline == "map_filter":
        return ext.MapFilterPipeline(local_rank, args.sources, args.batch, keys_per_source, args.inflight)
    elif args.pipeline == "win_agg":
        if args.count_based:
            win, slide, delay = args.win, args.slide, 0
        else:
            # one tuple per microsecond per source, keys round-robin: a key ticks every keys_per_source us
            win, slide, delay = args.win * keys_per_source, args.slide * keys_per_source, args.delay * keys_per_source
        transport = None
        if args.parallelism == "keyby":
            from windflow_amd import parallel as wfpar

            transport = (wfpar.ipc_transport if args.transport == "ipc" else wfpar.rccl_transport)(wfpar.DistEnv(), local_rank)
        # n_keys: per source when the sources own disjoint key sets, the whole key space (of this GPU) when they share it
        return ext.WinAggPipeline(local_rank, args.sources, args.batch, keys_per_source * args.sources if args.shared_keys else keys_per_source,
                                  win, slide, not args.count_based, delay, args.inflight, transport, args.agg_replicas, KEY_DISTS[args.key_dist],
                                  args.zipf_alpha, args.shared_keys, args.deterministic)
    elif args.pipeline == "win_seq":
        # BASELINE.json config 3: count-based sliding windows, non-incremental user function.  A single key cannot be split
        # over sources that own disjoint key sets: every source then emits it and one replica aggregates
        shared = args.shared_keys or args.keys < args.sources
        replicas = args.replicas or (min(args.sources, max(1, args.keys)) if shared else 0)
        # n_keys: per source when the sources own disjoint key sets, the whole key space when they share it
        return ext.WinSeqPipeline(local_rank, args.sources, args.batch, max(1, args.keys) if shared else max(1, args.keys // args.sources),
                                  args.win, args.slide, False, args.inflight, replicas, args.wave_reduce, KEY_DISTS[args.key_dist],
                                  args.zipf_alpha, shared)
    elif args.pipeline == "ysb":
        transport = None
        if args.parallelism == "keyby":
            from windflow_amd import parallel as wfpar

            transport = (wfpar.ipc_transport if args.transport == "ipc" else wfpar.rccl_transport)(wfpar.DistEnv(), local_rank)
        return ext.YsbPipeline(local_rank, args.sources, args.batch, args.ysb_period_us, args.ysb_delay_us, args.inflight,
                               args.agg_replicas or 1, transport, args.ysb_split_merge)
    else:
        raise SystemExit("unknown pipeline %r" % args.pipeline)


def run_preflight(kind, world, local_rank, args):
    """Child-process body of the transport self-test: a tiny keyby pipeline over `kind`, then two steps of the pipeline that
    is going to be timed, at its real size (slab sizes, round sizes and the number of lanes are what a fabric or a memory
    limit would object to); exit status 0 = both moved their tuples between the ranks and lost none."""
    from windflow_amd import load_ext, parallel as wfpar

    env = wfpar.init_distributed()
    ext = load_ext()

    def agreed(ok):
        t = torch.tensor([1 if ok else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    transport = (wfpar.ipc_transport if kind == "ipc" else wfpar.rccl_transport)(env, local_rank)
    pipe = ext.WinAggPipeline(local_rank, 2, 4096, 64, 8, 2, False, 0, 4, transport, 2)
    pipe.step(40_000)
    pipe.step(40_000)
    pipe.close()
    st = pipe.stats()
    ok = agreed(st["overflow"] == 0 and st["table_full"] == 0 and st["exchange_tuples_sent"] == 160_000 and st["sink_tuples"] > 0)
    del pipe, transport
    if ok and args.pipeline in ("win_agg", "ysb") and not os.environ.get("WF_BENCH_SHORT_PREFLIGHT"):
        args.parallelism, args.transport = "keyby", kind
        per_source = max(1, args.tuples_per_step // args.sources)
        pipe = make_pipeline(args, ext, local_rank)
        pipe.step(per_source)
        pipe.step(per_source)
        pipe.close()
        st = pipe.stats()
        ok = agreed(st.get("overflow", 0) == 0 and st.get("table_full", 0) == 0 and st["sink_tuples"] > 0
                    and st.get("exchange_tuples_sent", 1) > 0)
    wfpar.shutdown()
    return 0 if ok else 1


def preflight_ok(kind, world, port_offset, limit_s=150):
    """Run the self-test of transport `kind` in a child of every rank; True only if every rank's child succeeded."""
    import subprocess

    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + port_offset)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--preflight", kind, "--gpus", str(world)]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit_s)
        ok, why = r.returncode == 0, (r.stderr or "")[-200:]
    except subprocess.TimeoutExpired:
        ok, why = False, "timeout"
    if not ok:
        print("[bench] %s self-test failed on rank %s: %s" % (kind, os.environ.get("RANK", "0"), why), file=sys.stderr, flush=True)
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def main():
    args = parse_args()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.parallelism == "auto":
        # on several GPUs the headline is the pipeline that really couples them: the keyed operator sits behind the
        # cross-GPU KeyBy.  Pipelines without an exchange stage (win_seq, map_filter) are data-parallel by construction.
        args.parallelism = "keyby" if (world_env > 1 and args.pipeline in ("win_agg", "ysb")) else "dp"
    if args.win <= 0:
        args.win = 1024 if args.pipeline == "win_seq" else 16
    if args.slide <= 0:
        args.slide = 128 if args.pipeline == "win_seq" else 4
    if args.batch <= 0:
        args.batch = 128 * 1024
    if args.delay < 0:
        # a source emits one tuple per microsecond; window, slide and delay are scaled by keys / sources microseconds (the
        # pace of a key over all its sources)
        args.delay = -(-16 * args.batch // max(1, args.keys // args.sources)) \
            if (args.shared_keys and not args.count_based and not args.deterministic) else 0
    if args.inflight <= 0:
        args.inflight = 4
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cpu_slot = local_rank
    if os.environ.get("WF_BENCH_DEVICE"):  # rehearsal of an N-rank launch on a box with fewer GPUs: every rank uses this GPU
        local_rank = int(os.environ["WF_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo prints a connection banner on stdout; stdout is reserved for the JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    if args.preflight:
        raise SystemExit(run_preflight(args.preflight, world, local_rank, args))
    fallback_note = None
    if args.parallelism == "keyby" and args.transport == "auto":
        if world == 1:
            args.transport = "rccl"
        else:
            # the data plane between distinct GPUs cannot be rehearsed on a one-GPU development box: try it in child processes
            # first, so that a fabric problem (an RCCL abort, a hang inside a collective) costs a fallback, not the run
            chosen = None
            for i, kind in enumerate(("rccl", "ipc")):
                if preflight_ok(kind, world, 23 + 7 * i):
                    chosen = kind
                    break
            if chosen is None:
                fallback_note = "keyby self-tests failed for rccl and ipc: ran dp (no exchange)"
                args.parallelism = "dp"
            else:
                args.transport = chosen
    if world > 1 and not args.no_cpu_partition:
        # Give every rank its own contiguous slice of the CPUs this job may use (the runtime's source, operator
        # and sink threads inherit it): on a two-socket node that keeps the pinned slabs and the threads that
        # fill them on the socket the rank's GPU hangs off, and ranks do not steal each other's cores.
        try:
            os.sched_setaffinity(0, rank_cpus(cpu_slot, world))
        except (AttributeError, OSError):
            pass

    import windflow_amd

    ext = windflow_amd.load_ext()
    per_source = max(1, args.tuples_per_step // args.sources)
    tuples_per_step_gpu = per_source * args.sources
    keys_per_source = max(1, args.keys // args.sources)
    pipe = make_pipeline(args, ext, local_rank)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        pipe.step(per_source)
    pipe.reset_latency()
    elapsed = timed_steps(pipe, per_source, args.steps, barrier, args.drain_every_step)
    pipe.close()  # (the device counters - late tuples, overflows - are collected when the operators terminate)
    stats = pipe.stats()
    t = torch.tensor([elapsed], dtype=torch.float64)
    lat = torch.tensor([stats["latency_p99_us"]], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(lat, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # Secondary measurement (never part of `value`): the same job without the exchange (keys partitioned at ingestion),
    # in CHILD processes (one per rank, own rendezvous port) under a wall-clock limit.
    probe = None
    if (world > 1 or os.environ.get("WF_BENCH_FORCE_PROBE")) and args.pipeline in ("win_agg", "ysb") and args.parallelism == "keyby" \
            and args.dp_probe_steps > 0:
        probe = run_dp_probe(args, world)
    if rank == 0:
        total_tuples = tuples_per_step_gpu * world * args.steps
        value = total_tuples / elapsed
        base = BASELINES.get(args.pipeline)
        print(json.dumps({
            "metric": "input events/sec (whole node)" if args.pipeline == "ysb" else "input tuples/sec (whole node)",
            "value": value,
            "unit": "tuples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": (value / base) if base else None,
            "dtype": "u64 / u32 fields (56-byte events)" if args.pipeline == "ysb" else "fp64 values / u64 keys (40-byte tuples)",
            "data": "synthetic tuple stream generated on the host (random-free, deterministic)",
            "p99_latency_us": float(lat.item()),
            "p50_latency_us": stats["latency_p50_us"],
            "config": {
                "model": {"win_agg": "Key_FFAT_GPU keyed sliding-window sum",
                          "ysb": "Yahoo Streaming Benchmark: %sFilter_GPU -> Map_GPU(join) -> %sKey_FFAT_GPU 10 s tumbling count per campaign"
                                 % ("split_gpu(ad type) x2 -> " if args.ysb_split_merge else "", "merge -> " if args.ysb_split_merge else ""),
                          "win_seq": "Key_Farm_GPU non-incremental sliding-window sum (%s)" % ("wave_reduce" if args.wave_reduce else "opaque functor")}
                .get(args.pipeline, args.pipeline),
                "global_batch": tuples_per_step_gpu * world,
                "seq_len": args.batch,
                "parallelism": "%s%d" % (args.parallelism, world),
                **({"transport": args.transport} if args.parallelism == "keyby" else {}),
                **({"fallback": fallback_note} if fallback_note else {}),
                "sources_per_gpu": args.sources,
                "keys_per_gpu": args.keys if args.pipeline == "win_seq" else keys_per_source * args.sources,
                "key_distribution": args.key_dist + (" alpha=%g" % args.zipf_alpha if args.key_dist == "zipf" else ""),
                **({"graph_mode": "DETERMINISTIC"} if args.deterministic else {}),
                "shared_keys": bool(args.shared_keys or (args.pipeline == "win_seq" and args.keys < args.sources)),
                "window": "%s win=%d slide=%d%s (tuples of a key)" % ("CB" if (args.count_based or args.pipeline == "win_seq") else "TB", args.win,
                                                                       args.slide, " delay=%d" % args.delay if args.delay else ""),
                "latency_samples": stats.get("latency_samples"),
                "steps_streamed": "drained after every step" if args.drain_every_step else "back to back, drained once at the end of the timed region",
                "sink_results_rank0": stats["sink_tuples"],
                "overflow": stats.get("overflow", 0),
                "dropped": stats.get("dropped", 0),
                **({"counted_views_rank0": stats["counted_views"], "tuple_bytes": 56} if args.pipeline == "ysb" else {}),
            },
            **({"exchange": {k[len("exchange_"):]: stats[k] for k in sorted(stats) if k.startswith("exchange_")}}
               if "exchange_rounds" in stats else {}),
            **({"dp_no_exchange_probe": probe} if probe is not None else {}),
        }), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
