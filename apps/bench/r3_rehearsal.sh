#!/bin/bash
# round 3: the driver's multi-rank launch of bench.py rehearsed on one GPU (every rank uses GPU 0; functional, not a scaling number)
set -o pipefail
O=${1:-gpurun_out/r3/rehearsal}
mkdir -p $O
export WF_QUIET=1 WF_BENCH_DEVICE=0
for n in 2 4; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) bench.py --gpus $n --steps 3 --warmup 1 > $O/bench_n$n.json 2> $O/bench_n$n.err || { tail -30 $O/bench_n$n.err; echo "FAILED n=$n"; exit 1; }
  grep -a '^{' $O/bench_n$n.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n=%d value %.1f M/s p99 %.1f ms metric=%s parallelism=%s transport=%s probe=%s devprobe=%s' % (d['n_gpus'], d['value']/1e6, d.get('p99_latency_us',0)/1e3, d['metric'], d['config'].get('parallelism'), d['config'].get('transport'), d.get('dp_no_exchange_probe'), d.get('device_source_probe')))
"
done
