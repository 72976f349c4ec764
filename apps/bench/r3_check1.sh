#!/bin/bash
# round 3, first GPU pass: MultiPipe GPU operators through the Batcher path, tuple-fed window benchmarks, flagship + device probe
set -o pipefail
O=gpurun_out/r3/check1
mkdir -p $O
export WF_QUIET=1
if [ -z "$SKIP_PYTEST" ]; then
timeout -k 10 600 python -u -m pytest tests/test_gpu_operators.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest_ops.log 2>&1 || { tail -30 $O/pytest_ops.log; echo "pytest failed"; exit 1; }
tail -3 $O/pytest_ops.log
fi
rm -f $O/win_tuple_fed.txt
for mode in "" "-D"; do
  for op in kf_gpu wf_gpu kff_gpu; do
    timeout -k 10 60 apps/bin/bench_win_gpu -o $op -s 4 -n 4 -k 64 -b 65536 -t 3 $mode >> $O/win_tuple_fed.txt 2>&1 || { tail -5 $O/win_tuple_fed.txt; echo "bench_win_gpu $op $mode failed"; exit 1; }
  done
done
cat $O/win_tuple_fed.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; echo "bench failed"; exit 1; }
cat $O/bench_default.json
