#!/bin/bash
# round 3: gather of long carries spread over waves - engine tests, tuple-fed rows, traces
set -o pipefail
O=${1:-gpurun_out/r3/check21}
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests/test_gpu_win_batch.py tests/test_gpu_operators.py -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; echo "pytest failed"; exit 1; }
tail -2 $O/pytest.log
B=apps/bin/bench_win_gpu
: > $O/rows.txt
for op in kf_gpu wf_gpu pf_gpu wmr_gpu pf_gg wmr_gg kff_gpu; do
  for D in "" "-D"; do
    timeout -k 10 120 $B -o $op -s 4 -n 4 -k 64 -b 65536 -t 3 $D >> $O/rows.txt 2>&1 || { tail -5 $O/rows.txt; echo "FAILED bench $op $D"; exit 1; }
  done
done
cat $O/rows.txt
bash apps/bench/r3_check20.sh $O/traces
