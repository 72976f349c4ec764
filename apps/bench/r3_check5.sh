#!/bin/bash
# round 3: batch path of the two-stage GPU operators
set -o pipefail
O=gpurun_out/r3/check5
mkdir -p $O
export WF_QUIET=1
timeout -k 10 1100 python -u -m pytest tests/test_gpu_operators.py -m gpu -x -q --timeout 150 --timeout-method thread -k "window_operators" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; echo "pytest failed"; exit 1; }
tail -3 $O/pytest.log
rm -f $O/win_batch_fed.txt
for op in kf_gpu wf_gpu pf_gpu wmr_gpu kff_gpu; do
  timeout -k 10 60 apps/bin/bench_win_gpu -B -o $op -s 4 -n 4 -k 256 -b 131072 -t 3 >> $O/win_batch_fed.txt 2>&1 || { tail -5 $O/win_batch_fed.txt; echo "bench_win_gpu -B $op failed"; exit 1; }
done
cat $O/win_batch_fed.txt
