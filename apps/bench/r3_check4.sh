#!/bin/bash
# round 3: pre-exchange combiner (skew)
set -o pipefail
O=gpurun_out/r3/check4
mkdir -p $O
export WF_QUIET=1
timeout -k 10 600 python -u -m pytest tests/test_gpu_dist.py -m gpu -x -q --timeout 150 --timeout-method thread -k combiner > $O/pytest.log 2>&1 || { cat $O/pytest.log | tail -60; echo "pytest failed rc=$?"; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 400 python apps/bench/skew_rehearsal.py --world 4 > $O/skew_world4.jsonl 2> $O/skew_world4.err || { tail -20 $O/skew_world4.err; echo "rehearsal failed"; exit 1; }
cat $O/skew_world4.jsonl
