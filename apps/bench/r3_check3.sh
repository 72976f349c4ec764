#!/bin/bash
# round 3: deterministic ordering across ranks
set -o pipefail
O=gpurun_out/r3/check3
mkdir -p $O
export WF_QUIET=1
timeout -k 10 900 python -u -m pytest tests/test_gpu_dist.py tests/test_gpu_operators.py -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; echo "pytest failed"; exit 1; }
tail -3 $O/pytest.log
