#!/bin/bash
# round 3, final check on one MI355X: smoke(), the whole GPU suite, the driver's bench command
set -o pipefail
O=${1:-gpurun_out/r3/final}
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; echo "smoke failed"; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; echo "pytest failed"; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; echo "bench failed"; exit 1; }
cat $O/bench.json
timeout -k 10 400 python bench.py > $O/bench_noflags.json 2> $O/bench_noflags.err || { tail -20 $O/bench_noflags.err; echo "bench (no flags) failed"; exit 1; }
cat $O/bench_noflags.json | python3 apps/bench/jline.py
