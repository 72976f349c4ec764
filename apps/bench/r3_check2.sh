#!/bin/bash
# round 3: disorder handling (timestamp pre-sort + merge gather): engine tests, tuple-fed DEFAULT/TB benches, shared-key regimes
set -o pipefail
O=gpurun_out/r3/check2
mkdir -p $O
export WF_QUIET=1
if [ -z "$SKIP_PYTEST" ]; then
timeout -k 10 900 python -u -m pytest tests/test_gpu_win_batch.py tests/test_gpu_windows.py tests/test_gpu_operators.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; echo "pytest failed"; exit 1; }
tail -3 $O/pytest.log
fi
rm -f $O/win_tuple_fed.txt
for op in kf_gpu wf_gpu kff_gpu; do
  timeout -k 10 60 apps/bin/bench_win_gpu -o $op -s 4 -n 4 -k 64 -b 65536 -t 3 -D >> $O/win_tuple_fed.txt 2>&1 || { tail -5 $O/win_tuple_fed.txt; echo "bench_win_gpu $op -D failed"; exit 1; }
done
cat $O/win_tuple_fed.txt
for keys in 64 1024; do
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --shared-keys --keys $keys --device-probe-steps 0 > $O/shared_keys_$keys.json 2> $O/shared_keys_$keys.err || { tail -20 $O/shared_keys_$keys.err; echo "bench failed"; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/shared_keys_$keys.json"))
print("shared keys=$keys DEFAULT: %.1f M t/s p50 %.1f ms p99 %.1f ms dropped %s" % (d["value"]/1e6, d["p50_latency_us"]/1e3, d["p99_latency_us"]/1e3, d["config"]["dropped"]))
PY
done
