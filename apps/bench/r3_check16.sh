#!/bin/bash
# round 3: huge-run fix proportional to the run (long triggering delays), end-of-stream flush over the compacted key list
set -o pipefail
O=${1:-gpurun_out/r3/check16}
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
if [ -z "$SKIP_PYTEST" ]; then
timeout -k 10 900 python -u -m pytest tests/test_gpu_windows.py tests/test_gpu_operators.py tests/test_gpu_kernels.py tests/test_python_api.py -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; echo "pytest failed"; exit 1; }
tail -3 $O/pytest.log
fi
: > $O/rows.txt
row() {
    local label=$1; shift
    local line
    line=$(timeout -k 10 150 python3 bench.py --steps 5 --warmup 1 --tuples-per-step 134217728 --device-probe-steps 0 "$@" 2>> "$O/bench.err" | grep -a '^{') || { echo "FAILED $label" >&2; tail -5 $O/bench.err; exit 1; }
    printf '%-58s ' "$label" >> "$O/rows.txt"
    echo "$line" | python3 apps/bench/jline.py >> "$O/rows.txt"
}
row "shared keys=64 DEFAULT" --keys 64 --shared-keys
row "shared keys=1024 DEFAULT" --keys 1024 --shared-keys
row "shared keys=64 DETERMINISTIC" --keys 64 --shared-keys --deterministic
row "shared keys=1024 DETERMINISTIC" --keys 1024 --shared-keys --deterministic
row "flagship host-fed"
row "flagship device source 128Ki" --source device
cat $O/rows.txt
bash apps/bench/r3_check15.sh $O/traces
