#!/bin/bash
# GPU window-operator matrix against the sequential oracle (tests/cpp/test_mp_windows_gpu); stops at the first crash / hang.
#   apps/bench/winmatrix.sh [ops...]
B=tests/cpp/bin/test_mp_windows_gpu
export WF_QUIET=1
OPS=${@:-seq kf wf kf_wr wf_wr pf_plq pf_wlq wmr_map wmr_red kf+pf wf+wmr}
fail=0
run() {
  echo "== $*"
  timeout -k 5 90 $B "$@" 2>&1 | tail -3
  rc=${PIPESTATUS[0]}
  echo "rc=$rc"
  [ $rc -ne 0 ] && fail=1
  if [ $rc -ge 124 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then echo "STOP"; exit 1; fi
}
for op in $OPS; do
  run -t $op -m cb -r 2 -l 1200 -k 5 -w 14 -s 4
  run -t $op -m tb -r 2 -l 1200 -k 5 -w 1000 -s 300
  run -t $op -m tbd -r 2 -l 1000 -k 4 -w 900 -s 300
done
# hopping windows, tumbling windows, many keys, long windows
for op in seq kf wf kf_wr; do
  run -t $op -m cb -r 2 -l 1500 -k 5 -w 3 -s 10
  run -t $op -m tb -r 2 -l 1500 -k 5 -w 300 -s 1000
  run -t $op -m cb -r 1 -l 600 -k 3 -w 7 -s 7
  run -t $op -m cb -r 1 -l 3000 -k 37 -w 500 -s 20
done
echo "matrix done fail=$fail"
exit $fail
