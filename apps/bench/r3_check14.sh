#!/bin/bash
# round 3: Pane_Farm_GPU with the PLQ on the device and the WLQ on the host gets a batch path - regression cases and tuple-fed rows
set -o pipefail
O=${1:-gpurun_out/r3/check14}
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
T=tests/cpp/bin/test_mp_windows_gpu
for m in ; do
  case $m in cb) W="-w 14 -s 4";; tb) W="-w 1000 -s 300";; *) W="-w 900 -s 300";; esac
  for op in pf_plq pf_wlq kf+pf wf+pf pf_gg; do
    timeout -k 10 120 $T -t $op -m $m -r 4 -l 1200 -k 5 $W > $O/t_${op}_$m.log 2>&1 || { cat $O/t_${op}_$m.log; echo "FAILED $op $m"; exit 1; }
    tail -1 $O/t_${op}_$m.log
  done
done
B=apps/bin/bench_win_gpu
: > $O/rows.txt
for op in pf_gpu wmr_gpu pf_gg wmr_gg kf_gpu; do
  for D in "" "-D"; do
    echo "== $op $D" >> $O/rows.txt
    timeout -k 10 120 $B -o $op -s 4 -n 4 -k 64 -b 65536 -t 3 $D >> $O/rows.txt 2>&1 || { tail -5 $O/rows.txt; echo "FAILED bench $op $D"; exit 1; }
  done
done
cat $O/rows.txt
