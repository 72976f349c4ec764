#!/bin/bash
# round 3: Win_MapReduce_GPU with the MAP stage on the device and the REDUCE stage on the host on the batch path
set -o pipefail
O=${1:-gpurun_out/r3/check18}
mkdir -p $O
export WF_QUIET=1 TMPDIR=/tmp
T=tests/cpp/bin/test_mp_windows_gpu
for m in cb tb tbd; do
  case $m in cb) W="-w 14 -s 4";; tb) W="-w 1000 -s 300";; *) W="-w 900 -s 300";; esac
  for op in wmr_map wmr_red kf+wmr wf+wmr wmr_gg pf_plq; do
    timeout -k 10 120 $T -t $op -m $m -r 4 -l 1200 -k 5 $W > $O/t_${op}_$m.log 2>&1 || { cat $O/t_${op}_$m.log; echo "FAILED $op $m"; exit 1; }
    tail -1 $O/t_${op}_$m.log
  done
done
timeout -k 10 120 $T -t wmr_map -m cb -r 3 -l 200000 -k 100 -w 1000 -s 100 > $O/t_big_cb.log 2>&1 || { cat $O/t_big_cb.log; echo "FAILED big cb"; exit 1; }
tail -1 $O/t_big_cb.log
timeout -k 10 120 $T -t wmr_map -m tb -c -r 3 -l 1200 -k 5 -w 1000 -s 300 > $O/t_chain.log 2>&1 || { cat $O/t_chain.log; echo "FAILED chaining"; exit 1; }
tail -1 $O/t_chain.log
B=apps/bin/bench_win_gpu
: > $O/rows.txt
for op in wmr_gpu pf_gpu; do
  for D in "" "-D"; do
    echo "== $op $D" >> $O/rows.txt
    timeout -k 10 120 $B -o $op -s 4 -n 4 -k 64 -b 65536 -t 3 $D >> $O/rows.txt 2>&1 || { tail -5 $O/rows.txt; echo "FAILED bench $op $D"; exit 1; }
  done
done
cat $O/rows.txt
