#!/bin/bash
# round 3: host memory of the two-stage GPU operators over time (a leak shows as a slope)
set -o pipefail
O=${1:-gpurun_out/r3/check33}
mkdir -p $O
export WF_QUIET=1
: > $O/rss.txt
run() {
  echo "== $*" >> $O/rss.txt
  "$@" > $O/out.txt 2>&1 &
  local pid=$!
  local t=0
  while kill -0 $pid 2>/dev/null; do
    sleep 1; t=$((t + 1))
    if [ $((t % 4)) -eq 0 ]; then echo "  t=${t}s rss_kb=$(awk '/VmRSS/ {print $2}' /proc/$pid/status 2>/dev/null)" >> $O/rss.txt; fi
  done
  wait $pid
  echo "  rc=$? $(grep -a 'tuples/s' $O/out.txt | tail -1 | cut -c1-150)" >> $O/rss.txt
}
B=apps/bin/bench_win_gpu
run env WF_COALESCE_TUPLES=0 $B -o wmr_gpu -s 4 -n 4 -k 64 -b 65536 -t 12 -D
run $B -o kf_gpu -s 4 -n 1 -k 64 -b 65536 -t 12 -D
run env WF_COALESCE_TUPLES=0 $B -o kf_gpu -s 4 -n 1 -k 64 -b 65536 -t 12 -D
run $B -o kff_gpu -s 4 -n 1 -k 64 -b 65536 -t 12 -D
run $B -o wmr_gpu -s 4 -n 4 -k 64 -b 65536 -t 12 -D -d 100000
cat $O/rss.txt
