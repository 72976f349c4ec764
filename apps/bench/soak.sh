#!/bin/bash
# long runs with the resident set size of each program sampled once a second (stability / leak check)
# usage: apps/bench/soak.sh [quick]
cd "$(dirname "$0")/../.."
export WF_QUIET=1
run() {
  local limit=$1
  shift
  "$@" > /tmp/soak_out.txt 2>/dev/null &
  local pid=$!
  local peak=0 first=0 t=0 rss=0
  while kill -0 $pid 2>/dev/null; do
    rss=$(awk '/VmRSS/ {print $2}' /proc/$pid/status 2>/dev/null)
    if [ -n "$rss" ]; then
      [ "$rss" -gt "$peak" ] && peak=$rss
      [ "$t" -eq 10 ] && first=$rss
    fi
    sleep 1
    t=$((t + 1))
    [ $((t % 10)) -eq 0 ] && echo "  ... t=${t}s rss_kb=$rss"
    if [ "$t" -ge "$limit" ]; then
      kill $pid
      break
    fi
  done
  wait $pid
  echo "rc=$? rss_at_10s_kb=$first peak_rss_kb=$peak : $(tail -n 1 /tmp/soak_out.txt | cut -c1-200)"
}
run 300 python bench.py --steps 1200 --warmup 5
run 120 apps/bin/bench_ops_gpu -o map_stateful -s 8 -k 1024 -b 262144 -t 30
run 120 apps/bin/bench_ops_gpu -o filter_stateless -s 8 -k 1024 -b 65536 -t 30
[ "$1" = quick ] && exit 0
run 120 apps/bin/bench_win_gpu -o kf_gpu -s 2 -n 4 -k 64 -t 20
run 120 apps/bin/bench_win_gpu -o kff -s 2 -n 4 -k 64 -t 20
run 120 apps/bin/ysb_gpu -l 20 -n 4 -m 2
