#!/bin/bash
# Round-3 soak: the device pipelines run for a minute each with the resident set size of the process sampled once a second
# (host memory must stay flat: everything the GPU operators keep lives in HBM or in bounded pools of pinned slabs).
# usage: bash apps/bench/soak_r2.sh [out file]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/r3/soak_r3.txt}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
export WF_QUIET=1
run() {
  local limit=$1
  shift
  echo "== $*" | tee -a "$OUT"
  "$@" > /tmp/soak_out.txt 2>/dev/null &
  local pid=$!
  local peak=0 first=0 last=0 t=0 rss=0
  while kill -0 $pid 2>/dev/null; do
    rss=$(awk '/VmRSS/ {print $2}' /proc/$pid/status 2>/dev/null)
    if [ -n "$rss" ]; then
      [ "$rss" -gt "$peak" ] && peak=$rss
      [ "$t" -eq 10 ] && first=$rss
      last=$rss
    fi
    sleep 1
    t=$((t + 1))
    [ $((t % 10)) -eq 0 ] && echo "  ... t=${t}s rss_kb=$rss" | tee -a "$OUT"
    if [ "$t" -ge "$limit" ]; then
      kill $pid
      break
    fi
  done
  wait $pid
  echo "rc=$? rss_at_10s_kb=$first rss_at_end_kb=$last peak_rss_kb=$peak : $(grep -a -e '^{' -e 'throughput' /tmp/soak_out.txt | tail -n 1 | cut -c1-160)" | tee -a "$OUT"
}
run 150 python3 bench.py --steps 300 --warmup 5 --device-probe-steps 0
run 100 python3 bench.py --steps 150 --warmup 5 --batch 16384 --device-probe-steps 0
run 100 python3 bench.py --steps 120 --warmup 5 --keys 64 --shared-keys --device-probe-steps 0
run 100 python3 bench.py --steps 150 --warmup 5 --parallelism keyby --device-probe-steps 0
run 100 python3 bench.py --steps 150 --warmup 5 --pipeline ysb
run 100 python3 bench.py --steps 150 --warmup 5 --pipeline win_seq --keys 1024
run 100 python3 bench.py --steps 150 --warmup 5 --pipeline win_seq --keys 1
run 60 apps/bin/bench_win_gpu -o pf_gpu -s 4 -n 4 -k 64 -b 65536 -t 40
run 60 apps/bin/bench_win_gpu -o wmr_gpu -s 4 -n 4 -k 64 -b 65536 -t 40 -D
run 60 apps/bin/bench_ops_gpu -o map_stateful -s 14 -n 14 -k 1000 -b 10000 -t 40
