#!/bin/bash
# Round-3 leak hunt: every pipeline family for ~20 s, resident set size at 6 s and at the end (a leak shows as a difference of
# tens of MB; pinned pools and HBM-side state are allocated in the first seconds)
set -o pipefail
O=${1:-gpurun_out/r3/soak_matrix}
mkdir -p $O
export WF_QUIET=1
: > $O/rss.txt
run() {
  "$@" > $O/out.txt 2>&1 &
  local pid=$!
  local t=0 first=0 last=0 rss=0
  while kill -0 $pid 2>/dev/null; do
    sleep 1; t=$((t + 1))
    rss=$(awk '/VmRSS/ {print $2}' /proc/$pid/status 2>/dev/null)
    if [ -n "$rss" ]; then [ "$t" -eq 6 ] && first=$rss; last=$rss; fi
  done
  wait $pid
  local rc=$?
  printf 'rc=%d rss_6s=%8d rss_end=%8d delta_kb=%8d after %3ds : %s\n' $rc $first $last $((last - first)) $t "$*" | tee -a $O/rss.txt
}
P="python3 bench.py --warmup 2 --device-probe-steps 0 --dp-probe-steps 0"
run $P --steps 100 --pipeline win_seq --keys 1
run $P --steps 100 --pipeline win_seq --keys 65536
run $P --steps 100 --pipeline map_filter
run $P --steps 80 --pipeline ysb --ysb-split-merge --agg-replicas 2
run $P --steps 80 --pipeline ysb --parallelism keyby --agg-replicas 2
run $P --steps 40 --source device
run $P --steps 60 --keys 1024 --shared-keys --deterministic
run $P --steps 60 --keys 1024 --shared-keys
run $P --steps 60 --key-dist zipf
run $P --steps 60 --keys 8
run $P --steps 60 --count-based --win 1024 --slide 1 --keys 1024
run apps/bin/ysb_gpu -l 20 -n 4 -m 2 -b 65536
run apps/bin/ysb_gpu_dist -l 20 -n 4 -m 2 -b 65536 -t local -r 2
O2=apps/bin/bench_ops_gpu
run $O2 -o map_stateful -s 8 -n 8 -k 1024 -t 20
run $O2 -o map_stateful -s 2 -n 2 -k 1024 -t 20
run $O2 -o filter_stateful -s 8 -n 8 -k 1024 -t 20
run $O2 -o map_scan -s 8 -n 8 -k 64 -t 20
run $O2 -o keyby -s 4 -n 8 -k 16384 -b 262144 -t 20
run $O2 -o map_stateless -s 1 -n 1 -k 500 -b 10000 -t 20
run $O2 -o filter_stateless -s 8 -n 8 -b 262144 -t 20
W=apps/bin/bench_win_gpu
for op in kf_gpu wf_gpu pf_gpu wmr_gpu kff_gpu; do run $W -o $op -s 4 -n 4 -k 256 -b 131072 -B -t 15; done
run $W -o kf_gpu -s 4 -n 4 -k 64 -b 65536 -t 15
run $W -o kff_gpu -s 4 -n 4 -k 64 -b 65536 -t 15
run $W -o wmr_gpu -s 4 -n 4 -k 64 -b 65536 -t 15
run $W -o pf_gg -s 4 -n 4 -k 64 -b 65536 -t 15 -D
run $W -o wf_gpu -s 4 -n 4 -k 64 -b 65536 -t 15 -D
run $W -o kf_gpu -s 2 -n 2 -k 64 -b 16384 -t 15 -B
