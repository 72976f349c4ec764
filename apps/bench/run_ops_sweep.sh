#!/bin/bash
# Sweep of the operator benchmarks (the reference's run_tests.py loops over -s -k -b -n in the same spirit).
# usage: apps/bench/run_ops_sweep.sh [seconds per point]   -> CSV-like lines on stdout
T=${1:-3}
BIN=$(dirname "$0")/../bin/bench_ops_gpu
export WF_QUIET=1
for op in map_stateless filter_stateless; do
  for b in 65536 262144; do
    timeout -k 10 60 $BIN -o $op -s 8 -n 8 -k 16384 -b $b -t $T || exit 1
  done
done
for op in map_stateful filter_stateful; do
  for k in 64 1024 16384 131072; do
    timeout -k 10 60 $BIN -o $op -s 8 -n 8 -k $k -b 262144 -t $T || exit 1
  done
done
timeout -k 10 60 $BIN -o keyby -s 4 -n 8 -k 16384 -b 262144 -t $T || exit 1
