#!/bin/bash
# is the slow state of the 64-shared-keys row CPU throttling?  cgroup statistics around every run
set -o pipefail
O=${1:-gpurun_out/r3/check50}
mkdir -p $O
export TMPDIR=/tmp
{
echo "nproc $(nproc)  affinity $(python3 -c 'import os; print(len(os.sched_getaffinity(0)))')"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
cat /proc/self/cgroup | head -3
} > $O/cpu.txt 2>&1
cat $O/cpu.txt
stat() { grep -E "nr_periods|nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
for i in 1 2 3 4 5 6; do
  echo "before: $(stat)" >> $O/cpu.txt
  timeout -k 10 150 python bench.py --steps 5 --warmup 1 --keys 64 --shared-keys --tuples-per-step 134217728 --device-probe-steps 0 > $O/j.json 2> $O/j.err || { tail -5 $O/j.err; echo failed; exit 1; }
  echo "after:  $(stat)" >> $O/cpu.txt
  python3 apps/bench/jline.py < $O/j.json >> $O/cpu.txt
  tail -3 $O/cpu.txt
done
