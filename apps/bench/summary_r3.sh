#!/bin/bash
# End-of-round measurement set on one MI355X (round 3): every number quoted in README.md / docs/BENCHMARKS.md.
#   bash apps/bench/summary_r3.sh [out dir]
set -o pipefail
OUT=${1:-gpurun_out/r3/summary}
mkdir -p "$OUT" || exit 1
export WF_QUIET=1
line() {  # line <name> <bench.py args...>
    local name=$1; shift
    echo "== bench.py $*" >&2
    timeout -k 10 200 python3 bench.py "$@" 2>> "$OUT/bench.err" | grep -a '^{' > "$OUT/$name.json" || { echo "FAILED $name" >&2; exit 1; }
}
line flagship_dp --steps 20 --warmup 3
line flagship_device --steps 10 --warmup 2 --source device --device-probe-steps 0
line flagship_device_512k --steps 10 --warmup 2 --source device --batch 524288 --device-probe-steps 0
line flagship_device_cb --steps 10 --warmup 2 --source device --count-based --device-probe-steps 0
line flagship_device_uniform --steps 10 --warmup 2 --source device --key-dist uniform --device-probe-steps 0
line flagship_16k --steps 10 --warmup 2 --batch 16384 --tuples-per-step 134217728 --device-probe-steps 0
line flagship_4k --steps 10 --warmup 2 --batch 4096 --tuples-per-step 33554432 --device-probe-steps 0
line winseq_1key_tumbling4096 --steps 10 --warmup 2 --pipeline win_seq --keys 1 --win 4096 --slide 4096
line winseq_1key_win64_slide1 --steps 10 --warmup 2 --pipeline win_seq --keys 1 --win 64 --slide 1
line winseq_64ki --steps 10 --warmup 2 --pipeline win_seq --keys 65536
line shared64_default --steps 5 --warmup 1 --keys 64 --shared-keys --tuples-per-step 134217728 --device-probe-steps 0
line shared1ki_default --steps 5 --warmup 1 --keys 1024 --shared-keys --tuples-per-step 134217728 --device-probe-steps 0
line shared64_deterministic --steps 5 --warmup 1 --keys 64 --shared-keys --deterministic --tuples-per-step 134217728 --device-probe-steps 0
line shared1ki_deterministic --steps 5 --warmup 1 --keys 1024 --shared-keys --deterministic --tuples-per-step 134217728 --device-probe-steps 0
line flagship_keyby --steps 20 --warmup 3 --parallelism keyby
line flagship_64keys --steps 10 --warmup 2 --keys 64
line flagship_8keys --steps 10 --warmup 2 --keys 8
line flagship_1key_1src --steps 10 --warmup 2 --keys 1 --sources 1
line winseq_1key --steps 10 --warmup 2 --pipeline win_seq --keys 1
line winseq_1key_wr --steps 10 --warmup 2 --pipeline win_seq --keys 1 --wave-reduce
line winseq_1ki --steps 10 --warmup 2 --pipeline win_seq --keys 1024
line winseq_1ki_wr --steps 10 --warmup 2 --pipeline win_seq --keys 1024 --wave-reduce
line ysb --steps 10 --warmup 2 --pipeline ysb
line ysb_split_merge --steps 10 --warmup 2 --pipeline ysb --ysb-split-merge --agg-replicas 2
line ysb_keyby --steps 10 --warmup 2 --pipeline ysb --parallelism keyby --agg-replicas 2
line map_filter --steps 10 --warmup 2 --pipeline map_filter
ops() { echo "== bench_ops_gpu $*" >&2; timeout -k 10 60 apps/bin/bench_ops_gpu "$@" -t 3 2>&1 | tail -1 >> "$OUT/ops.txt" || { echo "FAILED ops $*" >&2; exit 1; }; }
: > "$OUT/ops.txt"
ops -o map_stateless -s 8 -n 8 -b 262144
ops -o filter_stateless -s 8 -n 8 -b 262144
for k in 8 64 512 1024 16384 131072; do ops -o map_stateful -s 8 -n 8 -k $k; done
ops -o map_stateful -s 4 -n 4 -k 512
ops -o filter_stateful -s 8 -n 8 -k 1024
for k in 1 8 64 512; do ops -o map_scan -s 8 -n 8 -k $k; done
for k in 8 64 512; do ops -o map_sum -s 8 -n 8 -k $k; done
ops -o keyby -s 4 -n 8 -k 16384 -b 262144
ops -o keyby -s 8 -n 4 -k 16384
win() { echo "== bench_win_gpu $*" >&2; timeout -k 10 60 apps/bin/bench_win_gpu "$@" -t 3 2>&1 | tail -1 >> "$OUT/win.txt" || { echo "FAILED win $*" >&2; exit 1; }; }
: > "$OUT/win.txt"
# tuple-fed (CPU Source -> GPU window operator: the reference's programming model; the producers fill batches), DETERMINISTIC and DEFAULT
for op in kf_gpu wf_gpu pf_gpu wmr_gpu pf_gg wmr_gg kff_gpu; do win -o $op -s 4 -n 4 -k 64 -b 65536; done
for op in kf_gpu wf_gpu pf_gpu wmr_gpu pf_gg wmr_gg kff_gpu; do win -o $op -s 4 -n 4 -k 64 -b 65536 -D; done
# batch-fed
for op in kf_gpu wf_gpu pf_gpu wmr_gpu kff_gpu; do win -o $op -s 4 -n 4 -k 256 -b 131072 -B; done
for op in kf_gpu wf_gpu kff_gpu; do win -o $op -s 4 -n 4 -k 4096 -b 4096 -B; done
ops -o map_stateful -s 14 -n 14 -k 1000 -b 10000
ops -o win_agg -s 8 -n 8 -k 131072 -b 131072 -d
echo "== ysb apps" >&2
timeout -k 10 60 apps/bin/ysb_gpu -l 5 -n 4 -m 2 -b 65536 > "$OUT/ysb_gpu.txt" 2>&1 || exit 1
timeout -k 10 60 apps/bin/ysb_gpu_dist -l 5 -n 4 -m 2 -b 65536 -t local -r 2 > "$OUT/ysb_gpu_dist_local2.txt" 2>&1 || exit 1
timeout -k 10 60 apps/bin/ysb_gpu_dist -l 5 -n 4 -m 2 -b 65536 -t local -r 2 -x 1 > "$OUT/ysb_gpu_dist_local2_split.txt" 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
print("%-22s %10s %9s %9s %8s" % ("bench.py", "M/s", "p50 ms", "p99 ms", "timed s"))
for f in sorted(glob.glob(out + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-22s %10.1f %9.2f %9.2f %8.2f" % (os.path.basename(f)[:-5], d["value"] / 1e6, d["p50_latency_us"] / 1e3, d["p99_latency_us"] / 1e3,
                                        d["ms_per_step"] * d["steps"] / 1e3))
for t in ("ops.txt", "win.txt", "ysb_gpu.txt", "ysb_gpu_dist_local2.txt", "ysb_gpu_dist_local2_split.txt"):
    print("--", t)
    for l in open(os.path.join(out, t)):
        if "throughput" in l or "tuples/s" in l or "latency" in l:
            print("  ", l.rstrip())
PY
