#!/bin/bash
# Kernel + memory-copy trace of bench.py without its own HIP-event profiler, reduced to the idle-time report (tools/gap_report.py).
#   gpurun --timeout 600 -- 'bash tools/gap_pass.sh [mode]'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
m=${1:-f16x3}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gap_$m
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gap_$m -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile --gemm-mode $m > $OUT/gap_bench_$m.json 2> /tmp/gap_$m.err || { tail -5 /tmp/gap_$m.err; exit 1; }
KT=$(ls /tmp/gap_$m/*/*kernel_trace.csv | head -1)
MC=$(ls /tmp/gap_$m/*/*memory_copy_trace.csv 2>/dev/null | head -1)
python3 $R/tools/gap_report.py $KT 10 $MC $OUT/r04_idle_gaps_$m.json > /dev/null && cat $OUT/gap_bench_$m.json | cut -c1-300
rm -rf /tmp/gap_$m
