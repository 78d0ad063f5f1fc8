#!/bin/bash
# Kernel-trace leg of tools/refresh_profiles.sh alone (no PMC passes): gpurun --timeout 600 -- 'bash tools/trace_only.sh [mode]'
set -o pipefail
RND=r02
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
m=${1:-bf16x3}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$m
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --gemm-mode $m > $OUT/${RND}_bench_under_rocprof_$m.json 2> /tmp/prof_$m.err || { tail -5 /tmp/prof_$m.err; exit 1; }
cp $(ls /tmp/prof_$m/*/*kernel_stats.csv | head -1) $OUT/${RND}_kernel_stats_$m.csv
python3 $R/tools/trace_summary.py $(ls /tmp/prof_$m/*/*kernel_trace.csv | head -1) $OUT/${RND}_kernel_trace_by_grid_$m.csv
python3 $R/tools/timed_epoch_summary.py $(ls /tmp/prof_$m/*/*kernel_trace.csv | head -1) 10 $OUT/${RND}_bench_under_rocprof_$m.json $OUT/${RND}_kernel_timed_epoch_$m.json
