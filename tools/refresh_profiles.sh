#!/bin/bash
# Regenerates profiles/r01_* on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh'
# Everything large stays in /tmp; only the reduced summaries are written under gpurun_out/profiles/ (copy them to profiles/).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in bf16x3 bf16x6 f32; do
  extra="--no-cpu-baseline"; [ $m = bf16x3 ] && extra=""
  timeout -k 10 400 python3 $R/bench.py --steps 10 --warmup 2 --gemm-mode $m $extra > $OUT/r01_bench_$m.json 2> /tmp/bench_$m.err || { tail -5 /tmp/bench_$m.err; exit 1; }
  echo "bench $m done" 
  rm -rf /tmp/prof_$m
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --gemm-mode $m > $OUT/r01_bench_under_rocprof_$m.json 2> /tmp/prof_$m.err || { tail -5 /tmp/prof_$m.err; exit 1; }
  cp $(ls /tmp/prof_$m/*/*kernel_stats.csv | head -1) $OUT/r01_kernel_stats_$m.csv
  python3 $R/tools/trace_summary.py $(ls /tmp/prof_$m/*/*kernel_trace.csv | head -1) $OUT/r01_kernel_trace_by_grid_$m.csv
  echo "rocprof stats $m done"
  rm -rf /tmp/pmcf_$m /tmp/pmcw_$m
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmcf_$m --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile --gemm-mode $m > /dev/null 2> /tmp/pmcf_$m.err || { tail -5 /tmp/pmcf_$m.err; exit 1; }
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmcw_$m --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile --gemm-mode $m > /dev/null 2> /tmp/pmcw_$m.err || { tail -5 /tmp/pmcw_$m.err; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/pmcf_$m /tmp/pmcw_$m 4 $OUT/r01_pmc_traffic_$m.json
  rm -rf /tmp/pmcc_$m
  timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmcc_$m --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile --gemm-mode $m > /dev/null 2> /tmp/pmcc_$m.err || { tail -5 /tmp/pmcc_$m.err; exit 1; }
  python3 $R/tools/pmc_clock.py /tmp/pmcc_$m $OUT/r01_pmc_clock_$m.json > /dev/null
  rm -rf /tmp/pmcc_$m
  echo "pmc $m done"
  rm -rf /tmp/prof_$m /tmp/pmcf_$m /tmp/pmcw_$m
done
ls -la $OUT
