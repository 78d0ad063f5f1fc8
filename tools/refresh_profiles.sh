#!/bin/bash
# Regenerates profiles/<round>_* on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh [modes...]'      (default mode list: f16x3)
# Everything large stays in /tmp; only the reduced summaries are written under gpurun_out/profiles/ (copy them to profiles/).
# Order per mode: kernel trace + stats, traffic / clock counter passes, SQ passes (f16x3; SQ=0 skips), then the plain bench, whose
# line quotes the counter files just produced (MERGEREC_COUNTER_DIR).  Any failing step ends the script (set -e).
set -eo pipefail
RND=r04
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
MODES=${@:-f16x3}
cd /tmp && export TMPDIR=/tmp
for m in $MODES; do
  rm -rf /tmp/prof_$m
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --gemm-mode $m > $OUT/${RND}_bench_under_rocprof_$m.json 2> /tmp/prof_$m.err || { tail -5 /tmp/prof_$m.err; exit 1; }
  cp $(ls /tmp/prof_$m/*/*kernel_stats.csv | head -1) $OUT/${RND}_kernel_stats_$m.csv
  python3 $R/tools/trace_summary.py $(ls /tmp/prof_$m/*/*kernel_trace.csv | head -1) $OUT/${RND}_kernel_trace_by_grid_$m.csv
  python3 $R/tools/timed_epoch_summary.py $(ls /tmp/prof_$m/*/*kernel_trace.csv | head -1) 10 $OUT/${RND}_bench_under_rocprof_$m.json $OUT/${RND}_kernel_timed_epoch_$m.json > /dev/null
  echo "rocprof stats $m done"
  rm -rf /tmp/pmcf_$m /tmp/pmcw_$m /tmp/pmcl_$m
  B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile --gemm-mode $m"
  timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmcf_$m --output-format csv -- $B > /dev/null 2> /tmp/pmcf_$m.err || { tail -5 /tmp/pmcf_$m.err; exit 1; }
  timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmcw_$m --output-format csv -- $B > /dev/null 2> /tmp/pmcw_$m.err || { tail -5 /tmp/pmcw_$m.err; exit 1; }
  timeout -k 10 500 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/pmcl_$m --output-format csv -- $B > /dev/null 2> /tmp/pmcl_$m.err || { tail -5 /tmp/pmcl_$m.err; echo "(no L2 hit/miss pass)"; rm -rf /tmp/pmcl_$m; }
  L2=""; [ -d /tmp/pmcl_$m ] && L2=/tmp/pmcl_$m
  python3 $R/tools/pmc_summary.py /tmp/pmcf_$m /tmp/pmcw_$m 10 $OUT/${RND}_pmc_traffic_$m.json $L2 > /dev/null
  rm -rf /tmp/pmcc_$m
  timeout -k 10 500 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmcc_$m --output-format csv -- $B > /dev/null 2> /tmp/pmcc_$m.err || { tail -5 /tmp/pmcc_$m.err; exit 1; }
  python3 $R/tools/pmc_clock.py /tmp/pmcc_$m $OUT/${RND}_pmc_clock_$m.json > /dev/null
  echo "pmc $m done"
  rm -rf /tmp/prof_$m /tmp/pmcf_$m /tmp/pmcw_$m /tmp/pmcl_$m /tmp/pmcc_$m
  # SQ passes (matrix-pipe busy) for the headline mode, then the plain bench LAST: its line reads the counter files of THIS source
  # tree from profiles/ (roofline.traffic, roofline.mfma_busy; a hash of csrc/ flags them stale otherwise)
  if [ $m = f16x3 ] && [ "${SQ:-1}" = 1 ]; then bash $R/tools/sq_pass.sh $m > /tmp/sq_$m.log 2>&1 || { tail -8 /tmp/sq_$m.log; exit 1; }; echo "sq $m done"; fi
  # the bench reads the counter reductions of THIS call from $OUT (recorded in its line as traffic_source); the tracked profiles/ directory
  # is only ever written by a reviewed copy + commit afterwards
  extra="--no-cpu-baseline"; [ $m = f16x3 ] && extra=""
  MERGEREC_COUNTER_DIR=$OUT timeout -k 10 400 python3 $R/bench.py --steps 10 --warmup 2 --gemm-mode $m $extra > $OUT/${RND}_bench_$m.json 2> /tmp/bench_$m.err || { tail -5 /tmp/bench_$m.err; exit 1; }
  echo "bench $m done"
done
ls -la $OUT
