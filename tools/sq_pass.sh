#!/bin/bash
# SQ-counter evidence for the matrix pipe (VERDICT r02 Next #2), run through gpurun from the repo root:
#   gpurun --timeout 1100 -- 'bash tools/sq_pass.sh [mode]'
# Two rocprofv3 passes of the bench command (8 SQ slots per pass; GRBM_GUI_ACTIVE rides along in the GRBM block; --kernel-trace only,
# no other trace domain), reduced by tools/pmc_sq_summary.py into gpurun_out/profiles/<round>_pmc_sq_<mode>.json (copy to profiles/).
set -o pipefail
RND=${RND:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
m=${1:-f16x3}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $OUT/${RND}_sq_counters_available.txt || true
echo "SQ counters listed: $(wc -l < $OUT/${RND}_sq_counters_available.txt)"
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile --gemm-mode $m"
rm -rf /tmp/sq1_$m /tmp/sq2_$m
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d /tmp/sq1_$m --output-format csv -- $B > /dev/null 2> /tmp/sq1_$m.err || { tail -8 /tmp/sq1_$m.err; exit 1; }
echo "sq pass 1 done"
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/sq2_$m --output-format csv -- $B > /dev/null 2> /tmp/sq2_$m.err || { tail -8 /tmp/sq2_$m.err; exit 1; }
echo "sq pass 2 done"
python3 $R/tools/pmc_sq_summary.py 10 $OUT/${RND}_pmc_sq_$m.json /tmp/sq1_$m /tmp/sq2_$m > $OUT/${RND}_pmc_sq_$m.txt
rm -rf /tmp/sq1_$m /tmp/sq2_$m
tail -60 $OUT/${RND}_pmc_sq_$m.txt
