#!/bin/bash
# A/B of gemm_bf16.hip build flags on the GPU box: builds the library once per flag set (each argument is one set, quoted; "" = base) into
# separate directories and runs the same microbench on all of them, twice, alternating.
#   gpurun --timeout 900 -- 'bash tools/ab_gemm_flag.sh "" "-DSOME_FLAG=1"'
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python -m mergerec_amd.build > /dev/null 2>&1   # the other objects (mergerec_amd/lib/obj/ does not travel)
i=0
for defs in "$@"; do
  mkdir -p /tmp/lib_$i
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $defs -c mergerec_amd/csrc/gemm_bf16.hip -o /tmp/lib_$i/gemm_bf16.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$i/libmergerec_hip.so /tmp/lib_$i/gemm_bf16.o $(ls mergerec_amd/lib/obj/*.o | grep -v gemm_bf16.o)
  i=$((i + 1))
done
for r in 1 2; do
  i=0
  for defs in "$@"; do
    echo "== [$defs]"
    MERGEREC_HIP_LIB=/tmp/lib_$i/libmergerec_hip.so GB_MODE=${GB_MODE:-bf16x3} GB_ROUNDS=8 python tools/gemm_bench.py 2>&1 | grep x3
    i=$((i + 1))
  done
done
