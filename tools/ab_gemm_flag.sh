#!/bin/bash
# A/B of a gemm_bf16.hip build flag on the GPU box: builds the library twice into separate directories and runs the same microbench on both
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
FLAG=$1
for v in base flag; do
  mkdir -p /tmp/lib_$v
  defs=""; [ $v = flag ] && defs="$FLAG"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $defs -c mergerec_amd/csrc/gemm_bf16.hip -o /tmp/lib_$v/gemm_bf16.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$v/libmergerec_hip.so /tmp/lib_$v/gemm_bf16.o $(ls mergerec_amd/lib/obj/*.o | grep -v gemm_bf16.o)
done
for r in 1 2; do for v in base flag; do echo "== $v"; MERGEREC_HIP_LIB=/tmp/lib_$v/libmergerec_hip.so GB_MODE=bf16x3 GB_ROUNDS=8 python tools/gemm_bench.py 2>&1 | grep bf16x3; done; done
