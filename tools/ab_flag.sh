#!/bin/bash
# Same-box A/B of a compile-time flag of ONE kernel source: builds the library twice into /tmp (base, flag) and runs the given command on
# both, twice, alternating.   usage (through gpurun, from the repo root):  bash tools/ab_flag.sh attn_bf16.hip "-DMR_ATTN_PRIO" python tools/attn_rate_curve.py 3
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
SRC=$1; FLAG=$2; shift 2
extra=""; [ $SRC = merge.hip ] && extra="-ffp-contract=off"
for v in base flag; do
  mkdir -p /tmp/lib_$v
  defs=""; [ $v = flag ] && defs="$FLAG"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $extra $defs -c mergerec_amd/csrc/$SRC -o /tmp/lib_$v/${SRC%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$v/libmergerec_hip.so /tmp/lib_$v/${SRC%.hip}.o $(ls mergerec_amd/lib/obj/*.o | grep -v "/${SRC%.hip}.o")
done
for r in 1 2; do for v in base flag; do echo "== $v ($FLAG)"; MERGEREC_HIP_LIB=/tmp/lib_$v/libmergerec_hip.so "$@"; done; done
