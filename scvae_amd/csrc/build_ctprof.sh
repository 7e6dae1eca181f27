#!/usr/bin/env bash
# Probe build: libscvae_hip_ctprof.so = the regular library with count_gemm.hip compiled -DCT_PROF=1
# (per-section s_memtime sums of count_tiles_fwd_kernel, scvae_ct_prof_dump; tools/ct_prof.py).
set -euo pipefail
cd "$(dirname "$0")"
bash build.sh > /dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DCT_PROF=1 ${CT_EXTRA:-} -c count_gemm.hip -o /tmp/count_gemm_ctprof.o
objs=$(ls build/*.o | grep -v "count_gemm.o")
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs /tmp/count_gemm_ctprof.o -o ${CT_OUT:-libscvae_hip_ctprof.so}
echo "built $(pwd)/${CT_OUT:-libscvae_hip_ctprof.so}"
