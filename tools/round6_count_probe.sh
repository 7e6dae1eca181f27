#!/usr/bin/env bash
# Round 6: where a chunk of the count products' forward kernel goes.
#   tools/round6_count_probe.sh build      (here: the probe builds, in-tree, they travel with gpurun)
#   gpurun -- bash tools/round6_count_probe.sh run   -> gpurun_out/r06_count_probe.txt
# Probe builds: scvae_amd/csrc/build_ctprof.sh (-DCT_PROF=1: s_memtime per section of
# count_fwd2_kernel, workgroup (0, 0)); -DCT_EXP=4 / 5 / 7 switch one of the two roles off
# (wrong results, timing only).
set -uo pipefail
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd "$repo"
if [ "${1:-run}" = build ]; then
  bash scvae_amd/csrc/build_ctprof.sh
  for e in 4 5 7; do
    CT_EXTRA="-DCT_EXP=$e" CT_OUT=libscvae_hip_ctexp$e.so bash scvae_amd/csrc/build_ctprof.sh
  done
  exit 0
fi
out=gpurun_out/r06_count_probe.txt
mkdir -p gpurun_out
{
  echo "# Round 6: count_fwd2_kernel (two roles: waves 0-3 multiply, waves 4-7 stage the x side, all"
  echo "# eight move the W planes), 4096 cells x 32 738 genes x 100 units, one MI355X box, one gpurun call."
  echo "# s_memtime cycles per chunk (a chunk = 32 genes of the contraction for 256 cells: 48 MFMAs"
  echo "# 32x32x16 per multiplying wave = 1536 cycles of its SIMD's matrix pipe), probe build"
  echo "# (-DCT_PROF=1; every stamp costs ~60 cycles).  Sections -- multiplying waves: work (fragments,"
  echo "# MFMAs, their three pieces of the W planes), barrier; staging waves: wait, un-scatter (tiles)"
  echo "# or conversion + hi plane (uint16 batch), scatter, lo plane + flag, requests, W planes, barrier."
  echo
  echo "## x from the count tiles (scvae_count_gemm_tiles)"
  python tools/ct_prof.py 4096 2>&1 | grep -E "^count|^wave"
  echo
  echo "## x from the dense uint16 batch, SCVAE_CG_FWD2=1 (scvae_count_gemm_u16)"
  SCVAE_CG_FWD2=1 python tools/ct_prof.py 4096 dense 2>&1 | grep -E "^count|^wave"
  echo
  echo "## one role at a time (tiles; wrong results, timing only)"
  echo "# -DCT_EXP=4: the multiplying waves issue no MFMAs (the staging waves alone)"
  CT_LIB=libscvae_hip_ctexp4.so python tools/ct_prof.py 4096 2>&1 | grep -E "^wave [04]"
  echo "# -DCT_EXP=5: the staging waves only meet the barrier (the multiplying waves alone, with their W pieces)"
  CT_LIB=libscvae_hip_ctexp5.so python tools/ct_prof.py 4096 2>&1 | grep -E "^wave [04]"
  echo "# -DCT_EXP=7: as 5, and no W pieces in the multiplying waves (fragment reads + 48 MFMAs + barrier)"
  CT_LIB=libscvae_hip_ctexp7.so python tools/ct_prof.py 4096 2>&1 | grep -E "^wave [04]"
  echo
  echo "## kernel times, regular build (rocprofv3 --kernel-trace --stats of tools/time_count_tiles.py;"
  echo "## columns: grid, calls, total us, average us, minimum us, % of the run)"
  bash tools/prof_stats.sh r06ct python tools/time_count_tiles.py > /dev/null 2>&1
  grep -E "count_fwd2|count_gemm_fwd|count_tiles_dw|count_gemm_dw|csr_count_tiles|csr_densify_u16|split3_transpose|count_gemm_reduce" \
    gpurun_out/r06ct_kernel_stats.txt | cut -c1-165
  echo "# the same with SCVAE_CG_FWD2=1 (the dense forward product through count_fwd2_kernel<2, false>)"
  SCVAE_CG_FWD2=1 bash tools/prof_stats.sh r06ct2 python tools/time_count_tiles.py > /dev/null 2>&1
  grep -E "count_fwd2|count_gemm_fwd" gpurun_out/r06ct2_kernel_stats.txt | cut -c1-165
  echo
  echo "## tools/time_count_tiles.py (split + kernel + reduce, 20 launches each)"
  python tools/time_count_tiles.py 2>&1 | tail -3
} > "$out" 2>&1
cat "$out"
