#!/usr/bin/env bash
# Round 6: the per-phase breakdown of decoder_head4_kernel as a tracked profile (VERDICT round 5,
# next 3): s_memtime sums per section and 32-row tile for the waves of workgroup 0 (probe build,
# scvae_amd/csrc/build_prof.sh: run it first, here or on the box), two heads (NB, eight producer
# waves) and three (ZINB, four), dd through atomics as the step runs it -- and the SQ counters of
# the same launches from the REGULAR build (instruction mix, MFMA busy, where the waves wait).
#   tools/round6_head_profile.sh      ->  gpurun_out/r06_head4_phases.txt, r06_head4_pmc.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r06_head4_phases.txt
{
  echo "# decoder_head4_kernel, 4096 rows x 32 738 genes, H = 100, uint16 targets, dd through XCD-local atomics"
  echo "# cycles (s_memtime) per 32-row tile and section, waves of workgroup 0; probe build (-DD4_PROF=1: the"
  echo "# stamps themselves cost ~10 % of a tile).  Producers: gemm1 = GEMM1 of the NEXT tile (9 bf16 MFMAs per"
  echo "# product), dense = likelihood of the tile's elements, walk = the non-zero queue (lgamma / digamma),"
  echo "# sum+G = row sums + the cut of G into bf16 planes, barrier = wait for the consumers."
  echo "# Consumers: prep = fragments of G / d, gemm3 = dd += G W^T, ddstore = atomic adds of dd,"
  echo "# gemm2 = dW += d^T G, barrier = wait for the producers."
  for lk in "negative binomial" "zero-inflated negative binomial" "poisson"; do
    echo
    TIME_HEAD_FLAGS=0x400 python tools/d4_prof.py "$lk" 4096 2>&1 | grep -v amdgpu.ids
  done
  echo
  echo "# kernel time of the same launches, regular build (tools/time_head.py, 20 launches):"
  for lk in "negative binomial" "zero-inflated negative binomial" "poisson"; do
    TIME_HEAD_FLAGS=0x400 python tools/time_head.py 4096 32738 100 "$lk" 20 2>&1 | tail -1
  done
} > $out 2>&1
cat $out
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
G2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES SQ_IFETCH"
for lk in "negative binomial" "zero-inflated negative binomial"; do
  tag=$(echo "$lk" | tr ' -' '__')
  TIME_HEAD_FLAGS=0x400 tools/prof_pmc.sh r06_head4_$tag "$G1" "$G2" "$G3" -- python tools/time_head.py 4096 32738 100 "$lk" 3 2>&1 | grep -E "^kernel|decoder_head4" 
done
