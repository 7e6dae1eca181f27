#!/usr/bin/env bash
# SQ counters of the decoder-head training kernel under both schedules (one rocprofv3 --pmc pass per group)
cd "$(dirname "$0")/.."
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
G2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES SQ_IFETCH"
for cfg in "3 8" "4 4" "4 8"; do
  set -- $cfg
  export SCVAE_D3_SCHEDULE=$1 SCVAE_D4_PRODUCERS=$2
  tools/prof_pmc.sh d4pmc_s$1_p$2 "$G1" "$G2" "$G3" -- python tools/time_head.py 4096 32738 100 "negative binomial" 3 2>&1 | grep -E "^kernel|decoder_head"
done
