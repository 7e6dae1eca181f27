#!/usr/bin/env bash
# instruction counts of decoder_head4_kernel's parts (probe flags SCVAE_D3_DEBUG: 1 consumers idle, 2 producers idle, 4 no walk)
cd "$(dirname "$0")/.."
G2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for cfg in "8 0" "8 1" "8 2" "8 4" "4 0" "4 4"; do
  set -- $cfg
  export SCVAE_D3_SCHEDULE=4 SCVAE_D4_PRODUCERS=$1 SCVAE_D3_DEBUG=$2
  echo "producers $1 debug $2"
  tools/prof_pmc.sh d4pmc2 "$G2" -- python tools/time_head.py 4096 32738 100 "negative binomial" 3 2>&1 | grep -E "decoder_head"
done
