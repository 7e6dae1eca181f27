#!/usr/bin/env bash
# ablations of decoder_head4_kernel (probe build): SCVAE_D3_DEBUG bits
#   1 consumers idle (barriers only)  2 producers idle  4 no non-zero walk  8 no dd stores
#   16 no priorities  32 younger producers prio 2, older 1  64 younger producers prio 1 only
cd "$(dirname "$0")/.."
for name in "negative binomial"; do
for dbg in ${DBGS:-0 16 32 64 0 16 32 64}; do
  echo -n "dbg $dbg: "
  SCVAE_D3_DEBUG=$dbg python tools/time_head.py 4096 32738 100 "$name" 20 2>&1 | tail -1 | cut -c1-90
done
done
