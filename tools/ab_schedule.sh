#!/usr/bin/env bash
# A/B of the decoder-head training schedules on one box: SCVAE_D3_SCHEDULE=3 (all waves in one
# phase) against 4 (producer / consumer waves), NB / ZINB / Poisson at 4096 x 32 738, alternating.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for sch in 3 4; do
    for name in "negative binomial" "zero-inflated negative binomial" "poisson"; do
      echo -n "schedule $sch: "
      SCVAE_D3_SCHEDULE=$sch python tools/time_head.py 4096 32738 100 "$name" 20 2>&1 | tail -1
    done
  done
done
