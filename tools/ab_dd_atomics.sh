#!/usr/bin/env bash
# A/B on one box: dd through per-strip slabs + reduce (default) against XCD-local fp32 atomics
# (SCVAE_HEADS_DD_ATOMICS), at 4096 x 32 738; "kernel + reduces" is what counts.
#   tools/ab_dd_atomics.sh ["likelihood" ...]      (SCVAE_D3_SCHEDULE etc. pass through)
cd "$(dirname "$0")/.."
if [ $# -eq 0 ]; then set -- "negative binomial" "poisson"; fi
for rep in 1 2; do
  for flags in 0 0x400; do
    for name in "$@"; do
      echo -n "flags $flags: "
      TIME_HEAD_FLAGS=$flags python tools/time_head.py 4096 32738 100 "$name" 20 2>&1 | tail -1
    done
  done
done
