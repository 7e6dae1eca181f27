#!/usr/bin/env bash
# A/B of builds of the head kernel: NB / ZINB, atomics, alternating (tools/time_head.py)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for lik in "negative binomial" "zero-inflated negative binomial"; do
for lib in "$@"; do
    echo -n "$(basename $lib) $lik: "
    SCVAE_HIP_LIBRARY=$(realpath $lib) TIME_HEAD_FLAGS=0x400 python tools/time_head.py 4096 32738 100 "$lik" 20 2>&1 | tail -1 | cut -c1-110
done
done
done
