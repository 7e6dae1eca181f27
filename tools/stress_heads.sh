#!/usr/bin/env bash
# Run-to-run repeatability (ll / db bitwise over 30 launches) of the training head kernel on the GPU box:
# three likelihoods x {nine-term, six-term} x {dd slabs, dd atomics}; flags as in scvae_amd/_lib.py.
for lik in "negative binomial" "zero-inflated negative binomial" "poisson"; do
for fl in 0x200 0x600 0x800 0xC00; do
  echo -n "$lik flags $fl: "
  TIME_HEAD_STRESS=30 TIME_HEAD_FLAGS=$fl python tools/time_head.py 4096 32738 100 "$lik" 5 2>&1 | grep -E "stress|rror" | tail -1
done; done
