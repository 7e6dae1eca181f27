#!/usr/bin/env bash
# time the head kernel (kernel + reduces, NB, atomics and slabs) of several builds, alternating
cd "$(dirname "$0")/.."
for rep in 1 2; do
for lib in "$@"; do
  for flags in 0x400 0; do
    echo -n "$(basename $lib) flags $flags: "
    SCVAE_HIP_LIBRARY=$(realpath $lib) TIME_HEAD_FLAGS=$flags python tools/time_head.py 4096 32738 100 "${LIKELIHOOD:-negative binomial}" 20 2>&1 | tail -1 | cut -c1-95
  done
done
done
