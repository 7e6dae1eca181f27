#!/usr/bin/env bash
# A/B builds of ONE kernel file: tools/ab_variants.sh <file.hip> <MACRO> <v0> <v1> ...
# compiles scvae_amd/csrc/<file.hip> with -D<MACRO>=<v> for every value and links it with the
# other objects of the regular build into scvae_amd/csrc/build/ab/libscvae_hip_<v>.so (run here:
# hipcc cross-compiles; the variants travel to the GPU box with the snapshot).  Load one with
# SCVAE_HIP_LIBRARY=<path>.
set -euo pipefail
cd "$(dirname "$0")/../scvae_amd/csrc"
src=$1; macro=$2; shift 2
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
bash build.sh > /dev/null
mkdir -p ab
pids=()
for v in "$@"; do
  ( $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -D$macro=$v -c "$src" -o ab/${src%.hip}_$v.o
    objs=$(ls build/*.o | grep -v "build/${src%.hip}.o")
    $HIPCC --offload-arch=gfx950 -shared -fPIC $objs ab/${src%.hip}_$v.o -o ab/libscvae_hip_$v.so ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
ls -la ab/*.so
