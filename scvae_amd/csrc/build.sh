#!/usr/bin/env bash
# Build libscvae_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p build
pids=()
for src in *.hip; do
  obj=build/${src%.hip}.o
  if [[ ! -f $obj || $src -nt $obj || common.hpp -nt $obj || kernels.hpp -nt $obj || likelihood.hpp -nt $obj || plan.hpp -nt $obj || ../../include/scvae_hip.h -nt $obj ]]; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -o libscvae_hip.so
echo "built $(pwd)/libscvae_hip.so"
