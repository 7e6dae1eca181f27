#!/usr/bin/env bash
# Probe build: libscvae_hip_tcprobe.so = the regular library with tilechain.hip compiled
# -DSCVAE_TC_PROBE (per-phase s_memtime sums of the resident tile-chain kernels, tools/tc_probe.py).
set -euo pipefail
cd "$(dirname "$0")"
bash build.sh > /dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DSCVAE_TC_PROBE=1 ${EXTRA:-} -c tilechain.hip -o /tmp/tilechain_probe.o
objs=$(ls build/*.o | grep -v tilechain.o)
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs /tmp/tilechain_probe.o -o libscvae_hip_tcprobe.so
echo "built $(pwd)/libscvae_hip_tcprobe.so"
