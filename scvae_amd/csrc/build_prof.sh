#!/usr/bin/env bash
# Probe build: libscvae_hip_prof.so = the regular library with decoder_fused3.hip compiled
# -DD4_PROF=1 (per-section s_memtime sums of decoder_head4_kernel, scvae_d4_prof_dump).
set -euo pipefail
cd "$(dirname "$0")"
bash build.sh > /dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DD4_PROF=1 ${EXTRA:-} -c decoder_fused3.hip -o build/decoder_fused3_prof.o.tmp
objs=$(ls build/*.o | grep -v decoder_fused3.o)
cp build/decoder_fused3_prof.o.tmp /tmp/decoder_fused3_prof.o
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs /tmp/decoder_fused3_prof.o -o libscvae_hip_prof.so
rm -f build/decoder_fused3_prof.o.tmp
echo "built $(pwd)/libscvae_hip_prof.so"
