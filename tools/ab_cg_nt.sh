#!/usr/bin/env bash
# A/B of non-temporal requests for x in the input layer's count kernels (-DSCVAE_CG_NT=1: forward
# product, 2: weight gradient, 3: both; variant libraries linked next to the build):
#   tools/ab_cg_nt.sh            stand-alone kernels (three alternations), then the step
cd "$(dirname "$0")/.."
L=scvae_amd/csrc
libs="$L/libscvae_hip.so $L/libscvae_hip_nt1.so $L/libscvae_hip_nt2.so $L/libscvae_hip_nt3.so"
for r in 1 2 3; do
  for l in $libs; do echo -n "$(basename $l)  "; env SCVAE_HIP_LIBRARY=$(pwd)/$l python tools/time_count_gemm.py; done
done
run() {
  env SCVAE_HIP_LIBRARY=$1 python bench.py --no-other-workloads --no-cpu-baseline --steps 300 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],4), 'median', round(d['step_ms_median'],4), 'head us', round(d['roofline']['launch_us'],1), 'rest', round(d['ms_per_step']*1e3-d['roofline']['launch_us'],1))"
}
for r in 1 2 3; do
  for l in $libs; do echo -n "$(basename $l)"; run $(pwd)/$l; done
done
