#!/usr/bin/env bash
# A/B of two builds of the library on one box: tools/ab_libs_cg.sh <a.so> <b.so> [env...]
a=$1; b=$2; shift 2
run() {
  env SCVAE_HIP_LIBRARY=$1 "${@:2}" python bench.py --no-other-workloads --no-cpu-baseline --steps 300 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],4), 'median', round(d['step_ms_median'],4), 'head us', round(d['roofline']['launch_us'],1), 'rest', round(d['ms_per_step']*1e3-d['roofline']['launch_us'],1))"
}
for r in 1 2 3; do
  for l in $a $b; do echo "$l"; env SCVAE_HIP_LIBRARY=$(pwd)/$l "$@" python tools/time_count_gemm.py; done
done
for r in 1 2 3; do
  for l in $a $b; do echo "$l"; run $(pwd)/$l "$@"; done
done
