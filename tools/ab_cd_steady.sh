#!/usr/bin/env bash
# A/B on one box: the weight-gradient count kernel with its unconditional main loop and the dA
# requests issued ahead of the x requests (SCVAE_CD_STEADY=1, the build) against the loop whose
# wait for dA also waited for every x request (0): stand-alone, then in the step
for r in 1 2 3; do
  for v in 0 1; do
    echo "SCVAE_CD_STEADY=$v"; SCVAE_CD_STEADY=$v python tools/time_count_gemm.py
  done
done
run() {
  python bench.py --no-other-workloads --no-cpu-baseline --steps 300 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],4), 'median', round(d['step_ms_median'],4), 'head us', round(d['roofline']['launch_us'],1), 'rest', round(d['ms_per_step']*1e3-d['roofline']['launch_us'],1))"
}
for r in 1 2 3; do
  for v in 0 1; do echo "SCVAE_CD_STEADY=$v"; SCVAE_CD_STEADY=$v run; done
done
