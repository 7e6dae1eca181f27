#!/usr/bin/env bash
# A/B on one box: the weight-gradient count kernel with 512-gene workgroups of eight waves, one
# per CU (SCVAE_CD_WAVES=8: the dA planes of a chunk fetched once for twice the genes) against
# 256-gene workgroups, two per CU (the build): stand-alone, then in the step
for r in 1 2 3; do
  for v in 4 8; do
    echo "SCVAE_CD_WAVES=$v"; SCVAE_CD_WAVES=$v python tools/time_count_gemm.py
  done
done
run() {
  python bench.py --no-other-workloads --no-cpu-baseline --steps 300 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],4), 'median', round(d['step_ms_median'],4), 'head us', round(d['roofline']['launch_us'],1), 'rest', round(d['ms_per_step']*1e3-d['roofline']['launch_us'],1))"
}
for r in 1 2 3; do
  for v in 4 8; do echo "SCVAE_CD_WAVES=$v"; SCVAE_CD_WAVES=$v run; done
done
