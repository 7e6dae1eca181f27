#!/usr/bin/env bash
# A/B on one box: row groups of the producer / consumer head kernel (SCVAE_D4_ROW_GROUPS=1: off)
run() {
  python bench.py --no-other-workloads --no-cpu-baseline --steps 20 "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],4), 'median', round(d['step_ms_median'],4), 'head us', round(d['roofline']['launch_us'],1), 'frac', round(d['roofline']['frac'],3))"
}
for v in 1 0; do
  echo "SCVAE_D4_ROW_GROUPS=$v (0: chosen by the launcher)"
  echo " cfg5 shape (ZINB GMVAE K=20, F=27998, 512 cells)"
  SCVAE_D4_ROW_GROUPS=$v run --model gmvae --likelihood "zero-inflated negative binomial" --features 27998 --cells 16384 --latent 100 --batch 512
  echo " cfg4 (NB GMVAE K=20, 512 cells)"
  SCVAE_D4_ROW_GROUPS=$v run --model gmvae --latent 100 --batch 512
  echo " headline"
  SCVAE_D4_ROW_GROUPS=$v run
done
