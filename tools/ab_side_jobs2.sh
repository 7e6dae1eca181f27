#!/usr/bin/env bash
# SCVAE_SIDE_JOBS_AT = 1 against 2 (the carried fetch beside the hidden layers' backward pass /
# beside the input layer's weight gradient) over several workloads: tools/ab_side_jobs2.sh
cd "$(dirname "$0")/.."
run() {
  python bench.py --no-other-workloads --no-cpu-baseline "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],4), 'median', round(d['step_ms_median'],4))"
}
ab() {
  echo "== $*"
  for r in 1 2 3; do
    for v in ${VALUES:-1 2}; do echo "SCVAE_SIDE_JOBS_AT=$v"; SCVAE_SIDE_JOBS_AT=$v run "$@"; done
  done
}
if [[ -n "${QUICK:-}" ]]; then ab --steps 300; ab --steps 150 --batch 8192; exit 0; fi
ab --steps 300
ab --steps 200 --likelihood "zero-inflated negative binomial" --latent 100
ab --steps 300 --likelihood poisson
ab --steps 300 --batch 1024
ab --steps 300 --batch 2048
ab --steps 60 --batch 16384
