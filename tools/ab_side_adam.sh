#!/usr/bin/env bash
# Where the heads' share of clip + Adam leaves the step's stream (SCVAE_SIDE_ADAM_AT: 1 beside the
# hidden layers' backward pass, 2 beside the input layer's weight gradient) x where the carried
# fetch does (SCVAE_SIDE_JOBS_AT), 4096-cell headline step: tools/ab_side_adam.sh
cd "$(dirname "$0")/.."
run() {
  python bench.py --no-other-workloads --no-cpu-baseline --steps 300 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],4), 'median', round(d['step_ms_median'],4), 'head us', round(d['roofline']['launch_us'],1), 'rest', round(d['ms_per_step']*1e3-d['roofline']['launch_us'],1))"
}
for r in 1 2 3; do
  for c in "1 2" "2 2" "2 1" "1 1"; do
    set -- $c
    echo "ADAM_AT=$1 JOBS_AT=$2"; SCVAE_SIDE_ADAM_AT=$1 SCVAE_SIDE_JOBS_AT=$2 run
  done
done
