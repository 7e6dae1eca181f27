#!/usr/bin/env bash
# A/B on one box: where the carried fetch + noise leave the step's stream (SCVAE_SIDE_JOBS_AT:
# 0 at the start of the step, beside the input layer; 1 after the head kernel, beside the
# backward pass of the hidden layers; 3: never -- in line at the very end of the step, behind Adam,
# so that the next step's input layer finds the minibatch in the infinity cache; the heads' share
# of Adam still forks).  tools/ab_side_jobs.sh [values...]   (default: 1 0)
run() {
  python bench.py --no-other-workloads --no-cpu-baseline --steps 300 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],4), 'median', round(d['step_ms_median'],4), 'head us', round(d['roofline']['launch_us'],1), 'rest', round(d['ms_per_step']*1e3-d['roofline']['launch_us'],1))"
}
vals=${@:-1 0}
for r in 1 2 3; do
  for v in $vals; do echo -n "SCVAE_SIDE_JOBS_AT=$v"; SCVAE_SIDE_JOBS_AT=$v run; done
done
