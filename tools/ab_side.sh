#!/usr/bin/env bash
# A/B on one box: the work a step carries (next fetch + noise, clip + Adam) on the plan's second
# stream (SCVAE_SIDE_STREAM=1; the heads' part of Adam forked at point SCVAE_SIDE_ADAM_AT: 1 after
# the head kernel, 2 before x^T dA) against the end of the step on its own stream
run() {
  python bench.py --no-other-workloads --no-cpu-baseline --steps 300 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],4), 'median', round(d['step_ms_median'],4), 'head us', round(d['roofline']['launch_us'],1), 'rest', round(d['ms_per_step']*1e3-d['roofline']['launch_us'],1))"
}
for r in 1 2 3; do
  echo "SCVAE_SIDE_STREAM=0"; SCVAE_SIDE_STREAM=0 run
  echo "SCVAE_SIDE_STREAM=1 SCVAE_SIDE_ADAM_AT=1"; SCVAE_SIDE_STREAM=1 SCVAE_SIDE_ADAM_AT=1 run
done
