#!/usr/bin/env bash
# A/B on one box: tile-chain stages launched in segments (own-rows stages share a launch) vs one
# launch per stage
for r in 1 2; do
for v in 0 1; do
  echo "SCVAE_TILE_SEGMENTS=$v"
  SCVAE_TILE_SEGMENTS=$v python bench.py --no-other-workloads --no-cpu-baseline --steps 40 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],4), 'median', round(d['step_ms_median'],4), 'head us', round(d['roofline']['launch_us'],1))"
done
done
