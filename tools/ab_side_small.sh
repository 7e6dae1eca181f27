#!/usr/bin/env bash
# The second stream for small minibatches (SCVAE_SIDE_STREAM=1 forces it on; default: from 1024
# cells): tools/ab_side_small.sh [batch sizes...]
cd "$(dirname "$0")/.."
run() {
  python bench.py --no-other-workloads --no-cpu-baseline "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],4), 'median', round(d['step_ms_median'],4))"
}
for b in ${@:-100 256 512}; do
  echo "== $b cells"
  for r in 1 2 3; do
    for v in 0 1; do echo "SCVAE_SIDE_STREAM=$v"; SCVAE_SIDE_STREAM=$v run --batch $b --steps 600; done
  done
done
