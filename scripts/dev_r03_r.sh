#!/bin/bash
# k_stage2 grid transposed for batches (block-major dispatch): A/B on the batched configurations
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
for t in 0 1 0 1; do
  for cfg in "8 1" "8 4" "16 4" "32 4"; do
    set -- $cfg
    FOHO_S2_TRANSPOSE=$t timeout 300 python bench.py --steps 200 --warmup 20 --images-per-gpu $1 --streams $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('transpose $t images $1 streams $2:', round(o['value']), 'steps/s', o['kernel_ms'].get('k_stage2') if isinstance(o.get('kernel_ms'), dict) else '')"
  done
done
