#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in 256 128 64 32 256 64; do
for cfg in "8 1" "8 4" "16 4" "32 4"; do set -- $cfg
FOHO_TMP_GTILES=$v timeout 300 python bench.py --steps 200 --warmup 20 --images-per-gpu $1 --streams $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gtiles $v images $1 streams $2:', round(o['value']), 'steps/s', o['kernel_ms'].get('k_pix_bwd'))"
done; done
