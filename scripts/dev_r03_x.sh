#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py tests/test_edge_gpu.py -m gpu -q -x > gpurun_out/r03/x_tests.log 2>&1; tail -n 3 gpurun_out/r03/x_tests.log
for v in 1 4 2 8 1 4; do
for cfg in "8 1" "8 4" "16 4" "32 4"; do set -- $cfg
FOHO_TMP_GT=$v timeout 300 python bench.py --steps 200 --warmup 20 --images-per-gpu $1 --streams $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gt $v images $1 streams $2:', round(o['value']), 'steps/s', o['kernel_ms'].get('k_resolve'))"
done; done
