#!/bin/bash
# enumerate diet: parity tests, batched bench lines, VALU by phase
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py tests/test_edge_gpu.py tests/test_ops_gpu.py -m gpu -q -x > $O/w_tests.log 2>&1; tail -n 4 $O/w_tests.log
for cfg in "1 1" "8 1" "8 4" "16 4" "32 4"; do set -- $cfg
timeout 300 python bench.py --steps 200 --warmup 20 --images-per-gpu $1 --streams $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('images $1 streams $2:', round(o['value']), 'steps/s', o['kernel_ms'].get('k_stage2'))"
done
bash scripts/dev_r03_v.sh
