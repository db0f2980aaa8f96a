#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py tests/test_edge_gpu.py tests/test_ops_gpu.py tests/test_facade_gpu.py tests/test_inputs.py tests/test_flexi.py -m gpu -q -x > gpurun_out/r03/z_tests.log 2>&1; tail -n 2 gpurun_out/r03/z_tests.log
for i in 1 2 3; do
timeout 300 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1:', round(o['value']), 'steps/s', o['ms_per_step'], o['kernel_ms'])"
done
timeout 300 python bench.py --steps 200 --warmup 20 --images-per-gpu 16 --streams 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('16x4:', round(o['value']))"
