#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r03l_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03l_tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03l_bench.json 2> gpurun_out/r03l_bench.err
for n in 16 32; do timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 50 --images-per-gpu $n > gpurun_out/r03l_b$n.json 2>/dev/null; done
tail -n 4 gpurun_out/r03l_tests.log
python - <<'PY'
import json
o=json.loads(open('gpurun_out/r03l_bench.json').read().strip().splitlines()[-1])
print(round(o['value']), o['kernel_ms_median'], round(o['batched']['value']), o['job']['in_flight_1']['ms_per_image'], o['job']['in_flight_8']['ms_per_image'], o['job']['in_flight_16']['ms_per_image'])
for n in (16,32):
    o=json.loads(open(f'gpurun_out/r03l_b{n}.json').read().strip().splitlines()[-1]); print(n, round(o['value']))
PY
