#!/bin/bash
# final verification of the round: GPU suite, smoke, the driver's bench invocation and the default one
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03/final_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03/final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/final_smoke.log 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_driver_invocation.json 2> gpurun_out/r03/bench_driver_invocation.err
timeout 900 python bench.py > gpurun_out/r03/bench_b1.json 2> gpurun_out/r03/bench_b1.err
tail -n 3 gpurun_out/r03/final_tests.log; tail -n 1 gpurun_out/r03/final_smoke.log
python - <<'PY'
import json
for f in ("bench_driver_invocation", "bench_b1"):
    o=json.loads(open(f'gpurun_out/r03/{f}.json').read().strip().splitlines()[-1])
    print(f, round(o['value']), o['ms_per_step'], o['repeats'], o['roofline']['kernel'], round(o['roofline']['frac'],4), o['roofline']['traffic'], o['roofline']['dominant_by_trace'], o.get('step_traffic_MB'), round(o['batched']['value']), o['batched'].get('step_traffic_frac'))
PY
