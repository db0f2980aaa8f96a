#!/bin/bash
# round 3, pass d: byte plane for full-coverage fragments + division-light winner distance: parity suite, bench, batch timelines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r03d_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03d_tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench.err
timeout 300 python scripts/dev_spans_batch.py 8 > gpurun_out/r03d_spans8.log 2>&1
timeout 300 python scripts/dev_spans_batch.py 2 > gpurun_out/r03d_spans2.log 2>&1
tail -n 5 gpurun_out/r03d_tests.log
