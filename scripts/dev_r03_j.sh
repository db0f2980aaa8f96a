#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r03j_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03j_tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03j_bench.json 2> gpurun_out/r03j_bench.err
tail -n 40 gpurun_out/r03j_tests.log
