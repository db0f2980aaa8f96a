#!/bin/bash
# pipelined slot runner: tests of the drivers, job rates, end-to-end driver on files
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_inputs.py tests/test_step_gpu.py tests/test_bench_gpu.py -m gpu -q > gpurun_out/r03/o_tests.log 2>&1
tail -n 15 gpurun_out/r03/o_tests.log
timeout 600 python scripts/dev_job.py > gpurun_out/r03/o_job.log 2>&1; tail -n 12 gpurun_out/r03/o_job.log
timeout 600 python scripts/dev_driver_e2e.py 48 > gpurun_out/r03/o_e2e.log 2>&1; grep -v "^ \|^$" gpurun_out/r03/o_e2e.log | tail -n 8
