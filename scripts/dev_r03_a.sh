#!/bin/bash
# round 3, first GPU pass: the whole GPU suite, the default bench line, the per-image job
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --deselect tests/test_bench_gpu.py > gpurun_out/r03a_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03a_tests.log
timeout 600 python -m pytest tests/test_bench_gpu.py tests/test_inputs.py -m gpu -q > gpurun_out/r03a_tests2.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03a_tests2.log
timeout 600 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
timeout 400 python scripts/dev_job.py > gpurun_out/r03a_job.log 2>&1
tail -n 5 gpurun_out/r03a_tests.log gpurun_out/r03a_tests2.log
