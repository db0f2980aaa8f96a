#!/bin/bash
# round 3, third GPU pass: raster-role ablations at 8 images, the changed tests, the job
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_inputs.py tests/test_fullsize_gpu.py tests/test_bench_gpu.py -m gpu -q --maxfail=12 > gpurun_out/r03c_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03c_tests.log
NB=8 MASKS=0,1024,2048,3072,4096,8192,15360,61,1085,2109,3133,15421 timeout 600 python scripts/dev_roles.py > gpurun_out/r03c_roles8.log 2>&1
NB=1 MASKS=0,1024,2048,3072,4096,8192,15360,61,1085,2109,3133,15421 timeout 600 python scripts/dev_roles.py > gpurun_out/r03c_roles1.log 2>&1
timeout 400 python scripts/dev_job.py > gpurun_out/r03c_job.log 2>&1
tail -n 5 gpurun_out/r03c_tests.log
