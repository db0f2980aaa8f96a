#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python scripts/dev_streams_queues.py > gpurun_out/r03f_queues.log 2>&1
GPU_MAX_HW_QUEUES=8 timeout 300 python scripts/dev_streams_queues.py > gpurun_out/r03f_queues8.log 2>&1
tail -n 22 gpurun_out/r03f_queues.log
