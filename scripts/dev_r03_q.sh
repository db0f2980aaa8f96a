#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_inputs.py -m gpu -q > gpurun_out/r03/q_tests.log 2>&1; tail -n 8 gpurun_out/r03/q_tests.log
bash scripts/dev_r03_p.sh "$@"
