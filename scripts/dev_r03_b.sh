#!/bin/bash
# round 3, second GPU pass: the whole GPU suite, determinism probe, role ablation at 8 images, device timelines of phases A / C
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r03b_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03b_tests.log
timeout 600 python scripts/dev_determinism.py > gpurun_out/r03b_determinism.log 2>&1
NB=8 timeout 600 python scripts/dev_roles.py > gpurun_out/r03b_roles8.log 2>&1
timeout 300 python scripts/dev_spans.py 20k A > gpurun_out/r03b_spansA.log 2>&1
timeout 300 python scripts/dev_spans.py 20k C > gpurun_out/r03b_spansC.log 2>&1
tail -n 5 gpurun_out/r03b_tests.log
