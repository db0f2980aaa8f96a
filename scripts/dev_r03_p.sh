#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
for n in "$@"; do
timeout 900 python scripts/dev_driver_e2e.py $n > gpurun_out/r03/p_e2e_$n.log 2>&1; grep "in flight\|fixtures" gpurun_out/r03/p_e2e_$n.log
awk '/function calls/{p=1} p{print}' gpurun_out/r03/p_e2e_$n.log | head -34 | cut -c1-180
done
