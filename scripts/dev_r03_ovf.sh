#!/bin/bash
# how wide may k_resolve_ovf be?  32 / 128 / 256 workgroups per row (they leave at once on the benchmark frames) and the dense launch
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
LOG=gpurun_out/r03_ovf.log
rm -f $LOG
B="python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 50 --images-per-gpu 32"
run() { label=$1; shift
  env "$@" timeout 200 $B 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(o['value']))" >> $LOG 2>&1
}
for rep in 1 2 3; do
run dense FOHO_LISTED_CAP=0
run ovf32 X=1
run ovf128 FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_ovf128.so
run ovf256 FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_ovf256.so
run ovf256_cap128 FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_ovf256.so FOHO_LISTED_CAP=128
run ovf32_cap64 FOHO_LISTED_CAP=64
done
cat $LOG
