#!/bin/bash
# nearest-neighbour role: wave-per-query form against the LDS-tiled form of the previous build (libfoho_hip_base.so), back to back
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/r03_knn.log
rm -f $LOG
timeout 600 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $LOG
B="python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 50"
run() { # label, env...
  label=$1; shift
  for ipg in 1 8 16 32; do
    env "$@" timeout 300 $B --images-per-gpu $ipg 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label ipg $ipg', round(o['value']), o.get('kernel_ms'))" >> $LOG
  done
}
for rep in 1 2; do
run base FOHO_HIP_SO=$PWD/followmyhold_amd/libfoho_hip_base.so
run new X=1
done
cat $LOG
