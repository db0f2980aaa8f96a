#!/bin/bash
# what would k_resolve's empty workgroups cost?  Development build, k_resolve launched over a window of the frame's tiles only
# (FOHO_DEBUG_RWIN; results invalid outside the window -- timing only) against the full frame
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/r03_rwin.log
rm -f $LOG
export FOHO_HIP_SO=$PWD/followmyhold_amd/libfoho_hip_stamps.so
B="python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 50"
run() { label=$1; shift
  for ipg in 1 8 16 32; do
    env "$@" timeout 200 $B --images-per-gpu $ipg 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label ipg $ipg', round(o['value']), o['kernel_ms']['k_resolve'], o['hit_pixels'])" >> $LOG 2>&1 || echo "$label ipg $ipg FAILED" >> $LOG
  done
}
for rep in 1 2; do
run full X=1
run win256 FOHO_DEBUG_RWIN=4,16,8,32
run win144 FOHO_DEBUG_RWIN=5,20,6,24
done
cat $LOG
