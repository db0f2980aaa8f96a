#!/bin/bash
# sweep of k_stage2's workgroup shapes in the batch regime (development build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export FOHO_HIP_SO=$PWD/followmyhold_amd/libfoho_hip_stamps.so
B="python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 50"
run() { # label, env...
  label=$1; shift
  for ipg in 8 16; do
    env "$@" timeout 300 $B --images-per-gpu $ipg 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label ipg $ipg', round(o['value']))" >> gpurun_out/r03h_sweep.log
  done
}
rm -f gpurun_out/r03h_sweep.log
run base X=1
run rfh8 FOHO_DEBUG_RFH=8
run rfh4 FOHO_DEBUG_RFH=4
run rfh16 FOHO_DEBUG_RFH=16
run rfo32 FOHO_DEBUG_RFO=32
run lean_h128 FOHO_DEBUG_LEAN=128,64,64
run lean_all256 FOHO_DEBUG_LEAN=256,256,256
run lean_o128 FOHO_DEBUG_LEAN=256,128,128
run ifh8 FOHO_DEBUG_IFH=8
run ifh32 FOHO_DEBUG_IFH=32
run rfh8_leanh128 FOHO_DEBUG_RFH=8 FOHO_DEBUG_LEAN=128,64,64
cat gpurun_out/r03h_sweep.log
