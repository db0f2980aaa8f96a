#!/bin/bash
# k_stage2's LDS per workgroup against throughput: 9.6 KB (shipped) / 8.3 KB (raster queue 512) / 7.6 KB (+ nearest-neighbour stages of 256) / 6.6 KB (queue 256)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/r03_lds.log
rm -f $LOG
B="python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 50"
run() { # label, env...
  label=$1; shift
  for ipg in 1 8 16 32; do
    env "$@" timeout 200 $B --images-per-gpu $ipg 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label ipg $ipg', round(o['value']))" >> $LOG 2>&1 || echo "$label ipg $ipg FAILED" >> $LOG
  done
}
for rep in 1 2; do
run base X=1
for v in q512 q512s256 q256s256; do run $v FOHO_HIP_SO=$PWD/followmyhold_amd/libfoho_hip_$v.so; done
done
cat $LOG
