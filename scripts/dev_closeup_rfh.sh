#!/bin/bash
# close-up frames (fov 22 / 30) at 32 images in flight: hand faces per raster workgroup (development build, FOHO_DEBUG_RFH)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FOHO_HIP_SO=$PWD/followmyhold_amd/libfoho_hip_stamps.so
LOG=gpurun_out/r03_closeup_rfh.log
rm -f $LOG
for v in default 4 8 16 64; do
  if [ $v = default ]; then unset FOHO_DEBUG_RFH; else export FOHO_DEBUG_RFH=$v; fi
  echo "RFH $v" >> $LOG
  timeout 300 python scripts/dev_closeup.py 60 22 2>&1 | grep "None" >> $LOG
done
cat $LOG
