# Round-4, GPU call F: per-term gradient diagnostic (totals), hand faces per raster workgroup on crops (development build).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout 600 python scripts/diag_closeup_terms.py 9 2>&1 | grep -v "amdgpu.ids\|Warning\|print(f" > $O/diag_terms.log
cat $O/diag_terms.log
make -C followmyhold_amd/csrc STAMPS=1 -s 2>&1 | grep -E "error"
for rfh in 1 2 4 8; do
  for rfo in 32 64; do
    echo "rf_h=$rfh rf_o=$rfo" >> $O/rfh.log
    FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_stamps.so FOHO_DEBUG_RFH=$rfh FOHO_DEBUG_RFO=$rfo timeout 200 python scripts/run_steps.py --crop hoi --steps 300 2>&1 | grep "steps/s" >> $O/rfh.log
    FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_stamps.so FOHO_DEBUG_RFH=$rfh FOHO_DEBUG_RFO=$rfo timeout 200 python scripts/run_steps.py --crop hoi --images 32 --streams 4 --steps 100 2>&1 | grep "steps/s" >> $O/rfh.log
  done
done
cat $O/rfh.log
