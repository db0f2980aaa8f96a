R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
timeout 600 python scripts/diag_closeup_extrema.py 7 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/diag_extrema.log
cat $O/diag_extrema.log
timeout 600 python -m pytest tests/test_facade_gpu.py -m gpu -q -x 2>&1 | tail -15
make -C followmyhold_amd/csrc STAMPS=1 -s 2>&1 | grep -E "error"
for v in "4 256" "2 256" "2 512" "4 512"; do set -- $v
    echo "rf_h=$1 gtiles=$2" >> $O/sweep.log
    FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_stamps.so FOHO_DEBUG_RFH=$1 FOHO_DEBUG_GTILES=$2 timeout 200 python scripts/run_steps.py --steps 300 2>&1 | grep "steps/s" >> $O/sweep.log
    FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_stamps.so FOHO_DEBUG_RFH=$1 FOHO_DEBUG_GTILES=$2 timeout 200 python scripts/run_steps.py --crop hoi --steps 300 2>&1 | grep "steps/s" >> $O/sweep.log
    FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_stamps.so FOHO_DEBUG_RFH=$1 FOHO_DEBUG_GTILES=$2 timeout 200 python scripts/run_steps.py --obj 40k --steps 300 2>&1 | grep "steps/s" >> $O/sweep.log
done
cat $O/sweep.log
