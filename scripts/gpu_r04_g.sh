# Round-4, GPU call G: geometry decoder with LDS-DMA GEMM + leaner softmax; k_pix_bwd workgroups per render on crops (development build).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_geo_decode.py -m gpu -q -x > $O/pytest_geo.log 2>&1
tail -5 $O/pytest_geo.log
timeout 600 python scripts/geo_bench.py --parts > $O/geo_bench.log 2>&1
grep -v amdgpu.ids $O/geo_bench.log | tail -8
make -C followmyhold_amd/csrc STAMPS=1 -s 2>&1 | grep -E "error"
for gt in 128 256 512 1024; do
    echo "gtiles=$gt" >> $O/gtiles.log
    FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_stamps.so FOHO_DEBUG_GTILES=$gt FOHO_DEBUG_RFH=2 timeout 200 python scripts/run_steps.py --crop hoi --steps 300 2>&1 | grep "steps/s" >> $O/gtiles.log
    FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_stamps.so FOHO_DEBUG_GTILES=$gt timeout 200 python scripts/run_steps.py --crop hoi --images 32 --streams 4 --steps 100 2>&1 | grep "steps/s" >> $O/gtiles.log
    FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_stamps.so FOHO_DEBUG_GTILES=$gt timeout 200 python scripts/run_steps.py --steps 300 2>&1 | grep "steps/s" >> $O/gtiles.log
done
cat $O/gtiles.log
