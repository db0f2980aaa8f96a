# Round-4, GPU call H: geometry decoder (LDS-DMA GEMM, compiler-visible max3) tests + timings; close-up test with the exact-sum oracle variant.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
for rep in 1 2 3; do timeout 900 python -m pytest tests/test_geo_decode.py -m gpu -q -x 2>&1 | tail -3; done > $O/pytest_geo.log 2>&1
cat $O/pytest_geo.log
timeout 600 python scripts/geo_bench.py --parts > $O/geo_bench.log 2>&1
grep -v amdgpu.ids $O/geo_bench.log | tail -8
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x > $O/pytest_fullsize.log 2>&1
tail -15 $O/pytest_fullsize.log
