# Round-4, GPU call B: geometry decoder tests + timings, the close-up test again.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_geo_decode.py -m gpu -q -x > $O/pytest_geo.log 2>&1
tail -25 $O/pytest_geo.log
timeout 600 python scripts/geo_bench.py --parts > $O/geo_bench.log 2>&1
grep -v amdgpu.ids $O/geo_bench.log | tail -12
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "closeup or config1 or config0" > $O/pytest_closeup.log 2>&1
tail -15 $O/pytest_closeup.log
