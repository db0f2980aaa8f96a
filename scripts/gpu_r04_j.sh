# Round-4, GPU call J: 256 x 256 GEMM tiles; hand_faces_per_block / gtiles defaults; facade silhouette backward; close-up test.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
for rep in 1 2; do timeout 900 python -m pytest tests/test_geo_decode.py -m gpu -q -x 2>&1 | tail -3; done > $O/pytest_geo.log 2>&1
cat $O/pytest_geo.log
timeout 600 python scripts/geo_bench.py --parts > $O/geo_bench.log 2>&1
grep -v amdgpu.ids $O/geo_bench.log | tail -8
timeout 600 python -m pytest tests/test_facade_gpu.py -m gpu -q -x 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_step_gpu.py tests/test_edge_gpu.py -m gpu -q -x > $O/pytest_step.log 2>&1
tail -8 $O/pytest_step.log
for a in "" "--crop hoi" "--obj 40k"; do timeout 200 python scripts/run_steps.py $a --steps 500 2>&1 | grep "steps/s"; done
timeout 200 python scripts/run_steps.py --crop hoi --images 32 --streams 4 --steps 200 2>&1 | grep "steps/s"
