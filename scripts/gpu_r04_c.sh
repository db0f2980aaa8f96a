# Round-4, GPU call C: geometry decoder after the softmax / GELU diet; the scatter's deep-inside shortcut (all raster parity tests + close-up numbers).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_geo_decode.py -m gpu -q -x > $O/pytest_geo.log 2>&1
tail -5 $O/pytest_geo.log
timeout 600 python scripts/geo_bench.py --parts > $O/geo_bench.log 2>&1
grep -v amdgpu.ids $O/geo_bench.log | tail -12
timeout 1800 python -m pytest tests -m gpu -q -x --deselect tests/test_geo_decode.py > $O/pytest_gpu.log 2>&1
tail -8 $O/pytest_gpu.log
timeout 300 python scripts/run_steps.py --crop hoi > $O/closeup_b1.log 2>&1
timeout 300 python scripts/run_steps.py --crop hoi --images 32 --streams 4 --steps 200 > $O/closeup_b32.log 2>&1
timeout 300 python scripts/run_steps.py > $O/bench_scene_b1.log 2>&1
timeout 300 python scripts/run_steps.py --images 32 --streams 4 --steps 200 > $O/bench_scene_b32.log 2>&1
grep -h "steps/s" $O/closeup_b1.log $O/closeup_b32.log $O/bench_scene_b1.log $O/bench_scene_b32.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c1 -- python $R/scripts/run_steps.py --crop hoi --steps 200 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c8 -- python $R/scripts/run_steps.py --crop hoi --images 8 --streams 1 --steps 100 > /dev/null 2>&1
cd $R
for t in c1 c8; do find $O/kt_$t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_closeup_$t.csv; done
rm -rf $O/kt_c1 $O/kt_c8
head -8 $O/kernel_stats_closeup_c1.csv | cut -c1-110; head -10 $O/kernel_stats_closeup_c8.csv | cut -c1-110
