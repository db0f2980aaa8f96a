# Round-4, GPU call A: MFMA layout probe, the GPU test suite, baseline numbers of the close-up (crop) regime with kernel traces.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04a
mkdir -p $O
export TMPDIR=/tmp
cd $R
./scripts/micro/mfma_probe > $O/mfma_probe.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
timeout 300 python scripts/run_steps.py --crop hoi > $O/closeup_b1.log 2>&1
timeout 300 python scripts/run_steps.py --crop hoi --images 32 --streams 4 --steps 200 > $O/closeup_b32.log 2>&1
timeout 300 python scripts/run_steps.py > $O/bench_scene_b1.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c1 -- python $R/scripts/run_steps.py --crop hoi --steps 200 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c8 -- python $R/scripts/run_steps.py --crop hoi --images 8 --streams 1 --steps 100 > /dev/null 2>&1
cd $R
for t in c1 c8; do find $O/kt_$t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_closeup_$t.csv; done
rm -rf $O/kt_c1 $O/kt_c8
cat $O/mfma_probe.log $O/closeup_b1.log $O/closeup_b32.log $O/bench_scene_b1.log
head -12 $O/kernel_stats_closeup_c1.csv | cut -c1-110; head -12 $O/kernel_stats_closeup_c8.csv | cut -c1-110
