# Round-2 measurement set (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats, HBM traffic (two PMC
# passes) and SQ counters (two PMC passes) of the one-image step; 8-image variants of the stats and the SQ counters.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python bench.py > $O/bench_b1.json 2> $O/bench_b1.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --obj 40k > $O/bench_40k.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 50 --images-per-gpu 8 > $O/bench_b8.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 50 --images-per-gpu 32 > $O/bench_b32.json 2>/dev/null
cd /tmp
B1="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
NG="python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-graph"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b1 -- $B1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b8 -- $B1 --images-per-gpu 8 --streams 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $NG > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $NG > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq_a_b1 -- $NG > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $O/sq_b_b1 -- $NG > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq_a_b8 -- $NG --images-per-gpu 8 --streams 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $O/sq_b_b8 -- $NG --images-per-gpu 8 --streams 1 > /dev/null 2>&1
cd $R
for t in b1 b8; do find $O/kt_$t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$t.csv; done
python scripts/summarize_pmc.py $(find $O/pmc_fetch $O/pmc_write -name "*counter_collection.csv") > $O/pmc_fetch_write_b1.csv
python scripts/summarize_pmc.py $(find $O/sq_a_b1 $O/sq_b_b1 -name "*counter_collection.csv") > $O/sq_counters_b1.csv
python scripts/summarize_pmc.py $(find $O/sq_a_b8 $O/sq_b_b8 -name "*counter_collection.csv") > $O/sq_counters_b8.csv
rm -rf $O/kt_b1 $O/kt_b8 $O/pmc_fetch $O/pmc_write $O/sq_a_b1 $O/sq_b_b1 $O/sq_a_b8 $O/sq_b_b8
ls -la $O; tail -c 600 $O/bench_b1.json
