# Round-3 measurement set (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats, HBM traffic (two PMC
# passes each) of the one-image step and of ONE 8-image batch on one stream, SQ counters of both.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python bench.py > $O/bench_b1.json 2> $O/bench_b1.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 50 --images-per-gpu 8 > $O/bench_b8.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 50 --images-per-gpu 16 > $O/bench_b16.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 50 --images-per-gpu 32 > $O/bench_b32.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extras --obj 40k > $O/bench_40k.json 2>/dev/null
cd /tmp
B1="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
NG="python $R/bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --no-graph"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b1 -- $B1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b8 -- $B1 --images-per-gpu 8 --streams 1 > /dev/null 2>&1
for t in b1 b8; do
  X=""; [ $t = b8 ] && X="--images-per-gpu 8 --streams 1"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/d_fetch_$t -- $NG $X > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/d_write_$t -- $NG $X > /dev/null 2>&1
  # the L2's memory-side requests by size class (exact bytes: 32 / 64 / 128-byte reads, 64 / 32-byte writes, atomics = 32-byte writes)
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/d_eard_$t -- $NG $X > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum --output-format csv -d $O/d_eawr_$t -- $NG $X > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/d_sqa_$t -- $NG $X > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $O/d_sqb_$t -- $NG $X > /dev/null 2>&1
done
cd $R
for t in b1 b8; do
  find $O/kt_$t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$t.csv
  python scripts/summarize_pmc.py $(find $O/d_fetch_$t $O/d_write_$t -name "*counter_collection.csv") > $O/pmc_fetch_write_$t.csv
  python scripts/summarize_pmc.py $(find $O/d_sqa_$t $O/d_sqb_$t -name "*counter_collection.csv") > $O/sq_counters_$t.csv
  python scripts/summarize_pmc.py $(find $O/d_eard_$t $O/d_eawr_$t -name "*counter_collection.csv") > $O/ea_requests_$t.csv
done
# calibration of those counters on known byte counts (scripts/micro/ea_calib.hip: 1 GiB read in three shapes, 4 M 12-byte gathers,
# 1 GiB written, 4 M 64-bit atomics)
(cd scripts/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 ea_calib.hip -o ea_calib 2>/dev/null)
cd /tmp
for p in "rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum" "fs FETCH_SIZE" "ws WRITE_SIZE"; do
  set -- $p; tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/d_cal_$tag -- $R/scripts/micro/ea_calib > $O/ea_calib.log 2>&1
done
cd $R
python scripts/summarize_pmc.py $(find $O/d_cal_rd $O/d_cal_wr $O/d_cal_fs $O/d_cal_ws -name "*counter_collection.csv") | grep -v fillBuffer > $O/ea_calibration.csv
rm -rf $O/kt_b1 $O/kt_b8 $O/d_fetch_* $O/d_write_* $O/d_sqa_* $O/d_sqb_* $O/d_eard_* $O/d_eawr_* $O/d_cal_*
ls -la $O; cat $O/pmc_fetch_write_b1.csv; cat $O/pmc_fetch_write_b8.csv; head -8 $O/kernel_stats_b1.csv | cut -c1-120; head -8 $O/kernel_stats_b8.csv | cut -c1-120
