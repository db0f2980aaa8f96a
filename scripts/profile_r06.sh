# Round-6 measurement set (run on the GPU box through gpurun): bench lines; rocprofv3 kernel stats + HBM traffic (two PMC passes each) + SQ
# counters of the one-image step and of ONE 8-image batch on one stream; kernel stats of the crop (close-up) regime, of the geometry
# decoder, of the ShapeVAE transformer (foho_vae_fwd / _bwd) and of ONE PIPELINE ITERATION (transformer -> decoder -> FlexiCubes -> step ->
# backward); matrix-core counters of the decoder and of the transformer.  Summaries land in gpurun_out/r06/ -- copy them to profiles/r06_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python bench.py > $O/bench_b1.json 2> $O/bench_b1.err
cp gpurun_out/bench_detail.json $O/bench_b1_detail.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_invocation.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 50 --images-per-gpu 8 > $O/bench_b8.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 50 --images-per-gpu 32 > $O/bench_b32.json 2>/dev/null
cd /tmp
B1="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
NG="python $R/bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --no-graph"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b1 -- $B1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b8 -- $B1 --images-per-gpu 8 --streams 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c1 -- python $R/scripts/run_steps.py --crop hoi --steps 300 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c8 -- python $R/scripts/run_steps.py --crop hoi --images 8 --streams 1 --steps 100 > /dev/null 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_geo -- python $R/scripts/dev/dev_geo_trace.py > $O/geo_trace_run.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_vae -- python $R/scripts/dev/vae_trace.py 1 12 > $O/vae_trace_run.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_pipe -- python $R/scripts/dev/dev_pipe_trace.py > $O/pipe_trace_run.log 2>&1
for t in b1 b8; do
  X=""; [ $t = b8 ] && X="--images-per-gpu 8 --streams 1"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/d_fetch_$t -- $NG $X > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/d_write_$t -- $NG $X > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/d_eard_$t -- $NG $X > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum --output-format csv -d $O/d_eawr_$t -- $NG $X > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/d_sqa_$t -- $NG $X > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $O/d_sqb_$t -- $NG $X > /dev/null 2>&1
done
# matrix-core busy cycles, VALU / LDS instruction mix, waits, HBM bytes: the geometry decoder and the VAE transformer
for t in geo vae; do
  GB="python $R/scripts/dev/dev_geo_trace.py"; [ $t = vae ] && GB="python $R/scripts/dev/vae_trace.py 1 5"
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/d_${t}1 -- $GB > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/d_${t}2 -- $GB > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/d_${t}3 -- $GB > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/d_${t}4 -- $GB > /dev/null 2>&1
done
cd $R
for t in b1 b8 c1 c8 geo vae pipe; do find $O/kt_$t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$t.csv; done
python scripts/dev/trace_summary.py $(find $O/kt_pipe -name "*kernel_trace.csv" | head -1) k_vae_rowstat > $O/pipeline_iteration_by_kernel.txt
python scripts/dev/trace_summary.py $(find $O/kt_vae -name "*kernel_trace.csv" | head -1) k_vae_rowstat > $O/vae_transformer_by_kernel.txt
for t in b1 b8; do
  python scripts/summarize_pmc.py $(find $O/d_fetch_$t $O/d_write_$t -name "*counter_collection.csv") > $O/pmc_fetch_write_$t.csv
  python scripts/summarize_pmc.py $(find $O/d_sqa_$t $O/d_sqb_$t -name "*counter_collection.csv") > $O/sq_counters_$t.csv
  python scripts/summarize_pmc.py $(find $O/d_eard_$t $O/d_eawr_$t -name "*counter_collection.csv") > $O/ea_requests_$t.csv
done
for t in geo vae; do python scripts/summarize_pmc.py $(find $O/d_${t}1 $O/d_${t}2 $O/d_${t}3 $O/d_${t}4 -name "*counter_collection.csv") > $O/${t}_counters.csv; done
rm -rf $O/kt_* $O/d_*
ls -la $O; head -9 $O/kernel_stats_b1.csv | cut -c1-110; cat $O/pipeline_iteration_by_kernel.txt | head -30
