# PMC passes over the batched step (8 images, one stream, no graph): where do k_pix_bwd / k_stage2 wait?
set -x
R=/root/repo
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph --images-per-gpu 8 --streams 1"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc8_a -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc8_b -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc8_c -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc8_d -- $CMD > /dev/null 2>&1
cd $R
python scripts/summarize_pmc.py $(find gpurun_out/pmc8_a gpurun_out/pmc8_b gpurun_out/pmc8_c gpurun_out/pmc8_d -name "*counter_collection.csv") > gpurun_out/pmc8_summary.csv
grep "k_pix_bwd\|k_stage2\|k_loss\|k_vert_bwd\|k_resolve" gpurun_out/pmc8_summary.csv
