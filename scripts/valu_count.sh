# Vector instructions per launch of every kernel of the step (rocprofv3 --pmc SQ_INSTS_VALU, eager launches):
#   bash scripts/valu_count.sh [run_steps.py arguments, default: --images 8 --streams 1; e.g. --crop hoi --images 8]
# the yardstick of the batch regime's instruction diets (NOTEBOOK.md).  Run on the GPU box (through gpurun).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/valu_count
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
ARGS="${@:---images 8 --streams 1}"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $O/pmc -- python $R/scripts/run_steps.py --eager --steps 30 $ARGS > /dev/null 2>&1
cd $R
python scripts/summarize_pmc.py $(find $O/pmc -name "*counter_collection.csv") | tee $O/summary.csv | awk -F, 'NR>1 && ($1=="k_xform"||$1=="k_stage2"||$1=="k_tile_list"||$1=="k_resolve"||$1=="k_resolve_ovf"||$1=="k_loss"||$1=="k_pix_bwd"||$1=="k_vert_bwd") {printf "%-16s %8d launches %12.0f\n", $1, $3, $4; s += $4} END {printf "step total %.0f\n", s}'
find $O/pmc -name "*counter_collection.csv" -delete
