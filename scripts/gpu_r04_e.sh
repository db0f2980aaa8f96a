# Round-4, GPU call E: where the time goes -- SQ counters of the geometry decoder's kernels; in-kernel stamps of the close-up step.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04e
mkdir -p $O
export TMPDIR=/tmp
cd $R
make -C followmyhold_amd/csrc STAMPS=1 -s 2>&1 | grep -E "error" 
CROP=hoi timeout 300 python scripts/dev_stamps.py 2>&1 | grep -v amdgpu.ids > $O/stamps_closeup.log
CROP=hoi timeout 300 python scripts/dev_spans.py 2>&1 | grep -v amdgpu.ids > $O/spans_closeup.log
tail -22 $O/stamps_closeup.log; grep -A12 "stage2" $O/spans_closeup.log | head -40; tail -4 $O/spans_closeup.log
cd /tmp
GB="python $R/scripts/geo_bench.py --parts"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_geo -- $GB > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/d_sq1 -- $GB > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/d_sq2 -- $GB > /dev/null 2>&1
cd $R
find $O/kt_geo -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_geo.csv
python - <<'PY' > $O/sq_geo.csv
import collections, csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04e"
agg = collections.defaultdict(list)
for path in glob.glob(O + "/d_sq*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        agg[(row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])].append(float(row["Counter_Value"]))
print("kernel,counter,launches,mean_per_launch")
for (k, cn), v in sorted(agg.items()):
    if "geo" in k:
        print(f"{k},{cn},{len(v)},{sum(v) / len(v):.1f}")
PY
rm -rf $O/kt_geo $O/d_sq1 $O/d_sq2
grep geo $O/kernel_stats_geo.csv | cut -c1-120; cat $O/sq_geo.csv
