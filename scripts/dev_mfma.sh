set -x
R=/root/repo
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mfma_kt -- python $R/scripts/dev_mfma.py > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $R/gpurun_out/mfma_pmc -- python $R/scripts/dev_mfma.py > $R/gpurun_out/mfma_pmc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/mfma_pmc2 -- python $R/scripts/dev_mfma.py > $R/gpurun_out/mfma_pmc2.log 2>&1
cd $R
find gpurun_out/mfma_kt -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/mfma_kernel_trace.csv
python - <<'PY'
import csv, glob, collections
rows = list(csv.DictReader(open("gpurun_out/mfma_kernel_trace.csv")))
d = collections.defaultdict(list)
for r in rows:
    if "poseblend_mfma" in r["Kernel_Name"]:
        d[(r["Grid_Size_X"], r["Grid_Size_Y"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()): print("kernel_trace grid", k, "n", len(v), "mean ns", sum(v) / len(v), "min", min(v))
for pat in ("gpurun_out/mfma_pmc", "gpurun_out/mfma_pmc2"):
    for f in glob.glob(pat + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "poseblend_mfma" in r["Kernel_Name"]:
                acc[(r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()): print("pmc", k, "n", len(v), "mean", sum(v) / len(v))
PY
tail -3 gpurun_out/mfma_pmc.log gpurun_out/mfma_pmc2.log
