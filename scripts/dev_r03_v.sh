#!/bin/bash
# VALU instructions of the raster role by phase (8 images, STAMPS build ablations under rocprofv3 --pmc)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03/roles
mkdir -p $O
export TMPDIR=/tmp
make -C $R/followmyhold_amd/csrc STAMPS=1 > /dev/null 2>&1
cd /tmp
for m in 61 0; do
  rm -rf $O/d_$m
  ROLE_MASK=$m timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/d_$m -- python $R/scripts/dev_role_valu.py > /dev/null 2>&1
  python - $m $(find $O/d_$m -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
m, path = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(path)) if r["Kernel_Name"].startswith("k_stage2")]
by = collections.defaultdict(list)
for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("mask", m, {k: round(sum(v[-20:]) / 20) for k, v in by.items()})
PY
  rm -rf $O/d_$m
done
