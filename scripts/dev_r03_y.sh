#!/bin/bash
# VALU instructions of k_pix_bwd / k_vert_bwd with the development ablations (8 images, STAMPS build under rocprofv3 --pmc)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03/roles
mkdir -p $O
export TMPDIR=/tmp
make -C $R/followmyhold_amd/csrc STAMPS=1 > /dev/null 2>&1
cd /tmp
for m in "$@"; do
  rm -rf $O/d_$m
  ROLE_MASK=$m timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/d_$m -- python $R/scripts/dev_role_valu.py > /dev/null 2>&1
  python - $m $(find $O/d_$m -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
m, path = sys.argv[1], sys.argv[2]
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    by[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("k_xform", "k_stage2", "k_resolve", "k_loss<128>", "k_pix_bwd", "k_vert_bwd<128>"):
    if k in by: print("mask", m, k, {c: round(sum(v[-20:]) / 20) for c, v in by[k].items()})
PY
  rm -rf $O/d_$m
done
