#!/bin/bash
# nearest-neighbour pruning: parity tests, bench lines, VALU count of the role
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r03
mkdir -p $O/roles
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py tests/test_edge_gpu.py -m gpu -q -x > $O/t_tests.log 2>&1; tail -n 5 $O/t_tests.log
timeout 600 python bench.py --no-cpu-baseline > $O/t_bench.json 2> $O/t_bench.err
python - <<'PY'
import json
o=json.loads(open('gpurun_out/r03/t_bench.json').read().strip().splitlines()[-1])
print('b1', round(o['value']), o['kernel_ms'], 'batched', round(o['batched']['value']), 'job', {k: round(v['images_per_s'],1) for k, v in o['job'].items() if isinstance(v, dict)})
PY
for cfg in "8 1" "16 4" "32 4"; do set -- $cfg
timeout 300 python bench.py --steps 200 --warmup 20 --images-per-gpu $1 --streams $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('images $1 streams $2:', round(o['value']), 'steps/s', o['kernel_ms'].get('k_stage2'))"
done
export TMPDIR=/tmp
make -C $R/followmyhold_amd/csrc STAMPS=1 > /dev/null 2>&1
cd /tmp
for m in 0 59; do
  rm -rf $O/roles/d_$m
  ROLE_MASK=$m timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/roles/d_$m -- python $R/scripts/dev_role_valu.py > /dev/null 2>&1
  python - $m $(find $O/roles/d_$m -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
m, path = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(path)) if r["Kernel_Name"].startswith("k_stage2")]
by = collections.defaultdict(list)
for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("mask", m, {k: round(sum(v[-20:]) / 20) for k, v in by.items()})
PY
  rm -rf $O/roles/d_$m
done
