# default bench (two streams) of library variants at 8 / 32 images, twice each (development aid)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for so in "$@"; do
  for n in 8 32; do
    for rep in 1 2; do
    echo -n "$so n=$n: "
    FOHO_HIP_SO=$PWD/$so timeout 120 python bench.py --no-cpu-baseline --no-extras --images-per-gpu $n --steps 300 2>&1 | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t); print(round(d['value']), {k[2:]: round(v*1e3,1) for k,v in d['kernel_ms'].items()})
except Exception as e: print('ERR', t[-300:])"
    done
  done
done
