# bench value + per-kernel times of library variants (FOHO_HIP_SO hook) at 1 / 8 images (development aid)
# usage: bash scripts/dev_variants.sh hip p32 ...   (followmyhold_amd/libfoho_<name>.so)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for so in "$@"; do
  for n in 1 8; do
    s=2000; [ $n -gt 1 ] && s=300
    for rep in 1 2; do
    echo -n "$so n=$n: "
    FOHO_HIP_SO=$PWD/followmyhold_amd/libfoho_$so.so timeout 120 python bench.py --no-cpu-baseline --no-extras --images-per-gpu $n --steps $s 2>&1 | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t); print(round(d['value']), {k[2:]: round(v*1e3,1) for k,v in d['kernel_ms'].items()})
except Exception as e: print('ERR', t[-300:])"
    done
  done
done
