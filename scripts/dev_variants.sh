# bench value of library variants followmyhold_amd/libfoho_var*.so at 1 / 8 / 32 images (development aid)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for so in followmyhold_amd/libfoho_var*.so; do
  for n in 1 8 32; do
    s=2000; [ $n -gt 1 ] && s=200
    echo -n "$so n=$n: "
    FOHO_HIP_SO=$PWD/$so timeout 200 python bench.py --no-cpu-baseline --no-extras --images-per-gpu $n --steps $s 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k[2:]: round(v*1e3,1) for k,v in d['kernel_ms'].items()})"
  done
done
