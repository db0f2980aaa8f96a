# repeated batch bench in fresh processes: counts runs that end with a raised flag (development aid)
cd ${GRAFT_REPO_ROOT:-/root/repo}
so=${1:-hip}; n=${2:-25}; steps=${3:-300}
for rep in $(seq 1 $n); do
  FOHO_HIP_SO=$PWD/followmyhold_amd/libfoho_$so.so timeout 120 python bench.py --no-cpu-baseline --no-extras --images-per-gpu 8 --steps $steps 2>&1 | tail -1 | cut -c1-60
done | cut -c1-40 | sort | uniq -c
