#!/bin/bash
# listed k_resolve (k_tile_list + a quarter of the tile workgroups + k_resolve_ovf; default from four images per launch on) against
# the dense launch (FOHO_LISTED_CAP=0), same build: batched steps and the per-phase cost of a job
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
LOG=gpurun_out/r03_list.log
rm -f $LOG
B="python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 50"
run() { label=$1; shift
  for ipg in 16 32; do
    env "$@" timeout 200 $B --images-per-gpu $ipg 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label ipg $ipg', round(o['value']))" >> $LOG 2>&1 || echo "$label ipg $ipg FAILED" >> $LOG
  done
}
for rep in 1 2 3; do
run dense FOHO_LISTED_CAP=0
run default X=1
done
echo dense >> $LOG; FOHO_LISTED_CAP=0 timeout 300 python scripts/dev_phases_batch.py 16 32 2>&1 | grep images >> $LOG
echo default >> $LOG; timeout 300 python scripts/dev_phases_batch.py 16 32 2>&1 | grep images >> $LOG
cat $LOG
