#!/bin/bash
for v in "" _wpe6 _wpe5 _wpe4; do
  export FOHO_HIP_SO=$GRAFT_REPO_ROOT/followmyhold_amd/libfoho_hip$v.so
  echo "== lib$v"
  for a in "--images 1 --streams 1" "--images 32 --streams 4" "--crop hoi --images 1 --streams 1" "--crop hoi --images 32 --streams 4"; do
    timeout 300 python scripts/run_steps.py $a --steps 2000 2>&1 | tail -1
  done
done
