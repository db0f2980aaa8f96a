cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python scripts/dev_driver_e2e.py 48 2>&1 | grep -v "Processing\|Reconstructed\|Warning\|warn" > gpurun_out/r03n_e2e.log
tail -n 45 gpurun_out/r03n_e2e.log | cut -c1-200
