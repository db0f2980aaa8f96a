R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log
