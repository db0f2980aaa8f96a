R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_fullsize_gpu.py tests/test_edge_gpu.py tests/test_step_gpu.py -m gpu -q -x 2>&1 | tail -4
for a in "--crop hoi --images 32 --streams 4 --steps 200" "--fov 22 --images 32 --streams 4 --steps 200" "--images 32 --streams 4 --steps 200" "--images 8 --streams 4 --steps 200" "--crop hoi --images 8 --streams 4 --steps 200"; do python scripts/run_steps.py $a 2>&1 | grep steps/s; done
python scripts/dev_closeup.py 22 2>&1 | grep fov
