R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_geo_decode.py -m gpu -q -x 2>&1 | tail -3
for v in 0 1; do echo "attention variant $v (0: max folded into the accumulator init, 1: scalar max)"; GEO_ATTN_VARIANT=$v timeout 600 python scripts/geo_bench.py --parts 2>&1 | grep "attention\|fwd_ms" | cut -c1-200; done
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k closeup 2>&1 | tail -3
