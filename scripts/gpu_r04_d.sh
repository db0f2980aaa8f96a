R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
timeout 600 python scripts/diag_closeup_terms.py 9 2>&1 | grep -v amdgpu.ids > $O/diag_terms.log
cat $O/diag_terms.log
timeout 900 python -m pytest tests/test_geo_decode.py -m gpu -q -x > $O/pytest_geo.log 2>&1
tail -5 $O/pytest_geo.log
timeout 600 python scripts/geo_bench.py --parts > $O/geo_bench.log 2>&1
grep -v amdgpu.ids $O/geo_bench.log | tail -12
