R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
NG="python $R/bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --no-graph"
for t in b1 b8; do
  X=""; [ $t = b8 ] && X="--images-per-gpu 8 --streams 1"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/d_fetch_$t -- $NG $X > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/d_write_$t -- $NG $X > /dev/null 2>&1
done
cd $R
for t in b1 b8; do
  python scripts/summarize_pmc.py $(find $O/d_fetch_$t $O/d_write_$t -name "*counter_collection.csv") > $O/pmc_fetch_write_$t.csv
done
rm -rf $O/d_fetch_* $O/d_write_*
cat $O/pmc_fetch_write_b1.csv $O/pmc_fetch_write_b8.csv
