#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04o
timeout 600 python -m pytest tests/test_geo_decode.py -x -q -m gpu 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04o/prof_fb -- python $R/scripts/geo_bench.py --fb > $R/gpurun_out/r04o/fb.log 2>&1
cd $R
grep "forward + backward" gpurun_out/r04o/fb.log
f=$(find gpurun_out/r04o/prof_fb -name "*kernel_stats.csv" | head -1); head -5 $f | cut -c1-150
find gpurun_out/r04o/prof_fb -name "*kernel_trace.csv" -delete
