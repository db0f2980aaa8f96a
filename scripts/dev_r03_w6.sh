#!/bin/bash
# k_stage2 at 6 waves per SIMD (80 VGPRs, no scratch) against the shipped 7 (72 VGPRs, 28 bytes of scratch per lane whose dirty lines
# are written back at the end of the launch): throughput and memory-side write requests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r03w6
rm -rf $O; mkdir -p $O
LOG=$O/w6.log
B="python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 50"
run() { label=$1; shift
  for ipg in 1 8 16 32; do
    env "$@" timeout 200 $B --images-per-gpu $ipg 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label ipg $ipg', round(o['value']))" >> $LOG 2>&1 || echo "$label ipg $ipg FAILED" >> $LOG
  done
}
for rep in 1 2; do
run w7 X=1
run w6 FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_w6.so
done
export TMPDIR=/tmp
cd /tmp
WR="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum"
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
NG="python $R/bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --no-graph"
for v in w7 w6; do
  SO=$R/followmyhold_amd/libfoho_hip.so; [ $v = w6 ] && SO=$R/followmyhold_amd/libfoho_hip_w6.so
  for i in 1 2; do
    FOHO_HIP_SO=$SO timeout 300 rocprofv3 --pmc $WR --output-format csv -d $O/d_$v$i -- $NG > /dev/null 2>&1
    (cd $R; echo "$v wr $i" >> $LOG; python scripts/summarize_pmc.py $(find $O/d_$v$i -name "*counter_collection.csv") | grep k_stage2 >> $LOG)
  done
  FOHO_HIP_SO=$SO timeout 300 rocprofv3 --pmc $RD --output-format csv -d $O/d_rd_$v -- $NG > /dev/null 2>&1
  (cd $R; echo "$v rd" >> $LOG; python scripts/summarize_pmc.py $(find $O/d_rd_$v -name "*counter_collection.csv") | grep k_stage2 >> $LOG)
done
rm -rf $O/d_*
cat $LOG
