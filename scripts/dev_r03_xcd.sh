#!/bin/bash
# XCD-aware block order inside k_stage2's roles against the round-robin order (libfoho_hip_base.so): tests, throughput, memory-side requests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r03xcd
rm -rf $O; mkdir -p $O
LOG=$O/xcd.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 >> $LOG
B="python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 50"
run() { label=$1; shift
  for ipg in 1 8 16 32; do
    env "$@" timeout 200 $B --images-per-gpu $ipg 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label ipg $ipg', round(o['value']), o['kernel_ms']['k_stage2'])" >> $LOG 2>&1 || echo "$label ipg $ipg FAILED" >> $LOG
  done
}
for rep in 1 2; do
run base FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_base.so
run xcd X=1
done
export TMPDIR=/tmp
cd /tmp
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
WR="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum"
NG="python $R/bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --no-graph"
for v in base xcd; do
  SO=$R/followmyhold_amd/libfoho_hip.so; [ $v = base ] && SO=$R/followmyhold_amd/libfoho_hip_base.so
  for t in b1 b8; do
    X=""; [ $t = b8 ] && X="--images-per-gpu 8 --streams 1"
    FOHO_HIP_SO=$SO timeout 300 rocprofv3 --pmc $RD --output-format csv -d $O/d_rd_$v$t -- $NG $X > /dev/null 2>&1
    FOHO_HIP_SO=$SO timeout 300 rocprofv3 --pmc $WR --output-format csv -d $O/d_wr_$v$t -- $NG $X > /dev/null 2>&1
    (cd $R; echo "$v $t" >> $LOG; python scripts/summarize_pmc.py $(find $O/d_rd_$v$t $O/d_wr_$v$t -name "*counter_collection.csv") | grep -E "k_stage2|k_resolve|k_pix_bwd|k_vert_bwd" | grep -E "RDREQ_128B|RDREQ_64B|WRREQ_sum|WRREQ_64B" >> $LOG)
  done
done
rm -rf $O/d_*
cat $LOG
