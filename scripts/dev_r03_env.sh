#!/bin/bash
# HIP runtime knobs against the hipGraph replay of the step (one image: the gaps between the six nodes; 16 images: four streams)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/r03_env.log
rm -f $LOG
B="python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 50"
run() { # label, env...
  label=$1; shift
  for ipg in 1 16; do
    env "$@" timeout 200 $B --images-per-gpu $ipg 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label ipg $ipg', round(o['value']))" >> $LOG 2>&1 || echo "$label ipg $ipg FAILED" >> $LOG
  done
}
run base X=1
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run optflush0 AMD_OPT_FLUSH=0
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run sysscope0 ROC_SYSTEM_SCOPE_SIGNAL=0
run graphbatch1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run graphbatch1000 DEBUG_HIP_GRAPH_BATCH_SIZE=1000
run kernargopt0 DEBUG_HIP_KERNARG_COPY_OPT=0
run fgs1 ROC_USE_FGS_KERNARG=1
run fgs0 ROC_USE_FGS_KERNARG=0
run noscratchreclaim HSA_NO_SCRATCH_RECLAIM=1
run activewait ROC_ACTIVE_WAIT_TIMEOUT=1000
run base2 X=1
cat $LOG
