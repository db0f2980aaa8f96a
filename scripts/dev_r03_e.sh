#!/bin/bash
# round 3, pass e: 8 KB of LDS per k_stage2 workgroup (staged nearest-neighbour role, 1024-entry raster queue); stream / queue settings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_edge_gpu.py tests/test_fullsize_gpu.py -m gpu -q --maxfail=12 > gpurun_out/r03e_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03e_tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03e_bench.json 2> gpurun_out/r03e_bench.err
timeout 300 python scripts/dev_spans_batch.py 8 > gpurun_out/r03e_spans8.log 2>&1
B="python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 50"
for q in 4 8; do for st in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 $B --images-per-gpu 8 --streams $st 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q ipg 8 streams $st', round(o['value']))" >> gpurun_out/r03e_streams.log
done; done
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 $B --images-per-gpu 16 --streams 8 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q ipg 16 streams 8', round(o['value']))" >> gpurun_out/r03e_streams.log
  GPU_MAX_HW_QUEUES=$q timeout 300 $B --images-per-gpu 16 --streams 4 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q ipg 16 streams 4', round(o['value']))" >> gpurun_out/r03e_streams.log
done
timeout 300 python - > gpurun_out/r03e_jobseeds.log 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from followmyhold_amd import engine as E, synthetic, inputs
rf = E.hip_render_fn("cuda")
for base in (0, 200):
    scs = [synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, seed=base + s) for s in range(8)]
    r = inputs.MeshGuidanceRunner(in_flight=8)
    r.run(scs); torch.cuda.synchronize()
    for rep in range(3):
        todo = [scs[j % 8] for j in range(16)]
        torch.cuda.synchronize(); t0 = time.perf_counter(); res = r.run(todo); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("seeds", base, "rep", rep, "%.1f ms per image" % (dt * 1e3 / 16), "flags", sorted(set(x["flags"] for x in res)), flush=True)
PY
tail -n 3 gpurun_out/r03e_tests.log; cat gpurun_out/r03e_streams.log
