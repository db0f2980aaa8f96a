#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python - > gpurun_out/r03g_jobseeds.log 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from followmyhold_amd import engine as E, synthetic, inputs
rf = E.hip_render_fn("cuda")
for base in (0, 200, 0):
  for nf in (8, 16):
    scs = [synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, seed=base + s) for s in range(nf)]
    r = inputs.MeshGuidanceRunner(in_flight=nf)
    r.run(scs); torch.cuda.synchronize()
    for rep in range(2):
        todo = [scs[j % nf] for j in range(2 * nf)]
        torch.cuda.synchronize(); t0 = time.perf_counter(); res = r.run(todo); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("seeds", base, "in flight", nf, "rep", rep, "%.2f ms per image" % (dt * 1e3 / len(todo)), "flags", sorted(set(x["flags"] for x in res)), flush=True)
PY
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 50 --images-per-gpu 16 > gpurun_out/r03g_b16.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 50 --images-per-gpu 32 > gpurun_out/r03g_b32.json 2>/dev/null
grep -v Warn gpurun_out/r03g_jobseeds.log | tail -12
