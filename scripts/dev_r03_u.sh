#!/bin/bash
# job rate at 32 images in flight; end-to-end driver at 16 / 32 in flight, 192 folders
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
python - <<'PY'
import time, torch, sys
sys.path.insert(0, '.')
from followmyhold_amd import engine as E, synthetic, inputs
rf = E.hip_render_fn("cuda")
scs = [synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, seed=200 + s) for s in range(16)]
for in_flight, n in ((16, 64), (32, 128), (32, 128)):
    r = inputs.MeshGuidanceRunner(in_flight=in_flight)
    r.run(scs[:16] * (in_flight // 16)); torch.cuda.synchronize()
    for rep in range(2):
        todo = [scs[j % 16] for j in range(n)]
        torch.cuda.synchronize(); t0 = time.perf_counter(); res = r.run(todo); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"in_flight {in_flight}: {n} images, {dt*1e3/n:.2f} ms per image, {n/dt:.1f} images/s, per_slot {r.per_slot}, streams {r.n_streams}, ok {sum(x['ok'] for x in res)}", flush=True)
PY
