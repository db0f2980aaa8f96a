"""Where the per-image job's wall time goes (run on the GPU box): the exact-size single-image driver
(inputs.run_mesh_guidance), the slot runner the product entry point uses (inputs.MeshGuidanceRunner) at 1 and 8 images in
flight, and the per-iteration time of each phase inside its 10-iteration graphs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic, inputs
rf = E.hip_render_fn("cuda")
scs = [synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, seed=s) for s in range(8)]
sc = scs[0]
inputs.run_mesh_guidance([sc]); torch.cuda.synchronize()
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); inputs.run_mesh_guidance([sc]); torch.cuda.synchronize()
    print("run_mesh_guidance (exact size, captures per call): %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for in_flight, n in ((1, 4), (8, 16)):
    r = inputs.MeshGuidanceRunner(in_flight=in_flight)
    r.run(scs[:in_flight]); torch.cuda.synchronize()
    for _ in range(2):
        todo = [scs[j % 8] for j in range(n)]
        torch.cuda.synchronize(); t0 = time.perf_counter(); res = r.run(todo); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"MeshGuidanceRunner in_flight={in_flight}: {dt*1e3/n:.1f} ms per image ({n/dt:.1f} images/s), stats {r.stats}, ok {sum(x['ok'] for x in res)}", flush=True)
# per-iteration time of each phase on the slot (one image), graphs of 10 iterations
r = inputs.MeshGuidanceRunner(in_flight=1)
r.run([sc]); slot = r.slots[0]; gb = slot.gb
for phase, iters, di in (("A", 200, 9), ("B", 100, 10), ("C", 50, 15), ("C", 50, 19)):
    cfg, _ = E.phase_cfg(phase, r.config, denoise_i=di, do_update=True)
    gr = r._graph_for(slot, cfg, 10)
    ts = []
    for rep in range(5):
        gb.reset_optimizer(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters // 10): gr.replay()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / iters)
    print(f"phase {phase} (denoise {di}): {min(ts)*1e6:.1f} us per iteration", flush=True)
