"""Step time of phases A (hand only), B (object only) and C (joint) on the bench scene, 50-iteration hipGraphs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
for phase in ("A", "B", "C"):
    gb = E.GuidanceBatch([sc])
    cfg, nr = E.phase_cfg(phase, denoise_i=19, do_update=True)
    gb.set_n_renders(nr)
    g = gb.capture(cfg, steps_per_graph=50)
    p0 = gb.params.clone()
    for _ in range(2): g.replay()
    ts = []
    for rep in range(5):
        gb.params.copy_(p0); gb.reset_optimizer()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 50)
    prof = {}
    for _ in range(10):
        for k, v in gb.step_profiled(cfg).items(): prof[k] = prof.get(k, 0) + v / 10
    print(f"phase {phase}: {min(ts)*1e6:6.1f} us/step   " + " ".join(f"{k[2:]} {v*1e3:.1f}" for k, v in prof.items()), flush=True)
