"""Is a job's result a function of its inputs?  Same image set through (a) a fresh runner twice, (b) the same runner after
other image sets, (c) the exact-size driver twice; max |difference| of the final hand vertices, at the reference's learning
rates and at tame ones (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import engine as E, synthetic, inputs
rf = E.hip_render_fn("cuda")
mk = lambda kind, seed: synthetic.build_scene(rf, obj_kind=kind, H=64, W=64, seed=seed)
X = [mk("ico3", 1), mk("ico2", 2)]; Y = [mk("ico2", 3), mk("ico3", 4)]
def cfg(div, a, b, c):
    o = E.OptimizationConfig()
    o.optimization_steps_hand, o.optimization_steps_scale, o.optimization_steps_joint = a, b, c
    for n in ("phase1_hand_lrs", "phase2_hand_lrs", "obj_2half_lrs", "obj_lrs"):
        setattr(o, n, {k: v / div for k, v in getattr(o, n).items()})
    return o
d = lambda r, s: max(float(np.abs(a["hand"][0] - b["hand"][0]).max()) for a, b in zip(r, s))
for div in (1.0, 50.0):
    for sched in ((1, 0, 0), (2, 0, 0), (4, 0, 0), (4, 2, 2)):
        c = cfg(div, *sched)
        r1 = inputs.MeshGuidanceRunner(c, in_flight=2, grid_res=16); a = r1.run(X); a2 = r1.run(X); r1.run(Y); a3 = r1.run(X)
        r2 = inputs.MeshGuidanceRunner(c, in_flight=2, grid_res=16); b = r2.run(X)
        e = []
        for _ in range(2):
            gb = inputs.run_mesh_guidance([X[0]], c)
            m = gb.meta[0]; e.append(gb.region("world", torch.float32, (-1, 3))[:m["Vh"]].cpu().numpy())
        print(f"lr/{div:g} schedule {sched}: same runner again {d(a, a2):.2e}  after another set {d(a, a3):.2e}  fresh runner {d(a, b):.2e}  "
              f"exact driver twice {np.abs(e[0] - e[1]).max():.2e}  exact vs runner {np.abs(e[0] - a[0]['hand'][0]).max():.2e}", flush=True)
