"""Experiment: iterations per hipGraph (1, 5, 10, 25, 50) for the one-image loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
ident = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0] * 2, dtype=torch.float32, device="cuda")
for spg in [1, 5, 10, 25, 50, 1]:
    gb = E.GuidanceBatch([sc])
    g = gb.capture(cfg, steps_per_graph=spg)
    def run(n):
        for i in range(0, n, spg):
            if i % 50 == 0:
                gb.params.copy_(ident.expand_as(gb.params)); gb.reset_optimizer()
            g.replay()
    run(100); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(400); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("steps per graph %2d: %.0f steps/s (%.1f us/step)" % (spg, 400 / dt, dt / 400 * 1e6), flush=True)
