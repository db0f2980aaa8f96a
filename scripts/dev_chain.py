"""Timing of one iteration of the latent-side chain: SDF grid (res 64) -> FlexiCubes -> topology tables -> joint guidance
step at 512x512 -> dL/dSDF."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, ops, synthetic, facade
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
gb = E.GuidanceBatch([sc]); cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
res = 64
x = facade.FlexiCubes("cuda").construct_voxel_grid(res)[0] * 0.26
import numpy as np
r0 = float(np.linalg.norm(sc["obj_verts"], axis=1).mean())
s = (x.norm(dim=1) - r0 * (1.0 + 0.1 * torch.sin(40 * x[:, 0]) * torch.cos(33 * x[:, 1]))).requires_grad_(True)
def it():
    s.grad = None
    v, f, _ = ops.flexicubes(x, s, res)
    loss = gb.objective(v, f, cfg)
    loss.backward()
    return len(v), len(f), float(loss)
for _ in range(3): nv, nf, l = it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): it()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("chain iteration: %d verts / %d faces, loss %.3f: %.2f ms" % (nv, nf, l, dt * 1e3))
t0 = time.perf_counter()
for _ in range(20):
    v, f, _ = ops.flexicubes(x, s, res)
torch.cuda.synchronize(); print("  flexicubes fwd %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
for _ in range(20): gb.update_object(v.detach(), f)
torch.cuda.synchronize(); print("  update_object  %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
for _ in range(20): gb.step(cfg)
torch.cuda.synchronize(); print("  step (eager)   %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
