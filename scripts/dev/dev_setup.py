"""Per-image setup cost: GuidanceBatch construction (topology tables) and the target render of a MoGe-sized mesh."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from followmyhold_amd import engine as E, synthetic
render = E.hip_render_fn("cuda")
sc = synthetic.build_scene(render, obj_kind="20k", H=512, W=512, seed=0)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); gb = E.GuidanceBatch([sc]); torch.cuda.synchronize()
    print("GuidanceBatch(1 image, 22k faces): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
from test_inputs import _image_mesh
v, f = _image_mesh(512)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); render(v, f, 512, 512, 60.0); torch.cuda.synchronize()
    print("target render of a %d-face image mesh: %.1f ms" % (len(f), (time.perf_counter() - t0) * 1e3))
cfg, _ = E.phase_cfg("C")
g = gb.capture(cfg, steps_per_graph=50)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(15): g.replay()
torch.cuda.synchronize(); print("750 iterations: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for spg in (1, 10, 50, 200):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g = gb.capture(cfg, steps_per_graph=spg); torch.cuda.synchronize()
    t1 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); t2 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print("capture of %d iterations: %.1f ms; first replay %.1f ms, second %.1f ms" % (spg, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
from followmyhold_amd import inputs
torch.cuda.synchronize(); t0 = time.perf_counter()
inputs.run_mesh_guidance([sc]); torch.cuda.synchronize()
print("run_mesh_guidance (750 iterations, 11 phase loops): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
