"""Where the per-image job's wall time goes (inputs.run_mesh_guidance; run on the GPU box)."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic, inputs
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
inputs.run_mesh_guidance([sc]); torch.cuda.synchronize()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); inputs.run_mesh_guidance([sc]); torch.cuda.synchronize()
    print("run_mesh_guidance: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); inputs.run_mesh_guidance([sc]); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
