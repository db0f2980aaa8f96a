"""Where the construction of a capacity-mode slot (4 images, 512 x 512) and of a target renderer goes (run on the GPU box)."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic
rf = E.hip_render_fn("cuda")
scs = [synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, seed=s) for s in range(4)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gb = E.GuidanceBatch(scs, n_renders=2, obj_capacity=(12288, 24576)); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    rd = E.TargetRenderer(512, 512, 512 * 512, 2 * 511 * 511); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"slot: {1e3*(t1-t0):.1f} ms host, {1e3*(t2-t0):.1f} ms with the device; renderer: {1e3*(t3-t2):.1f} / {1e3*(t4-t2):.1f} ms; workspace {gb.workspace.numel()/1e6:.0f} MB, renderer workspace {rd.gb.workspace.numel()/1e6:.0f} MB", flush=True)
    del gb, rd
pr = cProfile.Profile(); pr.enable()
gb = E.GuidanceBatch(scs, n_renders=2, obj_capacity=(12288, 24576)); torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
