"""Per-kernel event timings of the joint step on close-up frames (fov 22: hand and object fill the crop) at 1 and 8 images per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic
rf = E.hip_render_fn("cuda")
for fov in (60.0, 22.0):
    for B in (1, 8):
        scenes = [synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, fov=fov, seed=100 + j) for j in range(B)]
        gb = E.GuidanceBatch(scenes)
        cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
        for _ in range(5): gb.step(cfg)
        acc = {}
        for _ in range(30):
            for k, v in gb.step_profiled(cfg).items(): acc[k] = acc.get(k, 0) + v / 30
        tot = sum(acc.values())
        print(f"fov {fov:.0f} B={B}: " + " ".join(f"{k[2:]} {v*1e3:.1f}" for k, v in acc.items()) + f" | sum {tot*1e3:.1f} us", flush=True)
