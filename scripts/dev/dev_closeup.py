"""Close-up frames (narrow field of view: hand and object fill the crop, like a cropped hand-object image) at 32 images in flight:
phase C with k_resolve dense (listed_cap = -1) and listed (default from eight images per launch on).  Usage: dev_closeup.py [fov ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic

rf = E.hip_render_fn("cuda")
ident = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0] * 2, dtype=torch.float32, device="cuda")
for fov in [float(a) for a in sys.argv[1:]] or [60.0, 30.0, 22.0]:
    scenes = [synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, fov=fov, seed=100 + j) for j in range(32)]
    for mode in (-1, 0):
        group = E.GuidanceGroup(scenes, 4, device="cuda")
        cfg, nr = E.phase_cfg("C", denoise_i=19, do_update=True)
        cfg.listed_cap = mode
        group.capture(cfg, steps_per_graph=50)
        ts = []
        for rep in range(5):
            group.restart(ident)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            group.run(cfg, 100)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 100)
        gb = group.batches[0]
        gb.raise_on_flags()
        P = 512 * 512
        p2f = gb.region("p2f", torch.int32, (2, gb.B, P))[1, 0]
        hit = (p2f >= 0).reshape(64, 8, 16, 32).any(3).any(1)
        act = int(gb.region_raw("act_count")[0]) if hasattr(gb, "region_raw") else -1
        print(f"fov {fov:4.0f}: {str(mode):6s} {32 / min(ts[1:]) / 1e3:6.1f} k steps/s; hit pixels {int((p2f >= 0).sum())}, hit tiles {int(hit.sum())} of 1024", flush=True)
        del group
