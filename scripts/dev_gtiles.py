"""Sweep of k_pix_bwd's grid shape (STAMPS build: FOHO_DEBUG_GTILES / FOHO_DEBUG_GFRAC) on the bench scene."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import _lib as L_
L_.SO_PATH = os.path.join(ROOT, "followmyhold_amd", "libfoho_hip_stamps.so")
from followmyhold_amd import engine as E, synthetic
NB = int(os.environ.get("NB", "1"))
scs = [synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=s) for s in range(NB)]
for gt in (64, 128, 256, 512, 1024):
    for gf in (8, 64):
        os.environ["FOHO_DEBUG_GTILES"], os.environ["FOHO_DEBUG_GFRAC"] = str(gt), str(gf)
        gb = E.GuidanceBatch(scs)
        cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
        g = gb.capture(cfg, steps_per_graph=50)
        for _ in range(2): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): g.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
        print(f"gtiles {gt:5d} gfrac {gf:3d}: {dt*1e6:7.2f} us/step  {NB/dt:9.0f} steps/s", flush=True)
