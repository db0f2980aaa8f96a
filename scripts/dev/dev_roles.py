"""Timing experiment: cost of each k_stage2 role at iteration 25 of the bench scene (run on the GPU box).
FOHO_DEBUG_SKIP_ROLES bits: 1 normals, 2 raster, 4 knn, 8 kps, 16 obj_local, 32 inside, 64 no stats atomics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from followmyhold_amd import _lib as L_
L_.SO_PATH = os.path.join(ROOT, "followmyhold_amd", "libfoho_hip_stamps.so")  # ablation hooks live in the STAMPS build
from followmyhold_amd import engine as E, synthetic
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
NB = int(os.environ.get("NB", "1"))
gb = E.GuidanceBatch([sc] * NB); cfgu, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
for _ in range(int(os.environ.get("NSTEP", "25"))): gb.step(cfgu)
torch.cuda.synchronize()
ndc = gb.region("ndc", torch.float32, (-1, 3)).cpu().numpy(); faces = gb.faces.cpu().numpy()
fv = ndc[faces]                                   # (F,3,3)
w = (fv[:, :, 0].max(1) - fv[:, :, 0].min(1)) * 256 + 1; h = (fv[:, :, 1].max(1) - fv[:, :, 1].min(1)) * 256 + 1
area = np.ceil(w) * np.ceil(h)
Fh = gb.meta[0]["Fh"]
print("zmin", fv[:, :, 2].min(), "hand box px: mean %.1f max %.0f sum %.0f | obj: mean %.1f max %.0f sum %.0f" % (
    area[:Fh].mean(), area[:Fh].max(), area[:Fh].sum(), area[Fh:].mean(), area[Fh:].max(), area[Fh:].sum()))
print("per-block T: hand(8) max %.0f  obj(64) max %.0f" % (max(area[i:i + 8].sum() for i in range(0, Fh, 8)),
      max(area[Fh + i:Fh + i + 64].sum() for i in range(0, len(area) - Fh, 64))))
masks = [int(x) for x in os.environ["MASKS"].split(",")] if os.environ.get("MASKS") else [0, 1, 2, 4, 32, 1 | 2, 2 | 32, 59, 61, 31, 62]
res = {m: [] for m in masks}
for rep in range(5):
    for m in masks:
        os.environ["FOHO_DEBUG_SKIP_ROLES"] = str(m)
        acc = {}
        for _ in range(10):
            for k, v in gb.step_profiled(cfg).items(): acc[k] = acc.get(k, 0) + v / 10
        res[m].append(acc)
for m in masks:
    ks = ["k_xform", "k_stage2", "k_resolve", "k_loss", "k_pix_bwd", "k_vert_bwd", "k_final"]
    print(m, {k: "%.1f/%.1f" % (min(a[k] for a in res[m]) * 1e3, sum(a[k] for a in res[m]) / len(res[m]) * 1e3) for k in ks if k in res[m][0]})
