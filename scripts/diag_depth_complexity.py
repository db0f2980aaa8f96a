"""Depth complexity of the naive, cull-free rasteriser on the bench and crop scenes: covering fragments (pixel centre inside the face, z
in front of the near plane) per hit pixel, per render -- what a two-pass "conservative z bound first" scatter could save (VERDICT r4
item 6: worth it from ~1.8 fragments per hit pixel).  Also: candidate (face, pixel-box) pairs per fragment = what the enumerate
stage visits.  python scripts/diag_depth_complexity.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import engine as E, synthetic
H = W = 512
dev = torch.device("cuda", 0)
ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
# pixel centres in NDC (pytorch3d: +x left, +y up)
xf = (1.0 - (2.0 * xs.float() + 1.0) / W).reshape(-1)
yf = (1.0 - (2.0 * ys.float() + 1.0) / H).reshape(-1)
for crop in (None, "hoi"):
    sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=H, W=W, seed=0, **({"crop": crop} if crop else {}))
    gb = E.GuidanceBatch([sc])
    cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
    gb.step(cfg); torch.cuda.synchronize()
    m = gb.meta[0]
    ndc = gb.region("ndc", torch.float32, (gb.Vtot, 3)).clone()
    faces = gb.faces.long().reshape(-1, 3)
    p2f = gb.region("p2f", torch.int32, (2, H * W))
    for r, (name, fsel) in enumerate((("hand render", faces[:m["Fh"]]), ("hand + object render", faces))):
        cover = torch.zeros(H * W, dtype=torch.int32, device=dev)
        boxpix = 0
        for c0 in range(0, fsel.shape[0], 64):
            f = fsel[c0:c0 + 64]
            v = ndc[f]                                   # (n,3,3)
            x0, y0, x1, y1, x2, y2 = v[:, 0, 0:1], v[:, 0, 1:2], v[:, 1, 0:1], v[:, 1, 1:2], v[:, 2, 0:1], v[:, 2, 1:2]
            e0 = (xf - x1) * (y2 - y1) - (yf - y1) * (x2 - x1)
            e1 = (xf - x2) * (y0 - y2) - (yf - y2) * (x0 - x2)
            e2 = (xf - x0) * (y1 - y0) - (yf - y0) * (x1 - x0)
            area = (x2 - x0) * (y1 - y0) - (y2 - y0) * (x1 - x0)
            inside = ((e0 * area > 0) & (e1 * area > 0) & (e2 * area > 0)) & (v[:, :, 2].min(1, keepdim=True)[0] > 0.005)
            cover += inside.sum(0).int()
            bx = ((xf >= v[:, :, 0].min(1, keepdim=True)[0]) & (xf <= v[:, :, 0].max(1, keepdim=True)[0]) &
                  (yf >= v[:, :, 1].min(1, keepdim=True)[0]) & (yf <= v[:, :, 1].max(1, keepdim=True)[0]))
            boxpix += int(bx.sum())
        hit = int((p2f[r] >= 0).sum()); frags = int(cover.sum()); hist = torch.bincount(cover.clamp(max=6), minlength=7).tolist()
        print(f"{'crop' if crop else 'bench'} scene, {name}: {fsel.shape[0]} faces, {hit} hit pixels, {frags} covering fragments = {frags / max(hit, 1):.2f} per hit pixel; "
              f"pixels by depth 0..5, 6+: {hist}; bounding-box candidates {boxpix} = {boxpix / max(frags, 1):.2f} per fragment", flush=True)
