"""Times the geometry decoder's routes at the Hunyuan3D-2 shape on the 65^3 grid: forward (plain / cached query side), backward
dense (kept activations / recomputed) and over the active rows with a FlexiCubes-produced gradient.  python scripts/dev_geo_rows.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from followmyhold_amd import ops, standins
from followmyhold_amd.facade import generate_dense_grid_points
from followmyhold_amd.geo_decode import HipGeoDecoder

dev = torch.device("cuda", 0)
torch.manual_seed(0)
vae = standins.StandInShapeVAE(num_latents=3072, embed_dim=64, width=1024, heads=16, layers=1, num_freqs=8)
hip = HipGeoDecoder.from_module(vae.geo_decoder.to(dev).eval(), device=dev)
xyz_np, gsz, _ = generate_dense_grid_points(np.full(3, -1.10), np.full(3, 1.10), octree_depth=5, octree_resolution=64, indexing="ij")
xyz = torch.as_tensor(xyz_np, dtype=torch.float32, device=dev)
n = xyz.shape[0]
lat = torch.randn(1, 3072, 1024, device=dev).half()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)


q_plain = xyz.half().float().unsqueeze(0).contiguous()
def fwd(q):
    hip._prepared = None
    with torch.no_grad():
        return hip(q, lat)
print(f"forward, plain:  {timed(lambda: fwd(q_plain)):.2f} ms", flush=True)
q = hip.grid_queries(xyz)
print(f"forward, cached: {timed(lambda: fwd(q)):.2f} ms", flush=True)
assert torch.equal(fwd(q), fwd(q_plain))

out = fwd(q).float().reshape(-1)
sdf = (-out).clone().requires_grad_(True)
verts, faces, _ = ops.flexicubes(xyz, sdf, 64)
(verts * torch.randn_like(verts)).sum().backward()
go = -sdf.grad
n_act = int((go != 0).sum())
print(f"surface: {verts.shape[0]} verts, {faces.shape[0]} faces; active rows {n_act} of {n} = {n_act / n:.3f}", flush=True)
god = torch.randn(n, device=dev)

def fb(mode, g):
    hip.backward_mode = mode
    l = lat.clone().requires_grad_(True)
    (hip(q, l).float().reshape(-1) * g).sum().backward()
    return l.grad
for mode, g, name in (("keep", god, "dense gradient, kept activations"), ("recompute", god, "dense gradient, recomputed"), ("rows", god, "dense gradient, rows route"),
                      ("keep", go, "FlexiCubes gradient, kept (dense)"), ("rows", go, "FlexiCubes gradient, rows route")):
    print(f"forward + backward, {name}: {timed(lambda: fb(mode, g)):.2f} ms", flush=True)
hip.set_kv(hip.kv_of(lat).detach())
for cap in (None, 65536, 49152):
    print(f"backward alone, rows route, row_cap {cap}: {timed(lambda: hip.decode_bwd_rows(q, go, row_cap=cap)):.2f} ms   stats {hip.last_row_stats.tolist()}", flush=True)
print(f"backward alone, dense recompute: {timed(lambda: hip.decode_bwd(q, god)):.2f} ms", flush=True)
