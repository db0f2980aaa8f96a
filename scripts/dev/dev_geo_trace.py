"""Five cached forwards + three forward / active-row-backward pairs of the geometry decoder at the Hunyuan3D-2 shape, 65^3 grid -- the
run to put under `rocprofv3 --kernel-trace --stats`.  python scripts/dev/dev_geo_trace.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
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
lat = torch.randn(1, 3072, 1024, device=dev).half()
q = hip.grid_queries(xyz)
with torch.no_grad():
    out = hip(q, lat)
sdf = (-out.float().reshape(-1)).clone().requires_grad_(True)
verts, faces, _ = ops.flexicubes(xyz, sdf, 64)
(verts * torch.randn_like(verts)).sum().backward()
go = -sdf.grad
torch.cuda.synchronize()
for _ in range(5):
    hip._prepared = None
    t0 = time.perf_counter()
    with torch.no_grad():
        hip(q, lat)
    torch.cuda.synchronize()
    print(f"forward (cached query side): {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
for _ in range(3):
    l = lat.clone().requires_grad_(True)
    t0 = time.perf_counter()
    (hip(q, l).float().reshape(-1) * go).sum().backward()
    torch.cuda.synchronize()
    print(f"forward + active-row backward: {(time.perf_counter() - t0) * 1e3:.2f} ms  rows {hip.last_row_stats.tolist()}", flush=True)
