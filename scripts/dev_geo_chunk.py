"""Forward time of the geometry decoder (cached query side, 65^3 grid) by row-block size.  python scripts/dev_geo_chunk.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import standins
from followmyhold_amd.facade import generate_dense_grid_points
from followmyhold_amd.geo_decode import HipGeoDecoder
dev = torch.device("cuda", 0)
torch.manual_seed(0)
vae = standins.StandInShapeVAE(num_latents=3072, embed_dim=64, width=1024, heads=16, layers=1, num_freqs=8)
dec = vae.geo_decoder.to(dev).eval()
xyz_np, gsz, _ = generate_dense_grid_points(np.full(3, -1.10), np.full(3, 1.10), octree_depth=5, octree_resolution=64, indexing="ij")
xyz = torch.as_tensor(xyz_np, dtype=torch.float32, device=dev)
lat = torch.randn(1, 3072, 1024, device=dev).half()
res = {}
chunks = [int(a) for a in sys.argv[1:]] or [16384, 32768, 49152, 65536, 92160, 137472, 274688]
hips = {c: HipGeoDecoder.from_module(dec, device=dev, chunk_rows=c) for c in chunks}
qs = {c: hips[c].grid_queries(xyz) for c in chunks}
for rnd in range(4):
    for c in chunks:
        hip, q = hips[c], qs[c]
        for _ in range(2):
            hip._prepared = None
            with torch.no_grad():
                hip(q, lat)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            hip._prepared = None
            with torch.no_grad():
                hip(q, lat)
        torch.cuda.synchronize(); res.setdefault(c, []).append((time.perf_counter() - t0) / 3 * 1e3)
for c in chunks:
    print(f"chunk {c:7d}: forward {min(res[c]):.2f} ms (median {sorted(res[c])[len(res[c]) // 2]:.2f})", flush=True)
