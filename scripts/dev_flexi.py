"""Timing of the iso-surfacing kernels at the pipeline resolution (res 64, PL:1126)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import facade, ops
for res in (64, 128, 384):
    fc = facade.FlexiCubes("cuda")
    x, _ = fc.construct_voxel_grid(res) if res <= 128 else (None, None)
    if x is None:
        G = res + 1
        lin = torch.linspace(-0.5, 0.5, G, device="cuda")
        x = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3)
    x = x * 2.2
    r = x.norm(dim=1)
    s = (r - 0.7 + 0.08 * torch.sin(9 * x[:, 0]) * torch.cos(7 * x[:, 1]) * torch.sin(5 * x[:, 2])).requires_grad_(True)
    v, f, _ = ops.flexicubes(x, s, res)
    g = torch.ones_like(v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        v, f, _ = ops.flexicubes(x, s, res, verts_cap=len(v) + 16, faces_cap=len(f) + 16)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        s.grad = None
        v.backward(g, retain_graph=True)
    torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 20
    G3 = (res + 1) ** 3
    print("res %d: %d verts %d faces | fwd %.3f ms (incl. the count read-back), bwd %.3f ms | grid %.1f MB -> %.0f GB/s over the 16 B/point the forward must read" % (
        res, len(v), len(f), tf * 1e3, tb * 1e3, G3 * 16 / 1e6, G3 * 16 / tf / 1e9))
