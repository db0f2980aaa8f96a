"""Stress case: render a MoGe-style image mesh (one vertex per pixel, 2 faces per pixel quad) into target maps."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import engine as E
def image_mesh(n, fov=60.0):
    ys, xs = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    t = np.tan(np.radians(fov) / 2)
    z = 0.5 + 0.1 * np.sin(xs / n * 6.0) * np.cos(ys / n * 5.0)            # depth map
    x = (xs + 0.5 - n / 2) / (n / 2) * t * z
    y = -(ys + 0.5 - n / 2) / (n / 2) * t * z
    v = np.stack([x, y, -z], -1).reshape(-1, 3).astype(np.float32)       # camera looks down -z
    i = (ys[:-1, :-1] * n + xs[:-1, :-1]).reshape(-1)
    f = np.concatenate([np.stack([i, i + n, i + 1], 1), np.stack([i + 1, i + n, i + n + 1], 1)], 0).astype(np.int64)
    return v, f
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
v, f = image_mesh(n)
print("image mesh", v.shape, f.shape)
render = E.hip_render_fn("cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    normal, disp, p2f = render(v, f, 512, 512, 60.0)
    torch.cuda.synchronize(); print("render %.1f ms, hit pixels %d, distinct faces %d" % ((time.perf_counter() - t0) * 1e3, (p2f >= 0).sum(), len(np.unique(p2f[p2f >= 0]))))
