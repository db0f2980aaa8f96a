"""Where the once-per-image target-map render (engine.hip_render_fn, 522 k-face MoGe image mesh at 512 x 512) spends its time."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import engine as E
n = 512
ys, xs = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
t = np.tan(np.radians(60.0) / 2)
z = 0.5 + 0.1 * np.sin(xs / n * 6.0) * np.cos(ys / n * 5.0)
x = (xs + 0.5 - n / 2) / (n / 2) * t * z * 0.9; y = -(ys + 0.5 - n / 2) / (n / 2) * t * z * 0.9
v = np.stack([x, y, -z], -1).reshape(-1, 3).astype(np.float32)
i = (ys[:-1, :-1] * n + xs[:-1, :-1]).reshape(-1)
f = np.concatenate([np.stack([i, i + n, i + 1], 1), np.stack([i + 1, i + n, i + n + 1], 1)], 0).astype(np.int64)
rf = E.hip_render_fn("cuda")
rf(v, f, 512, 512, 60.0); torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); rf(v, f, 512, 512, 60.0); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("render: %.1f ms (min of 5)" % (min(ts) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): rf(v, f, 512, 512, 60.0)
torch.cuda.synchronize(); pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(16)
