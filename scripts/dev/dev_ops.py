"""Timings of the stand-alone operators of the C ABI at the sizes the path uses them (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import ops, synthetic
def bench(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
rng = np.random.default_rng(0)
# ICP: 10k source samples vs 10k target samples (mesh_align.py samples 10 000 points), 30 iterations
src = rng.normal(size=(10000, 3)); tgt = src @ np.linalg.qr(rng.normal(size=(3, 3)))[0].T * 1.1 + 0.05
print("icp_points 10k x 10k, 30 iterations: %.2f ms (%.3f ms / iteration)" % ((lambda t: (t, t / 30))(bench(lambda: ops.icp_points(src, tgt, 30, n_outliers=500), n=5, warm=1))))
# LBS
model = ops.LbsModel(synthetic.mano_like_model())
for B in (1, 64):
    betas = torch.randn(B, 10, device="cuda", requires_grad=True)
    rot = torch.eye(3, device="cuda").repeat(B, 16, 1, 1).clone().requires_grad_(True)
    f = lambda: ops.lbs(betas, rot, model)
    def fb():
        v, j = ops.lbs(betas, rot, model); (v.sum() + j.sum()).backward()
    print("lbs B=%d: fwd %.3f ms, fwd+bwd %.3f ms" % (B, bench(f), bench(fb)))
# signed distance of the 65^3 grid to a 20k-face mesh (get_sdf_of_meshes, SDF:88-109): brute force, LDS tiled
ov, of = synthetic.make_object("20k")
v = torch.from_numpy(ov).cuda(); f = torch.from_numpy(of).int().cuda()
g = torch.stack(torch.meshgrid(*([torch.linspace(-1.2, 1.2, 65, device="cuda")] * 3), indexing="ij"), -1).reshape(-1, 3).contiguous()
t = bench(lambda: ops.point_mesh_dist(v, f, g), n=3, warm=1)
print("point_mesh_dist 65^3 x 20k faces: %.1f ms (%.0f G point-triangle tests/s)" % (t, 274625 * 20480 / t / 1e6))
t = bench(lambda: ops.inside_points(v, f, g), n=3, warm=1)
print("inside_points   65^3 x 20k faces: %.1f ms" % t)
p1 = torch.randn(778, 3, device="cuda"); p2 = torch.randn(10242, 3, device="cuda")
print("knn1 778 x 10242: %.3f ms" % bench(lambda: ops.knn1(p1, p2)))
