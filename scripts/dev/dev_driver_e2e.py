"""End-to-end time of the product driver on files (run on the GPU box): N scene folders at 512 x 512 with a 20 k-face object
and a 522 k-face MoGe image mesh each, `foho.guidance.run.run` with FOHO_MESH_LEVEL_GUIDANCE=1; where the wall time goes."""
import os, sys, time, tempfile, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from followmyhold_amd import engine as E, synthetic, inputs
from foho.guidance import run as G
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tmp = tempfile.mkdtemp()
names = ["cropped_obj_img_dir", "mask_dir", "moge_out_dir", "hunyuan_hoi_mesh_dir", "hamer_out_dir", "h2m_rt_dir", "aligned_mano_dir", "guidance_out_dir"]
d = {n: os.path.join(tmp, n) for n in names}
rf = E.hip_render_fn("cuda")
def image_mesh(n, fov=60.0):
    ys, xs = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    t = np.tan(np.radians(fov) / 2)
    z = 0.5 + 0.1 * np.sin(xs / n * 6.0) * np.cos(ys / n * 5.0)
    x = (xs + 0.5 - n / 2) / (n / 2) * t * z * 0.9; y = -(ys + 0.5 - n / 2) / (n / 2) * t * z * 0.9
    v = np.stack([x, y, -z], -1).reshape(-1, 3).astype(np.float32)
    i = (ys[:-1, :-1] * n + xs[:-1, :-1]).reshape(-1)
    f = np.concatenate([np.stack([i, i + n, i + 1], 1), np.stack([i + 1, i + n, i + n + 1], 1)], 0).astype(np.int64)
    return v, f
mv, mf = image_mesh(512)
t0 = time.perf_counter()
for k in range(N):
    sc = synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, seed=300 + k)
    inputs.save_scene_files(sc, mv, mf, {k2: v for k2, v in d.items() if k2 != "guidance_out_dir"}, f"{k:04d}")
jr = os.path.join(tmp, "J.npy"); np.save(jr, sc["J_regressor"])
print("fixtures written in %.1f s" % (time.perf_counter() - t0), flush=True)
os.environ["FOHO_J_REGRESSOR"] = jr; os.environ["FOHO_MESH_LEVEL_GUIDANCE"] = "1"
FLIGHTS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (16, 16, 1)
for rep, nfl in enumerate(FLIGHTS):
    os.environ["FOHO_IMAGES_IN_FLIGHT"] = str(nfl)
    if nfl == 0:      # the driver's own default (16, or 32 for long lists)
        os.environ.pop("FOHO_IMAGES_IN_FLIGHT")
    dd = dict(d, guidance_out_dir=os.path.join(tmp, f"out{rep}"))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if rep == 1:
        pr = cProfile.Profile(); pr.enable()
    tot = G.run(project_root=tmp, task_list_file=None, **dd)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if rep == 1:
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
    print(f"in flight {nfl}: {dt * 1e3 / N:.1f} ms per image end to end ({N / dt:.1f} images/s), n_images {tot['n_images']}, failed {tot['n_failed']}", flush=True)
