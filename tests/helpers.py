"""Shared test helpers: oracle-backed scene construction."""
import numpy as np
import torch

from followmyhold_amd import synthetic
from oracle import ref_ops as R


def oracle_render_fn(verts, faces, H, W, fov):
    """Ground-truth target renderer backed by the CPU oracle."""
    v = torch.from_numpy(np.asarray(verts, np.float32))
    f = torch.from_numpy(np.asarray(faces, np.int64))
    cam = R.Camera(fov, H, W)
    sel = R.rasterize_select(R.world_to_ndc(v, cam), f, H, W, R.blur_radius_from_sigma())
    rgba, zbuf = R.render_normals(v, f, cam, sel)
    nrm, disp = R.render_normal_and_disparity(rgba, zbuf)
    return nrm.numpy(), disp.numpy(), sel["pix_to_face"]


def make_scene(obj_kind="ico2", H=64, W=64, seed=0, **kw):
    sc = synthetic.build_scene(oracle_render_fn, obj_kind=obj_kind, H=H, W=W, seed=seed, **kw)
    return to_torch(sc)


def to_torch(sc):
    out = {}
    for k, v in sc.items():
        out[k] = torch.from_numpy(v) if isinstance(v, np.ndarray) else v
    return out


def icp_case_inputs(name):
    """Seeded point sets of the ICP golden cases (tests/golden/make_icp_golden.py generates tests/golden/ref_icp.npz from them with the
    REFERENCE's icp(); tests/test_icp_golden.py feeds the same sets to the oracle and to the GPU path).  -> dict(src, tgt, kw[, tgt_faces])"""
    cases = {   # name: (seed, source points, target points, iterations, icp() keyword arguments)
        "coarse": (1, 1000, 5000, 50, dict(outliers=0.2, min_scale=0.7, max_scale=3.0)),             # h2m.py:35-54's coarse stage
        "fine": (2, 5000, 10000, 8, dict(outliers=0.2, min_scale=0.7, max_scale=3.0)),               # ... a short fine stage
        "rotations": (3, 700, 3000, 8, dict(outliers=0.2, min_scale=0.7, max_scale=3.0, test_rotations=True)),
        "reflections_fixed_scale": (4, 600, 2500, 6, dict(outliers=0.1, test_reflections=True, fixed_scale=True)),
        "no_outliers_clipped": (5, 800, 2000, 10, dict(outliers=0, min_scale=0.95, max_scale=1.05)),
        "on_surface": (6, 300, 0, 6, dict(outliers=0.1, min_scale=0.7, max_scale=3.0, on_surface=True)),
    }
    seed, N, M, n_iter, kw = cases[name]
    rng = np.random.default_rng(seed)
    ov, of = synthetic.make_object("20k")
    allp = ov.astype(np.float64) * 3.0
    Mtx = np.eye(4)
    Mtx[:3, :3] = (0.9 if name != "no_outliers_clipped" else 0.8) * synthetic.axis_angle_matrix([0.05, -0.08, 0.04])
    Mtx[:3, 3] = [0.01, -0.015, 0.02]
    inv = np.linalg.inv(Mtx)
    out = dict(n_iter=n_iter, kw=kw)
    if name == "on_surface":
        tv, tf = synthetic.icosphere(3, 0.3)
        tv = (tv * (1 + 0.2 * np.sin(9 * tv[:, :1]) * np.cos(7 * tv[:, 1:2]))).astype(np.float64)
        bary = rng.dirichlet(np.ones(3), N)
        src0 = (tv[tf[rng.integers(0, len(tf), N)]] * bary[:, :, None]).sum(1)
        out.update(tgt=tv, tgt_faces=np.asarray(tf, np.int64))
    else:
        out["tgt"] = allp[rng.choice(len(allp), M, replace=False)]
        src0 = allp[rng.choice(len(allp), N, replace=False)] + rng.normal(size=(N, 3)) * 1e-3
    out["src"] = src0 @ inv[:3, :3].T + inv[:3, 3]
    return out


ICP_CASES = ("coarse", "fine", "rotations", "reflections_fixed_scale", "no_outliers_clipped", "on_surface")
