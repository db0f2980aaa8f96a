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
