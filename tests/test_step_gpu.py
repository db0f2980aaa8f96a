"""GPU parity of the HIP guidance step (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): face indices bit-exact; vertex positions, losses and gradients within
1e-4 relative in fp32.
"""
import numpy as np
import pytest
import torch

from helpers import make_scene
from oracle import ref_ops as R
from oracle import step_ref as S

gpu = pytest.mark.gpu
RTOL = 1e-4


def _np_scene(sc):
    return {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def small():
    """64x64 image, MANO-sized hand + icosphere(2) object, grid 16^3; oracle forward/backward of phase C."""
    sc = make_scene("ico2", 64, 64, seed=0)
    p = S.make_params(
        scale_hand=torch.tensor([1.02]), trans_hand=torch.tensor([0.004, -0.003, 0.002]),
        rot_hand=torch.tensor([0.999, 0.02, -0.01, 0.03]), scale_obj=torch.tensor([0.97]),
        trans_obj=torch.tensor([-0.002, 0.003, 0.001]), rot_obj=torch.tensor([0.998, -0.03, 0.02, 0.01]))
    st = S.JointStepper(sc, p, denoise_i=19, grid_res=16)
    total, terms, aux, grads = st.step(update=True)
    after = {k: v.detach().clone() for k, v in st.p.items()}
    return dict(scene=sc, params=p, total=total, terms=terms, aux=aux, grads=grads, after=after)


def _engine(sc, p, grid_res=16):
    from followmyhold_amd import engine as E
    gb = E.GuidanceBatch([_np_scene(sc)], grid_res=grid_res)
    gb.set_params(0, **{k: v.numpy() for k, v in p.items()})
    return E, gb


@gpu
def test_vertex_stage_bit_exact(small):
    """Similarity transform + projection: identical op order as the oracle -> identical bits."""
    E, gb = _engine(small["scene"], small["params"])
    cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
    gb.step(cfg, stages=E.L.STAGE_VERTEX)
    torch.cuda.synchronize()
    aux = small["aux"]
    world_ref = torch.cat([aux["hand"]["verts"], aux["obj_verts_t"]], 0).detach().numpy()
    world = gb.region("world", torch.float32, (-1, 3)).cpu().numpy()
    assert rel_err(world, world_ref) < 1e-6
    cam = R.Camera(small["scene"]["fov"], 64, 64)
    ndc_ref = R.world_to_ndc(torch.from_numpy(world_ref), cam).numpy()
    ndc = gb.region("ndc", torch.float32, (-1, 3)).cpu().numpy()
    assert np.array_equal(world, world_ref), f"world differs, max abs {np.abs(world - world_ref).max()}"
    assert np.array_equal(ndc, ndc_ref), f"ndc differs, max abs {np.abs(ndc - ndc_ref).max()}"
    vn_ref = torch.cat([R.vertex_normals(aux["hand"]["verts"].detach(), small["scene"]["hand_faces"]),
                        R.vertex_normals(aux["obj_verts_t"].detach(), small["scene"]["obj_faces"])], 0).numpy()
    vn = gb.region("vn", torch.float32, (-1, 3)).cpu().numpy()
    assert np.abs(vn - vn_ref).max() < 1e-6
    idx = gb.region("knn_idx", torch.int32)[:778].cpu().numpy()
    assert np.array_equal(idx, aux["knn_idx"].numpy())


@gpu
def test_raster_face_indices_bit_exact(small):
    E, gb = _engine(small["scene"], small["params"])
    cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
    gb.step(cfg, stages=E.L.STAGE_VERTEX | E.L.STAGE_RASTER)
    torch.cuda.synchronize()
    gb.raise_on_flags()
    P = 64 * 64
    p2f = gb.region("p2f", torch.int32, (2, P)).cpu().numpy()
    zb = gb.region("zbuf", torch.float32, (2, P)).cpu().numpy()
    sd = gb.region("sdist", torch.float32, (2, P)).cpu().numpy()
    prod = gb.region("prod", torch.float32, (2, P)).cpu().numpy()
    aux = small["aux"]
    for r, ren in enumerate([aux["hand"]["render"], aux["render"]]):
        sel = ren["sel"]
        ref = sel["pix_to_face"].reshape(-1)
        assert (p2f[r] >= 0).sum() > 20
        mism = int((p2f[r] != ref).sum())
        assert mism == 0, f"render {r}: {mism} face-index mismatches"
        assert np.array_equal(zb[r], sel["zbuf"].reshape(-1))
        assert np.array_equal(sd[r], sel["dists"].reshape(-1))
    sil_ref = aux["render"]["sil"].detach().numpy().reshape(-1)
    assert np.abs((1.0 - prod[1]) - sil_ref).max() < 1e-6


@gpu
def test_losses_and_gradients(small):
    E, gb = _engine(small["scene"], small["params"])
    cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
    gb.step(cfg)
    torch.cuda.synchronize()
    gb.raise_on_flags()
    l = gb.loss_dict(0)
    t = {k: float(v) for k, v in small["terms"].items()}
    assert int(l["n_intersect"]) == small["aux"]["n_int"]
    pairs = [("contact", "contact"), ("kps", "kps"), ("trans_hand", "trans_hand"), ("trans_obj", "trans_obj"),
             ("verts_obj", "verts_obj"), ("edge", "edge"), ("normal0", "normal_hand"), ("disp0", "disp_hand"),
             ("normal1", "normal_hoi"), ("disp1", "disp_hoi"), ("sil1", "sil_hoi")]
    for a, b in pairs:
        assert abs(l[a] - t[b]) <= RTOL * max(abs(t[b]), 1e-6), (a, l[a], t[b])
    assert abs(l["total"] - float(small["total"])) <= RTOL * abs(float(small["total"]))
    g = gb.grad_params[0].cpu().numpy()
    gref = np.concatenate([small["grads"][k].numpy().reshape(-1) for k in E.PARAM_NAMES])
    for k, sl in E.PARAM_SLICES.items():
        assert rel_err(g[sl], gref[sl]) < 5 * RTOL, (k, g[sl], gref[sl])
    gv = gb.grad_obj_verts(0).cpu().numpy()
    assert rel_err(gv, small["grads"]["obj_verts"].numpy()) < 5 * RTOL
    # AdamW update (PL:1478, 1601)
    p_after = gb.get_params(0)
    for k in E.PARAM_NAMES:
        ref = small["after"][k].numpy()
        assert np.abs(p_after[k].numpy() - ref).max() <= 2e-6 + RTOL * np.abs(ref).max() * 1e-2, (k, p_after[k], ref)
