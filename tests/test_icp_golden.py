"""The ICP loop against the REFERENCE's own `icp()` (src/foho/alignment/mesh_align.py:56-175, executed by tests/golden/make_icp_golden.py with
scipy's real cKDTree; trimesh's helpers bound to oracle/icp_ref.py): the numpy oracle on the CPU, `foho_icp_run*` through
foho.alignment.mesh_align / followmyhold_amd.ops on the GPU.  Six seeded cases: the coarse stage's sizes (50 x 1000 x 5000, 20 % trimmed,
scale clip 0.7 .. 3), a short fine stage (5000 x 10000), rotation starts, reflection starts with a fixed scale, an untrimmed run against the
scale clip, on_surface."""
import os

import numpy as np
import pytest

from helpers import ICP_CASES, icp_case_inputs
from oracle import icp_ref

gpu = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_icp.npz"))


def _oracle_kw(c):
    kw = dict(c["kw"])
    on_surface = kw.pop("on_surface", False)
    return dict(kw, target_faces=c["tgt_faces"] if on_surface else None)


@pytest.mark.parametrize("name", ICP_CASES)
def test_oracle_reproduces_the_references_icp(name):
    c = icp_case_inputs(name)
    rec = []
    T, cost = icp_ref.icp(c["src"], c["tgt"], c["n_iter"], record=rec, **_oracle_kw(c))
    assert abs(cost - float(G[f"{name}_cost"])) <= 1e-9 * abs(float(G[f"{name}_cost"]))
    assert np.allclose(T, G[f"{name}_T"], rtol=1e-9, atol=1e-11)
    # every iteration of every start: the cost the reference measured on its matched pairs
    assert len(rec) == len(G[f"{name}_iter_cost"])
    assert np.allclose([r[0] for r in rec], G[f"{name}_iter_cost"], rtol=1e-9, atol=1e-13)


def test_oracle_init_transform_and_start_matrices_are_the_references():
    c = icp_case_inputs("coarse")
    for fs in (False, True):
        got = icp_ref.compute_init_transform(c["src"] * 1.7 + 0.3, None, c["tgt"], None, fixed_scale=fs)
        assert np.allclose(got, G[f"init_fixed{int(fs)}"], rtol=1e-12, atol=1e-14)
    assert np.allclose(np.stack(icp_ref.axis_aligned_rotations()), G["rotations"], rtol=0, atol=1e-15)
    assert np.array_equal(np.stack(icp_ref.axis_aligned_reflections()), G["reflections"])
    from foho.alignment import mesh_align as MA      # the product's own start matrices (host numpy)
    assert np.allclose(np.stack(MA.get_all_axis_aligned_rotations()), G["rotations"], rtol=0, atol=1e-15)
    assert np.array_equal(np.stack(MA.get_all_axis_aligned_reflections()), G["reflections"])
    src, tgt = MA.Mesh(c["src"] * 1.7 + 0.3), MA.Mesh(c["tgt"])
    for fs in (False, True):
        assert np.allclose(MA.compute_init_transform(src, tgt, fs), G[f"init_fixed{int(fs)}"], rtol=1e-12, atol=1e-14)


@gpu
@pytest.mark.parametrize("name", ICP_CASES)
def test_hip_icp_reproduces_the_references_icp(name):
    """foho.alignment.mesh_align.icp (all starts in one foho_icp_run_batch / foho_icp_run_surface enqueue) -> the reference's best transform
    and cost; for the single-start cases also the cost of every iteration (foho_icp_run's history)."""
    from foho.alignment import mesh_align as MA
    from followmyhold_amd import ops
    c = icp_case_inputs(name)
    kw = dict(c["kw"])
    tgt = MA.Mesh(c["tgt"], c.get("tgt_faces"))
    T, cost = MA.icp(MA.Mesh(c["src"]), tgt, c["n_iter"], **kw)
    assert abs(cost - float(G[f"{name}_cost"])) <= 1e-9 * abs(float(G[f"{name}_cost"])) + 1e-14
    assert np.allclose(T, G[f"{name}_T"], rtol=1e-8, atol=1e-10)
    if not (kw.get("test_rotations") or kw.get("test_reflections")):
        n_out = int(kw.get("outliers", 0) * len(c["src"]))
        _, _, hist = ops.icp_points(c["src"], c["tgt"], n_iter=c["n_iter"], n_outliers=n_out, fixed_scale=kw.get("fixed_scale", False),
                                    min_scale=kw.get("min_scale", 0.5), max_scale=kw.get("max_scale", 2.0), return_history=True,
                                    **({"target_faces": c["tgt_faces"]} if kw.get("on_surface") else {}))
        assert np.allclose(hist, G[f"{name}_iter_cost"], rtol=1e-8, atol=1e-13)
