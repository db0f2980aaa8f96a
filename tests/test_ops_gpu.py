"""GPU parity of the stand-alone C-ABI operators (SDF pieces, MANO-shaped LBS fwd/bwd, ICP loop) vs the oracle."""
import numpy as np
import pytest
import torch

from followmyhold_amd import synthetic
from oracle import clib, icp_ref, lbs_ref
from oracle import ref_ops as R

gpu = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@gpu
def test_point_mesh_distance_and_inside_points_bit_exact():
    """kaolin point_to_mesh_distance / check_sign replacements (kaolin_sdf_ops.py:100-104) on a 17^3 grid."""
    from followmyhold_amd import ops
    hv, hf = synthetic.hand_template()
    ov, of = synthetic.make_object("ico2")
    ov = ov + np.array([0.02, 0.0, 0.01], np.float32)
    grid = R.joint_grid(torch.from_numpy(hv), torch.from_numpy(ov), 16)
    for v, f in ((hv, hf), (ov, of)):
        d2, fi = ops.point_mesh_dist(torch.from_numpy(v).cuda(), torch.from_numpy(f).cuda(), torch.from_numpy(grid).cuda())
        rd2, rfi = clib.point_mesh_dist(v, f.astype(np.int32), grid)
        assert np.array_equal(d2.cpu().numpy(), rd2)
        assert np.array_equal(fi.cpu().numpy(), rfi)
        ins = ops.inside_points(torch.from_numpy(v).cuda(), torch.from_numpy(f).cuda(), torch.from_numpy(grid).cuda())
        rins = clib.inside(v, f.astype(np.int32), grid)
        assert np.array_equal(ins.cpu().numpy(), rins) and rins.sum() > 10
        # signed distance as get_sdf_of_meshes builds it (kaolin_sdf_ops.py:100-107)
        sdf = torch.sqrt(d2) * torch.where(ins, -1.0, 1.0)
        assert np.allclose(sdf.cpu().numpy(), R.mesh_sdf(torch.from_numpy(v), torch.from_numpy(f), grid), rtol=0, atol=0)


@gpu
def test_fused_intersection_count_equals_full_sdf_route():
    """The step's column-parity count == honerf_intersection_loss over two full SDFs (pipelines.py:231-239)."""
    from followmyhold_amd import engine as E
    from followmyhold_amd import ops
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import make_scene
    sc = make_scene("ico2", 64, 64, seed=5)
    gb = E.GuidanceBatch([{k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}], grid_res=24)
    cfg, _ = E.phase_cfg("C", do_update=False)
    gb.step(cfg)
    torch.cuda.synchronize()
    world = gb.region("world", torch.float32, (-1, 3))
    hv, ov = world[:778], world[778:]
    grid = torch.from_numpy(R.joint_grid(hv.cpu(), ov.cpu(), 24)).cuda()
    ih = ops.inside_points(hv, sc["hand_faces"].cuda(), grid)
    io = ops.inside_points(ov, sc["obj_faces"].cuda(), grid)
    assert int(gb.loss_dict(0)["n_intersect"]) == int((ih & io).sum().item())


@gpu
@pytest.mark.parametrize("B,use_mfma", [(3, 0), (3, 1), (32, -1)])
def test_lbs_forward_and_backward(B, use_mfma):
    """MANO-shaped LBS: forward vs the smplx restatement, backward vs torch autograd (VALU and matrix-core paths)."""
    from followmyhold_amd import ops
    m = synthetic.mano_like_model(1)
    mt = {k: torch.from_numpy(np.asarray(v)) for k, v in m.items()}
    g = torch.Generator().manual_seed(B)
    betas = torch.randn(B, 10, generator=g)
    aa = torch.randn(B, 16, 3, generator=g) * 0.3
    rot = torch.stack([torch.stack([torch.from_numpy(synthetic.axis_angle_matrix(a.numpy())).float() for a in row]) for row in aa])
    bt, rt = betas.clone().requires_grad_(True), rot.clone().requires_grad_(True)
    v_ref, j_ref = lbs_ref.lbs(bt, rt, mt)
    gv, gj = torch.randn(v_ref.shape, generator=g), torch.randn(j_ref.shape, generator=g)
    ((v_ref * gv).sum() + (j_ref * gj).sum()).backward()

    model = ops.LbsModel(m)
    bd, rd = betas.cuda().requires_grad_(True), rot.cuda().requires_grad_(True)
    verts, joints = ops.lbs(bd, rd, model, use_mfma=use_mfma)
    assert rel_err(verts.detach().cpu().numpy(), v_ref.detach().numpy()) < 1e-5
    assert rel_err(joints.detach().cpu().numpy(), j_ref.detach().numpy()) < 1e-5
    ((verts * gv.cuda()).sum() + (joints * gj.cuda()).sum()).backward()
    assert rel_err(bd.grad.cpu().numpy(), bt.grad.numpy()) < 1e-4
    assert rel_err(rd.grad.cpu().numpy(), rt.grad.numpy()) < 1e-4
    # zero pose + zero betas -> template (K8)
    z = torch.zeros(1, 10, device="cuda")
    I = torch.eye(3, device="cuda").expand(1, 16, 3, 3).contiguous()
    v0, _ = ops.lbs(z, I, model, use_mfma=0)
    assert np.abs(v0[0].cpu().numpy() - m["v_template"]).max() < 1e-7


@gpu
@pytest.mark.parametrize("N,M,outliers", [(1000, 5000, 0.2), (2500, 4000, 0.0)])
def test_icp_loop_matches_the_numpy_reference(N, M, outliers):
    """Coarse-phase sizes of h2m.py:35-54 (1000 x 5000, 20 % trimmed): same transforms as the float64 restatement."""
    from followmyhold_amd import ops
    rng = np.random.default_rng(N)
    ov, of = synthetic.make_object("20k")
    tgt_all = ov.astype(np.float64) * 3.0
    tgt = tgt_all[rng.choice(len(tgt_all), M, replace=False)]
    src0 = tgt_all[rng.choice(len(tgt_all), N, replace=False)] + rng.normal(size=(N, 3)) * 1e-3
    Mtx = np.eye(4)
    Mtx[:3, :3] = 0.9 * synthetic.axis_angle_matrix([0.05, -0.08, 0.04])
    Mtx[:3, 3] = [0.01, -0.015, 0.02]
    src = icp_ref.transform_points(src0, np.linalg.inv(Mtx))
    n_out = int(outliers * N)
    rec = []
    T_ref, c_ref = icp_ref.icp_points(src, tgt, n_iter=12, outliers=outliers, min_scale=0.7, max_scale=3.0, record=rec)
    T, c, hist = ops.icp_points(src, tgt, n_iter=12, n_outliers=n_out, min_scale=0.7, max_scale=3.0, return_history=True)
    assert np.allclose(hist, [r[0] for r in rec], rtol=1e-9, atol=1e-12)
    assert abs(c - c_ref) <= 1e-9 * abs(c_ref) and np.allclose(T, T_ref, rtol=1e-8, atol=1e-10)
    # the alignment actually improves
    assert hist[-1] < 0.5 * hist[0]


@gpu
def test_icp_multi_start_equals_the_per_start_loop():
    """`for cube in cubes` (ICP:91-175: identity, 7 reflections, 9 rotations) as one batched enqueue: every start gets
    the transform / cost / history of its own single-start run, bit for bit, and matches the numpy restatement."""
    from followmyhold_amd import ops
    from foho.alignment import mesh_align as MA
    rng = np.random.default_rng(3)
    ov, _ = synthetic.make_object("20k")
    tgt_all = ov.astype(np.float64) * 3.0
    tgt = tgt_all[rng.choice(len(tgt_all), 3000, replace=False)]
    src = tgt_all[rng.choice(len(tgt_all), 700, replace=False)] * 1.1 + 0.01
    cubes = [np.eye(4)] + MA.get_all_axis_aligned_reflections() + MA.get_all_axis_aligned_rotations()
    assert len(cubes) == 17
    starts = np.stack([icp_ref.transform_points(src, c) for c in cubes])
    Ts, costs, hists = ops.icp_points_multi(starts, tgt, n_iter=8, n_outliers=140, min_scale=0.7, max_scale=3.0, return_history=True)
    assert Ts.shape == (17, 4, 4) and costs.shape == (17,) and hists.shape == (17, 8)
    for s in (0, 3, 9, 16):
        T1, c1, h1 = ops.icp_points(starts[s], tgt, n_iter=8, n_outliers=140, min_scale=0.7, max_scale=3.0, return_history=True)
        assert np.array_equal(T1, Ts[s]) and c1 == costs[s] and np.array_equal(h1, hists[s])
    T_ref, c_ref = icp_ref.icp_points(starts[5], tgt, n_iter=8, outliers=0.2, min_scale=0.7, max_scale=3.0)
    assert abs(costs[5] - c_ref) <= 1e-9 * abs(c_ref) and np.allclose(Ts[5], T_ref, rtol=1e-8, atol=1e-10)
    assert np.argmin(costs) == 0      # the unrotated start wins on an asymmetric object


@gpu
def test_icp_on_surface_matches_the_numpy_reference():
    """icp(..., on_surface=True) (ICP:106-107): source points are matched to the closest point ON the target triangles.
    Same cost history and transform as the float64 restatement (exhaustive Voronoi-region closest point)."""
    from followmyhold_amd import ops
    rng = np.random.default_rng(11)
    tv, tf = synthetic.icosphere(3, 0.3)                    # 1280 faces
    tv = (tv * (1 + 0.2 * np.sin(9 * tv[:, :1]) * np.cos(7 * tv[:, 1:2]))).astype(np.float64)
    bary = rng.dirichlet(np.ones(3), 400)
    src0 = (tv[tf[rng.integers(0, len(tf), 400)]] * bary[:, :, None]).sum(1)
    Mtx = np.eye(4)
    Mtx[:3, :3] = 0.93 * synthetic.axis_angle_matrix([0.06, -0.04, 0.08])
    Mtx[:3, 3] = [0.01, -0.012, 0.008]
    src = icp_ref.transform_points(src0, np.linalg.inv(Mtx))
    rec = []
    T_ref, c_ref = icp_ref.icp_points(src, tv, n_iter=10, outliers=0.1, min_scale=0.7, max_scale=3.0, record=rec, target_faces=tf)
    T, c, hist = ops.icp_points(src, tv, n_iter=10, n_outliers=40, min_scale=0.7, max_scale=3.0, return_history=True,
                                target_faces=tf)
    assert np.allclose(hist, [r[0] for r in rec], rtol=1e-8, atol=1e-13)
    assert abs(c - c_ref) <= 1e-8 * abs(c_ref) + 1e-14 and np.allclose(T, T_ref, rtol=1e-7, atol=1e-9)
    assert hist[-1] < 0.2 * hist[0]
    # against the sampled-point variant the surface residual is the smaller one
    _, c_pts = ops.icp_points(src, tv, n_iter=10, n_outliers=40, min_scale=0.7, max_scale=3.0)
    assert c < c_pts
    # batched starts share the triangles
    Ts, cs = ops.icp_points_multi(np.stack([src, src + 0.01]), tv, n_iter=4, n_outliers=40, target_faces=tf)
    T1, c1 = ops.icp_points(src + 0.01, tv, n_iter=4, n_outliers=40, target_faces=tf)
    assert np.array_equal(Ts[1], T1) and cs[1] == c1
    with pytest.raises(Exception):
        ops.icp_points(src, tv, n_iter=2, target_faces=np.array([[0, 1, 10 ** 6]]))


@gpu
def test_lbs_large_batch_matrix_core_tiles():
    """B >= 128 takes the 64 x 64 register-blocked pose-blend kernel (k_lbs_poseblend_mfma64): bit-identical to the
    16 x 16 kernel small batches use (same k-chunking), ragged last tile included, and 1e-5 from the VALU path."""
    from followmyhold_amd import ops
    model = ops.LbsModel(synthetic.mano_like_model())
    g = torch.Generator().manual_seed(0)
    B = 200
    betas = torch.randn(B, 10, generator=g).cuda()
    aa = torch.randn(B, 16, 3, generator=g) * 0.4
    rot = torch.stack([torch.from_numpy(np.stack([synthetic.axis_angle_matrix(a.numpy()) for a in row])) for row in aa]).float().cuda()
    v_big, j_big = ops.lbs(betas, rot, model, use_mfma=1)
    v_small, j_small = ops.lbs(betas[:100].contiguous(), rot[:100].contiguous(), model, use_mfma=1)      # B < 128: 16 x 16 kernel
    assert torch.equal(v_big[:100], v_small) and torch.equal(j_big[:100], j_small)
    v_tail, _ = ops.lbs(betas[150:].contiguous(), rot[150:].contiguous(), model, use_mfma=1)
    assert torch.equal(v_big[150:], v_tail)
    v_valu, _ = ops.lbs(betas, rot, model, use_mfma=0)
    assert (v_big - v_valu).abs().max() < 1e-5


@gpu
@pytest.mark.parametrize("seed", range(6))
def test_inside_and_distance_fuzz_with_grid_aligned_geometry(seed):
    """Worst case for a ray-parity inside test: an axis-aligned box whose vertices, edges and faces coincide with grid
    lines, planes through grid points, plus a sphere; query points on the same lattice (rays through vertices and along
    edges and faces).  Inside flags, squared distances and nearest-face ids equal the C oracle bit for bit."""
    from followmyhold_amd import ops
    rng = np.random.default_rng(900 + seed)
    n = 9 + seed
    lin = np.linspace(-1.0, 1.0, n).astype(np.float32)
    grid = np.stack(np.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3)
    lo, hi = lin[2], lin[-3]
    bv = np.array([[x, y, z] for x in (lo, hi) for y in (lo, hi) for z in (lo, hi)], np.float32)
    bf = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4],
                   [1, 5, 7], [1, 7, 3]], np.int64)
    sv, sf = synthetic.icosphere(1 + seed % 2, 0.45)
    sv = sv + rng.choice(lin, 3).astype(np.float32) * 0.25
    for v, f in ((bv, bf), (sv.astype(np.float32), sf), (np.concatenate([bv, sv]).astype(np.float32), np.concatenate([bf, sf + 8]))):
        tv, tf, tg = torch.from_numpy(v).cuda(), torch.from_numpy(f).cuda(), torch.from_numpy(grid).cuda()
        ins = ops.inside_points(tv, tf, tg).cpu().numpy()
        assert np.array_equal(ins, clib.inside(v, f.astype(np.int32), grid))
        d2, fi = ops.point_mesh_dist(tv, tf, tg)
        rd2, rfi = clib.point_mesh_dist(v, f.astype(np.int32), grid)
        assert np.array_equal(d2.cpu().numpy(), rd2) and np.array_equal(fi.cpu().numpy(), rfi)


@gpu
@pytest.mark.parametrize("res,shift", [(16, 0.0), (24, 0.013), (32, -0.02), (64, 0.0)])
def test_fused_intersection_count_with_an_axis_aligned_box_object(res, shift):
    """The joint grid spans the joint AABB, so the extreme faces of an axis-aligned box lie exactly ON boundary grid
    planes and its edges on grid lines whenever the box sets the AABB: the tie cases of the ray-parity test, inside the
    fused step's column-parity path.  The count equals the oracle's (kaolin check_sign restatement) exactly."""
    from followmyhold_amd import engine as E
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import make_scene
    sc = make_scene("ico2", 64, 64, seed=5)
    hv = sc["hand_verts"].numpy()
    lo, hi = hv.min(0) - 0.01, hv.max(0) + 0.01                   # the box encloses the hand: every hand-inside point counts
    lo[0] += shift
    bv = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], np.float32)
    bf = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4],
                   [1, 5, 7], [1, 7, 3]], np.int64)
    scn = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
    scn["obj_verts"], scn["obj_faces"], scn["T_h2m"] = bv, bf, np.eye(4, dtype=np.float32)
    gb = E.GuidanceBatch([scn], grid_res=res)
    cfg, _ = E.phase_cfg("C", do_update=False)
    gb.step(cfg)
    torch.cuda.synchronize()
    world = gb.region("world", torch.float32, (-1, 3)).cpu()
    assert np.abs(world[778:].numpy() - bv).max() < 1e-7           # identity transforms up to (v - c) + c rounding
    want = R.intersection_count(world[:778], sc["hand_faces"], world[778:], torch.from_numpy(bf), res)
    assert want > 0 and int(gb.loss_dict(0)["n_intersect"]) == want
