"""Analytic known-answer tests of the CPU oracle.

The geometry operators of the path live in un-vendored pytorch3d / kaolin, so the reference offers no golden
vectors for them ("parity unpinned"); these tests pin the restatement to closed-form answers instead.
"""
import math

import numpy as np
import pytest
import torch

from followmyhold_amd import synthetic
from oracle import clib
from oracle import ref_ops as R

BLUR = R.blur_radius_from_sigma()


def tri(v):
    return np.asarray(v, np.float32).reshape(-1, 3, 3)


def test_k1_single_triangle_coverage_and_plane_depth():
    """Right triangle covering the NDC lower-left half... pixel count ~ area, zbuf = plane equation."""
    H = W = 64
    fv = tri([[[-0.8, -0.8, 2.0], [0.8, -0.8, 2.0], [-0.8, 0.8, 4.0]]])
    p2f, zb, ba, di = clib.rasterize(fv, H, W, BLUR, K=1)
    hit = p2f[..., 0] >= 0
    area_px = 0.5 * (1.6 * W / 2) ** 2
    assert abs(hit.sum() - area_px) < 0.06 * area_px
    # barycentrics are perspective-correct: 1/z interpolates linearly in screen space
    ys, xs = np.nonzero(hit)
    yf = -1 + (2 * (H - 1 - ys) + 1) / H
    lam = (yf - (-0.8)) / 1.6                     # screen-space weight of the z=4 vertex
    z_expected = 1.0 / ((1 - lam) / 2.0 + lam / 4.0)
    assert np.abs(zb[ys, xs, 0] - z_expected).max() < 1e-3
    assert np.abs(ba[ys, xs, 0].sum(-1) - 1).max() < 1e-5
    # inside -> negative signed distance; the hypotenuse runs exactly through pixel centres, which are kept by the
    # blur radius with a (tiny) positive distance
    assert (di[ys, xs, 0] <= 0).mean() > 0.95 and di[ys, xs, 0].max() < BLUR
    assert np.all(p2f[~hit] == -1) and np.all(zb[~hit] == -1)


def test_k2_nearest_wins_and_equal_depth_keeps_lower_face_id():
    H = W = 32
    a = [[-0.5, -0.5, 3.0], [0.5, -0.5, 3.0], [0.0, 0.5, 3.0]]
    b = [[-0.5, -0.5, 2.0], [0.5, -0.5, 2.0], [0.0, 0.5, 2.0]]
    p2f, zb, _, _ = clib.rasterize(tri([a, b]), H, W, BLUR)
    hit = p2f[..., 0] >= 0
    assert hit.sum() > 50 and np.all(p2f[hit] == 1) and np.allclose(zb[hit], 2.0)
    p2f, _, _, _ = clib.rasterize(tri([a, a]), H, W, BLUR)   # identical depth: first (lowest id) stays
    assert np.all(p2f[p2f >= 0] == 0)
    p2f2, zb2, _, _ = clib.rasterize(tri([a, b]), H, W, BLUR, K=2)  # K-buffer sorted by z
    both = p2f2[..., 1] >= 0
    assert both.sum() > 50 and np.all(p2f2[both][:, 0] == 1) and np.all(p2f2[both][:, 1] == 0)
    assert np.all(zb2[both][:, 0] <= zb2[both][:, 1])


def test_k2b_near_plane_policy():
    """MeshRasterizer clips at z = znear / 2 = 0.005 (perspective camera, z_clip_value=None; RUN:84-105).  Restated policy:
    faces with a vertex nearer than the plane are culled (pytorch3d culls the fully-near ones and splits the
    straddling ones -- the latter are counted so that callers can flag them); a vertex ON the plane is kept."""
    H = W = 16
    xy = [[-0.5, -0.5], [0.5, -0.5], [0.0, 0.5]]
    mk = lambda zs: [[x, y, z] for (x, y), z in zip(xy, zs)]
    near, straddle, on_plane, far = mk([0.001, 0.002, 0.004]), mk([0.004, 0.3, 0.3]), mk([0.005, 0.3, 0.3]), mk([0.5, 0.5, 0.5])
    zc = np.float32(0.01) * np.float32(0.5)
    on_plane[0][2] = float(zc)
    for fv, visible in [(near, False), (straddle, False), (on_plane, True), (far, True)]:
        p2f, _, _, _ = clib.rasterize(tri([fv]), H, W, BLUR)
        assert (p2f >= 0).any() == visible
    assert clib.count_near_clipped(tri([near, straddle, on_plane, far])) == 1
    # a culled face in front does not hide the face behind it
    p2f, zb, _, _ = clib.rasterize(tri([straddle, far]), H, W, BLUR)
    assert np.all(p2f[p2f >= 0] == 1) and (p2f >= 0).sum() > 20
    try:        # plane disabled: the plain zmax < 0 / pz < 0 rules of the naive rasteriser remain
        clib.set_z_clip(-1e30)
        p2f, _, _, _ = clib.rasterize(tri([straddle]), H, W, BLUR)
        assert (p2f >= 0).any() and clib.count_near_clipped(tri([straddle])) == 0
    finally:
        clib.set_z_clip()
    assert clib.count_near_clipped(tri([straddle])) == 1


def test_k3_pixel_centre_convention_plus_x_is_left_plus_y_is_up():
    """pytorch3d NDC: +X left, +Y up -> a triangle in the x>0, y>0 quadrant lights the TOP-LEFT image quadrant."""
    H = W = 32
    fv = tri([[[0.1, 0.1, 1.0], [0.9, 0.1, 1.0], [0.1, 0.9, 1.0]]])
    p2f, _, _, _ = clib.rasterize(fv, H, W, BLUR)
    ys, xs = np.nonzero(p2f[..., 0] >= 0)
    assert len(ys) > 20 and ys.max() < H // 2 and xs.max() < W // 2
    # pixel (0,0) has centre (+1 - 1/W, +1 - 1/H)
    assert R.pix_ndc(torch.tensor(W - 1), W, H, torch.float32).item() == pytest.approx(1 - 1 / W)
    # non-square: the longer axis spans a wider NDC range
    assert R.pix_ndc(torch.tensor(0), 64, 32, torch.float32).item() == pytest.approx(-2 + 2 / 64)


def test_k4_icosphere_render_normalises_to_unit_range():
    v, f = synthetic.icosphere(4, 0.3)
    v = v + np.array([0, 0, -3.0], np.float32)
    vt, ft = torch.from_numpy(v), torch.from_numpy(f)
    cam = R.Camera(60.0, 96, 96)
    sel = R.rasterize_select(R.world_to_ndc(vt, cam), ft, 96, 96, BLUR)
    rgba, zb = R.render_normals(vt, ft, cam, sel)
    nn, dd = R.render_normal_and_disparity(rgba, zb)
    hit = torch.from_numpy(sel["pix_to_face"] >= 0)
    # projected disc radius = r / sqrt(d^2 - r^2) in tan units -> pixels
    r_px = 0.3 / math.sqrt(9 - 0.09) / math.tan(math.radians(30)) * 48
    assert abs(hit.sum().item() - math.pi * r_px ** 2) < 0.05 * math.pi * r_px ** 2
    assert torch.all(nn[~hit] == 0) and float(dd.min()) == 0.0 and float(dd.max()) == pytest.approx(1.0, abs=1e-5)
    assert float(nn.min()) >= 0 and float(nn.max()) <= 1.0
    # nearest point of the sphere is at view depth 2.7, and it projects to the image centre
    assert abs(float(zb[hit].min()) - 2.7) < 5e-3


def test_k5_quaternion_to_matrix():
    I = R.quaternion_to_matrix(torch.tensor([1.0, 0, 0, 0]))
    assert torch.equal(I, torch.eye(3))
    q = torch.tensor([0.3, -0.5, 0.2, 0.7])
    assert torch.allclose(R.quaternion_to_matrix(q), R.quaternion_to_matrix(2 * q), atol=1e-6)
    Rz = R.quaternion_to_matrix(torch.tensor([math.cos(math.pi / 4), 0, 0, math.sin(math.pi / 4)]))
    assert torch.allclose(Rz @ torch.tensor([1.0, 0, 0]), torch.tensor([0.0, 1.0, 0.0]), atol=1e-6)


def test_k6_knn_on_lattice_and_edge_loss_of_tetrahedron():
    g = np.stack(np.meshgrid(np.arange(4), np.arange(4), np.arange(4), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    q = g + np.float32(0.2)
    d2, idx = clib.knn1(q, g)
    assert np.array_equal(idx, np.arange(64)) and np.allclose(d2, 3 * 0.04, atol=1e-6)
    d2, idx = clib.knn1(np.array([[0.5, 0, 0]], np.float32), g)     # two candidates at equal distance: first wins
    assert idx[0] == 0 and d2[0] == pytest.approx(0.25)
    v = torch.tensor([[1.0, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]])
    e = R.unique_edges(torch.tensor([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]]))
    assert e.shape == (6, 2)
    assert float(R.mesh_edge_loss(v, e)) == pytest.approx(8.0, rel=1e-6)   # every edge has length^2 = 8


def test_k7_inside_count_of_cube_and_sphere():
    cube_v = np.array([[x, y, z] for x in (-0.5, 0.5) for y in (-0.5, 0.5) for z in (-0.5, 0.5)], np.float32)
    cube_f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6],
                       [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.int32)
    grid = R.dense_grid_points(np.array([-1.0] * 3, np.float32), np.array([1.0] * 3, np.float32), 20)
    ins = clib.inside(cube_v, cube_f, grid)
    expected = np.all(np.abs(grid) < 0.5, axis=1)
    on_surface = np.any(np.isclose(np.abs(grid), 0.5), axis=1) & np.all(np.abs(grid) <= 0.5 + 1e-6, axis=1)
    assert np.array_equal(ins[~on_surface], expected[~on_surface])
    sv, sf = synthetic.icosphere(3, 0.7)
    ins = clib.inside(sv, sf.astype(np.int32), grid)
    r = np.linalg.norm(grid, axis=1)
    assert np.all(ins[r < 0.68]) and not np.any(ins[r > 0.7])
    assert (r < 0.68).sum() <= ins.sum() <= (r <= 0.7).sum()
    # finer grid: the inside fraction approaches the analytic volume fraction of the ball in the cube
    fine = R.dense_grid_points(np.array([-1.0] * 3, np.float32), np.array([1.0] * 3, np.float32), 48)
    frac = clib.inside(sv, sf.astype(np.int32), fine).mean()
    assert abs(frac - (np.linalg.norm(fine, axis=1) < 0.7).mean()) < 0.004      # lattice count of the analytic ball
    assert abs(frac - (4 / 3 * math.pi * 0.7 ** 3) / 8) < 0.02                   # ~ volume fraction
    # winding independence: the flipped mesh is the same solid
    assert np.array_equal(clib.inside(sv, sf[:, ::-1].astype(np.int32).copy(), grid), ins)


def test_k7b_point_mesh_distance_of_a_sphere_mesh():
    sv, sf = synthetic.icosphere(3, 1.0)
    pts = np.array([[0, 0, 0], [2, 0, 0], [0, 0.5, 0], [0, 0, -3]], np.float32)
    d2, _ = clib.point_mesh_dist(sv, sf.astype(np.int32), pts)
    d = np.sqrt(d2)
    assert abs(d[1] - 1.0) < 5e-3 and abs(d[3] - 2.0) < 5e-3 and abs(d[2] - 0.5) < 1e-2 and 0.97 < d[0] <= 1.0
    sdf = R.mesh_sdf(torch.from_numpy(sv), torch.from_numpy(sf), pts)
    assert np.all(np.sign(sdf) == [-1, 1, -1, 1])


def test_k10_procrustes_and_icp_recover_a_known_similarity():
    from oracle import icp_ref
    rng = np.random.default_rng(0)
    a = rng.normal(size=(500, 3))
    Rm = synthetic.axis_angle_matrix([0.2, -0.4, 0.3])
    b = 1.7 * a @ Rm.T + np.array([0.3, -0.2, 0.5])
    T = icp_ref.procrustes(a, b, reflection=False, scale=True)
    assert np.allclose(T[:3, :3], 1.7 * Rm, atol=1e-9) and np.allclose(T[:3, 3], [0.3, -0.2, 0.5], atol=1e-9)
    src = rng.normal(size=(400, 3))
    M = np.eye(4)
    M[:3, :3] = 1.05 * synthetic.axis_angle_matrix([0.02, 0.03, -0.01])
    M[:3, 3] = [0.02, -0.01, 0.03]
    tgt = src @ M[:3, :3].T + M[:3, 3]
    T, cost = icp_ref.icp_points(src, tgt, n_iter=30, outliers=0.0, min_scale=0.7, max_scale=3.0)
    assert cost < 1e-6 and np.allclose(T, M, atol=1e-5)


def test_k12_float64_finite_differences_of_the_differentiable_rasteriser_path():
    """Autograd through the oracle's fragment re-evaluation + shading + normalisation vs central differences."""
    torch.manual_seed(0)
    v, f = synthetic.icosphere(1, 0.3)
    v = torch.from_numpy(v).double() + torch.tensor([0.02, -0.01, -1.5], dtype=torch.float64)
    f = torch.from_numpy(f)
    cam = R.Camera(60.0, 24, 24, dtype=torch.float64)
    sel = R.rasterize_select(R.world_to_ndc(v, cam), f, 24, 24, BLUR)
    tgt_n = torch.rand(24, 24, 3, dtype=torch.float64)
    tgt_d = torch.rand(24, 24, dtype=torch.float64)

    def loss_fn(vv):
        rgba, zb = R.render_normals(vv, f, cam, sel)
        nn, dd = R.render_normal_and_disparity(rgba, zb)
        return R.normal_alignment_loss(nn, tgt_n) + (dd - tgt_d).abs().mean()

    vv = v.clone().requires_grad_(True)
    loss_fn(vv).backward()
    g = vv.grad.clone()
    rng = np.random.default_rng(1)
    for _ in range(6):
        i, k = int(rng.integers(v.shape[0])), int(rng.integers(3))
        h = 1e-6
        vp, vm = v.clone(), v.clone()
        vp[i, k] += h
        vm[i, k] -= h
        fd = (float(loss_fn(vp)) - float(loss_fn(vm))) / (2 * h)
        assert abs(fd - float(g[i, k])) <= 2e-4 * max(1.0, abs(fd)), (i, k, fd, float(g[i, k]))


def test_closest_point_on_triangle_regions():
    """trimesh.proximity.closest_point restatement (icp on_surface, ICP:106-107): one query per Voronoi region of a
    triangle -- interior, three vertices, three edges -- and the nearer of two triangles wins."""
    from oracle import icp_ref as I
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 5], [1, 0, 5], [0, 1, 5]], float)
    f = np.array([[0, 1, 2], [3, 4, 5]])
    p = np.array([[0.2, 0.2, 1.0], [-1, -1, 0], [2, 0, 0], [0, 3, 1], [0.5, -1, 0], [1, 1, 0], [-1, 0.5, 0], [0.2, 0.2, 4.0]], float)
    q, d = I.closest_point(v, f, p)
    want = np.array([[0.2, 0.2, 0], [0, 0, 0], [1, 0, 0], [0, 1, 0], [0.5, 0, 0], [0.5, 0.5, 0], [0, 0.5, 0], [0.2, 0.2, 5]])
    assert np.allclose(q, want) and np.allclose(d, np.linalg.norm(p - want, axis=1))
    # a collinear triangle behaves like its longest segment (the edge regions catch every query)
    q2, d2 = I.closest_point(np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2], [0, 0, 1], [1, 0, 1], [0, 1, 1]], float),
                             np.array([[0, 1, 2], [3, 4, 5]]), np.array([[0.1, 0.1, 0.0], [0.1, 0.1, 0.9]]))
    assert np.allclose(q2, [[1 / 15] * 3, [0.1, 0.1, 1.0]]) and np.allclose(d2[1], 0.1)


def test_nearest_neighbour_restatement_equals_scipy_ckdtree():
    """The reference's ICP queries scipy.spatial.cKDTree (ICP:16, 91, 109); scipy is installed here, so the oracle's
    brute-force `nearest` is pinned against the real thing: same indices, same distances."""
    from scipy.spatial import cKDTree
    from oracle import icp_ref as I
    rng = np.random.default_rng(4)
    q = rng.normal(size=(3000, 3))
    p = rng.normal(size=(1200, 3)) * 1.3
    d_ref, i_ref = cKDTree(q).query(p)
    d, i = I.nearest(p, q)
    assert np.array_equal(i, i_ref) and np.allclose(d, d_ref, rtol=1e-12, atol=1e-15)
