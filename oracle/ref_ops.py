"""Torch-CPU restatement of the operators on FollowMyHold's guidance path (ORACLE; tests only).

Every function cites the reference site it follows (paths relative to /root/reference;
PL = third_party_patches/hy3dgen/shapegen/pipelines.py, RUN = src/foho/guidance/run.py,
SDF = third_party/utilz/kaolin_sdf_ops.py).  Operators whose arithmetic lives in
un-vendored pytorch3d/kaolin follow SURVEY.md Appendix A ("parity unpinned").

All arithmetic is written as explicit element-wise torch ops in a fixed
association order (no matmul, no fused ops) so that float32 results are plain
IEEE-754 and can be matched bit-for-bit by the HIP kernels where the domain is
discrete (face indices).  Works in float32 (parity) and float64 (finite-difference
gradient checks).
"""
import math

import numpy as np
import torch

from . import clib

K_EPS = 1e-8


# --------------------------------------------------------------------------------------
# transforms (pytorch3d.transforms.quaternion_to_matrix; PL:1324, PL:1408, PL:1484, PL:1524)
# --------------------------------------------------------------------------------------
def quaternion_to_matrix(q):
    """wxyz quaternion (not necessarily unit) -> 3x3; two_s = 2/|q|^2."""
    r, i, j, k = q[0], q[1], q[2], q[3]
    two_s = 2.0 / (((r * r + i * i) + j * j) + k * k)
    o = torch.stack([
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)])
    return o.reshape(3, 3)


def _rowmat(v, M):
    """v @ M.T written element-wise: out_i = (v0*M[i,0] + v1*M[i,1]) + v2*M[i,2]."""
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    return torch.stack([(x * M[i, 0] + y * M[i, 1]) + z * M[i, 2] for i in range(3)], dim=1)


def transform_hunyuan2moge(verts, RT):
    """PL:242-250: verts @ RT[:3,:3].T + RT[:3,3]."""
    return _rowmat(verts, RT[:3, :3]) + RT[:3, 3]


def bbox_center(verts):
    """PL:111: (min + max) / 2 over vertices; differentiable through min/max indices."""
    return (verts.min(dim=0)[0] + verts.max(dim=0)[0]) / 2.0


def transform_around_center_w_scale(verts, R, t, scale):
    """PL:108-118: (scale * (verts - center)) @ R.T + center + t."""
    center = bbox_center(verts)
    u = scale * (verts - center)
    return (_rowmat(u, R) + center) + t


# --------------------------------------------------------------------------------------
# camera (pytorch3d FoVPerspectiveCameras; RUN:84-90)
# --------------------------------------------------------------------------------------
class Camera:
    def __init__(self, fov_deg, H, W, R=None, T=None, znear=0.01, zfar=100.0, dtype=torch.float32):
        self.H, self.W, self.znear, self.zfar = int(H), int(W), float(znear), float(zfar)
        self.R = torch.tensor([[-1.0, 0, 0], [0, 1.0, 0], [0, 0, -1.0]], dtype=dtype) if R is None else R.to(dtype)
        self.T = torch.zeros(3, dtype=dtype) if T is None else T.to(dtype)
        self.k00, self.k11 = fov_focal(fov_deg)
        self.dtype = dtype


def fov_focal(fov_deg, aspect=1.0, znear=0.01):
    """K[0,0], K[1,1] of FoVPerspectiveCameras.compute_projection_matrix, in float32 steps."""
    f32 = np.float32
    fov = f32(np.pi / 180.0) * f32(fov_deg)
    tan_half = f32(np.tan(f32(fov / f32(2.0))))
    max_y = f32(tan_half * f32(znear))
    min_y = f32(-max_y)
    max_x = f32(max_y * f32(aspect))
    min_x = f32(-max_x)
    k00 = f32(f32(2.0) * f32(znear)) / f32(max_x - min_x)
    k11 = f32(f32(2.0) * f32(znear)) / f32(max_y - min_y)
    return float(f32(k00)), float(f32(k11))


def world_to_ndc(verts, cam):
    """MeshRasterizer.transform: view = X @ R + T ; ndc_xy = K * view_xy / view_z ; z = view_z."""
    R, T = cam.R.to(verts.dtype), cam.T.to(verts.dtype)
    x, y, z = verts[:, 0], verts[:, 1], verts[:, 2]
    vx = ((x * R[0, 0] + y * R[1, 0]) + z * R[2, 0]) + T[0]
    vy = ((x * R[0, 1] + y * R[1, 1]) + z * R[2, 1]) + T[1]
    vz = ((x * R[0, 2] + y * R[1, 2]) + z * R[2, 2]) + T[2]
    return torch.stack([(cam.k00 * vx) / vz, (cam.k11 * vy) / vz, vz], dim=1)


def ndc_to_screen(ndc, H, W):
    """cameras.transform_points_screen (PL:1336, PL:1491): x_px = W/2 - s*x_ndc, s = min(H,W)/2."""
    s = min(H, W) / 2.0
    return torch.stack([W / 2.0 - s * ndc[:, 0], H / 2.0 - s * ndc[:, 1]], dim=1)


# --------------------------------------------------------------------------------------
# vertex normals (pytorch3d Meshes.verts_normals_packed; PL:83)
# --------------------------------------------------------------------------------------
def cross3(a, b):
    return torch.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1],
                        a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                        a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], dim=1)


def vertex_normals(verts, faces):
    v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    fn = cross3(v2 - v1, v0 - v1)
    vn = torch.zeros_like(verts)
    vn = vn.index_add(0, faces[:, 0], fn)
    vn = vn.index_add(0, faces[:, 1], fn)
    vn = vn.index_add(0, faces[:, 2], fn)
    nrm = torch.sqrt((vn[:, 0] * vn[:, 0] + vn[:, 1] * vn[:, 1]) + vn[:, 2] * vn[:, 2])
    return vn / torch.clamp(nrm, min=1e-6)[:, None]


# --------------------------------------------------------------------------------------
# rasteriser: discrete selection in C, differentiable re-evaluation in torch
# (pytorch3d rasterize_meshes forward/backward)
# --------------------------------------------------------------------------------------
def pix_ndc(idx, S1, S2, dtype):
    rng = 2.0
    if S1 > S2:
        rng = (S1 * 2.0) / S2
    off = rng / 2.0
    return -off + (rng * idx.to(dtype) + off) / S1


def _edge(px, py, ax, ay, bx, by):
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def _seg_d2(px, py, ax, ay, bx, by):
    bax, bay = bx - ax, by - ay
    l2 = bax * bax + bay * bay
    safe = torch.where(l2 <= K_EPS, torch.ones_like(l2), l2)
    t = ((bax * (px - ax) + bay * (py - ay)) / safe).clamp(0.0, 1.0).detach()  # envelope: t constant in bwd
    qx, qy = ax + t * bax, ay + t * bay
    dx, dy = qx - px, qy - py
    d_seg = dx * dx + dy * dy
    ex, ey = px - bx, py - by
    return torch.where(l2 <= K_EPS, ex * ex + ey * ey, d_seg)


def clip_subtriangles(fv, sub, z_clip):
    """pytorch3d clip_faces (renderer/mesh/clip.py) for the rows of fv (n,3,3) [x_ndc, y_ndc, z_view] whose `sub` (n,) is
    0 or 1: the sub-triangle a face straddling z = z_clip is rasterised as -- differentiable w.r.t. fv, same operations and
    order as oracle/foho_oracle.c::clip_one_face (perspective camera).  Rows with sub < 0 come back unchanged.
      two vertices behind   p1 = the vertex in front, p2 / p3 the next two (cyclic): triangle (p4, p5, p1)
      one vertex behind     p1 = that vertex: sub 0 = (p4, p2, p5), sub 1 = (p5, p2, p3)
      p4 / p5               crossings of p1p2 / p1p3: w = (z1 - c) / (z1 - z_o), z = z1 (1 - w) + z_o w,
                            xy = ((xy1 z1)(1 - w) + (xy_o z_o) w) / c"""
    sel = sub >= 0
    if not bool(sel.any()):
        return fv
    rows = sel.nonzero(as_tuple=True)[0]
    f = fv[rows]
    c = torch.tensor(z_clip, dtype=torch.float32).to(fv.dtype)
    behind = f[:, :, 2] < c
    nb = behind.sum(1)
    if not bool(((nb == 1) | (nb == 2)).all()):
        raise ValueError("clip_subtriangles: a fragment's sub-triangle index does not match its face (not straddling the plane)")
    lone = torch.where((nb == 2)[:, None], ~behind, behind)          # the one vertex on its own side
    i1 = lone.to(torch.int64).argmax(1)
    i2, i3 = (i1 + 1) % 3, (i1 + 2) % 3
    ar = torch.arange(f.shape[0])
    p1, p2, p3 = f[ar, i1], f[ar, i2], f[ar, i3]

    def crossing(po):
        w = (p1[:, 2] - c) / (p1[:, 2] - po[:, 2])
        u = 1.0 - w
        z = p1[:, 2] * u + po[:, 2] * w
        x = ((p1[:, 0] * p1[:, 2]) * u + (po[:, 0] * po[:, 2]) * w) / c
        y = ((p1[:, 1] * p1[:, 2]) * u + (po[:, 1] * po[:, 2]) * w) / c
        return torch.stack([x, y, z], 1)

    p4, p5 = crossing(p2), crossing(p3)
    tri_two = torch.stack([p4, p5, p1], 1)
    tri_a = torch.stack([p4, p2, p5], 1)
    tri_b = torch.stack([p5, p2, p3], 1)
    s_ = sub[rows]
    out = torch.where((nb == 2)[:, None, None], tri_two, torch.where((s_ == 0)[:, None, None], tri_a, tri_b))
    return fv.index_put((rows,), out)


def eval_fragments(verts_ndc, faces, pix, face_idx, H, W, blur_radius=0.0, sub=None):
    """Differentiable per-(pixel, face) evaluation: returns zbuf, bary_clip (n,3), signed dist, inside.  `sub` (n,): the
    sub-triangle of a near-clipped face the fragment belongs to (rasterize_select's "sub" / third pair column), -1 / None
    for faces rasterised as they are; barycentrics are then those of the sub-triangle."""
    dt = verts_ndc.dtype
    yi = torch.div(pix, W, rounding_mode="floor")
    xi = pix - yi * W
    yf = pix_ndc(H - 1 - yi, H, W, dt)
    xf = pix_ndc(W - 1 - xi, W, H, dt)
    fv = verts_ndc[faces[face_idx]]  # (n,3,3)
    if sub is not None:
        fv = clip_subtriangles(fv, sub, clib.get_z_clip())
    x0, y0, z0 = fv[:, 0, 0], fv[:, 0, 1], fv[:, 0, 2]
    x1, y1, z1 = fv[:, 1, 0], fv[:, 1, 1], fv[:, 1, 2]
    x2, y2, z2 = fv[:, 2, 0], fv[:, 2, 1], fv[:, 2, 2]
    area = _edge(x2, y2, x0, y0, x1, y1) + K_EPS
    a0 = _edge(xf, yf, x1, y1, x2, y2) / area
    a1 = _edge(xf, yf, x2, y2, x0, y0) / area
    a2 = _edge(xf, yf, x0, y0, x1, y1) / area
    t0 = a0 * z1 * z2
    t1 = z0 * a1 * z2
    t2 = z0 * z1 * a2
    den = torch.clamp((t0 + t1) + t2, min=K_EPS)
    w0, w1, w2 = t0 / den, t1 / den, t2 / den
    c0, c1, c2 = w0.clamp(min=0.0), w1.clamp(min=0.0), w2.clamp(min=0.0)
    s = torch.clamp((c0 + c1) + c2, min=1e-5)
    c0, c1, c2 = c0 / s, c1 / s, c2 / s
    pz = (c0 * z0 + c1 * z1) + c2 * z2
    d01 = _seg_d2(xf, yf, x0, y0, x1, y1)
    d02 = _seg_d2(xf, yf, x0, y0, x2, y2)
    d12 = _seg_d2(xf, yf, x1, y1, x2, y2)
    dist = torch.minimum(torch.minimum(d01, d02), d12)
    inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
    sdist = torch.where(inside, -dist, dist)
    return pz, torch.stack([c0, c1, c2], dim=1), sdist, inside


def rasterize_select(verts_ndc, faces, H, W, blur_radius, K_sil=100):
    """Run the C oracle on detached float32 NDC vertices."""
    fv = verts_ndc.detach().to(torch.float32)[faces].numpy()
    return clib.render_pass(fv, H, W, float(blur_radius), K_sil)


# --------------------------------------------------------------------------------------
# shaders (PhongNormalShader PL:74-92 + pytorch3d softmax_rgb_blend; SoftSilhouetteShader RUN:113-116)
# --------------------------------------------------------------------------------------
def blur_radius_from_sigma(sigma=1e-8):
    """RUN:97: np.log(1/1e-4 - 1) * sigma (sigma is a float32 tensor there)."""
    return float(np.float32(np.log(1.0 / 1e-4 - 1.0) * np.float32(sigma)))


def render_normals(verts_world, faces, cam, sel, sigma=1e-8, gamma=1e-8):
    """renderer(mesh) of RUN:102-105 -> (H,W,4) RGBA 'normal colour' image + zbuf (H,W).

    sel = rasterize_select(...) result for the same mesh.  Fragments are re-evaluated
    differentiably from pix_to_face; K=1 blend per SURVEY A.4."""
    H, W = cam.H, cam.W
    dt = verts_world.dtype
    ndc = world_to_ndc(verts_world, cam)
    p2f = torch.from_numpy(sel["pix_to_face"]).reshape(-1)
    hit = (p2f >= 0).nonzero(as_tuple=True)[0]
    fidx = p2f[hit]
    sub = torch.from_numpy(sel["sub"]).reshape(-1).to(torch.int64)[hit] if "sub" in sel else None
    pz, bary, sdist, _ = eval_fragments(ndc, faces, hit, fidx, H, W, sub=sub)
    vn = vertex_normals(verts_world, faces)
    fn = vn[faces[fidx]]  # (n,3,3)
    col = (fn[:, 0] + fn[:, 1]) + fn[:, 2]  # interpolate_face_attributes with bary := ones (PL:85-88)
    sig = torch.tensor(sigma, dtype=torch.float32).to(dt)
    gam = torch.tensor(gamma, dtype=torch.float32).to(dt)
    eps = 1e-10
    prob = torch.sigmoid(-sdist / sig)
    z_inv = (cam.zfar - pz) / (cam.zfar - cam.znear)
    z_inv_max = z_inv.clamp(min=eps)
    wnum = prob * torch.exp((z_inv - z_inv_max) / gam)
    delta = torch.exp((eps - z_inv_max) / gam).clamp(min=eps)
    denom = wnum + delta
    rgb_hit = (wnum[:, None] * col + delta[:, None] * 1.0) / denom[:, None]
    rgba = torch.ones(H * W, 4, dtype=dt)
    rgba[:, 3] = 0.0
    rgba = rgba.index_put((hit,), torch.cat([rgb_hit, (1.0 - (1.0 - prob))[:, None]], dim=1))
    zbuf = torch.full((H * W,), -1.0, dtype=dt).index_put((hit,), pz)
    return rgba.reshape(H, W, 4), zbuf.reshape(H, W)


def render_silhouette(verts_world, faces, cam, sel, sigma=1e-8):
    """sil_renderer(mesh)[..., 3] of RUN:113-116: alpha = 1 - prod_k(1 - sigmoid(-d_k/sigma))."""
    H, W = cam.H, cam.W
    dt = verts_world.dtype
    ndc = world_to_ndc(verts_world, cam)
    pairs = torch.from_numpy(sel["pairs"])
    alpha = torch.zeros(H * W, dtype=dt)
    if pairs.shape[0] == 0:
        return alpha.reshape(H, W)
    pix, fidx = pairs[:, 0], pairs[:, 1]
    _, _, sdist, _ = eval_fragments(ndc, faces, pix, fidx, H, W, sub=pairs[:, 2] if pairs.shape[1] > 2 else None)
    sig = torch.tensor(sigma, dtype=torch.float32).to(dt)
    one_minus = 1.0 - torch.sigmoid(-sdist / sig)
    # dense (n_hit_pixels, Kmax) layout; pairs arrive grouped by pixel and sorted by z
    upix, inv, cnt = torch.unique_consecutive(pix, return_inverse=True, return_counts=True)
    start = torch.cumsum(cnt, 0) - cnt
    slot = torch.arange(pix.shape[0]) - start[inv]
    dense = torch.ones(upix.shape[0], int(cnt.max()), dtype=dt).index_put((inv, slot), one_minus)
    a = 1.0 - torch.prod(dense, dim=1)
    return alpha.index_put((upix,), a).reshape(H, W)


def render_normal_and_disparity(rgba, zbuf):
    """PL:272-289 on the (H,W,4) colour image and (H,W) zbuf of one mesh."""
    alpha = rgba[..., 3]
    mask = alpha > 0.0
    n = rgba[..., :3]
    nn = (n - n.min()) / (n.max() - n.min() + 1e-6)
    nn = torch.where(mask[..., None], nn, torch.zeros_like(nn))
    depth = torch.where(zbuf < 0, torch.full_like(zbuf, 10.0), zbuf)
    disp = 1 / (depth + 1e-6)
    disp = (disp - disp.min()) / (disp.max() - disp.min() + 1e-6)
    return nn, disp


# --------------------------------------------------------------------------------------
# loss heads (PL:178-186, PL:231-239, PL:1340-1342, PL:1529-1541, PL:1567-1576)
# --------------------------------------------------------------------------------------
def normal_alignment_loss(rendered, gt, valid_mask=None):
    """PL:178-186."""
    r = torch.nn.functional.normalize(rendered, dim=-1)
    g = torch.nn.functional.normalize(gt, dim=-1)
    loss = 1 - torch.sum(r * g, dim=-1)
    if valid_mask is not None:
        loss = loss[valid_mask]
    return loss.mean()


def honerf_intersection_loss(sdf_hand, sdf_obj):
    """PL:231-239 -- a count, no gradient."""
    inner = sdf_obj < 0
    return (sdf_hand[inner] < 0).sum() / 1000


def knn1(p1, p2):
    """pytorch3d knn_points(K=1) (PL:1529-1532): squared distances, differentiable to both clouds."""
    _, idx = clib.knn1(p1.detach().to(torch.float32).numpy(), p2.detach().to(torch.float32).numpy())
    idx = torch.from_numpy(idx)
    d = p1 - p2[idx]
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], idx


def unique_edges(faces):
    """pytorch3d Meshes.edges_packed(): unique undirected edges, sorted by (min*V + max)."""
    f = faces.numpy() if isinstance(faces, torch.Tensor) else faces
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
    e = np.sort(e, axis=1)
    e = np.unique(e, axis=0)
    return torch.from_numpy(e.astype(np.int64))


def mesh_edge_loss(verts, edges):
    """pytorch3d.loss.mesh_edge_loss(target_length=0) (PL:1430, PL:1575)."""
    d = verts[edges[:, 0]] - verts[edges[:, 1]]
    n = torch.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    return ((n - 0.0) ** 2.0).sum() / edges.shape[0]


def mano_vert_to_3dkps(verts, J_regressor):
    """PL:121-135."""
    tips = torch.tensor([744, 320, 443, 554, 671], dtype=torch.int64)
    order = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]
    # a two-hand scene (BASELINE config 4) stacks both hands; the (16,778) regressor covers the first one
    kp = torch.cat([J_regressor.to(verts.dtype) @ verts[:J_regressor.shape[1]], verts[tips]], dim=0)
    return kp[order, :]


# --------------------------------------------------------------------------------------
# SDF grid / inside test (SDF:131-160, SDF:88-109; kaolin check_sign = ray parity)
# --------------------------------------------------------------------------------------
def dense_grid_points(bmin, bmax, res):
    """SDF:26-45 / PL:341-360 with indexing='ij': (res+1)^3 points, x-major."""
    x = np.linspace(bmin[0], bmax[0], res + 1, dtype=np.float32)
    y = np.linspace(bmin[1], bmax[1], res + 1, dtype=np.float32)
    z = np.linspace(bmin[2], bmax[2], res + 1, dtype=np.float32)
    xs, ys, zs = np.meshgrid(x, y, z, indexing="ij")
    return np.stack((xs, ys, zs), axis=-1).reshape(-1, 3)


def joint_grid(v1, v2, res=64):
    """SDF:138-154: grid over the joint AABB of two meshes (detached)."""
    a, b = v1.detach().to(torch.float32), v2.detach().to(torch.float32)
    bmin = torch.minimum(a.min(0)[0], b.min(0)[0]).numpy()
    bmax = torch.maximum(a.max(0)[0], b.max(0)[0]).numpy()
    return dense_grid_points(bmin, bmax, res)


def mesh_sdf(verts, faces, grid):
    """SDF:88-109: sqrt(point_to_mesh_distance) * (-1 inside, +1 outside)."""
    v = verts.detach().to(torch.float32).numpy()
    f = faces.numpy().astype(np.int32)
    d2, _ = clib.point_mesh_dist(v, f, grid)
    ins = clib.inside(v, f, grid)
    return np.sqrt(d2) * np.where(ins, -1.0, 1.0).astype(np.float32)


def intersection_count(v1, f1, v2, f2, res=64):
    """PL:1553-1554 reduced to what the loss consumes: #grid points inside both meshes."""
    grid = joint_grid(v1, v2, res)
    a = clib.inside(v1.detach().to(torch.float32).numpy(), f1.numpy().astype(np.int32), grid)
    b = clib.inside(v2.detach().to(torch.float32).numpy(), f2.numpy().astype(np.int32), grid)
    return int(np.count_nonzero(a & b))
