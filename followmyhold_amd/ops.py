"""Python wrappers of the stand-alone C-ABI operators (torch tensors in, torch tensors out; no CPU fallback).

Each function states the reference interface it replaces.  All launches go to the current torch stream.
"""
import ctypes

import numpy as np
import torch

from . import _lib as L

P = ctypes.c_void_p


def _stream(t):
    return P(torch.cuda.current_stream(t.device).cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.FohoError("libfoho_hip operators need CUDA/HIP tensors (there is no CPU fallback)")


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------ rasteriser
def raster_fwd(verts_ndc, faces, H, W, blur_radius, sigma=1e-8, want_sil=True):
    """pytorch3d rasterize_meshes(K=1) (+ the K=100 silhouette product) for one mesh.
    Returns dict(pix_to_face (H,W) int64, zbuf, bary (H,W,3), dists, sil_prod (H,W) or None)."""
    _need_cuda(verts_ndc, faces)
    lib = L.lib()
    v, f = _f32(verts_ndc), faces.detach().to(torch.int32).contiguous()
    V, F = v.shape[0], f.shape[0]
    dev = v.device
    lib.foho_raster_workspace_bytes.restype = ctypes.c_size_t
    nws = lib.foho_raster_workspace_bytes(V, F, H, W)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    p2f = torch.empty(H, W, dtype=torch.int64, device=dev)
    zb, di = torch.empty(H, W, device=dev), torch.empty(H, W, device=dev)
    ba = torch.empty(H, W, 3, device=dev)
    pr = torch.empty(H, W, device=dev) if want_sil else None
    ov = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(lib.foho_raster_fwd(P(v.data_ptr()), P(f.data_ptr()), V, F, H, W, ctypes.c_float(blur_radius),
                                ctypes.c_float(sigma), P(p2f.data_ptr()), P(zb.data_ptr()), P(ba.data_ptr()),
                                P(di.data_ptr()), P(pr.data_ptr()) if want_sil else None, P(ov.data_ptr()),
                                P(ws.data_ptr()), ctypes.c_size_t(nws), _stream(v)), "foho_raster_fwd")
    return dict(pix_to_face=p2f, zbuf=zb, bary=ba, dists=di, sil_prod=pr, overflow=ov, _keep=(v, f, ws))


def raster_bwd(verts_ndc, faces, pix_to_face, grad_zbuf=None, grad_bary=None, grad_dists=None, blur_radius=0.0):
    """Backward of the K=1 fragments -> grad w.r.t. verts_ndc (V,3).  blur_radius: the forward call's."""
    lib = L.lib()
    v, f = _f32(verts_ndc), faces.detach().to(torch.int32).contiguous()
    H, W = pix_to_face.shape[-2:]
    g = torch.zeros_like(v)
    gz = _f32(grad_zbuf) if grad_zbuf is not None else None
    gb = _f32(grad_bary) if grad_bary is not None else None
    gd = _f32(grad_dists) if grad_dists is not None else None
    p2f = pix_to_face.contiguous()
    L.check(lib.foho_raster_bwd(P(v.data_ptr()), P(f.data_ptr()), v.shape[0], f.shape[0], H, W, P(p2f.data_ptr()),
                                P(gz.data_ptr()) if gz is not None else None, P(gb.data_ptr()) if gb is not None else None,
                                P(gd.data_ptr()) if gd is not None else None, P(g.data_ptr()), ctypes.c_float(blur_radius),
                                _stream(v)), "foho_raster_bwd")
    return g


def raster_sil_bwd(verts_ndc, faces, sil_prod, grad_prod, blur_radius, sigma):
    """Backward of the silhouette product of raster_fwd -> grad w.r.t. verts_ndc (V,3)."""
    lib = L.lib()
    v, f = _f32(verts_ndc), faces.detach().to(torch.int32).contiguous()
    H, W = sil_prod.shape[-2:]
    g = torch.zeros_like(v)
    pr, gp = _f32(sil_prod), _f32(grad_prod)
    L.check(lib.foho_raster_sil_bwd(P(v.data_ptr()), P(f.data_ptr()), v.shape[0], f.shape[0], H, W, P(pr.data_ptr()), P(gp.data_ptr()),
                                    P(g.data_ptr()), ctypes.c_float(blur_radius), ctypes.c_float(sigma), _stream(v)), "foho_raster_sil_bwd")
    return g


# ------------------------------------------------------------------------------------------------ knn / sdf
def knn1(p1, p2):
    """pytorch3d.ops.knn_points(K=1): (squared distances (N1,), indices (N1,) int64)."""
    _need_cuda(p1, p2)
    lib = L.lib()
    a, b = _f32(p1), _f32(p2)
    d2 = torch.empty(a.shape[0], device=a.device)
    idx = torch.empty(a.shape[0], dtype=torch.int64, device=a.device)
    L.check(lib.foho_knn1_fwd(P(a.data_ptr()), a.shape[0], P(b.data_ptr()), b.shape[0], P(d2.data_ptr()), P(idx.data_ptr()),
                              _stream(a)), "foho_knn1_fwd")
    return d2, idx


def point_mesh_dist(verts, faces, pts):
    """kaolin point_to_mesh_distance: (squared distance (N,), closest face (N,) int64)."""
    _need_cuda(verts, faces, pts)
    lib = L.lib()
    v, f, p = _f32(verts), faces.detach().to(torch.int32).contiguous(), _f32(pts)
    d2 = torch.empty(p.shape[0], device=p.device)
    fi = torch.empty(p.shape[0], dtype=torch.int64, device=p.device)
    L.check(lib.foho_point_mesh_dist(P(v.data_ptr()), P(f.data_ptr()), v.shape[0], f.shape[0], P(p.data_ptr()), p.shape[0],
                                     P(d2.data_ptr()), P(fi.data_ptr()), _stream(v)), "foho_point_mesh_dist")
    return d2, fi


def inside_points(verts, faces, pts):
    """kaolin check_sign: bool (N,) -- True inside the closed mesh."""
    _need_cuda(verts, faces, pts)
    lib = L.lib()
    v, f, p = _f32(verts), faces.detach().to(torch.int32).contiguous(), _f32(pts)
    out = torch.empty(p.shape[0], dtype=torch.uint8, device=p.device)
    L.check(lib.foho_inside_points(P(v.data_ptr()), P(f.data_ptr()), v.shape[0], f.shape[0], P(p.data_ptr()), p.shape[0],
                                   P(out.data_ptr()), _stream(v)), "foho_inside_points")
    return out.bool()


# ------------------------------------------------------------------------------------------------ LBS
class LbsModel:
    """Device copy of a MANO-shaped model (synthetic.mano_like_model() or the real MANO arrays)."""

    def __init__(self, model, device="cuda"):
        t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a)).to(dt).contiguous().to(device)
        self.v_template = t(model["v_template"])
        self.shapedirs = t(model["shapedirs"])
        self.posedirs = t(model["posedirs"])
        self.J_regressor = t(model["J_regressor"])
        self.lbs_weights = t(model["lbs_weights"])
        self.parents = t(model["parents"], torch.int32)
        self.V = self.v_template.shape[0]
        assert self.shapedirs.shape == (self.V, 3, 10) and self.posedirs.shape == (135, 3 * self.V)
        assert self.J_regressor.shape == (16, self.V) and self.lbs_weights.shape == (self.V, 16)

    def _args(self):
        return [P(x.data_ptr()) for x in (self.v_template, self.shapedirs, self.posedirs, self.J_regressor,
                                          self.lbs_weights, self.parents)] + [self.V]


class _LbsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, betas, rot, model, use_mfma):
        lib = L.lib()
        B = betas.shape[0]
        b, r = _f32(betas), _f32(rot).reshape(B, 16, 3, 3)
        lib.foho_lbs_workspace_bytes.restype = ctypes.c_size_t
        nws = lib.foho_lbs_workspace_bytes(B, model.V)
        ws = torch.empty(nws, dtype=torch.uint8, device=b.device)
        verts = torch.empty(B, model.V, 3, device=b.device)
        joints = torch.empty(B, 16, 3, device=b.device)
        L.check(lib.foho_lbs_fwd(*model._args(), P(b.data_ptr()), P(r.data_ptr()), B, int(use_mfma), P(verts.data_ptr()),
                                 P(joints.data_ptr()), P(ws.data_ptr()), ctypes.c_size_t(nws), _stream(b)), "foho_lbs_fwd")
        ctx.model, ctx.ws, ctx.nws, ctx.B = model, ws, nws, B
        ctx.save_for_backward(r)
        return verts, joints

    @staticmethod
    def backward(ctx, g_verts, g_joints):
        lib = L.lib()
        (r,) = ctx.saved_tensors
        B, model = ctx.B, ctx.model
        gv = _f32(g_verts) if g_verts is not None else torch.zeros(B, model.V, 3, device=r.device)
        gj = _f32(g_joints) if g_joints is not None else None
        gb = torch.empty(B, 10, device=r.device)
        gr = torch.empty(B, 16, 3, 3, device=r.device)
        L.check(lib.foho_lbs_bwd(*model._args(), P(r.data_ptr()), B, P(gv.data_ptr()), P(gj.data_ptr()) if gj is not None else None,
                                 P(gb.data_ptr()), P(gr.data_ptr()), P(ctx.ws.data_ptr()), ctypes.c_size_t(ctx.nws), _stream(r)),
                "foho_lbs_bwd")
        return gb, gr, None, None


def lbs(betas, rot_mats, model: LbsModel, use_mfma=-1):
    """smplx MANOLayer(pose2rot=False): betas (B,10), rot_mats (B,16,3,3) -> verts (B,V,3), posed joints (B,16,3).
    Differentiable w.r.t. betas and rot_mats (hand-derived backward kernels)."""
    _need_cuda(betas, rot_mats)
    return _LbsFn.apply(betas, rot_mats, model, use_mfma)


# ------------------------------------------------------------------------------------------------ ICP
def icp_points_multi(start_points, target_points, n_iter, n_outliers=0, fixed_scale=False, min_scale=0.5, max_scale=2.0,
                     device="cuda", return_history=False, target_faces=None):
    """The icp() loop of src/foho/alignment/mesh_align.py:91-142 for ALL start point sets (S, N, 3) against one target
    (float64), one enqueue and one synchronisation.  Returns (transforms (S,4,4), costs (S,)[, histories (S,n_iter)]).
    target_faces (F,3) switches to on_surface=True (ICP:106-107): target_points are then the target mesh's vertices and
    every source point is matched to the closest point on the triangles."""
    lib = L.lib()
    src = torch.as_tensor(np.ascontiguousarray(np.asarray(start_points, np.float64))).to(device)
    tgt = torch.as_tensor(np.ascontiguousarray(np.asarray(target_points, np.float64))).to(device)
    if src.dim() != 3 or src.shape[2] != 3 or tgt.dim() != 2 or tgt.shape[1] != 3:
        raise L.FohoError("icp_points_multi: expected (S,N,3) start points and (M,3) target points")
    S, N, M = src.shape[0], src.shape[1], tgt.shape[0]
    tf = None
    if target_faces is not None:
        tf = torch.as_tensor(np.ascontiguousarray(np.asarray(target_faces, np.int32))).to(device)
        if tf.dim() != 2 or tf.shape[1] != 3 or tf.shape[0] < 1:
            raise L.FohoError("icp_points_multi: target_faces must be (F,3)")
        if int(tf.min()) < 0 or int(tf.max()) >= M:
            raise L.FohoError("icp_points_multi: target face index out of range")
        lib.foho_icp_surface_workspace_bytes.restype = ctypes.c_size_t
        nws = lib.foho_icp_surface_workspace_bytes(S, N, tf.shape[0])
    else:
        lib.foho_icp_batch_workspace_bytes.restype = ctypes.c_size_t
        nws = lib.foho_icp_batch_workspace_bytes(S, N, M)
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=device)
    T = torch.zeros(S, 16, dtype=torch.float64, device=device)
    cost = torch.zeros(S, dtype=torch.float64, device=device)
    hist = torch.zeros(S, max(n_iter, 1), dtype=torch.float64, device=device)
    tail = (int(n_iter), int(n_outliers), int(bool(fixed_scale)), ctypes.c_double(min_scale), ctypes.c_double(max_scale),
            P(T.data_ptr()), P(cost.data_ptr()), P(hist.data_ptr()), P(ws.data_ptr()), ctypes.c_size_t(nws), _stream(src))
    if tf is not None:
        L.check(lib.foho_icp_run_surface(P(src.data_ptr()), S, N, P(tgt.data_ptr()), M, P(tf.data_ptr()), int(tf.shape[0]), *tail),
                "foho_icp_run_surface")
    else:
        L.check(lib.foho_icp_run_batch(P(src.data_ptr()), S, N, P(tgt.data_ptr()), M, *tail), "foho_icp_run_batch")
    out = (T.cpu().numpy().reshape(S, 4, 4), cost.cpu().numpy())      # the copies synchronise
    return out + (hist.cpu().numpy()[:, :n_iter],) if return_history else out


def icp_points(source_points, target_points, n_iter, n_outliers=0, fixed_scale=False, min_scale=0.5, max_scale=2.0,
               device="cuda", return_history=False, target_faces=None):
    """One start of icp_points_multi.  Returns (best_transform (4,4) np.float64, best_cost float[, cost history])."""
    out = icp_points_multi(np.asarray(source_points, np.float64)[None], target_points, n_iter, n_outliers, fixed_scale,
                           min_scale, max_scale, device, return_history, target_faces)
    return (out[0][0], float(out[1][0])) + ((out[2][0],) if return_history else ())


# ------------------------------------------------------------------------------------------------ iso-surfacing
class _FlexiFn(torch.autograd.Function):
    """kaolin FlexiCubes.__call__ with default weights (pipelines.py:1393, 1509): differentiable w.r.t. the SDF and the
    grid positions through the edge crossings."""

    @staticmethod
    def forward(ctx, x, s, res, verts_cap, faces_cap):
        _need_cuda(x, s)
        lib = L.lib()
        xx, ss = _f32(x), _f32(s).reshape(-1)
        G = res + 1
        if xx.shape != (G ** 3, 3) or ss.numel() != G ** 3:
            raise L.FohoError(f"flexicubes: expected {(G ** 3, 3)} grid positions and {G ** 3} SDF values")
        dev = xx.device
        lib.foho_flexi_workspace_bytes.restype = ctypes.c_size_t
        nws = lib.foho_flexi_workspace_bytes(res)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        counts = torch.zeros(3, dtype=torch.int32, device=dev)
        while True:
            verts = torch.empty(verts_cap, 3, device=dev)
            faces = torch.empty(faces_cap, 3, dtype=torch.int64, device=dev)
            ldev = torch.empty(verts_cap, device=dev)
            L.check(lib.foho_flexi_fwd(P(xx.data_ptr()), P(ss.data_ptr()), res, P(verts.data_ptr()), verts_cap,
                                       P(faces.data_ptr()), faces_cap, P(ldev.data_ptr()), P(counts.data_ptr()),
                                       P(ws.data_ptr()), ctypes.c_size_t(nws), _stream(xx)), "foho_flexi_fwd")
            nv, nf, over = counts.tolist()           # the output sizes are data dependent: one host sync per extraction
            if not over:
                break
            verts_cap, faces_cap = max(verts_cap, nv), max(faces_cap, nf)
        ctx.save_for_backward(xx, ss)
        ctx.res, ctx.ws, ctx.nv = res, ws, nv
        ctx.need_x = x.requires_grad
        ctx.s_meta, ctx.x_meta = (s.shape, s.dtype), (x.shape, x.dtype)
        f_out, l_out = faces[:nf], ldev[:nv]
        ctx.mark_non_differentiable(f_out, l_out)
        return verts[:nv], f_out, l_out

    @staticmethod
    def backward(ctx, g_verts, _gf, _gl):
        xx, ss = ctx.saved_tensors
        lib = L.lib()
        g = _f32(g_verts)
        gs = torch.zeros_like(ss)
        gx = torch.zeros_like(xx) if ctx.need_x else None
        L.check(lib.foho_flexi_bwd(P(xx.data_ptr()), P(ss.data_ptr()), ctx.res, P(g.data_ptr()), ctx.nv, P(gs.data_ptr()),
                                   P(gx.data_ptr()) if gx is not None else None, P(ctx.ws.data_ptr()),
                                   ctypes.c_size_t(ctx.ws.numel()), _stream(xx)), "foho_flexi_bwd")
        # gradients in the shape and dtype the caller's tensors have (a (G,G,G) or half-precision SDF is a legal input)
        gs = gs.view(ctx.s_meta[0]).to(ctx.s_meta[1])
        if gx is not None:
            gx = gx.view(ctx.x_meta[0]).to(ctx.x_meta[1])
        return gx, gs, None, None, None


def flexicubes(x, s, res, verts_cap=None, faces_cap=None):
    """(verts (V,3), faces (F,3) int64, l_dev (V,)) of the zero level set of s on the regular (res+1)^3 grid x."""
    verts_cap = verts_cap or 16 * res * res
    faces_cap = faces_cap or 32 * res * res
    return _FlexiFn.apply(x, s, int(res), int(verts_cap), int(faces_cap))
