"""pytorch3d / kaolin shaped operator facade over libfoho_hip.so (SURVEY.md 8(b) "Operator API the loop calls").

The reference's patched pipeline talks to its geometry libraries through a small surface:
`renderer(mesh)`, `renderer.rasterizer(mesh).zbuf`, `cameras.transform_points_screen`, `sil_renderer(mesh)[..., 3]`,
`Meshes(...)` + a handful of accessors, `join_meshes_as_scene`, `knn_points`, `mesh_edge_loss`,
`quaternion_to_matrix`, `load_ply`, `IO`, `kaolin_sdf.get_sdf_of_meshes` (pipelines.py:54-70, 272-289, 1223-1227,
1324-1336, 1529-1553; run.py:84-116).  This module answers that surface with N=1 batches (the reference's
batch size, guid_config.py:9).  Rasterisation, nearest neighbours and the SDF pieces are HIP operators behind
torch.autograd.Functions; the glue the reference itself writes in torch (shader blend, normalisation) stays in
torch ops on the device.  For throughput use followmyhold_amd.engine.GuidanceBatch (the fused step) instead.

The K=100 silhouette alpha is differentiable (foho_raster_sil_bwd: through every fragment of the pixels with fractional
coverage); its sigma comes from the shader's blend_params.
"""
import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import meshio, ops


# ------------------------------------------------------------------------------------------------ structures
class TexturesVertex:
    """Placeholder: the path's shaders never read vertex colours (pipelines.py:1224-1226, 1400-1402)."""

    def __init__(self, verts_features=None):
        self.verts_features = verts_features

    def to(self, device):
        return self


class Meshes:
    def __init__(self, verts, faces, textures=None):
        v = verts[0] if isinstance(verts, (list, tuple)) else (verts[0] if verts.dim() == 3 else verts)
        f = faces[0] if isinstance(faces, (list, tuple)) else (faces[0] if faces.dim() == 3 else faces)
        self._v, self._f, self.textures = v, f.to(torch.int64), textures
        self._edges = None

    device = property(lambda self: self._v.device)

    def to(self, device):
        return Meshes([self._v.to(device)], [self._f.to(device)], self.textures)

    def clone(self):
        return Meshes([self._v.clone()], [self._f.clone()], self.textures)

    def verts_padded(self):
        return self._v.unsqueeze(0)

    def verts_packed(self):
        return self._v

    def faces_padded(self):
        return self._f.unsqueeze(0)

    def faces_packed(self):
        return self._f

    def update_padded(self, new_verts_padded):
        m = Meshes([new_verts_padded[0]], [self._f], self.textures)
        m._edges = self._edges
        return m

    def edges_packed(self):
        if self._edges is None:
            f = self._f
            e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
            e = torch.sort(e, dim=1)[0]
            self._edges = torch.unique(e, dim=0)
        return self._edges

    def verts_normals_packed(self):
        """Area-weighted vertex normals, normalised with eps 1e-6 (pytorch3d Meshes._compute_vertex_normals)."""
        v, f = self._v, self._f
        fn = torch.cross(v[f[:, 2]] - v[f[:, 1]], v[f[:, 0]] - v[f[:, 1]], dim=1)
        vn = torch.zeros_like(v).index_add(0, f[:, 0], fn).index_add(0, f[:, 1], fn).index_add(0, f[:, 2], fn)
        return torch.nn.functional.normalize(vn, eps=1e-6, dim=1)


def join_meshes_as_scene(meshes: Sequence[Meshes], include_textures: bool = True) -> Meshes:
    vs, fs, off = [], [], 0
    for m in meshes:
        vs.append(m.verts_packed())
        fs.append(m.faces_packed() + off)
        off += m.verts_packed().shape[0]
    return Meshes([torch.cat(vs, 0)], [torch.cat(fs, 0)])


# ------------------------------------------------------------------------------------------------ transforms / camera
def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


class FoVPerspectiveCameras:
    def __init__(self, device="cuda", R=None, T=None, znear=0.01, zfar=100.0, fov=60.0, aspect_ratio=1.0, degrees=True):
        self.device = torch.device(device)
        self.R = (torch.eye(3)[None] if R is None else R).to(self.device, torch.float32)
        self.T = (torch.zeros(1, 3) if T is None else T).to(self.device, torch.float32)
        self.znear, self.zfar = float(znear), float(zfar)
        f = float(fov)
        self.fov = f if degrees else math.degrees(f)
        from .engine import fov_focal
        self.k00, self.k11 = fov_focal(self.fov, aspect_ratio, znear)

    def to(self, device):
        return self

    def view_points(self, pts):
        return pts @ self.R[0] + self.T[0]

    def transform_points_ndc(self, pts):
        v = self.view_points(pts)
        return torch.stack([self.k00 * v[..., 0] / v[..., 2], self.k11 * v[..., 1] / v[..., 2], v[..., 2]], -1)

    def transform_points_screen(self, pts, image_size=None, **_):
        H, W = image_size
        n = self.transform_points_ndc(pts)
        s = min(H, W) / 2.0
        return torch.stack([W / 2.0 - s * n[..., 0], H / 2.0 - s * n[..., 1], n[..., 2]], -1)


class BlendParams:
    def __init__(self, sigma=1e-4, gamma=1e-4, background_color=(1.0, 1.0, 1.0)):
        self.sigma, self.gamma, self.background_color = float(sigma), float(gamma), background_color


class RasterizationSettings:
    def __init__(self, image_size=256, blur_radius=0.0, faces_per_pixel=1, bin_size=None, max_faces_per_bin=None,
                 perspective_correct=None, clip_barycentric_coords=None, cull_backfaces=False):
        self.image_size = (image_size, image_size) if isinstance(image_size, int) else tuple(image_size)
        self.blur_radius, self.faces_per_pixel = float(blur_radius), int(faces_per_pixel)
        self.bin_size, self.max_faces_per_bin = bin_size, max_faces_per_bin


class Fragments:
    def __init__(self, pix_to_face, zbuf, bary_coords, dists, sil_prod=None):
        self.pix_to_face, self.zbuf, self.bary_coords, self.dists, self.sil_prod = pix_to_face, zbuf, bary_coords, dists, sil_prod


class _RasterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts_ndc, faces, H, W, blur, sigma, want_sil):
        out = ops.raster_fwd(verts_ndc, faces, H, W, blur, sigma, want_sil)
        ov = int(out["overflow"].item())
        if (ov & 4) and want_sil:
            # a pixel with 100 fractional-coverage fragments or more has its K = 100 buffer re-built on the device
            # (kbuffer_fix); this is what is left when even that is beyond its capacities (> 1024 fragments on a pixel, > 32
            # such pixels)
            raise ops.L.FohoError("K = 100 silhouette not reproduced: a pixel holds more than 1024 fragments, or more than 32 pixels "
                                  "hold 100 fractional-coverage fragments")
        ctx.blur, ctx.sigma = float(blur), float(sigma)
        prod = out["sil_prod"] if want_sil else torch.zeros(0, device=verts_ndc.device)
        ctx.save_for_backward(verts_ndc.detach(), faces, out["pix_to_face"], prod)
        ctx.mark_non_differentiable(out["pix_to_face"])
        return out["pix_to_face"], out["zbuf"], out["bary"], out["dists"], prod

    @staticmethod
    def backward(ctx, g_p2f, g_z, g_b, g_d, g_prod):
        v, f, p2f, prod = ctx.saved_tensors
        g = ops.raster_bwd(v, f, p2f, g_z, g_b, g_d, blur_radius=ctx.blur)
        if prod.numel() and g_prod is not None:      # SoftSilhouetteShader's alpha = 1 - prod (RUN:106-116, PL:1341, 1423, 1569)
            g = g + ops.raster_sil_bwd(v, f, prod, g_prod, ctx.blur, ctx.sigma)
        return g, None, None, None, None, None, None


class MeshRasterizer:
    def __init__(self, cameras=None, raster_settings=None):
        self.cameras, self.raster_settings = cameras, raster_settings or RasterizationSettings()

    def to(self, device):
        return self

    def transform(self, meshes, **kwargs):
        cams = kwargs.get("cameras", self.cameras)
        return cams.transform_points_ndc(meshes.verts_packed())

    def __call__(self, meshes, **kwargs) -> Fragments:
        rs = self.raster_settings
        H, W = rs.image_size
        ndc = self.transform(meshes, **kwargs)
        want_sil = rs.faces_per_pixel > 1
        # sigma of the silhouette product: the renderer hands over its shader's blend_params.sigma; a bare rasteriser call
        # takes the sigma its blur radius was derived from (blur_radius = log(1 / 1e-4 - 1) sigma, RUN:97)
        sigma = kwargs.get("sigma")
        if sigma is None:
            sigma = rs.blur_radius / math.log(1.0 / 1e-4 - 1.0) if rs.blur_radius > 0 else 1e-8
        p2f, z, b, d, prod = _RasterFn.apply(ndc, meshes.faces_packed(), H, W, rs.blur_radius, float(sigma), want_sil)
        return Fragments(p2f[None, ..., None], z[None, ..., None], b[None, :, :, None, :], d[None, ..., None],
                         prod[None] if want_sil else None)


def interpolate_face_attributes(pix_to_face, barycentric_coords, face_attributes):
    """pytorch3d.ops.interpolate_face_attributes: sum_k bary_k * attr[face, k]; zero for background."""
    mask = pix_to_face < 0
    idx = pix_to_face.clamp(min=0)
    attr = face_attributes[idx]                      # (N,H,W,K,3,D)
    out = (barycentric_coords[..., None] * attr).sum(-2)
    return torch.where(mask[..., None], torch.zeros_like(out), out)


def softmax_rgb_blend(colors, fragments, blend_params, znear=1.0, zfar=100.0):
    """pytorch3d.renderer.blending.softmax_rgb_blend."""
    N, H, W, K = fragments.pix_to_face.shape
    eps = 1e-10
    mask = fragments.pix_to_face >= 0
    prob = torch.sigmoid(-fragments.dists / blend_params.sigma) * mask
    alpha = torch.prod(1.0 - prob, dim=-1)
    z_inv = (zfar - fragments.zbuf) / (zfar - znear) * mask
    z_inv_max = torch.max(z_inv, dim=-1).values[..., None].clamp(min=eps)
    wnum = prob * torch.exp((z_inv - z_inv_max) / blend_params.gamma)
    delta = torch.exp((eps - z_inv_max) / blend_params.gamma).clamp(min=eps)
    denom = wnum.sum(dim=-1)[..., None] + delta
    bg = torch.tensor(blend_params.background_color, dtype=colors.dtype, device=colors.device)
    rgb = ((wnum[..., None] * colors).sum(dim=-2) + delta * bg) / denom
    return torch.cat([rgb, (1.0 - alpha)[..., None]], dim=-1)


class ShaderBase:
    def __init__(self, device="cpu", cameras=None, lights=None, materials=None, blend_params=None):
        self.cameras, self.blend_params = cameras, blend_params or BlendParams()

    def to(self, device):
        return self

    def __call__(self, fragments, meshes, **kwargs):
        return self.forward(fragments, meshes, **kwargs)


class PhongNormalShader(ShaderBase):
    """The reference's normal-map shader (pipelines.py:74-92): colour = sum of the hit face's 3 vertex normals."""

    def forward(self, fragments, meshes, **kwargs):
        cameras = kwargs.get("cameras", self.cameras)
        blend_params = kwargs.get("blend_params", self.blend_params)
        faces = meshes.faces_packed()
        faces_normals = meshes.verts_normals_packed()[faces]
        ones = torch.ones_like(fragments.bary_coords)
        pixel_normals = interpolate_face_attributes(fragments.pix_to_face, ones, faces_normals)
        return softmax_rgb_blend(pixel_normals, fragments, blend_params, znear=cameras.znear, zfar=cameras.zfar)


class SoftSilhouetteShader(ShaderBase):
    """alpha = 1 - prod_k(1 - sigmoid(-d_k / sigma)) over the K nearest fragments (run.py:113-116)."""

    def forward(self, fragments, meshes, **kwargs):
        if fragments.sil_prod is None:
            raise ValueError("SoftSilhouetteShader needs a rasterizer with faces_per_pixel > 1")
        a = (1.0 - fragments.sil_prod)
        rgb = torch.ones(a.shape + (3,), device=a.device, dtype=a.dtype)
        return torch.cat([rgb, a[..., None]], dim=-1)


class MeshRenderer:
    def __init__(self, rasterizer, shader):
        self.rasterizer, self.shader = rasterizer, shader

    def to(self, device):
        return self

    def __call__(self, meshes_world, **kwargs):
        bp = kwargs.get("blend_params", getattr(self.shader, "blend_params", None))
        if bp is not None and "sigma" not in kwargs:
            kwargs = dict(kwargs, sigma=bp.sigma)
        return self.shader(self.rasterizer(meshes_world, **kwargs), meshes_world, **kwargs)


# ------------------------------------------------------------------------------------------------ ops / losses
class _Knn1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p1, p2):
        d2, idx = ops.knn1(p1, p2)
        ctx.save_for_backward(p1.detach(), p2.detach(), idx)
        ctx.mark_non_differentiable(idx)
        return d2, idx

    @staticmethod
    def backward(ctx, g, _):
        p1, p2, idx = ctx.saved_tensors
        diff = 2.0 * (p1 - p2[idx]) * g[:, None]
        return diff, torch.zeros_like(p2).index_add(0, idx, -diff)


def knn_points(p1, p2, K=1, **_):
    """pytorch3d.ops.knn_points for N=1, K=1: returns (dists (1,P1,1) squared, idx (1,P1,1), nn=None)."""
    if K != 1 or p1.shape[0] != 1:
        raise NotImplementedError("the guidance path only uses K=1 with batch size 1 (pipelines.py:1529-1538)")
    d2, idx = _Knn1Fn.apply(p1[0], p2[0])
    return d2[None, :, None], idx[None, :, None], None


def mesh_edge_loss(meshes: Meshes, target_length: float = 0.0):
    e, v = meshes.edges_packed(), meshes.verts_packed()
    d = v[e[:, 0]] - v[e[:, 1]]
    return ((d.norm(dim=1, p=2) - target_length) ** 2.0).sum() / e.shape[0]


def load_ply(path):
    v, f = meshio.load_ply(path)
    return torch.from_numpy(v), torch.from_numpy(f)


class IO:
    def register_meshes_format(self, fmt):
        pass

    def load_mesh(self, path, **_):
        v, f = meshio.load_mesh(path)
        return Meshes([torch.from_numpy(v)], [torch.from_numpy(f)])

    def save_mesh(self, mesh: Meshes, path, **_):
        meshio.save_ply(path, mesh.verts_packed().detach().cpu().numpy(), mesh.faces_packed().cpu().numpy())


# ------------------------------------------------------------------------------------------------ kaolin_sdf_ops
def generate_dense_grid_points(bbox_min, bbox_max, octree_depth, indexing="ij", octree_resolution=None):
    """kaolin_sdf_ops.py:26-45 / pipelines.py:341-360."""
    length = bbox_max - bbox_min
    n = int(octree_resolution if octree_resolution is not None else np.exp2(octree_depth))
    axes = [np.linspace(bbox_min[k], bbox_max[k], n + 1, dtype=np.float32) for k in range(3)]
    xs, ys, zs = np.meshgrid(*axes, indexing=indexing)
    return np.stack((xs, ys, zs), axis=-1).reshape(-1, 3), [n + 1] * 3, length


def mesh2sdf(mesh: Meshes, grid_points=None, device="cuda", resolution=64):
    """kaolin_sdf_ops.py:88-109: sqrt(point_to_mesh_distance) * (-1 inside, +1 outside)."""
    v, f = mesh.verts_packed().detach(), mesh.faces_packed()
    d2, _ = ops.point_mesh_dist(v, f, grid_points)
    ins = ops.inside_points(v, f, grid_points)
    return torch.sqrt(d2) * torch.where(ins, -1.0, 1.0)


def get_sdf_of_meshes(mesh1: Meshes, mesh2: Meshes, device, resolution=64):
    """kaolin_sdf_ops.py:131-160: SDFs of two meshes on the grid over their joint AABB."""
    a, b = mesh1.verts_packed().detach(), mesh2.verts_packed().detach()
    bmin = torch.minimum(a.min(0)[0], b.min(0)[0]).cpu().numpy()
    bmax = torch.maximum(a.max(0)[0], b.max(0)[0]).cpu().numpy()
    grid, _, _ = generate_dense_grid_points(bmin, bmax, 5, "ij", resolution)
    grid = torch.from_numpy(grid).to(a.device)
    return mesh2sdf(mesh1, grid, device, resolution), mesh2sdf(mesh2, grid, device, resolution)


class FlexiCubes:
    """kaolin.non_commercial.FlexiCubes as the pipeline uses it (pipelines.py:1142-1143, 1393, 1509): default weights, no
    training mode -- Dual Marching Cubes on the regular grid, differentiable w.r.t. the SDF (libfoho_hip: k_flexi.inc)."""

    _CORNERS = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1)]

    def __init__(self, device="cuda", **_):
        self.device = torch.device(device)

    def construct_voxel_grid(self, res):
        """((res+1)^3, 3) grid points in [-0.5, 0.5]^3 (x-major) and the (res^3, 8) corner indices of the cubes."""
        G = res + 1
        lin = torch.linspace(-0.5, 0.5, G, device=self.device)
        xs, ys, zs = torch.meshgrid(lin, lin, lin, indexing="ij")
        verts = torch.stack([xs, ys, zs], -1).reshape(-1, 3)
        i, j, k = torch.meshgrid(*([torch.arange(res, device=self.device)] * 3), indexing="ij")
        base = ((i * G + j) * G + k).reshape(-1)
        offs = torch.tensor([(cx * G + cy) * G + cz for cx, cy, cz in self._CORNERS], device=self.device)
        return verts, base[:, None] + offs[None, :]

    def __call__(self, x_nx3, s_n, cube_fx8, res, beta_fx12=None, alpha_fx8=None, gamma_f=None, training=False, **_):
        if beta_fx12 is not None or alpha_fx8 is not None or gamma_f is not None or training:
            raise NotImplementedError("only the default-weight call of pipelines.py:1393 / 1509 is implemented")
        if cube_fx8 is not None and cube_fx8.shape[0] != res ** 3:
            raise ValueError("cube_fx8 must be the regular grid of construct_voxel_grid(res)")
        return ops.flexicubes(x_nx3, s_n, int(res))
