"""ORACLE (tests only): pytorch3d / kaolin shaped shims over oracle.ref_ops, on the CPU.

Purpose: let the REFERENCE's own loop code -- `Hunyuan3DDiTFlowMatchingPipeline_main.__call__`
(third_party_patches/hy3dgen/shapegen/pipelines.py:1044-1679) and `utilz.kaolin_sdf_ops.get_sdf_of_meshes` -- execute in
the build container with its un-vendored third-party operators (pytorch3d, kaolin) replaced by this
repository's CPU restatement of them.  tests/golden/make_pipeline_golden.py installs these classes under the third-party
module names, runs the reference loop on a small scene with stand-in networks and stores the trajectory; the GPU test
replays the same inputs through followmyhold_amd.pipeline.  What that pins is the ORCHESTRATION of the reference
(phase order, optimisers, schedules, detach/clone points, transforms, decode / final decode); the operators themselves
remain restatements (parity unpinned).

Only the API surface the reference loop touches is provided (SURVEY.md 8(b) "operator API the loop calls").
"""
import numpy as np
import torch

from . import flexi_ref as FR
from . import ref_ops as R
from . import clib


class TexturesVertex:
    def __init__(self, verts_features=None):
        self.verts_features = verts_features

    def to(self, device):
        return self


class Meshes:
    def __init__(self, verts, faces, textures=None):
        self._v = verts[0] if isinstance(verts, (list, tuple)) else verts
        self._f = faces[0] if isinstance(faces, (list, tuple)) else faces
        if self._v.dim() == 3:
            self._v = self._v[0]
        if self._f.dim() == 3:
            self._f = self._f[0]
        self._f = self._f.to(torch.int64)
        self.textures = textures

    def to(self, device):
        return self

    def clone(self):
        return Meshes(self._v.clone(), self._f.clone(), self.textures)

    def verts_padded(self):
        return self._v.unsqueeze(0)

    def verts_packed(self):
        return self._v

    def faces_padded(self):
        return self._f.unsqueeze(0)

    def faces_packed(self):
        return self._f

    def update_padded(self, new_verts_padded):
        return Meshes(new_verts_padded[0], self._f, self.textures)


def join_meshes_as_scene(meshes, include_textures=True):
    verts, faces, off = [], [], 0
    for m in meshes:
        verts.append(m.verts_packed())
        faces.append(m.faces_packed() + off)
        off += m.verts_packed().shape[0]
    return Meshes(torch.cat(verts, 0), torch.cat(faces, 0))


def quaternion_to_matrix(q):
    return R.quaternion_to_matrix(q)


def knn_points(p1, p2, K=1, **_):
    d, idx = R.knn1(p1[0], p2[0])
    return d.reshape(1, -1, 1), idx.reshape(1, -1, 1), p2[0][idx].reshape(1, -1, 1, 3)


def mesh_edge_loss(meshes, target_length=0.0):
    return R.mesh_edge_loss(meshes.verts_packed(), R.unique_edges(meshes.faces_packed()))


class _Cameras:
    def __init__(self, cam):
        self.cam = cam

    def transform_points_screen(self, pts, image_size=None, **_):
        H, W = image_size
        ndc = R.world_to_ndc(pts.reshape(-1, 3), self.cam)
        xy = R.ndc_to_screen(ndc, H, W)
        return torch.cat([xy, ndc[:, 2:3]], 1).reshape(*pts.shape[:-1], 3)


class _Fragments:
    def __init__(self, zbuf):
        self.zbuf = zbuf


class _Rasterizer:
    def __init__(self, owner):
        self.owner, self.cameras = owner, _Cameras(owner.cam)

    def __call__(self, mesh, **_):
        _, zbuf = self.owner.render(mesh)
        return _Fragments(zbuf.reshape(1, self.owner.cam.H, self.owner.cam.W, 1))


class NormalRenderer:
    """renderer of RUN:102-105: MeshRenderer(MeshRasterizer(K=1, blur from sigma=1e-8), PhongNormalShader)."""

    def __init__(self, fov, H, W):
        self.cam = R.Camera(fov, H, W)
        self.rasterizer = _Rasterizer(self)
        self._last = None

    def _select(self, mesh):
        v, f = mesh.verts_packed(), mesh.faces_packed()
        key = (id(mesh), v.data_ptr())
        if self._last is None or self._last[0] != key:
            sel = R.rasterize_select(R.world_to_ndc(v, self.cam), f, self.cam.H, self.cam.W, R.blur_radius_from_sigma())
            self._last = (key, sel, mesh)   # keeps the mesh alive so that id() stays unique
        return self._last[1]

    def render(self, mesh):
        return R.render_normals(mesh.verts_packed(), mesh.faces_packed(), self.cam, self._select(mesh))

    def __call__(self, mesh, **_):
        rgba, _ = self.render(mesh)
        return rgba.unsqueeze(0)


class SilhouetteRenderer(NormalRenderer):
    """sil_renderer of RUN:106-116: K=100 fragments, SoftSilhouetteShader; only [..., 3] is used."""

    def __call__(self, mesh, **_):
        alpha = R.render_silhouette(mesh.verts_packed(), mesh.faces_packed(), self.cam, self._select(mesh))
        out = torch.ones(1, self.cam.H, self.cam.W, 4, dtype=alpha.dtype)
        return torch.cat([out[..., :3], alpha.reshape(1, self.cam.H, self.cam.W, 1)], -1)


# ------------------------------------------------------------------------------------------------ kaolin
class FlexiCubes:
    def __init__(self, device="cpu", **_):
        self.device = device

    def construct_voxel_grid(self, res):
        return None, None      # the restatement derives cube corners from the grid resolution

    def __call__(self, x_nx3, s_n, cube_fx8, res, **_):
        return FR.flexicubes(x_nx3, s_n, res)


def index_vertices_by_faces(vertices, faces):
    return vertices[:, faces]          # (1, F, 3, 3)


def point_to_mesh_distance(points, face_vertices):
    fv = face_vertices[0].detach().to(torch.float32).numpy()
    F = fv.shape[0]
    v = fv.reshape(-1, 3)
    f = np.arange(3 * F, dtype=np.int32).reshape(F, 3)
    d2, idx = clib.point_mesh_dist(np.ascontiguousarray(v), f, np.ascontiguousarray(points[0].detach().to(torch.float32).numpy()))
    return torch.from_numpy(d2).unsqueeze(0), torch.from_numpy(np.asarray(idx)).unsqueeze(0), None


def check_sign(vertices, faces, points):
    ins = clib.inside(np.ascontiguousarray(vertices[0].detach().to(torch.float32).numpy()), faces.numpy().astype(np.int32),
                      np.ascontiguousarray(points[0].detach().to(torch.float32).numpy()))
    return torch.from_numpy(np.asarray(ins, dtype=bool)).unsqueeze(0)
