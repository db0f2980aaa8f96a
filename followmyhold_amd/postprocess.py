"""Mesh post-processing after the guided pipeline.

Reference: src/foho/guidance/run.py:159-164 --
    obj_mesh = FloaterRemover()(obj_mesh); obj_mesh = DegenerateFaceRemover()(obj_mesh); obj_mesh = FaceReducer()(obj_mesh)
with the three classes imported from hy3dgen.shapegen (RUN:33), thin wrappers over pymeshlab filters.  Neither hy3dgen
nor pymeshlab is available on the MI355X image, so the filters are restated (parity unpinned):

  FloaterRemover         compute_selection_by_small_disconnected_components_per_face(nbfaceratio=0.005) + delete: drop
                         every connected component with fewer faces than 0.5 % of the largest one
  DegenerateFaceRemover  a save / reload round trip whose effect is to drop degenerate faces (repeated indices, zero
                         area) and vertices no face uses
  FaceReducer            meshing_decimation_quadric_edge_collapse(targetfacenum=max_facenum=40000, ...) when the mesh has
                         more faces than that: `foho_mesh_decimate` (host C++ in libfoho_hip.so, csrc/mesh_decimate.inc)

All three take and return a `TriMesh` (vertices (V,3) float32, faces (F,3) int64, `.export(path)`), and accept anything
with `.vertices` / `.faces` or a (vertices, faces) pair.
"""
import ctypes

import numpy as np

from . import _lib as L
from . import meshio


class TriMesh:
    def __init__(self, vertices, faces):
        self.vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        self.faces = np.ascontiguousarray(faces, np.int64).reshape(-1, 3)

    def export(self, path):
        if str(path).lower().endswith(".obj"):
            meshio.save_obj(path, self.vertices, self.faces)
        else:
            meshio.save_ply(path, self.vertices, self.faces)


def _as_mesh(m):
    if isinstance(m, TriMesh):
        return m
    if isinstance(m, (tuple, list)):
        return TriMesh(m[0], m[1])
    return TriMesh(np.asarray(m.vertices), np.asarray(m.faces))


def _drop_unused_vertices(v, f):
    used = np.zeros(len(v), bool)
    used[f.reshape(-1)] = True
    remap = np.cumsum(used) - 1
    return v[used], remap[f]


def face_components(n_verts, faces):
    """Connected-component label of every face (faces that share a vertex are connected) and the component sizes."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    f = np.asarray(faces, np.int64)
    i = np.concatenate([f[:, 0], f[:, 1]])
    j = np.concatenate([f[:, 1], f[:, 2]])
    g = coo_matrix((np.ones(len(i), np.int8), (i, j)), shape=(n_verts, n_verts))
    _, vlabel = connected_components(g, directed=False)
    flabel = vlabel[f[:, 0]]
    uniq, inv, counts = np.unique(flabel, return_inverse=True, return_counts=True)
    return inv, counts


class FloaterRemover:
    def __init__(self, nbfaceratio=0.005):
        self.nbfaceratio = nbfaceratio

    def __call__(self, mesh):
        m = _as_mesh(mesh)
        if len(m.faces) == 0:
            return m
        label, counts = face_components(len(m.vertices), m.faces)
        keep = counts[label] >= self.nbfaceratio * counts.max()
        v, f = _drop_unused_vertices(m.vertices, m.faces[keep])
        return TriMesh(v, f)


class DegenerateFaceRemover:
    def __call__(self, mesh):
        m = _as_mesh(mesh)
        f, v = m.faces, m.vertices.astype(np.float64)
        ok = (f[:, 0] != f[:, 1]) & (f[:, 1] != f[:, 2]) & (f[:, 0] != f[:, 2])
        t = v[f]
        ok &= np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1) > 0
        v2, f2 = _drop_unused_vertices(m.vertices, f[ok])
        return TriMesh(v2, f2)


def decimate(vertices, faces, target_faces):
    """foho_mesh_decimate: quadric edge-collapse decimation to at most `target_faces` faces (fewer collapses when the
    link condition / normal-flip test leave no admissible edge)."""
    lib = L.lib()
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    f = np.ascontiguousarray(faces, np.int64).reshape(-1, 3)
    ov, of = np.empty_like(v), np.empty_like(f)
    counts = np.zeros(2, np.int32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    L.check(lib.foho_mesh_decimate(P(v), len(v), P(f), len(f), int(target_faces), P(ov), P(of), P(counts)), "foho_mesh_decimate")
    return ov[:counts[0]].copy(), of[:counts[1]].copy()


class FaceReducer:
    def __call__(self, mesh, max_facenum: int = 40000):
        m = _as_mesh(mesh)
        if len(m.faces) <= max_facenum:
            return m
        return TriMesh(*decimate(m.vertices, m.faces, max_facenum))
