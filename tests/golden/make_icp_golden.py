"""Golden vectors of the REFERENCE's ICP loop (src/foho/alignment/mesh_align.py:56-175 `icp`, :25-35 `compute_init_transform`).

Run in the build container only (needs /root/reference; the GPU box never runs this):
    python tests/golden/make_icp_golden.py [output directory]
It imports the reference's mesh_align.py itself.  What is not installed is bound to this repository's restatements (oracle/icp_ref.py),
exactly as make_pipeline_golden.py does for pytorch3d / kaolin: `trimesh` (PointCloud, transform_points, transformations.*,
registration.procrustes, proximity.closest_point), `pyvista` and `click` (decorators only).  What IS installed stays real: numpy, tqdm and
`scipy.spatial.cKDTree` -- the nearest-neighbour search of ICP:108 is scipy's own.  The reference's `icp()` then runs on seeded point clouds
(tests/helpers.py::icp_case_inputs) and its results go to tests/golden/ref_icp.npz: best transform, best cost, and the per-iteration
costs / transforms (recorded through the procrustes binding: the function keeps its records local).

What this pins: the reference's loop -- start transforms (identity, reflections, rotations), outlier trimming by sorted distance, the
cost-before / transform-after bookkeeping (ICP:129-142), the scale clip, best-of-all selection -- and scipy's kd-tree against the
restatement's exhaustive search.  trimesh's own arithmetic (procrustes, closest_point) is this repository's restatement (parity unpinned).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = sys.argv[1] if len(sys.argv) > 1 else HERE
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/src/foho/alignment/mesh_align.py"

from oracle import icp_ref   # noqa: E402

RECORD = []      # one (cost of the matched pairs, next_transform) per procrustes call


class PointCloud:
    def __init__(self, vertices):
        self.vertices = np.asarray(vertices, np.float64)


class TriMeshStub:
    """what `closest_point(target_mesh, p)` of ICP:107 is handed: vertices and faces"""

    def __init__(self, vertices, faces):
        self.vertices, self.faces = np.asarray(vertices, np.float64), np.asarray(faces, np.int64)


def _procrustes(a, b, reflection=True, translation=True, scale=True, return_cost=True):
    M = icp_ref.procrustes(a, b, reflection=reflection, scale=scale)
    RECORD.append((float(np.linalg.norm(np.asarray(a) - np.asarray(b), axis=1).mean()), M.copy()))
    return M if not return_cost else (M, None, None)


def _closest_point(mesh, points):
    q, dist = icp_ref.closest_point(mesh.vertices, mesh.faces, np.asarray(points, np.float64))
    return q, dist, None


def import_reference():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    ident = lambda *a, **k: (lambda f: f)
    mod("click", command=ident, argument=ident, option=ident, Path=lambda **k: None)
    mod("pyvista")
    tr = mod("trimesh.transformations", translation_matrix=icp_ref.translation_matrix, scale_matrix=icp_ref.scale_matrix,
             rotation_matrix=icp_ref.rotation_matrix)
    reg = mod("trimesh.registration", procrustes=_procrustes)
    prox = mod("trimesh.proximity", closest_point=_closest_point)
    mod("trimesh", PointCloud=PointCloud, transform_points=icp_ref.transform_points, transformations=tr, registration=reg, proximity=prox,
        sample=types.SimpleNamespace())
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_mesh_align", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    from helpers import ICP_CASES, icp_case_inputs
    ref = import_reference()
    out = {}
    for name in ICP_CASES:
        c = icp_case_inputs(name)
        kw = dict(c["kw"])
        src = PointCloud(c["src"])
        tgt = TriMeshStub(c["tgt"], c["tgt_faces"]) if kw.get("on_surface") else PointCloud(c["tgt"])
        if kw.get("on_surface"):        # ICP:83-86 samples a mesh target; on_surface never reads the samples (ICP:106-107), so hand it the vertices
            ref.tm.sample.sample_surface_even = lambda mesh, count: (mesh.vertices, None)
        del RECORD[:]
        T, cost = ref.icp(src, tgt, n_iter=c["n_iter"], **kw)
        out[f"{name}_T"], out[f"{name}_cost"] = np.asarray(T, np.float64), np.float64(cost)
        out[f"{name}_iter_cost"] = np.array([r[0] for r in RECORD])          # all starts, one after the other (starts x n_iter)
        out[f"{name}_iter_next"] = np.stack([r[1] for r in RECORD])
        print(f"{name}: {len(RECORD)} iterations, best cost {cost:.9g}", flush=True)
    # compute_init_transform (ICP:25-35) on point clouds, scaled and fixed-scale
    c = icp_case_inputs("coarse")
    for fs in (False, True):
        out[f"init_fixed{int(fs)}"] = ref.compute_init_transform(PointCloud(c["src"] * 1.7 + 0.3), PointCloud(c["tgt"]), fs)
    out["rotations"] = np.stack(ref.get_all_axis_aligned_rotations())
    out["reflections"] = np.stack(ref.get_all_axis_aligned_reflections())
    np.savez_compressed(os.path.join(OUT, "ref_icp.npz"), **out)
    print("wrote", os.path.join(OUT, "ref_icp.npz"))


if __name__ == "__main__":
    main()
