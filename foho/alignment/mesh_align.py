"""Two-stage trimmed ICP with scale -- drop-in for the reference's src/foho/alignment/mesh_align.py (ICP below).

Same callables and parameters: `get_centroid_scale`, `compute_init_transform` (ICP:18-35), `icp(...)` (ICP:56-175,
including the optional axis-aligned rotation / reflection starts) and `align_meshes_impl(...)` with its 18
positional parameters (ICP:178-217).  The per-iteration work (nearest neighbour, trimming, procrustes, scale
clipping, best-of bookkeeping) runs in libfoho_hip.so's device-resident loop (`foho_icp_run`, float64); what
stays on the host is file IO, the initial transform and the surface sampling.

Differences that are deliberate:
  * trimesh / pyvista are not available on the MI355X image: meshes are loaded by followmyhold_amd.meshio
    (.ply / .obj); `plot=True` raises NotImplementedError.  `on_surface=True` IS implemented (closest point on the
    target triangles, `foho_icp_run_surface`).
  * `trimesh.sample.sample_surface_even` is unseeded in the reference (ICP:79, ICP:85), so its results differ from run
    to run; here it is restated with a seedable generator and the runs are reproducible: `icp(..., seed=)`, the
    keyword-only `seed` of `align_meshes_impl` (coarse stage: seed, fine stage: seed + 1), the CLI's `--seed` and
    $FOHO_ICP_SEED (default 0; a negative value draws a fresh seed from the OS like the reference does).
"""
import time
from typing import Optional

import numpy as np

from followmyhold_amd import meshio


class Mesh:
    """Vertices + triangles; an empty face array makes it a point cloud (trimesh.PointCloud analogue)."""

    def __init__(self, vertices, faces=None):
        self.vertices = np.asarray(vertices, np.float64).reshape(-1, 3)
        self.faces = np.zeros((0, 3), np.int64) if faces is None else np.asarray(faces, np.int64).reshape(-1, 3)

    @property
    def is_point_cloud(self):
        return len(self.faces) == 0

    @property
    def triangles(self):
        return self.vertices[self.faces]

    @property
    def area_faces(self):
        t = self.triangles
        return 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1)

    def apply_transform(self, M):
        self.vertices = transform_points(self.vertices, M)
        return self


def load(path) -> Mesh:
    v, f = meshio.load_mesh(path)
    return Mesh(v, f)


def transform_points(p, M):
    return np.asarray(p, np.float64) @ M[:3, :3].T + M[:3, 3]


def translation_matrix(t):
    M = np.eye(4)
    M[:3, 3] = t
    return M


def scale_matrix(factor, origin):
    M = np.eye(4) * factor
    M[3, 3] = 1.0
    M[:3, 3] = np.asarray(origin, np.float64) * (1.0 - factor)
    return M


def rotation_matrix(angle, axis):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    M = np.eye(4)
    M[:3, :3] = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)
    return M


def get_centroid_scale(m: Mesh):
    """ICP:18-23.  Point cloud: vertex mean; mesh: area-weighted mean of the triangle centroids.  Scale: AABB diagonal."""
    scale = np.linalg.norm(m.vertices.max(axis=0) - m.vertices.min(axis=0))
    if m.is_point_cloud:
        return m.vertices.mean(axis=0), scale
    a = m.area_faces
    return (m.triangles.mean(axis=1) * a[:, None]).sum(axis=0) / a.sum(), scale


def compute_init_transform(source_mesh: Mesh, target_mesh: Mesh, fixed_scale: bool):
    """ICP:25-35."""
    sc, ss = get_centroid_scale(source_mesh)
    tc, ts = get_centroid_scale(target_mesh)
    T = translation_matrix(tc - sc)
    if fixed_scale:
        return T
    return T @ scale_matrix(ts / ss, sc)


def get_all_axis_aligned_rotations():
    """ICP:37-44."""
    out = []
    for coord in range(3):
        axis = np.zeros(3)
        axis[coord] = 1
        for angle in (-np.pi / 2, np.pi, np.pi / 2):
            out.append(rotation_matrix(angle, axis))
    return out


def get_all_axis_aligned_reflections():
    """ICP:46-54."""
    return [np.eye(4) * np.append(d, 1) for d in ([1, 1, -1], [1, -1, 1], [-1, 1, 1], [-1, -1, 1], [-1, 1, -1],
                                                   [1, -1, -1], [-1, -1, -1])]


def sample_surface_even(m: Mesh, count: int, rng: np.random.Generator):
    """trimesh.sample.sample_surface_even: 3*count area-weighted samples, then greedy removal of points closer than
    sqrt(area / (3 count)); may return fewer than `count` points."""
    area = m.area_faces
    radius = np.sqrt(area.sum() / (3 * count))
    n = count * 3
    cum = np.cumsum(area)
    fidx = np.searchsorted(cum, rng.random(n) * cum[-1])
    tri = m.triangles[fidx]
    r = rng.random((n, 2, 1))
    flip = r.sum(axis=1).reshape(-1) > 1.0
    r[flip] -= 1.0
    r = np.abs(r)
    pts = tri[:, 0] + ((tri[:, 1:] - tri[:, :1]) * r).sum(axis=1)
    return pts[greedy_remove_close(pts, radius)][:count]


def greedy_remove_close(pts, radius):
    """trimesh.points.remove_close: walk the points in order, keep a point unless an earlier KEPT point lies within
    `radius`.  That is the lexicographically first maximal independent set of the "closer than radius" graph; it is
    computed here in vectorised rounds over the pair list (a point is decided once all its lower-index neighbours
    are) instead of a Python loop over the points -- same mask."""
    from scipy.spatial import cKDTree
    n = len(pts)
    pairs = cKDTree(pts).query_pairs(radius, output_type="ndarray")      # (i < j)
    lo, hi = pairs[:, 0], pairs[:, 1]
    state = np.zeros(n, np.int8)                                         # 0 undecided, 1 kept, 2 removed
    while True:
        und = state == 0
        if not und.any():
            break
        removed = np.zeros(n, bool)
        removed[hi[state[lo] == 1]] = True
        state[und & removed] = 2
        waiting = np.zeros(n, bool)
        waiting[hi[state[lo] == 0]] = True
        state[(state == 0) & ~waiting] = 1
        live = (state[hi] == 0) & (state[lo] != 2)
        lo, hi = lo[live], hi[live]
    return state == 1


def icp(source_mesh: Mesh, target_mesh: Mesh, n_iter, count_source=5_000, count_target=5_000, test_reflections=False,
        test_rotations=False, fixed_scale=False, outliers=0, on_surface=False, min_scale=0.5, max_scale=2.0, plot=False,
        seed: Optional[int] = 0):
    """ICP:56-175.  Returns (best_of_all_transform (4,4), best_of_all_cost)."""
    if plot:
        raise NotImplementedError("plot needs pyvista (an interactive viewer), which this build does not ship")
    if on_surface and target_mesh.is_point_cloud:
        raise ValueError("on_surface needs a target mesh with faces (trimesh.proximity.closest_point, ICP:106-107)")
    from followmyhold_amd import ops
    rng = np.random.default_rng(seed)
    cubes = [np.eye(4)]
    if test_reflections:
        cubes += get_all_axis_aligned_reflections()
    if test_rotations:
        cubes += get_all_axis_aligned_rotations()
    if source_mesh.is_point_cloud:
        source_points = source_mesh.vertices
        count_source = len(source_points)
    else:
        source_points = sample_surface_even(source_mesh, count_source, rng)
    target_points = target_mesh.vertices if target_mesh.is_point_cloud else sample_surface_even(target_mesh, count_target, rng)
    n_outliers = int(outliers * count_source)   # ICP:89 uses the REQUESTED count, not len(source_points)
    n_outliers = min(n_outliers, len(source_points) - 2)
    # `for cube in cubes` (ICP:91) as one batched enqueue; the first strictly lowest cost wins, as in the loop
    starts = np.stack([transform_points(source_points, cube) for cube in cubes])
    if on_surface:      # closest point ON the target triangles instead of the nearest sampled target point (ICP:106-107)
        Ts, costs = ops.icp_points_multi(starts, target_mesh.vertices, n_iter, n_outliers=n_outliers, fixed_scale=fixed_scale,
                                         min_scale=min_scale, max_scale=max_scale, target_faces=target_mesh.faces)
    else:
        Ts, costs = ops.icp_points_multi(starts, target_points, n_iter, n_outliers=n_outliers, fixed_scale=fixed_scale,
                                         min_scale=min_scale, max_scale=max_scale)
    best_cost, best_T = np.inf, np.eye(4)
    for cube, T, cost in zip(cubes, Ts, costs):
        if cost < best_cost:
            best_cost, best_T = float(cost), T @ cube
    return best_T, best_cost


def align_meshes_impl(source_mesh_path, target_mesh_path, transform_path, transformed_mesh_path, fixed_scale, outliers,
                      test_rotations, test_reflections, on_surface, iterations_coarse, count_source_coarse,
                      count_target_coarse, iterations_fine, count_source_fine, count_target_fine, min_scale, max_scale, plot,
                      *, seed: Optional[int] = None):
    """ICP:178-217: init transform, coarse ICP, fine ICP; writes the 4x4 transform (np.save appends '.npy') and/or the
    transformed source mesh.  The 18 positional parameters are the reference's; `seed` (keyword only, default
    $FOHO_ICP_SEED or 0, negative = unseeded) fixes the surface sampling."""
    import os
    if seed is None:
        seed = int(os.environ.get("FOHO_ICP_SEED", "0"))
    seed_c, seed_f = (None, None) if seed < 0 else (seed, seed + 1)
    t0 = time.time()
    source_mesh, target_mesh = load(source_mesh_path), load(target_mesh_path)
    init_transform = compute_init_transform(source_mesh, target_mesh, fixed_scale)
    source_mesh.apply_transform(init_transform)
    transform_coarse, _ = icp(source_mesh, target_mesh, n_iter=iterations_coarse, count_source=count_source_coarse,
                              count_target=count_target_coarse, test_reflections=test_reflections,
                              test_rotations=test_rotations, fixed_scale=fixed_scale, outliers=outliers,
                              on_surface=on_surface, min_scale=min_scale, max_scale=max_scale, plot=plot, seed=seed_c)
    source_mesh.apply_transform(transform_coarse)
    transform_fine, _ = icp(source_mesh, target_mesh, n_iter=iterations_fine, count_source=count_source_fine,
                            count_target=count_target_fine, outliers=outliers, on_surface=on_surface, min_scale=min_scale,
                            max_scale=max_scale, plot=plot, seed=seed_f)
    source_mesh.apply_transform(transform_fine)
    final_transform = transform_fine @ transform_coarse @ init_transform
    if transform_path is not None:
        np.save(transform_path, final_transform)
    if transformed_mesh_path is not None:
        if str(transformed_mesh_path).lower().endswith(".obj"):
            meshio.save_obj(transformed_mesh_path, source_mesh.vertices, source_mesh.faces)
        else:
            meshio.save_ply(transformed_mesh_path, source_mesh.vertices, source_mesh.faces)
    print(f"Elapsed time: {time.time() - t0:.2f} seconds")
    return final_transform


def main(argv=None):
    """CLI with the reference's options (ICP:219-239)."""
    import argparse
    ap = argparse.ArgumentParser(description="Align two meshes with trimmed ICP")
    ap.add_argument("source_mesh_path")
    ap.add_argument("target_mesh_path")
    ap.add_argument("-tp", "--transform_path", default=None)
    ap.add_argument("-tmp", "--transformed_mesh_path", default=None)
    ap.add_argument("-fs", "--fixed_scale", action="store_true")
    ap.add_argument("-o", "--outliers", type=float, default=0.2)
    ap.add_argument("-trot", "--test_rotations", action="store_true")
    ap.add_argument("-tref", "--test_reflections", action="store_true")
    ap.add_argument("-os", "--on_surface", action="store_true")
    ap.add_argument("-ir", "--iterations_coarse", type=int, default=50)
    ap.add_argument("-csr", "--count_source_coarse", type=int, default=1_000)
    ap.add_argument("-ctr", "--count_target_coarse", type=int, default=5_000)
    ap.add_argument("-if", "--iterations_fine", type=int, default=100)
    ap.add_argument("-csf", "--count_source_fine", type=int, default=5_000)
    ap.add_argument("-ctf", "--count_target_fine", type=int, default=10_000)
    ap.add_argument("-mis", "--min_scale", type=float, default=0.7)
    ap.add_argument("-mas", "--max_scale", type=float, default=3.0)
    ap.add_argument("-p", "--plot", action="store_true")
    ap.add_argument("--seed", type=int, default=None, help="seed of the surface sampling (not in the reference, whose sampling "
                    "is unseeded; default $FOHO_ICP_SEED or 0, negative = unseeded)")
    a = ap.parse_args(argv)
    align_meshes_impl(a.source_mesh_path, a.target_mesh_path, a.transform_path, a.transformed_mesh_path, a.fixed_scale,
                      a.outliers, a.test_rotations, a.test_reflections, a.on_surface, a.iterations_coarse,
                      a.count_source_coarse, a.count_target_coarse, a.iterations_fine, a.count_source_fine,
                      a.count_target_fine, a.min_scale, a.max_scale, a.plot, **({} if a.seed is None else {"seed": a.seed}))


if __name__ == "__main__":
    main()
