"""ORACLE (tests only): numpy float64 restatement of the reference's ICP alignment
(src/foho/alignment/mesh_align.py; ICP below) and of the trimesh / scipy pieces it calls
(trimesh.registration.procrustes, Trimesh.centroid / .scale, transformations).

PARITY UNPINNED for trimesh internals (not installed here); `sample_surface_even` is additionally unseeded in
the reference (ICP:79, ICP:85), so parity is defined on the deterministic part: given the sampled source and
target point sets, the sequence of transforms is reproduced exactly.
"""
import numpy as np


def translation_matrix(t):
    M = np.eye(4)
    M[:3, 3] = t
    return M


def scale_matrix(factor, origin):
    """trimesh.transformations.scale_matrix(factor, origin): uniform scaling about `origin`."""
    M = np.eye(4) * factor
    M[3, 3] = 1.0
    M[:3, 3] = np.asarray(origin, np.float64) * (1.0 - factor)
    return M


def transform_points(p, M):
    return p @ M[:3, :3].T + M[:3, 3]


def mesh_centroid_scale(verts, faces=None):
    """ICP:18-23: PointCloud -> vertex mean & AABB diagonal; Trimesh -> area-weighted centroid & AABB diagonal."""
    v = np.asarray(verts, np.float64)
    scale = np.linalg.norm(v.max(0) - v.min(0))
    if faces is None or len(faces) == 0:
        return v.mean(0), scale
    t = v[np.asarray(faces)]
    area = 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1)
    cen = t.mean(1)
    return (cen * area[:, None]).sum(0) / area.sum(), scale


def compute_init_transform(src_v, src_f, tgt_v, tgt_f, fixed_scale=False):
    """ICP:25-35: translate centroids, scale by the AABB-diagonal ratio about the source centroid."""
    sc, ss = mesh_centroid_scale(src_v, src_f)
    tc, ts = mesh_centroid_scale(tgt_v, tgt_f)
    T = translation_matrix(tc - sc)
    if fixed_scale:
        return T
    return T @ scale_matrix(ts / ss, sc)


def procrustes(a, b, reflection=True, scale=True):
    """trimesh.registration.procrustes(a, b, reflection, scale, return_cost=False): 4x4 mapping a -> b."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    ac, bc = a.mean(0), b.mean(0)
    if scale:
        ascale = np.sqrt(((a - ac) ** 2).sum() / len(a))
        bscale = np.sqrt(((b - bc) ** 2).sum() / len(b))
    else:
        ascale = bscale = 1.0
    target = ((b - bc) / bscale).T @ ((a - ac) / ascale)
    u, s, vh = np.linalg.svd(target)
    if reflection:
        Rm = u @ vh
    else:
        Rm = u @ np.diag([1, 1, np.linalg.det(u @ vh)]) @ vh
    M = np.eye(4)
    M[:3, :3] = bscale / ascale * Rm
    M[:3, 3] = bc - (bscale / ascale) * (Rm @ ac)
    return M


def nearest(p, q):
    """Brute-force stand-in for scipy cKDTree(q).query(p): (distance, index), ties -> lowest index."""
    d2 = ((p[:, None, :] - q[None, :, :]) ** 2).sum(-1)
    idx = d2.argmin(1)
    return np.sqrt(d2[np.arange(len(p)), idx]), idx


def icp_points(source_points, target_points, n_iter, outliers=0.0, fixed_scale=False, min_scale=0.5, max_scale=2.0,
               record=None):
    """ICP:91-142 for one 'cube' (identity start), on already-sampled point sets.
    Returns (best_transform, best_cost) with the reference's bookkeeping: the cost of an iteration is measured
    BEFORE that iteration's update while `best_transform` stores the transform AFTER it (ICP:129, ICP:140-142)."""
    src = np.asarray(source_points, np.float64)
    tgt = np.asarray(target_points, np.float64)
    n_out = int(outliers * len(src))
    transform = np.eye(4)
    best_cost, best_transform = np.inf, transform.copy()
    for _ in range(n_iter):
        p = transform_points(src, transform)
        dist, qi = nearest(p, tgt)
        q = tgt[qi]
        if n_out > 0:
            order = np.argsort(dist)  # same default (quicksort) as the reference's np.argsort
            inl = order[:-n_out]
            cost = dist[inl].mean()
            p_in, q_in = p[inl], q[inl]
        else:
            p_in, q_in, cost = p, q, dist.mean()
        nxt = procrustes(p_in, q_in, reflection=False, scale=not fixed_scale)
        transform = nxt @ transform
        if not fixed_scale:
            s = np.linalg.norm(transform[:3, 0])
            transform[:3, :3] /= s
            transform[:3, :3] *= np.clip(s, min_scale, max_scale)
        if record is not None:
            record.append((cost, transform.copy()))
        if cost < best_cost:
            best_cost, best_transform = cost, transform
    return best_transform, best_cost
