"""ORACLE for the mesh decimator (test infrastructure, NOT product code; numpy, small meshes only).

`FaceReducer` of the reference (src/foho/guidance/run.py:163) is hy3dgen's pymeshlab call
meshing_decimation_quadric_edge_collapse(targetfacenum=40000, preserveboundary, boundaryweight=3, preservenormal,
preservetopology); neither package is available (parity unpinned).  This file restates the published algorithm (Garland &
Heckbert 1997) the way a text book would -- NO priority queue, NO lazy deletion, NO time stamps: before every collapse the
cost of EVERY current edge is recomputed from scratch and the cheapest admissible one is taken -- so that the product's
heap-based implementation (followmyhold_amd/csrc/mesh_decimate.inc) has an independent implementation to be compared with.
Same definitions as there: area-weighted plane quadrics, a constraint plane of weight 3 |e|^2 through every boundary edge,
optimal position by solving the 3x3 system (fallback: best of the end points and the mid point), link condition, normal-flip
test at cos < 0.2."""
import numpy as np


def _plane_quadric(n, d, w):
    p = np.array([n[0], n[1], n[2], d], np.float64)
    return w * np.outer(p, p)


def _face_normal(P, tri):
    return np.cross(P[tri[1]] - P[tri[0]], P[tri[2]] - P[tri[0]])


def initial_quadrics(V, F):
    P = np.asarray(V, np.float64)
    Q = np.zeros((len(P), 4, 4))
    for tri in F:
        n = _face_normal(P, tri)
        l = np.linalg.norm(n)
        if l <= 0:
            continue
        n = n / l
        q = _plane_quadric(n, -float(n @ P[tri[0]]), 0.5 * l)
        for v in tri:
            Q[v] += q
    e = np.sort(np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]]), 1)
    ue, cnt = np.unique(e, axis=0, return_counts=True)
    for (a, b) in ue[cnt == 1]:                         # boundary edges: a plane through the edge, perpendicular to its face
        f = next(t for t in F if a in t and b in t)
        n = _face_normal(P, f)
        d = P[b] - P[a]
        c = np.cross(d, n)
        l = np.linalg.norm(c)
        if l <= 0:
            continue
        c = c / l
        q = _plane_quadric(c, -float(c @ P[a]), 3.0 * float(d @ d))
        Q[a] += q
        Q[b] += q
    return Q


def _qeval(q, p):
    h = np.array([p[0], p[1], p[2], 1.0])
    return float(h @ q @ h)


def _candidate(P, Q, a, b):
    q = Q[a] + Q[b]
    A, r = q[:3, :3], -q[:3, 3]
    mid = 0.5 * (P[a] + P[b])
    len2 = float((P[a] - P[b]) @ (P[a] - P[b]))
    det, tr = np.linalg.det(A), np.trace(A)
    p = None
    if abs(det) > 1e-9 * tr ** 3 and tr > 0:
        x = np.linalg.solve(A, r)
        d2 = float((x - mid) @ (x - mid))
        if np.isfinite(d2) and d2 <= 4.0 * len2:
            p = x
    if p is None:
        ea, eb, em = _qeval(q, P[a]), _qeval(q, P[b]), _qeval(q, mid)
        p = mid if (em <= ea and em <= eb) else (P[a] if ea <= eb else P[b])
    return max(_qeval(q, p), 0.0), np.array(p, np.float64)


def _admissible(P, faces, a, b, p):
    fa = [f for f in faces if a in f]
    fb = [f for f in faces if b in f]
    na = {v for f in fa for v in f} - {a}
    nb = {v for f in fb for v in f} - {b}
    shared = [f for f in fa if b in f]
    common = len(na & nb)
    if len(shared) == 0 or len(shared) > 2 or common != len(shared):
        return False                                    # link condition: the collapse would pinch the surface
    if len(na) + len(nb) - common - 2 < 3:
        return False                                    # would close a tetrahedron
    for v, other, star in ((a, b, fa), (b, a, fb)):
        for f in star:
            if other in f:
                continue
            n0 = _face_normal(P, f)
            Pn = P.copy()
            Pn[v] = p
            n1 = _face_normal(Pn, f)
            l0, l1 = np.linalg.norm(n0), np.linalg.norm(n1)
            if not l1 > 0:
                return False
            if l0 > 0 and float(n0 @ n1) < 0.2 * l0 * l1:
                return False
    return True


def decimate_bruteforce(V, F, target_faces):
    """(verts float64, faces int64) with at most target_faces faces (or as few as admissible collapses allow); vertices and
    faces keep their relative order, collapsing (a, b), a < b, moves a to the optimal position and removes b."""
    P = np.array(V, np.float64)
    faces = [tuple(int(x) for x in f) for f in np.asarray(F) if len(set(f)) == 3]
    Q = initial_quadrics(P, np.array(faces, np.int64))
    while len(faces) > target_faces:
        edges = sorted({(min(f[i], f[(i + 1) % 3]), max(f[i], f[(i + 1) % 3])) for f in faces for i in range(3)})
        cands = sorted((_candidate(P, Q, a, b)[0], a, b) for a, b in edges)
        done = False
        for _, a, b in cands:
            cost, p = _candidate(P, Q, a, b)
            if not _admissible(P, faces, a, b, p):
                continue
            P[a] = p
            Q[a] = Q[a] + Q[b]
            faces = [tuple(a if v == b else v for v in f) for f in faces if not (a in f and b in f)]
            done = True
            break
        if not done:
            break
    used = sorted({v for f in faces for v in f})
    remap = {v: i for i, v in enumerate(used)}
    return P[used], np.array([[remap[v] for v in f] for f in faces], np.int64).reshape(-1, 3)


def mean_sq_distance_to_mesh(points, V, F):
    """Mean squared distance of `points` to the surface (V, F): exact point-triangle distance, brute force."""
    P = np.asarray(points, np.float64)
    T = np.asarray(V, np.float64)[np.asarray(F)]
    a, b, c = T[:, 0], T[:, 1], T[:, 2]
    out = np.full(len(P), np.inf)
    for k in range(len(T)):                              # Ericson, Real-Time Collision Detection 5.1.5, vectorised over points
        ab, ac, ap = b[k] - a[k], c[k] - a[k], P - a[k]
        d1, d2 = ap @ ab, ap @ ac
        bp = P - b[k]
        d3, d4 = bp @ ab, bp @ ac
        cp = P - c[k]
        d5, d6 = cp @ ab, cp @ ac
        vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
        den = va + vb + vc
        with np.errstate(divide="ignore", invalid="ignore"):
            v, w = vb / den, vc / den
            q = a[k] + np.outer(v, ab) + np.outer(w, ac)
            t_ab = np.clip(d1 / (d1 - d3), 0, 1)
            t_ac = np.clip(d2 / (d2 - d6), 0, 1)
            t_bc = np.clip((d4 - d3) / ((d4 - d3) + (d5 - d6)), 0, 1)
        inside = (va >= 0) & (vb >= 0) & (vc >= 0) & (den > 0)
        cand = [a[k] + np.outer(np.nan_to_num(t_ab), ab), a[k] + np.outer(np.nan_to_num(t_ac), ac),
                b[k] + np.outer(np.nan_to_num(t_bc), c[k] - b[k])]
        d = np.minimum.reduce([((P - x) ** 2).sum(1) for x in cand])
        d = np.where(inside, ((P - np.nan_to_num(q)) ** 2).sum(1), d)
        out = np.minimum(out, d)
    return float(out.mean())
