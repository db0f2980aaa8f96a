"""oracle/flexi_ref.py -- CPU restatement (torch, autograd) of the iso-surfacing step between the diffusion latent and
the guidance path: FlexiCubes with its default (unit) weights = Dual Marching Cubes on the regular SDF grid.
TEST INFRASTRUCTURE ONLY (imported by tests/, never by followmyhold_amd/ or foho/).

Reference call sites: `flexi = knc.FlexiCubes(device)`, `flexi.construct_voxel_grid(res)` (pipelines.py:1142-1143),
`obj_verts, obj_faces, _ = flexi(xyz_samples, sdf.flatten(), cube_indices, res)` (PL:1393, PL:1509) -- no weights, no
training mode.  The implementation lives in kaolin 0.17.0 (`kaolin/non_commercial/flexicubes`), which is not under
/root/reference and cannot be installed here: PARITY UNPINNED.  What is restated is the published algorithm
(Shen et al., "Flexible Isosurface Extraction for Gradient-Based Mesh Optimization", SIGGRAPH 2023, section 4, with
all weights at their defaults; Nielson, "Dual Marching Cubes", 2004):

  * a cube is a surface cube when its 8 corner signs differ; sign-change edges carry the zero crossing
    u_e = (x_a s_b - x_b s_a) / (s_b - s_a)
  * inside a cube the crossing edges are grouped into patches: on every face the crossing edges are joined pairwise,
    and on an ambiguous face (two diagonal inside corners) the pairs are the ones that separate the INSIDE corners;
    every patch yields one dual vertex = mean of its crossing points (beta = 1)
  * every interior sign-change grid edge yields one quad through the dual vertices of its four cubes, oriented from
    the inside end of the edge to the outside end, split along its first diagonal (gamma = 1)
  * L_dev of a dual vertex = mean absolute deviation of the distances to its crossing points

Orderings (this repo's convention, shared with the HIP implementation so that indices can be compared bit for bit):
grid point (i,j,k) -> (i*G + j)*G + k (x-major, generate_dense_grid_points PL:341-360); cube corners
[(0,0,0),(1,0,0),(0,1,0),(1,1,0),(0,0,1),(1,0,1),(0,1,1),(1,1,1)]; vertices ordered by (cube, patch) with patches
ordered by their smallest cube-edge id; faces ordered by (axis, i, j, k) of their grid edge.
"""
import numpy as np
import torch

CORNERS = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
# cube edges as corner pairs: 0-3 along x, 4-7 along y, 8-11 along z
EDGES = [(0, 1), (2, 3), (4, 5), (6, 7), (0, 2), (1, 3), (4, 6), (5, 7), (0, 4), (1, 5), (2, 6), (3, 7)]
# faces as cyclic corner quadruples
FACES = [(0, 2, 6, 4), (1, 3, 7, 5), (0, 1, 5, 4), (2, 3, 7, 6), (0, 1, 3, 2), (4, 5, 7, 6)]


def _edge_id(a, b):
    return EDGES.index((min(a, b), max(a, b)))


def patch_tables():
    """n_patch (256,) and edge_patch (256,12): patch id of every crossing cube edge (-1 elsewhere), derived by WALKING
    the cycles: from a crossing edge step across a face to its partner edge on that face, leave through the other face of
    that edge, until the walk closes."""
    n_patch = np.zeros(256, np.int8)
    edge_patch = -np.ones((256, 12), np.int8)
    # the two faces of every edge
    edge_faces = [[fi for fi, f in enumerate(FACES) if a in f and b in f] for a, b in EDGES]
    for case in range(256):
        inside = [(case >> c) & 1 for c in range(8)]
        crossing = [inside[a] != inside[b] for a, b in EDGES]

        def partner(e, fi):
            f = FACES[fi]
            on_face = [_edge_id(f[q], f[(q + 1) % 4]) for q in range(4)]
            cr = [x for x in on_face if crossing[x]]
            if len(cr) == 2:
                return cr[0] if cr[1] == e else cr[1]
            # ambiguous face: the partner is the other face edge at the INSIDE corner of e
            a, b = EDGES[e]
            corner = a if inside[a] else b
            cand = [x for x in on_face if x != e and corner in EDGES[x]]
            assert len(cand) == 1
            return cand[0]

        order = []
        seen = set()
        for e0 in range(12):
            if not crossing[e0] or e0 in seen:
                continue
            cyc, e, fi = [], e0, edge_faces[e0][0]
            while True:
                cyc.append(e)
                seen.add(e)
                e2 = partner(e, fi)
                fi = [x for x in edge_faces[e2] if x != fi][0]
                e = e2
                if e == e0:
                    break
            order.append(sorted(cyc))
        order.sort(key=lambda c: c[0])
        n_patch[case] = len(order)
        for pi, cyc in enumerate(order):
            for e in cyc:
                edge_patch[case, e] = pi
    return n_patch, edge_patch


def construct_voxel_grid(res):
    """(res+1)^3 grid points in [-0.5, 0.5]^3 (x-major) and the (res^3, 8) corner indices of every cube."""
    G = res + 1
    lin = torch.linspace(-0.5, 0.5, G)
    xs, ys, zs = torch.meshgrid(lin, lin, lin, indexing="ij")
    verts = torch.stack([xs, ys, zs], -1).reshape(-1, 3)
    i, j, k = torch.meshgrid(torch.arange(res), torch.arange(res), torch.arange(res), indexing="ij")
    base = ((i * G + j) * G + k).reshape(-1)
    offs = torch.tensor([(cx * G + cy) * G + cz for cx, cy, cz in CORNERS])
    return verts, base[:, None] + offs[None, :]


def flexicubes(x, s, res):
    """x (G^3,3) grid positions, s (G^3,) SDF (negative inside), res -> verts (V,3), faces (F,3) int64, l_dev (V,).
    Differentiable w.r.t. x and s through the crossing points."""
    G = res + 1
    n_patch, edge_patch = patch_tables()
    s3 = s.reshape(G, G, G)
    inside = (s3 < 0).numpy() if not s3.requires_grad else (s3.detach() < 0).numpy()
    case = np.zeros((res, res, res), np.int64)
    for c, (cx, cy, cz) in enumerate(CORNERS):
        case |= inside[cx:cx + res, cy:cy + res, cz:cz + res].astype(np.int64) << c
    case = case.reshape(-1)
    np_cube = n_patch[case].astype(np.int64)
    v_off = np.concatenate([[0], np.cumsum(np_cube)])
    cubes = np.flatnonzero(np_cube > 0)
    ci, cj, ck = np.unravel_index(cubes, (res, res, res))
    corner_idx = np.stack([((ci + cx) * G + (cj + cy)) * G + (ck + cz) for cx, cy, cz in CORNERS], 1)   # (S,8)
    # vertices: one per (cube, patch)
    verts, ldev = [], []
    xs = x
    for p in range(4):
        sel = np.flatnonzero(np_cube[cubes] > p)
        if len(sel) == 0:
            continue
        cs = case[cubes[sel]]
        mask = torch.from_numpy((edge_patch[cs] == p).astype(np.float32))                               # (n,12)
        acc = torch.zeros(len(sel), 3, dtype=x.dtype)
        ues = []
        for e, (a, b) in enumerate(EDGES):
            ia, ib = torch.from_numpy(corner_idx[sel, a]), torch.from_numpy(corner_idx[sel, b])
            sa, sb = s[ia], s[ib]
            den = sb - sa
            den = torch.where(mask[:, e] > 0, den, torch.ones_like(den))
            ue = (xs[ia] * sb[:, None] - xs[ib] * sa[:, None]) / den[:, None]
            ue = torch.where(mask[:, e, None] > 0, ue, torch.zeros_like(ue))
            ues.append(ue)
            acc = acc + ue
        cnt = mask.sum(1)
        v = acc / cnt[:, None]
        d = torch.stack([((u - v) ** 2).sum(1).sqrt() for u in ues], 1) * mask
        mean_d = d.sum(1) / cnt
        dev = ((d - mean_d[:, None]).abs() * mask).sum(1) / cnt
        verts.append((v_off[cubes[sel]] + p, v, dev))
    order = np.concatenate([t[0] for t in verts]) if verts else np.zeros(0, np.int64)
    V = torch.cat([t[1] for t in verts]) if verts else torch.zeros(0, 3)
    D = torch.cat([t[2] for t in verts]) if verts else torch.zeros(0)
    perm = torch.from_numpy(np.argsort(order, kind="stable"))
    V, D = V[perm], D[perm]
    # faces: interior sign-change grid edges, (axis, i, j, k) order
    faces = []
    cube_id = lambda i, j, k: (i * res + j) * res + k
    # for every axis: the four cubes around an edge (cyclic) and the local edge id of the grid edge inside each of them
    ring = {0: [((0, -1, -1), 3), ((0, 0, -1), 2), ((0, 0, 0), 0), ((0, -1, 0), 1)],
            1: [((-1, 0, -1), 7), ((-1, 0, 0), 5), ((0, 0, 0), 4), ((0, 0, -1), 6)],
            2: [((-1, -1, 0), 11), ((0, -1, 0), 10), ((0, 0, 0), 8), ((-1, 0, 0), 9)]}
    for axis in range(3):
        d = [(1, 0, 0), (0, 1, 0), (0, 0, 1)][axis]
        ni, nj, nk = G - d[0], G - d[1], G - d[2]
        a = inside[:ni, :nj, :nk]
        b = inside[d[0]:, d[1]:, d[2]:]
        ii, jj, kk = np.nonzero(a != b)
        for i, j, k in zip(ii, jj, kk):
            quad = []
            ok = True
            for (di, dj, dk), le in ring[axis]:
                ci_, cj_, ck_ = i + di, j + dj, k + dk
                if not (0 <= ci_ < res and 0 <= cj_ < res and 0 <= ck_ < res):
                    ok = False
                    break
                cid = cube_id(ci_, cj_, ck_)
                quad.append(v_off[cid] + edge_patch[case[cid], le])
            if not ok:
                continue
            if not inside[i, j, k]:           # orient from the inside end to the outside end of the edge
                quad = quad[::-1]
            faces.append([quad[0], quad[1], quad[2]])
            faces.append([quad[0], quad[2], quad[3]])
    F = torch.tensor(faces, dtype=torch.int64).reshape(-1, 3)
    return V, F, D
