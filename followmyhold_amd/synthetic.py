"""Seeded synthetic inputs for the guidance step (SURVEY.md 8(d) "Synthetic inputs").

Pure data generation (numpy): a closed MANO-sized hand surface (778 vertices / 1552 faces = MANO's
1538 faces + the 14 wrist-closing faces HaMeR adds, reference
third_party/estimator/hamer/hamer/utils/renderer.py:149-164), icosphere / uv-sphere objects, a
MANO-shaped linear-blend-skinning model with random but structured parameters, and a scene builder
that takes the renderer as a callback so the same scenes feed the HIP path, the tests and the
CPU baseline.
"""
import math

import numpy as np


# ----------------------------------------------------------------------------- meshes
def icosphere(level, radius=1.0):
    """Subdivided icosahedron: level 4 -> 2562 V / 5120 F, level 5 -> 10242 V / 20480 F."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        cache = {}
        verts = list(v)

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = (verts[a] + verts[b]) / 2.0
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(verts), np.array(nf, dtype=np.int64)
    return (v * radius).astype(np.float32), f


def uv_sphere(n_lat, n_lon, radius=1.0):
    """Closed lat/long sphere: 2 + (n_lat-1)*n_lon vertices, 2*n_lon*(n_lat-1) faces."""
    verts = [[0.0, 0.0, 1.0]]
    for i in range(1, n_lat):
        th = math.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * math.pi * j / n_lon
            verts.append([math.sin(th) * math.cos(ph), math.sin(th) * math.sin(ph), math.cos(th)])
    verts.append([0.0, 0.0, -1.0])
    faces = []
    for j in range(n_lon):
        faces.append([0, 1 + j, 1 + (j + 1) % n_lon])
    for i in range(n_lat - 2):
        r0, r1 = 1 + i * n_lon, 1 + (i + 1) * n_lon
        for j in range(n_lon):
            a, b = r0 + j, r0 + (j + 1) % n_lon
            c, d = r1 + j, r1 + (j + 1) % n_lon
            faces += [[a, c, d], [a, d, b]]
    last = len(verts) - 1
    r0 = 1 + (n_lat - 2) * n_lon
    for j in range(n_lon):
        faces.append([last, r0 + (j + 1) % n_lon, r0 + j])
    return (np.array(verts) * radius).astype(np.float32), np.array(faces, dtype=np.int64)


def torus(n_u, n_v, R=1.0, r=0.45):
    """Closed genus-1 quad-grid surface, every vertex has valence 6: n_u*n_v vertices, 2*n_u*n_v faces."""
    u = 2 * math.pi * np.arange(n_u) / n_u
    v = 2 * math.pi * np.arange(n_v) / n_v
    uu, vv = np.meshgrid(u, v, indexing="ij")
    x = (R + r * np.cos(vv)) * np.cos(uu)
    y = (R + r * np.cos(vv)) * np.sin(uu)
    z = r * np.sin(vv)
    verts = np.stack([x, y, z], -1).reshape(-1, 3)
    faces = []
    for i in range(n_u):
        for j in range(n_v):
            a, b = i * n_v + j, ((i + 1) % n_u) * n_v + j
            c, d = i * n_v + (j + 1) % n_v, ((i + 1) % n_u) * n_v + (j + 1) % n_v
            faces += [[a, b, d], [a, d, c]]
    return verts.astype(np.float32), np.array(faces, dtype=np.int64)


def _split_edges(v, f, n_split, rng):
    """Split n_split edges of a closed manifold mesh (+1 vertex, +2 faces each)."""
    v = [np.asarray(x, dtype=np.float64) for x in v]
    f = [list(map(int, x)) for x in f]
    for _ in range(n_split):
        fi = int(rng.integers(len(f)))
        a, b, c = f[fi]
        # the neighbour across edge (a,b) walks it as (b,a)
        fj = next(k for k, t in enumerate(f) if k != fi and any(t[i] == b and t[(i + 1) % 3] == a for i in range(3)))
        t = f[fj]
        i = next(i for i in range(3) if t[i] == b and t[(i + 1) % 3] == a)
        d = t[(i + 2) % 3]
        m = len(v)
        v.append((v[a] + v[b]) / 2.0)
        f[fi] = [a, m, c]
        f[fj] = [b, m, d]
        f.append([m, b, c])
        f.append([m, a, d])
    return np.array(v), np.array(f, dtype=np.int64)


def hand_template(seed=1):
    """Closed, star-shaped, hand-sized surface with MANO's counts: (778,3) float32, (1552,3) int64.
    Metres; palm in the xy plane, fingers towards +y, centred at the origin."""
    rng = np.random.default_rng(seed)
    v, f = icosphere(3)  # 642 V / 1280 F
    v, f = _split_edges(v.astype(np.float64), f, 136, rng)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    assert v.shape[0] == 778 and f.shape[0] == 1552
    ang = np.arctan2(v[:, 0], v[:, 1])  # 0 at +y
    r = np.ones(len(v))
    for k, c in enumerate(np.linspace(-0.9, 0.9, 5)):  # five finger lobes
        r += (0.85 + 0.1 * math.sin(k + seed)) * np.exp(-((ang - c) / 0.16) ** 2) * np.clip(v[:, 1], 0, None) ** 2
    out = v * r[:, None] * np.array([0.045, 0.05, 0.014])
    return out.astype(np.float32), f


def displaced_sphere(v, f, radius, seed, amp=0.15):
    """Low-frequency radial displacement (keeps the surface closed and star-shaped)."""
    rng = np.random.default_rng(seed)
    u = v.astype(np.float64)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    r = np.ones(len(u))
    for _ in range(6):
        k = rng.normal(size=3) * 2.0
        r += amp / 6.0 * np.sin(u @ k + rng.uniform(0, 2 * math.pi))
    return (u * r[:, None] * radius).astype(np.float32), f


def make_object(kind, seed=2):
    """'ico4' (cfg 1): 2562/5120;  '20k' (cfg 2,3,5): 10242/20480;  '40k' (cfg 4): 20160/40320 (a torus: no
    high-valence poles, so no pixel sees more than K=100 faces)."""
    if kind == "ico4":
        return icosphere(4, 0.05)
    if kind in ("ico1", "ico2", "ico3"):
        return icosphere(int(kind[3]), 0.05)
    if kind == "20k":
        v, f = icosphere(5)
        return displaced_sphere(v, f, 0.05, seed)
    if kind == "40k":
        v, f = torus(160, 126)
        return (v * 0.036).astype(np.float32), f
    raise ValueError(kind)


# ----------------------------------------------------------------------------- rotations
def axis_angle_matrix(aa):
    aa = np.asarray(aa, dtype=np.float64)
    th = np.linalg.norm(aa)
    if th < 1e-12:
        return np.eye(3)
    k = aa / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


def axis_angle_quat(aa):
    aa = np.asarray(aa, dtype=np.float64)
    th = np.linalg.norm(aa)
    if th < 1e-12:
        return np.array([1.0, 0, 0, 0])
    return np.concatenate([[math.cos(th / 2)], math.sin(th / 2) * aa / th])


# ----------------------------------------------------------------------------- MANO-shaped LBS model
MANO_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]  # smplx MANO kinematic tree


def mano_like_model(seed=1):
    """Synthetic parameters with MANO's shapes: v_template (778,3), shapedirs (778,3,10),
    posedirs (135,2334), J_regressor (16,778), lbs_weights (778,16), parents (16,), faces (1552,3)."""
    rng = np.random.default_rng(seed)
    vt, faces = hand_template(seed)
    V = vt.shape[0]
    # joints: wrist + 5 chains of 3 along the finger lobes
    joints = [np.array([0.0, -0.03, 0.0])]
    for c in np.linspace(-0.9, 0.9, 5):
        d = np.array([math.sin(c), math.cos(c), 0.0])
        for s in (0.35, 0.6, 0.8):
            joints.append(d * s * 0.09)
    joints = np.array(joints)  # (16,3)
    d2 = ((vt[:, None, :].astype(np.float64) - joints[None]) ** 2).sum(-1)
    w = np.exp(-d2 / (2 * 0.012 ** 2)) + 1e-6
    w /= w.sum(1, keepdims=True)
    jr = np.exp(-d2.T / (2 * 0.008 ** 2)) + 1e-9
    jr /= jr.sum(1, keepdims=True)
    shapedirs = rng.normal(size=(V, 3, 10)) * 0.002
    posedirs = rng.normal(size=(135, V * 3)) * 0.0005
    return dict(v_template=vt.astype(np.float32), shapedirs=shapedirs.astype(np.float32),
                posedirs=posedirs.astype(np.float32), J_regressor=jr.astype(np.float32),
                lbs_weights=w.astype(np.float32), parents=np.array(MANO_PARENTS, dtype=np.int64), faces=faces)


# ----------------------------------------------------------------------------- scene
FINGERTIPS = [744, 320, 443, 554, 671]  # reference pipelines.py:127
MANO_TO_OPENPOSE = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]  # pipelines.py:128


def similarity_about_center(v, scale, Rm, t):
    c = (v.min(0) + v.max(0)) / 2.0
    return (scale * (v - c)) @ Rm.T + c + t


def hoi_crop(points, src_width=640.0, src_fov=60.0, margin_px=10.0, factor=1.25):
    """The reference's input regime: its frames are CROPS around the hand and the object -- union of the two boxes, 10 px
    of margin on every side, squared up around its centre, times 1.25, resampled to 512 x 512
    (src/foho/preprocess/segment_hoi_sam2.py:108-124 `process_bbox`, :180-196) -- so the meshes fill most of the frame,
    where a hand at half a metre through a 60-degree lens covers 3 % of it.  `points` (N,3) in the camera's space (looking
    down -z) as a `src_width`-pixel, `src_fov`-degree source camera sees them -> (shift (3,), fov_degrees): the lateral
    shift that puts the crop's centre on the optical axis (the depth maps of the crop come from a network that assumes a
    centred principal point) and the field of view of the crop."""
    p = np.asarray(points, np.float64)
    px_per_tan = (src_width / 2.0) / math.tan(math.radians(src_fov) / 2.0)
    zc = float(np.mean(-p[:, 2]))
    shift = np.zeros(3)
    for _ in range(4):      # the box's edges are set by points at different depths: a few rounds centre it to a fraction of a pixel
        u, v = (p[:, 0] + shift[0]) / -p[:, 2], (p[:, 1] + shift[1]) / -p[:, 2]
        shift -= np.array([(u.max() + u.min()) / 2.0 * zc, (v.max() + v.min()) / 2.0 * zc, 0.0])
    u, v = (p[:, 0] + shift[0]) / -p[:, 2], (p[:, 1] + shift[1]) / -p[:, 2]
    w = (u.max() - u.min()) * px_per_tan + 2.0 * margin_px
    h = (v.max() - v.min()) * px_per_tan + 2.0 * margin_px
    side = max(w, h) * factor
    return shift, 2.0 * math.degrees(math.atan(side / 2.0 / px_per_tan))


def build_scene(render_fn, obj_kind="20k", H=512, W=512, fov=60.0, seed=0, two_hands=False, crop=None):
    """Scene dict of numpy arrays for one image.

    crop="hoi": the frame is the reference's crop around hand and object (`hoi_crop`; `fov` is then computed, not taken).

    render_fn(verts_world (V,3) f32, faces (F,3) i64, H, W, fov) -> (normal (H,W,3), disp (H,W), pix_to_face (H,W))
    renders the ground-truth pose to make the MoGe-style targets (reference pipelines.py:1247-1256).
    Ground truth lives in the "MoGe" world (x right, y up, camera looks down -z; run.py:84-90);
    the returned start point is the ground truth perturbed by scale 1.1, 5 degrees, 1 cm."""
    rng = np.random.default_rng(seed)
    model = mano_like_model(1)
    hv, hf = model["v_template"].astype(np.float64), model["faces"]
    Rh = axis_angle_matrix(rng.normal(size=3) * 0.2)
    hv = hv @ Rh.T + np.array([0.0, 0.0, -0.5])
    if two_hands:
        hv2 = model["v_template"].astype(np.float64) * np.array([-1.0, 1, 1])
        hv2 = hv2 @ axis_angle_matrix(rng.normal(size=3) * 0.2).T + np.array([0.09, 0.0, -0.52])
        hv = np.concatenate([hv, hv2], 0)
        hf = np.concatenate([hf, hf[:, ::-1] + 778], 0)
    ov, of = make_object(obj_kind, seed + 2)
    ov = ov.astype(np.float64)
    # Hunyuan space -> MoGe space similarity (what foho.alignment.h2m produces)
    s_h2m = 1.0 / 0.9
    R_h2m = axis_angle_matrix(np.array([0.1, -0.2, 0.05]))
    obj_center = np.array([0.02, -0.01, -0.5 + 0.062])  # in front of the palm, touching it
    T = np.eye(4)
    T[:3, :3] = s_h2m * R_h2m
    T[:3, 3] = obj_center
    ov_hy = ov / s_h2m  # object as Hunyuan emits it
    ov_moge = ov_hy @ T[:3, :3].T + T[:3, 3]
    if crop == "hoi":
        shift, fov = hoi_crop(np.concatenate([hv, ov_moge], 0))
        hv = hv + shift
        T[:3, 3] += shift
        ov_moge = ov_hy @ T[:3, :3].T + T[:3, 3]
    elif crop is not None:
        raise ValueError(crop)

    Vh = hv.shape[0]
    gt_v = np.concatenate([hv, ov_moge], 0).astype(np.float32)
    gt_f = np.concatenate([hf, of + Vh], 0)
    normal, disp, p2f = render_fn(gt_v, gt_f, H, W, fov)
    hand_mask = (p2f >= 0) & (p2f < hf.shape[0])
    obj_mask = p2f >= hf.shape[0]
    hoi = (hand_mask | obj_mask).astype(np.float32)
    moge_normal = normal * hoi[..., None]
    moge_disp = disp * hoi

    # 21 keypoints of the ground truth, projected, + N(0,1 px)
    jr = model["J_regressor"].astype(np.float64)
    kp3 = np.concatenate([jr @ hv[:778], hv[FINGERTIPS]], 0)[MANO_TO_OPENPOSE]
    tanh = math.tan(math.radians(fov) / 2)
    xn = (-kp3[:, 0]) / (tanh * -kp3[:, 2])
    yn = kp3[:, 1] / (tanh * -kp3[:, 2])
    s = min(H, W) / 2.0
    kps = np.stack([W / 2.0 - s * xn, H / 2.0 - s * yn], 1) + rng.normal(size=(21, 2))

    # start point: undo a similarity perturbation so that optimisation has something to do
    pert_R = axis_angle_matrix(np.array([0.0, 0.0, math.radians(5.0)]))
    hand_start = similarity_about_center(hv, 1.1, pert_R, np.array([0.01, 0.0, 0.0]))
    return dict(
        hand_verts=hand_start.astype(np.float32), hand_faces=hf.astype(np.int64),
        obj_verts=ov_hy.astype(np.float32), obj_faces=of.astype(np.int64), T_h2m=T.astype(np.float32),
        J_regressor=model["J_regressor"], kps_2d=kps.astype(np.float32),
        moge_normal=moge_normal.astype(np.float32), moge_disp=moge_disp.astype(np.float32),
        hand_mask=hand_mask, obj_mask=obj_mask, fov=float(fov), H=int(H), W=int(W),
        gt_hand_verts=hv.astype(np.float32),
    )
