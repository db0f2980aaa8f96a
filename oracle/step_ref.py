"""ORACLE restatement of the three optimisation-in-the-loop phases (tests / cpu_baseline only).

Follows third_party_patches/hy3dgen/shapegen/pipelines.py (PL) of the reference:
  phase A  hand only   PL:1295-1358   (Adam,  eps 1e-4)
  phase B  object only PL:1361-1453   (AdamW, eps 1e-4)
  phase C  joint       PL:1455-1601   (AdamW, eps 1e-4)  <- one iteration = one "guidance step"
with the object mesh as an input (latent2sdf + FlexiCubes, PL:1507-1509, are outside the
synthetic step): the gradient sink on the object side is `obj_verts`.

Scene dict (all torch CPU tensors):
  hand_verts (Vh,3)  hand mesh already in MoGe space (mano_mesh_moge, PL:1241), constant
  hand_faces (Fh,3) int64, obj_verts (Vo,3) Hunyuan space, obj_faces (Fo,3) int64, T_h2m (4,4)
  J_regressor (16,Vh), kps_2d (21,2), moge_normal (H,W,3), moge_disp (H,W),
  hand_mask (H,W) bool, obj_mask (H,W) bool, fov (deg), H, W
Params dict: scale_hand (1,), trans_hand (3,), rot_hand (4,), scale_obj, trans_obj, rot_obj.
"""
import torch

from . import ref_ops as R


def bce(pred, target):
    """F.binary_cross_entropy (mean), log clamped at -100."""
    return torch.nn.functional.binary_cross_entropy(pred, target)


def _cam(scene, dtype):
    return R.Camera(scene["fov"], scene["H"], scene["W"], dtype=dtype)


def hand_transform(scene, p):
    """PL:1483-1486."""
    Rm = R.quaternion_to_matrix(p["rot_hand"])
    return R.transform_around_center_w_scale(scene["hand_verts"].to(p["rot_hand"].dtype), Rm, p["trans_hand"], p["scale_hand"])


def obj_transform(scene, p, obj_verts):
    """PL:1520-1526."""
    moge = R.transform_hunyuan2moge(obj_verts, scene["T_h2m"].to(obj_verts.dtype))
    Rm = R.quaternion_to_matrix(p["rot_obj"])
    return R.transform_around_center_w_scale(moge, Rm, p["trans_obj"], p["scale_obj"])


def render_all(verts, faces, cam, blur, want_sil):
    sel = R.rasterize_select(R.world_to_ndc(verts, cam), faces, cam.H, cam.W, blur)
    rgba, zbuf = R.render_normals(verts, faces, cam, sel)
    nrm, disp = R.render_normal_and_disparity(rgba, zbuf)
    sil = R.render_silhouette(verts, faces, cam, sel) if want_sil else None
    return dict(sel=sel, rgba=rgba, zbuf=zbuf, normal=nrm, disp=disp, sil=sil)


def hand_losses(scene, p, cam, blur, want_sil):
    dt = p["rot_hand"].dtype
    hv = hand_transform(scene, p)
    r = render_all(hv, scene["hand_faces"], cam, blur, want_sil)
    kp3 = R.mano_vert_to_3dkps(hv, scene["J_regressor"])
    kp2 = R.ndc_to_screen(R.world_to_ndc(kp3, cam), cam.H, cam.W)
    hm = scene["hand_mask"]
    out = dict(
        verts=hv, render=r,
        kps=torch.nn.functional.mse_loss(kp2, scene["kps_2d"].to(dt)),
        normal=R.normal_alignment_loss(r["normal"], scene["moge_normal"].to(dt), valid_mask=hm),
        disp=torch.nn.functional.l1_loss(r["disp"], scene["moge_disp"].to(dt) * hm),
        trans=(p["trans_hand"] ** 2).mean(),
    )
    if want_sil:
        out["sil"] = bce(r["sil"], hm.to(dt))
    return out


def phase_a_loss(scene, p):
    """PL:1320-1349."""
    cam = _cam(scene, p["rot_hand"].dtype)
    blur = R.blur_radius_from_sigma()
    h = hand_losses(scene, p, cam, blur, True)
    total = 1e-2 * h["kps"] + 1 * h["normal"] + 10 * h["disp"] + 1 * h["sil"] + 1e-2 * h["trans"]
    terms = dict(kps=h["kps"], normal_hand=h["normal"], disp_hand=h["disp"], sil_hand=h["sil"], trans_hand=h["trans"])
    return total, terms, h


def phase_b_loss(scene, p, obj_verts, edges):
    """PL:1386-1440."""
    dt = obj_verts.dtype
    cam = _cam(scene, dt)
    blur = R.blur_radius_from_sigma()
    ov = obj_transform(scene, p, obj_verts)
    r = render_all(ov, scene["obj_faces"], cam, blur, True)
    om = scene["obj_mask"]
    terms = dict(
        edge=R.mesh_edge_loss(ov, edges),
        normal_obj=R.normal_alignment_loss(r["normal"], scene["moge_normal"].to(dt), valid_mask=om),
        disp_obj=torch.nn.functional.l1_loss(r["disp"], scene["moge_disp"].to(dt) * om),
        sil_obj=bce(r["sil"], om.to(dt)),
        verts_obj=ov.pow(2).mean(),
        trans_obj=(p["trans_obj"] ** 2).mean(),
    )
    total = (1 * terms["edge"] + 10 * terms["normal_obj"] + 10 * terms["disp_obj"] + 100 * terms["sil_obj"]
             + 1e-3 * terms["verts_obj"] + 1e-2 * terms["trans_obj"])
    return total, terms, dict(verts=ov, render=r)


def phase_c_loss(scene, p, obj_verts, edges, denoise_i=19, num_inference_steps=20, use_intersection=True,
                 grid_res=64):
    """PL:1480-1588: one joint guidance step's loss."""
    dt = obj_verts.dtype
    cam = _cam(scene, dt)
    blur = R.blur_radius_from_sigma()
    h = hand_losses(scene, p, cam, blur, False)
    hand_loss = 1e-4 * h["kps"] + 10 * h["normal"] + 10 * h["disp"] + 1e-2 * h["trans"]
    hv = h["verts"]
    ov = obj_transform(scene, p, obj_verts)

    d_ho, knn_idx = R.knn1(hv, ov)
    distance_loss = torch.clamp(d_ho - 0.01, min=0).mean()

    Vh = hv.shape[0]
    hoi_v = torch.cat([hv, ov], dim=0)  # join_meshes_as_scene: hand first (PL:1544)
    hoi_f = torch.cat([scene["hand_faces"], scene["obj_faces"] + Vh], dim=0)
    r = render_all(hoi_v, hoi_f, cam, blur, True)

    if use_intersection:
        n_int = R.intersection_count(hv, scene["hand_faces"], ov, scene["obj_faces"], grid_res)
        loss_int = torch.tensor(n_int / 1000, dtype=dt)
    else:
        n_int = 0
        loss_int = torch.tensor(0.0, dtype=dt)
    w_int = 1e-5 if (float(d_ho.detach().mean()) < 0.001 and denoise_i >= num_inference_steps - 3) else 1e-9

    hoi_mask = scene["hand_mask"] | scene["obj_mask"]
    terms = dict(
        intersection=loss_int, contact=distance_loss,
        normal_hoi=R.normal_alignment_loss(r["normal"], scene["moge_normal"].to(dt), valid_mask=hoi_mask),
        disp_hoi=torch.nn.functional.l1_loss(r["disp"], scene["moge_disp"].to(dt)),
        sil_hoi=bce(r["sil"], hoi_mask.to(dt)),
        verts_obj=ov.pow(2).mean(), edge=R.mesh_edge_loss(ov, edges),
        trans_obj=(p["trans_obj"] ** 2).mean(), hand_loss=hand_loss,
        kps=h["kps"], normal_hand=h["normal"], disp_hand=h["disp"], trans_hand=h["trans"],
    )
    total = (w_int * loss_int + 10 * distance_loss + 10 * terms["normal_hoi"] + 10 * terms["disp_hoi"]
             + 10 * terms["sil_hoi"] + 1e-3 * terms["verts_obj"] + 1 * terms["edge"] + 1e-3 * terms["trans_obj"]
             + 1e-3 * hand_loss)
    aux = dict(hand=h, obj_verts_t=ov, render=r, knn_idx=knn_idx, n_int=n_int, w_int=w_int, hoi_faces=hoi_f)
    return total, terms, aux


PARAM_KEYS = ["scale_hand", "trans_hand", "rot_hand", "scale_obj", "trans_obj", "rot_obj"]
SIL_TERM = {"A": ("sil_hand", 1.0), "B": ("sil_obj", 100.0), "C": ("sil_hoi", 10.0)}     # PL:1346, 1436, 1584


def make_params(dtype=torch.float32, **over):
    """PL:1207-1215 identity start."""
    p = dict(scale_hand=torch.tensor([1.0]), trans_hand=torch.zeros(3), rot_hand=torch.tensor([1.0, 0, 0, 0]),
             scale_obj=torch.tensor([1.0]), trans_obj=torch.zeros(3), rot_obj=torch.tensor([1.0, 0, 0, 0]))
    p.update(over)
    return {k: v.detach().clone().to(dtype) for k, v in p.items()}


def leafify(p, keys):
    return {k: (v.detach().clone().requires_grad_(True) if k in keys else v.detach().clone()) for k, v in p.items()}


class JointStepper:
    """Drives phase-C iterations with the real torch.optim.AdamW (PL:1478), obj_verts as grad sink."""

    # guid_config.py:21-26
    PHASE2_HAND_LRS = {"scale": 1e-4, "trans": 1e-4, "rot": 1e-2}
    OBJ_LRS = {"scale": 5e-2, "trans": 1e-2, "rot": 1e-2}

    def __init__(self, scene, params, denoise_i=19, num_inference_steps=20, grid_res=64):
        self.scene = scene
        self.grid_res = grid_res
        self.p = leafify(params, PARAM_KEYS)
        self.obj_verts = scene["obj_verts"].detach().clone().requires_grad_(True)
        self.edges = R.unique_edges(scene["obj_faces"])
        self.i, self.n = denoise_i, num_inference_steps
        h, o = self.PHASE2_HAND_LRS, self.OBJ_LRS
        groups = [
            {"params": [self.p["scale_hand"]], "lr": h["scale"]}, {"params": [self.p["trans_hand"]], "lr": h["trans"]},
            {"params": [self.p["rot_hand"]], "lr": h["rot"]}, {"params": [self.p["scale_obj"]], "lr": o["scale"]},
            {"params": [self.p["trans_obj"]], "lr": o["trans"]}, {"params": [self.p["rot_obj"]], "lr": o["rot"]},
        ]
        self.opt = torch.optim.AdamW(groups, eps=1e-4)

    def step(self, update=True):
        self.opt.zero_grad()
        self.obj_verts.grad = None
        total, terms, aux = phase_c_loss(self.scene, self.p, self.obj_verts, self.edges, self.i, self.n,
                                         grid_res=self.grid_res)
        if torch.isnan(total):
            return total, terms, aux, None
        total.backward()
        grads = {k: self.p[k].grad.detach().clone() for k in PARAM_KEYS}
        grads["obj_verts"] = self.obj_verts.grad.detach().clone()
        if update:
            self.opt.step()
        return total.detach(), {k: v.detach() for k, v in terms.items()}, aux, grads


class PhaseStepper:
    """Phase A (hand only, torch.optim.Adam, PL:1318) or phase B (object only, torch.optim.AdamW, PL:1384)."""

    PHASE1_HAND_LRS = {"scale": 1e-2, "trans": 1e-2, "rot": 0.5}      # guid_config.py:21
    OBJ_2HALF_LRS = {"scale": 1e-2, "trans": 1e-2, "rot": 1e-2}       # guid_config.py:23

    def __init__(self, phase, scene, params):
        assert phase in ("A", "B")
        self.phase, self.scene = phase, scene
        self.p = leafify(params, PARAM_KEYS)
        self.obj_verts = scene["obj_verts"].detach().clone().requires_grad_(phase == "B")
        self.edges = R.unique_edges(scene["obj_faces"])
        if phase == "A":
            l = self.PHASE1_HAND_LRS
            groups = [{"params": [self.p["scale_hand"]], "lr": l["scale"]}, {"params": [self.p["trans_hand"]], "lr": l["trans"]},
                      {"params": [self.p["rot_hand"]], "lr": l["rot"]}]
            self.opt = torch.optim.Adam(groups, eps=1e-4)
        else:
            l = self.OBJ_2HALF_LRS
            groups = [{"params": [self.p["scale_obj"]], "lr": l["scale"]}, {"params": [self.p["trans_obj"]], "lr": l["trans"]},
                      {"params": [self.p["rot_obj"]], "lr": l["rot"]}]
            self.opt = torch.optim.AdamW(groups, eps=1e-4)

    def step(self, update=True):
        self.opt.zero_grad()
        self.obj_verts.grad = None
        if self.phase == "A":
            total, terms, aux = phase_a_loss(self.scene, self.p)
            keys = ["scale_hand", "trans_hand", "rot_hand"]
        else:
            total, terms, aux = phase_b_loss(self.scene, self.p, self.obj_verts, self.edges)
            keys = ["scale_obj", "trans_obj", "rot_obj"]
        total.backward()
        grads = {k: self.p[k].grad.detach().clone() for k in keys}
        if self.phase == "B":
            grads["obj_verts"] = self.obj_verts.grad.detach().clone()
        if update:
            self.opt.step()
        return total.detach(), {k: v.detach() for k, v in terms.items()}, aux, grads


def loss_without_silhouette(phase, scene, params, denoise_i=19, num_inference_steps=20, grid_res=64):
    """Total loss of one iteration of `phase` WITHOUT its silhouette BCE term, and its gradients, at `params` (no update).

    The BCE of a silhouette pixel whose alpha rounds to exactly 1 is clamped (log(0) -> -100, gradient / 1e-12): a jump the
    last bit of one expf decides.  Parity tests compare steps that hold such a pixel on both sides of the clamp through
    this function -- everything but that term -- instead of skipping them."""
    p = leafify(params, PARAM_KEYS)
    ov = scene["obj_verts"].detach().clone().requires_grad_(phase != "A")
    edges = R.unique_edges(scene["obj_faces"])
    if phase == "A":
        total, terms, _ = phase_a_loss(scene, p)
        keys = ["scale_hand", "trans_hand", "rot_hand"]
    elif phase == "B":
        total, terms, _ = phase_b_loss(scene, p, ov, edges)
        keys = ["scale_obj", "trans_obj", "rot_obj"]
    else:
        total, terms, _ = phase_c_loss(scene, p, ov, edges, denoise_i, num_inference_steps, grid_res=grid_res)
        keys = PARAM_KEYS
    name, w = SIL_TERM[phase]
    rest = total - w * terms[name]
    rest.backward()
    grads = {k: p[k].grad.detach().clone() for k in keys}
    if phase != "A":
        grads["obj_verts"] = ov.grad.detach().clone()
    return rest.detach(), grads


def grads_f64(phase, scene, params, renders, denoise_i=19, num_inference_steps=20, grid_res=64, without_silhouette=False, knn_idx=None):
    """THE gradient referee of the parity tests: d total / d (similarity parameters, obj_verts) of one iteration of `phase` with the
    differentiable part of this oracle in FLOAT64, on the fragments of the float32 run.  `renders`: the render records of that run in
    call order (phase A / B: [aux["render"]]; phase C: [aux["hand"]["render"], aux["render"]]).  Two things of the float32 run are
    injected, because the reference's objective is DEFINED through them:
      * the rasteriser's selection (`sel`: face ids, K-buffers) -- so all sides differentiate the very same fragments;
      * the VALUES of the silhouette alphas (`sil`), straight-through: BCE'(alpha) = 1 / (1 - alpha) is evaluated at the float32 alpha the
        reference's loss sees (a pixel whose alpha sits one float32 step below 1 carries 1e4 of phase A's translation gradient -- a
        float64 alpha there is a different objective, 1e3 x off), while d alpha / d vertices is formed in float64.
      * (phase C, `knn_idx`: aux["knn_idx"]) the nearest object vertex of every hand vertex: on the regular synthetic meshes two object
        vertices are often equally near, and which one wins decides where the contact term's gradient lands -- a discrete choice like the
        rasteriser's, not something a float64 run may re-decide.
    float32 torch autograd through normalize / cross / index_add loses 3-4 digits on the vertices with the largest normal gradient
    (NOTEBOOK round 5); the kernels' hand-derived backward does not, so a float32-vs-float32 comparison measures the ORACLE's error there.
    -> (total (float64 tensor), {name: float64 tensor}); without_silhouette: the total minus the phase's silhouette BCE term
    (loss_without_silhouette's comparison)."""
    sc64 = {k: (v.double() if isinstance(v, torch.Tensor) and v.dtype == torch.float32 else v) for k, v in scene.items()}
    sel_q = [r["sel"] for r in renders]
    sil_q = [r["sil"] for r in renders if r.get("sil") is not None]
    real_sel, real_sil, real_knn, real_center = R.rasterize_select, R.render_silhouette, R.knn1, R.bbox_center
    # ... and the vertices that ARE the bounding box (PL:111: the similarity's centre is (min + max) / 2 over the vertices, so its gradient
    # lands on the six extremal vertices; on the synthetic meshes several vertices share an extreme coordinate and rounding picks): the
    # float32 run's choice, found by repeating its float32 arithmetic -- hand vertices as given, object vertices through T_h2m
    centers = []
    with torch.no_grad():
        clouds = []
        if phase in ("A", "C"):
            clouds.append(scene["hand_verts"].float())
        if phase in ("B", "C"):
            clouds.append(R.transform_hunyuan2moge(scene["obj_verts"].detach().float(), scene["T_h2m"].float()))
        for v32 in clouds:
            centers.append((v32.min(dim=0)[1], v32.max(dim=0)[1]))

    def center_injected(verts):
        imin, imax = centers.pop(0)
        cols = torch.arange(3)
        return (verts[imin, cols] + verts[imax, cols]) / 2.0

    def sil_straight_through(*a, **kw):
        alpha = real_sil(*a, **kw)
        a32 = sil_q.pop(0).detach().to(alpha.dtype).reshape(alpha.shape)
        return alpha + (a32 - alpha).detach()

    # ... and the SIGN of every disparity residual (F.l1_loss, PL:1340, 1422, 1497, 1568): the synthetic targets are this oracle's own
    # renders, so at the first step of a scene most residuals are EXACTLY zero in float32 (subgradient 0) where a float64 render differs
    # from the float32 target in its last bits (subgradient +-1 at random): 11 % of the disparity term's vertex gradient
    disp_q = [r["disp"] for r in renders]
    real_l1 = torch.nn.functional.l1_loss

    def l1_injected(pred, target):
        r32 = disp_q.pop(0).detach().float().reshape(pred.shape)
        sign = torch.sign(r32 - target.detach().float()).to(pred.dtype)
        return (sign * (pred - target)).mean()

    def knn_injected(p1, p2):
        idx = knn_idx if isinstance(knn_idx, torch.Tensor) else torch.as_tensor(knn_idx)
        d = p1 - p2[idx.long()]
        return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], idx

    R.rasterize_select = lambda *a, **kw: sel_q.pop(0)
    R.render_silhouette = sil_straight_through
    R.bbox_center = center_injected
    torch.nn.functional.l1_loss = l1_injected
    if knn_idx is not None:
        R.knn1 = knn_injected
    try:
        p64 = leafify({k: v.detach().double() for k, v in params.items()}, PARAM_KEYS)
        ov64 = sc64["obj_verts"].detach().clone().requires_grad_(phase != "A")
        edges = R.unique_edges(scene["obj_faces"])
        if phase == "A":
            total, terms, _ = phase_a_loss(sc64, p64)
            keys = ["scale_hand", "trans_hand", "rot_hand"]
        elif phase == "B":
            total, terms, _ = phase_b_loss(sc64, p64, ov64, edges)
            keys = ["scale_obj", "trans_obj", "rot_obj"]
        else:
            total, terms, _ = phase_c_loss(sc64, p64, ov64, edges, denoise_i, num_inference_steps, grid_res=grid_res)
            keys = PARAM_KEYS
        if without_silhouette:
            name, w = SIL_TERM[phase]
            total = total - w * terms[name]
        total.backward()
    finally:
        R.rasterize_select, R.render_silhouette, R.knn1, R.bbox_center = real_sel, real_sil, real_knn, real_center
        torch.nn.functional.l1_loss = real_l1
    grads = {k: p64[k].grad.detach().clone() for k in keys}
    if phase != "A":
        grads["obj_verts"] = ov64.grad.detach().clone()
    return total.detach(), grads


def referee_grads(phase, scene, params, renders, grads32, denoise_i=19, num_inference_steps=20, grid_res=64, knn_idx=None):
    """The gradient reference of the parity tests for a WHOLE iteration: every term but the silhouette's from `grads_f64` (float64
    derivatives on the float32 run's fragments), the silhouette term's own gradient from the float32 oracle -- `grads32` (the float32
    autograd of the total, a Stepper's `grads`) minus the float32 autograd of the total without it.  The silhouette BCE at sigma = 1e-8 is
    a function of float32 ROUNDINGS (sigmoids that are exactly 0 or 1 in float32 are not in float64, 1 / (1 - alpha) at alphas one float32
    step below 1): there is no float64 version of it to referee with, and its backward is plain arithmetic on edge distances (no
    cancellation to lose digits in).  What float32 autograd does lose -- the normal / disparity / key-point / edge routes through
    normalize, cross, index_add -- is float64 here."""
    _, g32_rest = loss_without_silhouette(phase, scene, params, denoise_i, num_inference_steps, grid_res)
    _, g64_rest = grads_f64(phase, scene, params, renders, denoise_i, num_inference_steps, grid_res, without_silhouette=True, knn_idx=knn_idx)
    return {k: g64_rest[k] + (grads32[k].double() - g32_rest[k].double()) for k in g64_rest}
