"""Batched guidance-step engine: host side of libfoho_hip.so's foho_step_run.

A `GuidanceBatch` packs B independent images (one guidance loop each; the reference processes them one at a
time, src/foho/guidance/run.py:208-259) into device buffers owned by PyTorch-ROCm and drives the HIP kernels
through the C ABI.  PyTorch is only plumbing here: allocation, the current stream, host<->device copies.

Phase recipes follow third_party_patches/hy3dgen/shapegen/pipelines.py (PL) and
src/foho/configs/guid_config.py (CFG) of the reference:
  phase "A"  hand only    PL:1320-1349, Adam  (CFG phase1_hand_lrs)
  phase "B"  object only  PL:1386-1440, AdamW (CFG obj_2half_lrs)
  phase "C"  joint        PL:1480-1588, AdamW (CFG phase2_hand_lrs + obj_lrs)  <- the "guidance step"
"""
import ctypes
import os
import math

import numpy as np
import torch

from . import _lib as L

PARAM_NAMES = ["scale_hand", "trans_hand", "rot_hand", "scale_obj", "trans_obj", "rot_obj"]
PARAM_SLICES = {"scale_hand": slice(0, 1), "trans_hand": slice(1, 4), "rot_hand": slice(4, 8),
                "scale_obj": slice(8, 9), "trans_obj": slice(9, 12), "rot_obj": slice(12, 16)}


def fov_focal(fov_deg, aspect=1.0, znear=0.01):
    """K[0,0], K[1,1] of pytorch3d FoVPerspectiveCameras (RUN:90) in float32 steps."""
    f32 = np.float32
    fov = f32(np.pi / 180.0) * f32(fov_deg)
    tan_half = f32(np.tan(f32(fov / f32(2.0))))
    max_y = f32(tan_half * f32(znear))
    min_y = f32(-max_y)
    max_x = f32(max_y * f32(aspect))
    min_x = f32(-max_x)
    k00 = f32(f32(2.0) * f32(znear)) / f32(max_x - min_x)
    k11 = f32(f32(2.0) * f32(znear)) / f32(max_y - min_y)
    return float(f32(k00)), float(f32(k11))


def blur_radius_from_sigma(sigma=1e-8):
    """RUN:97: np.log(1/1e-4 - 1) * sigma."""
    return float(np.float32(np.log(1.0 / 1e-4 - 1.0) * np.float32(sigma)))


def unique_edges(faces):
    """Unique undirected edges (pytorch3d Meshes.edges_packed semantics: sorted (min,max) pairs)."""
    f = np.asarray(faces, dtype=np.int64)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
    e = np.sort(e, axis=1)
    return np.unique(e, axis=0)


def incidence_csr(faces, n_verts):
    """vertex -> (face<<2 | corner), ordered corner-major then face (the accumulation order of
    pytorch3d's three index_add calls in verts_normals_packed)."""
    f = np.asarray(faces, dtype=np.int64)
    F = f.shape[0]
    v = f.T.reshape(-1)                       # corner-major
    corner = np.repeat(np.arange(3), F)
    face = np.tile(np.arange(F), 3)
    order = np.lexsort((face, corner, v))
    off = np.zeros(n_verts + 1, np.int64)
    np.add.at(off, v + 1, 1)
    off = np.cumsum(off)
    fc = (face[order] << 2) | corner[order]
    return off.astype(np.int32), fc.astype(np.int32)


def neighbour_csr(edges, n_verts):
    e = np.asarray(edges, dtype=np.int64)
    src = np.concatenate([e[:, 0], e[:, 1]])
    dst = np.concatenate([e[:, 1], e[:, 0]])
    order = np.lexsort((dst, src))
    off = np.zeros(n_verts + 1, np.int64)
    np.add.at(off, src + 1, 1)
    off = np.cumsum(off)
    return off.astype(np.int32), dst[order].astype(np.int32)


class OptimizationConfig:
    """Same attributes and call-returns-self behaviour as src/foho/configs/guid_config.py:6-32."""

    def __init__(self):
        self.obj_guidance_scale = 5.0
        self.batch_size = 1
        self.optimization_steps_hand = 200
        self.optimization_steps_joint = 50
        self.optimization_steps_scale = 100
        self.num_inference_steps = 20
        self.guidance_start_step = self.num_inference_steps // 2
        self.handopt_start_step = self.guidance_start_step - 1
        self.guidance_end_step = self.num_inference_steps
        self.phase1_hand_lrs = {"scale": 1e-2, "trans": 1e-2, "rot": 0.5}
        self.phase2_hand_lrs = {"scale": 1e-4, "trans": 1e-4, "rot": 1e-2}
        self.obj_2half_lrs = {"scale": 1e-2, "trans": 1e-2, "rot": 1e-2}
        self.obj_lrs = {"scale": 5e-2, "trans": 1e-2, "rot": 1e-2}
        self.noise_obj_lr1 = 1e-4
        self.noise_obj_lr2 = 1e-2
        self.use_intersection_loss = True

    def __call__(self):
        return self


def _lr16(hand=None, obj=None):
    lr = [0.0] * 16
    if hand:
        lr[0] = hand["scale"]
        lr[1:4] = [hand["trans"]] * 3
        lr[4:8] = [hand["rot"]] * 4
    if obj:
        lr[8] = obj["scale"]
        lr[9:12] = [obj["trans"]] * 3
        lr[12:16] = [obj["rot"]] * 4
    return lr


def phase_cfg(phase, config=None, denoise_i=19, do_update=True, sigma=1e-8, gamma=1e-8):
    """foho_step_cfg for one inner-loop iteration of phase 'A', 'B' or 'C'."""
    config = config or OptimizationConfig()
    c = L.FohoStepCfg()
    c.sigma, c.gamma = sigma, gamma
    c.blur_radius = blur_radius_from_sigma(sigma)
    c.beta1, c.beta2, c.eps = 0.9, 0.999, 1e-4
    c.contact_margin = 0.01
    c.w_int_near, c.w_int_far, c.int_gate = 1e-5, 1e-9, 0.001
    c.do_update = int(do_update)
    r0, r1 = c.render[0], c.render[1]
    if phase == "A":      # PL:1343-1349, torch.optim.Adam (PL:1318)
        r0.face_set, r0.normal_mask, r0.disp_mask, r0.sil_mask = L.FACES_HAND, L.MASK_HAND, L.MASK_HAND, L.MASK_HAND
        r0.w_normal, r0.w_disp, r0.w_sil = 1.0, 10.0, 1.0
        c.w_kps, c.w_trans_hand = 1e-2, 1e-2
        c.weight_decay = 0.0
        lr = _lr16(hand=config.phase1_hand_lrs)
        n_renders = 1
    elif phase == "B":    # PL:1433-1440, torch.optim.AdamW (PL:1384)
        r0.face_set, r0.normal_mask, r0.disp_mask, r0.sil_mask = L.FACES_OBJ, L.MASK_OBJ, L.MASK_OBJ, L.MASK_OBJ
        r0.w_normal, r0.w_disp, r0.w_sil = 10.0, 10.0, 100.0
        c.w_edge, c.w_verts_obj, c.w_trans_obj = 1.0, 1e-3, 1e-2
        c.weight_decay = 0.01
        lr = _lr16(obj=config.obj_2half_lrs)
        n_renders = 1
    elif phase == "C":    # PL:1499-1504 nested in PL:1578-1588, torch.optim.AdamW (PL:1478)
        f32 = np.float32
        r0.face_set, r0.normal_mask, r0.disp_mask, r0.sil_mask = L.FACES_HAND, L.MASK_HAND, L.MASK_HAND, L.MASK_NONE
        r0.w_normal, r0.w_disp, r0.w_sil = float(f32(1e-3) * f32(10)), float(f32(1e-3) * f32(10)), 0.0
        r1.face_set, r1.normal_mask, r1.disp_mask, r1.sil_mask = L.FACES_ALL, L.MASK_HOI, L.MASK_NONE, L.MASK_HOI
        r1.w_normal, r1.w_disp, r1.w_sil = 10.0, 10.0, 10.0
        c.w_kps = float(f32(1e-3) * f32(1e-4))
        c.w_trans_hand = float(f32(1e-3) * f32(1e-2))
        c.w_contact, c.w_verts_obj, c.w_edge, c.w_trans_obj = 10.0, 1e-3, 1.0, 1e-3
        c.use_intersection = int(bool(config.use_intersection_loss))
        c.int_gate_step_ok = int(denoise_i >= config.num_inference_steps - 3)
        c.weight_decay = 0.01
        lr = _lr16(hand=config.phase2_hand_lrs, obj=config.obj_lrs)
        n_renders = 2
    else:
        raise ValueError(f"Unknown phase {phase}. Expected 'A' (hand only), 'B' (object only) or 'C' (joint).")
    for i, v in enumerate(lr):
        c.lr[i] = v
    # phases A and B run ONE render on whatever the workspace is laid out for (no re-allocation between the phases of a
    # job), and roles without a weight in this recipe are not launched (foho_step_cfg.n_active_renders)
    c.n_active_renders = n_renders
    return c, n_renders


def morton_order(points):
    """Permutation that visits `points` (N, 3) along a Morton (Z-order) curve of their bounding box: neighbours in the order
    are neighbours in space."""
    p = np.asarray(points, np.float64).reshape(-1, 3)
    if len(p) == 0:
        return np.zeros(0, np.int64)
    lo, hi = p.min(0), p.max(0)
    q = ((p - lo) / np.maximum(hi - lo, 1e-30) * 1023.0).astype(np.int64).clip(0, 1023)

    def spread(x):
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        return (x | (x << 2)) & 0x09249249
    return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")


_WARMED_DEVICES = set()     # GPUs on which this process has launched the step once (GuidanceBatch.capture)


def _same_regressor(jr, first_scene, b):
    """A batch shares ONE joint regressor (foho_step_desc.J_regressor): an image that brings another one would silently be
    regressed with the first image's."""
    j0 = np.asarray(first_scene["J_regressor"], np.float32)
    if jr.shape != j0.shape or not np.array_equal(jr, j0):
        raise L.FohoError(f"GuidanceBatch: image {b} has a J_regressor that differs from image 0's; a batch shares one "
                          "(put images of different hand models into different batches)")


class GuidanceBatch:
    """Device-resident state of B guidance loops.

    scenes: list of dicts with numpy arrays hand_verts (Vh,3) [MoGe space], hand_faces (Fh,3),
    obj_verts (Vo,3) [Hunyuan space], obj_faces (Fo,3), T_h2m (4,4), J_regressor (16,778), kps_2d (21,2),
    moge_normal (H,W,3), moge_disp (H,W), hand_mask (H,W) bool, obj_mask (H,W) bool, fov (deg), H, W.
    """

    def __init__(self, scenes, device="cuda", grid_res=64, frac_cap=1 << 18, n_renders=2, topology="auto", obj_capacity=None,
                 gbuf_f16=False):
        """topology: "auto" builds the tables on the device (foho_topology_tables; closed manifold object meshes) and falls
        back to the host builders when the validity flag says so; "host" always uses the numpy builders; "render" builds
        the incidence lists only (target-map renders need no edge tables); "deferred" allocates the lists and leaves them
        to the caller.

        obj_capacity=(verts, faces): CAPACITY MODE (foho_object_update): every image gets that many object vertex / face
        slots, the actual object of an iteration -- its counts live on the device only -- is installed by
        `SdfObjective` without a host round trip; the scenes' own obj_verts / obj_faces are ignored.

        gbuf_f16: store the depth and colour planes of the G-buffer in half precision (BASELINE configs[4]); face ids, edge
        distances, silhouette products and every accumulation stay fp32."""
        self.lib = L.lib()
        self.device = torch.device(device)
        self.B = B = len(scenes)
        H, W = int(scenes[0]["H"]), int(scenes[0]["W"])
        assert all(int(s["H"]) == H and int(s["W"]) == W for s in scenes), "one image size per batch"
        self.H, self.W = H, W
        self.obj_capacity = None if obj_capacity is None else (int(obj_capacity[0]), int(obj_capacity[1]))
        if self.obj_capacity is not None:
            vcap, fcap = self.obj_capacity
            if vcap < 1 or fcap < 2:
                raise ValueError("obj_capacity: at least 1 vertex and 2 faces")
            scenes = [dict(s, obj_verts=np.zeros((vcap, 3), np.float32), obj_faces=np.zeros((fcap, 3), np.int64)) for s in scenes]
            topology = "capacity"
        verts, faces, images = [], [], []
        v_off = f_off = 0
        self.meta = []
        for s in scenes:
            hv, ov = np.asarray(s["hand_verts"], np.float32), np.asarray(s["obj_verts"], np.float32)
            hf, of = np.asarray(s["hand_faces"], np.int64), np.asarray(s["obj_faces"], np.int64)
            Vh, Vo, Fh, Fo = len(hv), len(ov), len(hf), len(of)
            verts += [hv, ov]
            faces += [hf + v_off, of + v_off + Vh]
            im = L.FohoImage()
            im.v_off, im.Vh, im.Vo, im.f_off, im.Fh, im.Fo = v_off, Vh, Vo, f_off, Fh, Fo
            im.n_edges = 0
            jr = np.asarray(s["J_regressor"], np.float32)
            _same_regressor(jr, scenes[0], len(images))
            im.jcols = jr.shape[1]
            im.k00, im.k11 = fov_focal(float(s["fov"]))
            R = np.asarray(s.get("cam_R", np.diag([-1.0, 1.0, -1.0])), np.float32).reshape(-1)  # RUN:84-90
            T = np.asarray(s.get("cam_T", np.zeros(3)), np.float32)
            for k in range(9):
                im.cam_R[k] = float(R[k])
            for k in range(3):
                im.cam_T[k] = float(T[k])
            im.znear, im.zfar = 0.01, 100.0
            M = np.asarray(s["T_h2m"], np.float32)[:3, :4].reshape(-1)
            for k in range(12):
                im.T_h2m[k] = float(M[k])
            images.append(im)
            self.meta.append(dict(v_off=v_off, Vh=Vh, Vo=Vo, f_off=f_off, Fh=Fh, Fo=Fo, n_edges=0, fov=float(s["fov"])))
            v_off += Vh + Vo
            f_off += Fh + Fo
        self.Vtot, self.Ftot = v_off, f_off
        verts = np.concatenate(verts, 0)
        faces = np.concatenate(faces, 0)
        dev = self.device
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
        self.verts_in = t(verts, torch.float32)
        self.faces = t(faces, torch.int32)
        obj_flag = np.zeros(self.Vtot, np.uint8)
        for m in self.meta:
            obj_flag[m["v_off"] + m["Vh"]:m["v_off"] + m["Vh"] + m["Vo"]] = 1
        ok = False
        if topology == "capacity":      # tables are (re)built on the device by foho_object_update; sized for the capacity
            ok = True
            self.obj_flag = t(obj_flag, torch.uint8)
            self.inc_off = torch.zeros(self.Vtot + 1, dtype=torch.int32, device=dev)
            self.inc_fc = torch.zeros(3 * self.Ftot, dtype=torch.int32, device=dev)
            self.nbr_off, self.nbr_idx = self.inc_off, torch.zeros(3 * self.Ftot, dtype=torch.int32, device=dev)
            self.obj_counts = torch.zeros(B, 3, dtype=torch.int32, device=dev)           # actual (Vo, Fo, overflow bits)
            self.obj_faces64 = torch.zeros(B, self.obj_capacity[1], 3, dtype=torch.int64, device=dev)
            self.lib.foho_object_workspace_bytes.restype = ctypes.c_size_t
            self.obj_ws = torch.zeros(self.lib.foho_object_workspace_bytes(self.Vtot, self.Ftot), dtype=torch.uint8, device=dev)
        elif topology == "deferred":    # the caller builds the incidence lists itself before the first step (TargetRenderer)
            ok = True
            self.inc_off = torch.zeros(self.Vtot + 1, dtype=torch.int32, device=dev)
            self.inc_fc = torch.zeros(3 * max(self.Ftot, 1), dtype=torch.int32, device=dev)
            self.nbr_off, self.nbr_idx = torch.zeros(self.Vtot + 1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        elif topology in ("auto", "render") and self.Ftot > 0:
            ok = self._device_topology(t(obj_flag, torch.uint8) if topology == "auto" else None)
            if ok and topology == "auto":
                ok = all(m["Fo"] % 2 == 0 for m in self.meta)
        if ok and topology == "capacity":
            for m, im in zip(self.meta, images):     # no object yet: the records are filled in on the device
                m.update(Vcap=m["Vo"], Fcap=m["Fo"], Vo=0, Fo=0)
                im.Vo = im.Fo = im.n_edges = 0
        elif ok:
            for m, im in zip(self.meta, images):
                m["n_edges"] = im.n_edges = (3 * m["Fo"] // 2) if topology == "auto" else 0
        else:   # general meshes (boundaries, non-manifold edges): sort-based builders on the host
            inc_off, inc_fc = incidence_csr(faces, self.Vtot)
            edges_all = []
            for m, im in zip(self.meta, images):
                lo = m["f_off"] + m["Fh"]
                e = unique_edges(faces[lo:lo + m["Fo"]]) if m["Fo"] else np.zeros((0, 2), np.int64)
                m["n_edges"] = im.n_edges = len(e)
                edges_all.append(e)
            edges = np.concatenate(edges_all, 0) if edges_all else np.zeros((0, 2), np.int64)
            nbr_off, nbr_idx = neighbour_csr(edges, self.Vtot)
            if len(nbr_idx) == 0:
                nbr_idx = np.zeros(1, np.int32)
            self.inc_off, self.inc_fc = t(inc_off, torch.int32), t(inc_fc, torch.int32)
            self.nbr_off, self.nbr_idx = t(nbr_off, torch.int32), t(nbr_idx, torch.int32)
        self.J = t(np.asarray(scenes[0]["J_regressor"], np.float32), torch.float32)
        self._images_host = images
        img_bytes = b"".join(bytes(im) for im in images)
        self.images = torch.frombuffer(bytearray(img_bytes), dtype=torch.uint8).to(dev)
        if all("moge_normal" in s for s in scenes):
            self.tgt_normal = t(np.stack([s["moge_normal"] for s in scenes]), torch.float32)
            self.tgt_disp = t(np.stack([s["moge_disp"] for s in scenes]), torch.float32)
        else:       # target maps come from a TargetRenderer, on the device
            self.tgt_normal = torch.zeros(B, H, W, 3, device=dev)
            self.tgt_disp = torch.zeros(B, H, W, device=dev)
        mask = np.stack([(np.asarray(s["hand_mask"]).astype(np.uint8) | (np.asarray(s["obj_mask"]).astype(np.uint8) << 1))
                         for s in scenes])
        self.mask = t(mask, torch.uint8)
        self.kps_2d = t(np.stack([s["kps_2d"] for s in scenes]), torch.float32)

        ident = np.array([1, 0, 0, 0, 1, 0, 0, 0] * 2, np.float32)  # PL:1207-1215: s=1, t=0, q=(1,0,0,0)
        self.params = t(np.tile(ident, (B, 1)), torch.float32)
        self.adam_m = torch.zeros(B, 16, device=dev)
        self.adam_v = torch.zeros(B, 16, device=dev)
        self.adam_t = torch.zeros(B, dtype=torch.int32, device=dev)
        self.losses = torch.zeros(B, L.N_LOSS, device=dev)
        self.grad_params = torch.zeros(B, 16, device=dev)
        self.grad_verts_in = torch.zeros(self.Vtot, 3, device=dev)
        self.flags = torch.zeros(B, dtype=torch.int32, device=dev)

        d = L.FohoDims()
        d.B, d.H, d.W, d.Vtot, d.Ftot = B, H, W, self.Vtot, self.Ftot
        d.Vmax = max(m["Vh"] + m.get("Vcap", m["Vo"]) for m in self.meta)
        d.Fmax = max(m["Fh"] + m.get("Fcap", m["Fo"]) for m in self.meta)
        d.Vh_max = max(m["Vh"] for m in self.meta)
        d.Vo_max = max(m.get("Vcap", m["Vo"]) for m in self.meta)
        d.Fh_max = max(m["Fh"] for m in self.meta)
        d.Fo_max = max(m.get("Fcap", m["Fo"]) for m in self.meta)
        d.grid_res, d.frac_cap, d.n_renders = grid_res, frac_cap, n_renders
        d.gbuf_f16 = int(bool(gbuf_f16))     # BASELINE configs[4]: depth / colour planes of the G-buffer in fp16, sums in fp32
        self.dims = d
        self._pinned, self._uploaded = {}, None      # page-locked upload mirrors of load_scenes()
        self._hand_order = None
        self._alloc_workspace()
        self._set_hand_order([np.asarray(s["hand_verts"], np.float32) for s in scenes])
        if self.obj_capacity is not None:
            self.adopt_objects()        # hands only for now: tables, pair table and AABB of the (still empty) scene
            self.flags.zero_()          # ... which is not an "empty iso-surface" event (flag bit 6)

    # ------------------------------------------------------------------ plumbing
    def _alloc_workspace(self):
        nbytes = int(self.lib.foho_step_workspace_bytes(ctypes.byref(self.dims)))
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self._desc = None
        # the AABB of verts_in lives in the workspace (FOHO_STAGE_BBOX); capacity mode recomputes it in adopt_objects()
        self._bbox_dirty = getattr(self, "obj_capacity", None) is None
        self._targets_dirty = True   # ... and so do the static loss sums of the target maps (FOHO_STAGE_TARGETS)
        if getattr(self, "_hand_order", None) is not None:
            self._upload_hand_order()

    def _set_hand_order(self, hand_verts_per_image):
        """Lane -> hand vertex table of the nearest-neighbour role (workspace region "hand_order", stored as a delta on the
        lane slot): the hand's vertices in Morton order, so that a wave's 64 vertices are neighbours in space and agree more
        often on the candidate runs they can skip (k_vertex.inc).  Any permutation gives the same results."""
        vh_max = max(int(self.dims.Vh_max), 1)
        tab = np.zeros((self.B, vh_max), np.int32)
        for b, hv in enumerate(hand_verts_per_image):
            n = len(hv)
            if n:
                tab[b, :n] = morton_order(hv) - np.arange(n)
        self._hand_order = tab
        self._desc = None                 # hand_order_valid / hand_faces_per_block are derived fields of the descriptor
        self._upload_hand_order()

    def _upload_hand_order(self):
        reg = self.region("hand_order", torch.int32, self._hand_order.shape)
        buf = self._pinned.get("hand_order")
        if buf is None or tuple(buf.shape) != tuple(self._hand_order.shape):
            buf = self._pinned["hand_order"] = torch.empty(self._hand_order.shape, dtype=torch.int32, pin_memory=True)
        if self._uploaded is not None:
            self._uploaded.synchronize()
        buf.numpy()[...] = self._hand_order
        reg.copy_(buf, non_blocking=True)
        if self._uploaded is None:
            self._uploaded = torch.cuda.Event()
        self._uploaded.record()

    def adopt_objects(self, stream=None):
        """Capacity mode: install the object meshes whose vertices sit in the object slots of verts_in, whose mesh-local
        faces sit in obj_faces64 and whose counts sit in obj_counts (all on the device, as foho_flexi_fwd leaves them):
        image records, global face ids, topology tables, pair table, AABB.  Asynchronous, no host round trip."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        P = ctypes.c_void_p
        self.lib.foho_object_update.restype = ctypes.c_int
        L.check(self.lib.foho_object_update(ctypes.byref(self.desc()), P(self.obj_counts.data_ptr()), P(self.obj_faces64.data_ptr()),
                                            ctypes.c_int32(self.obj_capacity[1]), P(self.obj_flag.data_ptr()), P(self.obj_ws.data_ptr()),
                                            ctypes.c_size_t(self.obj_ws.numel()), P(stream)), "foho_object_update")

    def obj_slot(self, b):
        """(first vertex slot, vertex capacity) of image b's object in verts_in / grad_verts_in."""
        m = self.meta[b]
        return m["v_off"] + m["Vh"], m.get("Vcap", m["Vo"])

    def set_n_renders(self, n):
        """Make sure the workspace holds `n` renders.  It only ever grows: a step whose cfg activates fewer renders
        (foho_step_cfg.n_active_renders, set by phase_cfg) runs on the larger layout as it is, so the phases of a job share
        one workspace -- AABB, pair table, clean scatter planes and static target sums included."""
        if n > self.dims.n_renders:
            self.dims.n_renders = n
            self._alloc_workspace()

    def desc(self):
        if self._desc is None:
            d = L.FohoStepDesc()
            d.dims = self.dims
            for name in ["images", "verts_in", "faces", "inc_off", "inc_fc", "nbr_off", "nbr_idx", "tgt_normal",
                         "tgt_disp", "mask", "kps_2d", "params", "adam_m", "adam_v", "adam_t", "losses",
                         "grad_params", "grad_verts_in", "flags", "workspace"]:
                setattr(d, name, getattr(self, name).data_ptr())
            d.J_regressor = self.J.data_ptr()
            d.workspace_bytes = self.workspace.numel()
            d.hand_order_valid = int(getattr(self, "_hand_order", None) is not None)   # the table is (re-)uploaded with every workspace
            # crops (the reference's frames: fov ~ 25 degrees, big hand faces): 2 hand faces per raster workgroup at one image
            fovs = [float(m["fov"]) for m in self.meta if "fov" in m]
            d.hand_faces_per_block = 2 if (self.B == 1 and fovs and max(fovs) < 40.0) else 0
            self._desc = d
        return self._desc

    def region(self, name, dtype, shape=None):
        """View of a named workspace region (parity tests inspect intermediates through this)."""
        n = ctypes.c_int64(0)
        off = self.lib.foho_step_workspace_region(ctypes.byref(self.dims), L.WS_REGIONS.index(name), ctypes.byref(n))
        assert off >= 0
        v = self.workspace[off:off + n.value].view(dtype)
        return v.reshape(shape) if shape is not None else v

    def prepare(self, stream=None):
        """Run the per-input stages the step relies on (AABB / pair table / clean planes: FOHO_STAGE_BBOX; static sums of the
        target maps: FOHO_STAGE_TARGETS) now, eagerly, if they are due -- a captured hipGraph only holds the step itself."""
        stages = 0
        if self._bbox_dirty:
            stages |= L.STAGE_BBOX
            self._bbox_dirty = False
        if self._targets_dirty:
            stages |= L.STAGE_TARGETS
            self._targets_dirty = False
        if stages:
            if stream is None:
                stream = torch.cuda.current_stream(self.device).cuda_stream
            cfg, _ = phase_cfg("C", do_update=False)
            L.check(self.lib.foho_step_run(ctypes.byref(self.desc()), ctypes.byref(cfg), int(stages), ctypes.c_void_p(stream)),
                    "foho_step_run(prepare)")

    def refresh_world(self, stream=None):
        """Recompute the world-space vertices (workspace region "world") from the CURRENT parameters.  A step transforms the
        vertices first and updates the parameters last, so after a loop the region is one optimiser update behind; the
        reference builds its output meshes from the final parameters (PL:1614-1618, 1653-1657).  One vertex-stage call."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        self.prepare(stream)
        cfg, _ = phase_cfg("C", do_update=False)
        L.check(self.lib.foho_step_run(ctypes.byref(self.desc()), ctypes.byref(cfg), int(L.STAGE_VERTEX), ctypes.c_void_p(stream)),
                "foho_step_run(VERTEX)")

    def fits(self, scenes):
        """True when `scenes` can be loaded into this (capacity-mode) batch in place."""
        if self.obj_capacity is None or len(scenes) != self.B:
            return False
        vcap, fcap = self.obj_capacity
        for s, m in zip(scenes, self.meta):
            if (int(s["H"]), int(s["W"])) != (self.H, self.W) or len(s["hand_verts"]) != m["Vh"] or len(s["hand_faces"]) != m["Fh"]:
                return False
            if len(s["obj_verts"]) > vcap or len(s["obj_faces"]) > fcap or np.asarray(s["J_regressor"]).shape[1] != self.J.shape[1]:
                return False
        return True

    def load_scenes(self, scenes):
        """CAPACITY MODE: put B new images into this batch IN PLACE -- same buffers, same addresses, so every hipGraph captured
        on it stays valid and a per-image job pays for its captures once per process, not once per image (the reference
        rebuilds renderer, pipeline and optimisers per image, RUN:140).  Geometry, camera, targets and key points are
        overwritten, the objects (any vertex / face count within the capacity) are installed on the device
        (foho_object_update: image records, topology tables, pair table, AABB), pose parameters return to the identity
        (PL:1207-1215) and the optimiser is reset.  Asynchronous on the current stream after the host-to-device copies.
        `load_flags` keeps what the installation reported (bit 4 capacity, bit 5 not a closed manifold, bit 6 empty)."""
        if not self.fits(scenes):
            raise L.FohoError("load_scenes: needs a capacity-mode batch and scenes of the same image size / hand topology "
                              "whose objects fit the capacity")
        dev = self.device
        vcap, fcap = self.obj_capacity
        verts = np.zeros((self.Vtot, 3), np.float32)
        hfaces = []
        faces64 = np.zeros((self.B, fcap, 3), np.int64)
        counts = np.zeros((self.B, 3), np.int32)
        for b, (s, m, im) in enumerate(zip(scenes, self.meta, self._images_host)):
            ov, of = np.asarray(s["obj_verts"], np.float32).reshape(-1, 3), np.asarray(s["obj_faces"], np.int64).reshape(-1, 3)
            nv, nf = len(ov), len(of)
            lo = m["v_off"]
            verts[lo:lo + m["Vh"]] = np.asarray(s["hand_verts"], np.float32)
            verts[lo + m["Vh"]:lo + m["Vh"] + nv] = ov
            hfaces.append(np.asarray(s["hand_faces"], np.int64) + lo)
            faces64[b, :nf] = of
            counts[b, :2] = (nv, nf)
            m.update(Vo=nv, Fo=nf, n_edges=3 * nf // 2, fov=float(s["fov"]))
            im.k00, im.k11 = fov_focal(float(s["fov"]))
            R = np.asarray(s.get("cam_R", np.diag([-1.0, 1.0, -1.0])), np.float32).reshape(-1)
            T = np.asarray(s.get("cam_T", np.zeros(3)), np.float32)
            M = np.asarray(s["T_h2m"], np.float32)[:3, :4].reshape(-1)
            for k in range(9):
                im.cam_R[k] = float(R[k])
            for k in range(3):
                im.cam_T[k] = float(T[k])
            for k in range(12):
                im.T_h2m[k] = float(M[k])
            im.Vo = im.Fo = im.n_edges = 0        # the device-side records get the actual counts from foho_object_update
        # Uploads go through page-locked mirrors owned by this batch and are ASYNCHRONOUS on the current stream: the host does
        # not wait behind whatever that stream still has queued (inputs.MeshGuidanceRunner keeps a second slot's whole job
        # queued on it).  A mirror is rewritten only after the copies of the previous load have run (`_uploaded`).
        if self._uploaded is not None:
            self._uploaded.synchronize()

        def up(key, dst, fill):
            buf = self._pinned.get(key)
            if buf is None:
                buf = self._pinned[key] = torch.empty(dst.shape, dtype=dst.dtype, pin_memory=True)
            fill(buf.numpy())
            dst.copy_(buf, non_blocking=True)

        def put(a):
            def fill(view):
                view[...] = np.asarray(a).reshape(view.shape)
            return fill

        def per_image(get):
            def fill(view):
                for b, s in enumerate(scenes):
                    view[b] = get(s)
            return fill

        up("verts", self.verts_in, put(verts))
        for b, (m, hf) in enumerate(zip(self.meta, hfaces)):
            up(("hand_faces", b), self.faces[m["f_off"]:m["f_off"] + m["Fh"]], put(hf))
        up("obj_faces", self.obj_faces64, put(faces64))
        up("counts", self.obj_counts, put(counts))
        up("images", self.images, put(np.frombuffer(b"".join(bytes(im) for im in self._images_host), np.uint8)))
        if all("moge_normal" in s for s in scenes):      # else: the caller renders them into tgt_normal / tgt_disp (TargetRenderer)
            up("tgt_normal", self.tgt_normal, per_image(lambda s: s["moge_normal"]))
            up("tgt_disp", self.tgt_disp, per_image(lambda s: s["moge_disp"]))
        up("mask", self.mask, per_image(lambda s: np.asarray(s["hand_mask"]).astype(np.uint8) | (np.asarray(s["obj_mask"]).astype(np.uint8) << 1)))
        up("kps", self.kps_2d, per_image(lambda s: s["kps_2d"]))
        # the joint regressor is shared by the batch (foho_step_desc.J_regressor; MANO's is a constant of the hand model): the new
        # image set's, not the first set's -- and one for all of its images, or the set is refused
        for b, s in enumerate(scenes):
            _same_regressor(np.asarray(s["J_regressor"], np.float32), scenes[0], b)
        up("J", self.J, put(np.asarray(scenes[0]["J_regressor"], np.float32)))
        self._desc = None                 # the descriptor's derived fields (faces per raster workgroup: by field of view) follow the new set
        up("params", self.params, put(np.tile(np.array([1, 0, 0, 0, 1, 0, 0, 0] * 2, np.float32), (self.B, 1))))
        vh_max = max(int(self.dims.Vh_max), 1)

        def order(view):
            view[...] = 0
            for b, s in enumerate(scenes):
                n = len(s["hand_verts"])
                view[b, :n] = morton_order(s["hand_verts"]) - np.arange(n)
            self._hand_order = view.copy()
        up("hand_order", self.region("hand_order", torch.int32, (self.B, vh_max)), order)
        if self._uploaded is None:
            self._uploaded = torch.cuda.Event()
        self._uploaded.record()
        self.reset_optimizer()
        self.adopt_objects()
        self.load_flags = self.flags.clone()
        self.flags.zero_()
        self._targets_dirty = True
        if all("moge_normal" in s for s in scenes):
            self.prepare()

    # ------------------------------------------------------------------ state
    def reset_optimizer(self):
        """A fresh torch.optim.Adam/AdamW is created at every denoising step (PL:1318, 1384, 1478)."""
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.adam_t.zero_()
        self.flags.zero_()

    def set_params(self, b, **kw):
        for k, v in kw.items():
            self.params[b, PARAM_SLICES[k]] = torch.as_tensor(v, dtype=torch.float32, device=self.device).reshape(-1)

    def get_params(self, b):
        p = self.params[b].detach().cpu()
        return {k: p[s].clone() for k, s in PARAM_SLICES.items()}

    def set_obj_verts(self, b, verts):
        m = self.meta[b]
        lo = m["v_off"] + m["Vh"]
        self.verts_in[lo:lo + m["Vo"]] = torch.as_tensor(verts, dtype=torch.float32, device=self.device)
        self.refresh_bbox()       # eagerly, so that an already captured graph sees the new centre

    def refresh_bbox(self, stream=None):
        """Recompute the AABB of the input meshes (centre of the similarity transform, PL:111).  Needed once and after
        every change of verts_in; the per-iteration step (STAGE_STEP) reuses it."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        cfg, _ = phase_cfg("C", do_update=False)
        L.check(self.lib.foho_step_run(ctypes.byref(self.desc()), ctypes.byref(cfg), int(L.STAGE_BBOX), ctypes.c_void_p(stream)),
                "foho_step_run(BBOX)")
        self._bbox_dirty = False

    def update_object(self, verts, faces, assume_manifold=True):
        """New object mesh with a NEW TOPOLOGY (the FlexiCubes output of this iteration, PL:1393 / 1509) for a one-image
        batch -- the reference's batch size.  The topology tables (vertex -> incident faces in pytorch3d's index_add
        order, unique edges, neighbour lists) are rebuilt on the GPU: by foho_topology_tables (k_topo.inc) for closed
        manifold meshes -- what FlexiCubes emits --, by general torch sorts otherwise; one host read-back (validity flag).
        The next step recomputes the cached AABB and clears the rasteriser's planes (FOHO_STAGE_BBOX)."""
        if self.obj_capacity is not None:
            raise L.FohoError("update_object: this batch is in capacity mode (use SdfObjective / adopt_objects)")
        if self.B != 1:
            raise L.FohoError("update_object: exact-size topology updates are implemented for one-image batches; capacity mode "
                              "(obj_capacity=...) handles any batch size")
        dev = self.device
        v = torch.as_tensor(verts, dtype=torch.float32, device=dev).detach().reshape(-1, 3).contiguous()
        f = torch.as_tensor(faces, device=dev).detach().to(torch.int64).reshape(-1, 3)
        m = self.meta[0]
        Vh, Fh = m["Vh"], m["Fh"]
        Vo, Fo = int(v.shape[0]), int(f.shape[0])
        Vtot, Ftot = Vh + Vo, Fh + Fo
        self.verts_in = torch.cat([self.verts_in[:Vh], v], 0).contiguous()
        faces_all = torch.cat([self.faces[:Fh].to(torch.int64), f + Vh], 0)
        self.faces = faces_all.to(torch.int32).contiguous()
        self.Vtot, self.Ftot = Vtot, Ftot          # _device_topology sizes its tables from these
        n_edges = -1
        if assume_manifold and Fo > 0 and Fo % 2 == 0:
            # dual-marching-cubes output is a closed oriented 2-manifold: tables straight from the incidence lists
            # (foho_topology_tables, k_topo.inc); the flag tells when the assumption does not hold
            flags = torch.zeros(Vtot, dtype=torch.uint8, device=dev)
            flags[Vh:] = 1
            if self._device_topology(flags):
                n_edges = 3 * Fo // 2
        if n_edges < 0:
            self._topology_by_sort(faces_all, f, Vh, Vtot, Ftot)
            n_edges = self._n_edges
        # image record, sizes, outputs
        im = self._images_host[0]
        im.Vo, im.Fo, im.n_edges = Vo, Fo, n_edges
        self.images = torch.frombuffer(bytearray(bytes(im)), dtype=torch.uint8).to(dev)
        m.update(Vo=Vo, Fo=Fo, n_edges=n_edges)
        self.Vtot, self.Ftot = Vtot, Ftot
        d = self.dims
        d.Vtot, d.Ftot, d.Vmax, d.Fmax, d.Vo_max, d.Fo_max = Vtot, Ftot, Vtot, Ftot, Vo, Fo
        self.grad_verts_in = torch.zeros(Vtot, 3, device=dev)
        need = int(self.lib.foho_step_workspace_bytes(ctypes.byref(d)))
        if need > self.workspace.numel():
            self.workspace = torch.zeros(int(need * 1.25), dtype=torch.uint8, device=dev)   # head-room: sizes drift slowly
            self._targets_dirty = True
        self._desc = None
        self._bbox_dirty = True     # the workspace layout moved with the sizes: AABB + clean scatter planes again

    def _device_topology(self, obj_flag):
        """foho_topology_tables on self.faces: incidence lists (+ neighbour lists at the same offsets when obj_flag marks the
        object vertices).  Returns False when the validity flag asks for the general path (one host read-back)."""
        lib, dev = self.lib, self.device
        Vtot, Ftot = int(self.Vtot), int(self.faces.shape[0])
        lib.foho_topology_workspace_bytes.restype = ctypes.c_size_t
        nws = lib.foho_topology_workspace_bytes(Vtot)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        inc_off = torch.empty(Vtot + 1, dtype=torch.int32, device=dev)
        inc_fc = torch.empty(3 * Ftot, dtype=torch.int32, device=dev)
        nbr_idx = torch.empty(3 * Ftot if obj_flag is not None else 1, dtype=torch.int32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        P = ctypes.c_void_p
        L.check(lib.foho_topology_tables(P(self.faces.data_ptr()), Vtot, Ftot, P(obj_flag.data_ptr()) if obj_flag is not None else None,
                                         P(inc_off.data_ptr()), P(inc_fc.data_ptr()), P(nbr_idx.data_ptr()), P(flag.data_ptr()),
                                         P(ws.data_ptr()), ctypes.c_size_t(nws),
                                         P(torch.cuda.current_stream(dev).cuda_stream)), "foho_topology_tables")
        if int(flag.item()) != 0:
            return False
        self.inc_off, self.inc_fc = inc_off, inc_fc
        if obj_flag is not None:
            self.nbr_off, self.nbr_idx = inc_off, nbr_idx  # neighbour lists share the incidence offsets
        else:                                              # no edge tables: every neighbour list is empty
            self.nbr_off, self.nbr_idx = torch.zeros(Vtot + 1, dtype=torch.int32, device=dev), nbr_idx
        return True

    def _topology_by_sort(self, faces_all, f, Vh, Vtot, Ftot):
        """General path (any mesh): torch sorts on the device."""
        dev = self.device
        # vertex -> (face << 2 | corner), ordered by (vertex, corner, face): unique keys, one sort
        vv = faces_all.t().reshape(-1)
        corner = torch.arange(3, device=dev).repeat_interleave(Ftot)
        face = torch.arange(Ftot, device=dev).repeat(3)
        order = torch.argsort((vv * 3 + corner) * Ftot + face)
        self.inc_fc = ((face[order] << 2) | corner[order]).to(torch.int32).contiguous()
        off = torch.zeros(Vtot + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.bincount(vv, minlength=Vtot).cumsum(0)
        self.inc_off = off.to(torch.int32).contiguous()
        # unique undirected edges of the OBJECT mesh (pytorch3d edges_packed) and the neighbour lists over them
        fo = f + Vh
        e = torch.cat([fo[:, [0, 1]], fo[:, [1, 2]], fo[:, [2, 0]]], 0)
        key = torch.unique(e.min(1).values * Vtot + e.max(1).values)
        ea, eb = key // Vtot, key % Vtot
        self._n_edges = int(key.numel())
        src, dst = torch.cat([ea, eb]), torch.cat([eb, ea])
        order = torch.argsort(src * Vtot + dst)
        noff = torch.zeros(Vtot + 1, dtype=torch.int64, device=dev)
        noff[1:] = torch.bincount(src, minlength=Vtot).cumsum(0)
        self.nbr_off = noff.to(torch.int32).contiguous()
        self.nbr_idx = (dst[order] if self._n_edges else torch.zeros(1, dtype=torch.int64, device=dev)).to(torch.int32).contiguous()

    def objective(self, verts, faces, cfg):
        """Differentiable scalar: total loss of one iteration on the object mesh (verts, faces); backward() delivers
        dL/d verts (so that an upstream FlexiCubes / VAE receives its gradient, PL:1600).  The pose parameters are
        updated by the iteration when cfg.do_update is set, exactly like gb.step()."""
        return _ObjectiveFn.apply(verts, faces, self, cfg)

    def grad_obj_verts(self, b):
        m = self.meta[b]
        lo = m["v_off"] + m["Vh"]
        return self.grad_verts_in[lo:lo + m["Vo"]]

    # ------------------------------------------------------------------ the step
    def step(self, cfg, stages=L.STAGE_STEP, stream=None):
        """One iteration for every image of the batch; asynchronous on the current stream."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        if self._bbox_dirty and (stages & L.STAGE_VERTEX):
            stages |= L.STAGE_BBOX
            self._bbox_dirty = False
        if self._targets_dirty and (stages & L.STAGE_LOSS):
            stages |= L.STAGE_TARGETS
            self._targets_dirty = False
        L.check(self.lib.foho_step_run(ctypes.byref(self.desc()), ctypes.byref(cfg), int(stages), ctypes.c_void_p(stream)),
                "foho_step_run")

    def loss_dict(self, b=0):
        l = self.losses[b].detach().cpu().tolist()
        return dict(zip(L.LOSS_NAMES, l))

    def raise_on_flags(self, strict_k=True, ignore_images=()):
        """bit1: fractional-fragment list overflow; bit2: the K = 100 buffer of a pixel with 100 fractional-coverage fragments
        or more could not be re-built on the device (more than 1024 fragments on the pixel, or more than 32 such pixels in a
        render): the silhouette over all fragments may then differ from the reference's 100 nearest ones; bits 4-6: capacity
        mode (foho_object_update).  (Bit 3 is retired: faces across the near plane are clipped like pytorch3d clips them.)
        Fails loudly instead of deviating; strict_k=False downgrades bit2 to a warning (a collapsing object -- thousands
        of sub-pixel faces on one pixel -- is outside any regime where the K=100 cut-off is meaningful).  ignore_images: slots whose
        flags the caller has dealt with (images that left a batch and stay frozen)."""
        f = self.flags.detach().cpu().numpy()
        if len(ignore_images):
            f = f.copy()
            f[list(ignore_images)] = 0
        if (f & 2).any():
            raise L.FohoError("fractional-coverage fragment list overflowed: raise frac_cap")
        if (f & 16).any():
            raise L.FohoError(f"object capacity exceeded (images {np.flatnonzero(f & 16).tolist()}): enlarge obj_capacity")
        if (f & 64).any():
            raise L.FohoError(f"empty iso-surface (images {np.flatnonzero(f & 64).tolist()}): the iteration was skipped")
        if (f & 32).any():
            raise L.FohoError(f"object mesh is not a closed 2-manifold (images {np.flatnonzero(f & 32).tolist()}): the on-device "
                              "edge tables assume it; use the exact-size path (GuidanceBatch.objective)")
        if (f & 4).any():
            msg = "K = 100 silhouette not reproduced: more than 1024 fragments on a pixel, or more than 32 pixels with 100+ fractional-coverage fragments"
            if strict_k:
                raise L.FohoError(msg)
            import warnings
            warnings.warn(msg + f" (images {np.flatnonzero(f & 4).tolist()})")
        return f

    # ------------------------------------------------------------------ HIP graph + per-kernel timing
    def capture(self, cfg, steps_per_graph=1):
        """Capture `steps_per_graph` consecutive iterations into one hipGraph (the step has no host sync:
        NaN break, intersection-weight gate and Adam state all live on the device)."""
        self.prepare()      # per-input stages that are due run now, eagerly: the graph holds the step only
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        if dev_index not in _WARMED_DEVICES:
            # first capture of the process on this GPU: one launch outside capture (the code object is loaded lazily, which a
            # capture does not allow); its optimiser update is undone.  The only device-wide synchronisation of a capture.
            s = torch.cuda.Stream(self.device)
            s.wait_stream(torch.cuda.current_stream(self.device))
            state = [t.clone() for t in (self.params, self.adam_m, self.adam_v, self.adam_t, self.flags)]
            with torch.cuda.stream(s):
                self.step(cfg)
            torch.cuda.current_stream(self.device).wait_stream(s)
            torch.cuda.synchronize(self.device)
            for t, saved in zip((self.params, self.adam_m, self.adam_v, self.adam_t, self.flags), state):
                t.copy_(saved)
            torch.cuda.synchronize(self.device)
            _WARMED_DEVICES.add(dev_index)
        # Several iterations in one graph: every iteration but the last leaves its final stage (loss assembly, parameter
        # gradients, Adam) to the prologue of the next one (foho_step_cfg.deferred_update) -- the serial last-workgroup
        # tail of k_vert_bwd disappears from the chain -- and one foho_step_finalize launch closes the graph.
        deferred = steps_per_graph > 1 and os.environ.get("FOHO_NO_DEFERRED_UPDATE") != "1"
        def numbered(k):        # deferred_update = 1 + parity of the iteration: its accumulators are double buffered
            c2 = L.FohoStepCfg.from_buffer_copy(bytes(cfg))
            c2.deferred_update = 1 + (k & 1)
            return c2
        # Recorded on a stream of its own with capture_begin / capture_end: `with torch.cuda.graph(g)` synchronises the DEVICE
        # and empties the allocator cache on entry, i.e. a capture would wait for every other slot's running job.  Nothing
        # executes and nothing is allocated during the recording.
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(torch.cuda.Stream(self.device)):
            g.capture_begin()
            try:
                for k in range(steps_per_graph):
                    self.step(numbered(k) if deferred else cfg)
                if deferred:
                    self.finalize(numbered(steps_per_graph - 1))
            finally:
                g.capture_end()
        return g

    def finalize(self, cfg, stream=None):
        """Apply the update a deferred_update step left pending (no-op when nothing is pending)."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.foho_step_finalize(ctypes.byref(self.desc()), ctypes.byref(cfg), ctypes.c_void_p(stream)),
                "foho_step_finalize")

    def step_profiled(self, cfg, deferred=False):
        """Per-kernel durations {kernel name: milliseconds} of one iteration (hipEvents after every launch).  The library
        runs an un-timed iteration first and times the one behind it, so TWO iterations are executed.  deferred=True times
        the iteration the way it runs inside a multi-iteration hipGraph (foho_step_cfg.deferred_update: the final stage of
        an iteration rides in the next k_xform); a finalize closes the pair."""
        lib = self.lib
        self.prepare()
        lib.foho_step_run_profiled.restype = ctypes.c_int
        lib.foho_kernel_name.restype = ctypes.c_char_p
        ms = (ctypes.c_float * L.N_KERNELS)()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        c2 = cfg
        if deferred:
            c2 = L.FohoStepCfg.from_buffer_copy(bytes(cfg))
            c2.deferred_update = 2
        L.check(lib.foho_step_run_profiled(ctypes.byref(self.desc()), ctypes.byref(c2), ctypes.c_void_p(stream), ms),
                "foho_step_run_profiled")
        names = [lib.foho_kernel_name(i).decode() for i in range(L.N_KERNELS)]
        out = {n: float(ms[i]) for i, n in enumerate(names) if n}
        if deferred:
            self.finalize(c2)
        return out


class _ObjectiveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, faces, gb, cfg):
        gb.update_object(verts, faces)
        gb.step(cfg)
        ctx.gb = gb
        return gb.losses[0, 0].clone()

    @staticmethod
    def backward(ctx, g):
        return ctx.gb.grad_obj_verts(0) * g, None, None, None


class SdfObjective:
    """SDF grid -> FlexiCubes mesh -> one guidance iteration -> dL/dSDF, as ONE hipGraph replay (PL:1507-1601 without the
    VAE decode).  The reference rebuilds the object mesh from the latent in every iteration of phases B and C, so the
    object's vertex count, face count and connectivity change every time; here the whole chain -- iso-surfacing
    (foho_flexi_fwd, 4 launches), installing the new object (foho_object_update, 5 launches: image records, topology
    tables, pair table, AABB), the fused step (6 launches) and the iso-surface backward (2 launches) -- runs over
    fixed-capacity buffers with the actual counts in device memory, so it is captured once per step recipe and replayed.

    gb must be a capacity-mode GuidanceBatch (obj_capacity=...); xyz = (res+1)^3 grid positions shared by the batch.
    `obj(sdf, cfg)` with sdf (B, (res+1)^3) or ((res+1)^3,) returns the total loss per image (autograd: backward() puts
    dL/dSDF into sdf.grad); `status()` reads back (vertex count, face count, flags) of the last call -- one host sync, only
    when the caller wants to know (empty mesh, capacity overflow, non-manifold output)."""

    def __init__(self, gb, xyz, res):
        if gb.obj_capacity is None:
            raise L.FohoError("SdfObjective needs a capacity-mode GuidanceBatch (obj_capacity=(verts, faces))")
        self.gb, self.res = gb, int(res)
        dev = gb.device
        G = self.res + 1
        self.xyz = torch.as_tensor(xyz, dtype=torch.float32, device=dev).reshape(-1, 3).contiguous()
        if self.xyz.shape[0] != G ** 3:
            raise L.FohoError(f"SdfObjective: expected {G ** 3} grid positions")
        self.sdf = torch.zeros(gb.B, G ** 3, device=dev)
        self.grad_sdf = torch.zeros(gb.B, G ** 3, device=dev)
        lib = gb.lib
        lib.foho_flexi_workspace_bytes.restype = ctypes.c_size_t
        self.nws = int(lib.foho_flexi_workspace_bytes(self.res))
        self.flexi_ws = torch.zeros(gb.B, self.nws, dtype=torch.uint8, device=dev)
        self._graphs = {}
        self._stream = torch.cuda.Stream(dev)

    def enqueue(self, cfg):
        """The launch sequence of one iteration on the current stream (eager; the graph captures exactly this)."""
        gb, lib, P = self.gb, self.gb.lib, ctypes.c_void_p
        stream = P(torch.cuda.current_stream(gb.device).cuda_stream)
        vcap, fcap = gb.obj_capacity
        for b in range(gb.B):
            lo, _ = gb.obj_slot(b)
            L.check(lib.foho_flexi_fwd(P(self.xyz.data_ptr()), P(self.sdf[b].data_ptr()), self.res, P(gb.verts_in[lo:].data_ptr()), vcap,
                                       P(gb.obj_faces64[b].data_ptr()), fcap, None, P(gb.obj_counts[b].data_ptr()),
                                       P(self.flexi_ws[b].data_ptr()), ctypes.c_size_t(self.nws), stream), "foho_flexi_fwd")
        gb.adopt_objects()
        gb.step(cfg)
        self.grad_sdf.fill_(0.0)        # a fill kernel (memset nodes inside a captured graph are not reliably ordered on this stack)
        for b in range(gb.B):
            lo, _ = gb.obj_slot(b)
            L.check(lib.foho_flexi_bwd(P(self.xyz.data_ptr()), P(self.sdf[b].data_ptr()), self.res, P(gb.grad_verts_in[lo:].data_ptr()),
                                       vcap, P(self.grad_sdf[b].data_ptr()), None, P(self.flexi_ws[b].data_ptr()),
                                       ctypes.c_size_t(self.nws), stream), "foho_flexi_bwd")

    def graph(self, cfg):
        key = (bytes(cfg), self.gb.workspace.data_ptr())     # a re-allocated workspace (set_n_renders) needs a new capture
        g = self._graphs.get(key)
        if g is None:
            gb = self.gb
            st = self._stream
            st.wait_stream(torch.cuda.current_stream(gb.device))
            state = [t.clone() for t in (gb.params, gb.adam_m, gb.adam_v, gb.adam_t, gb.flags)]
            with torch.cuda.stream(st):
                self.enqueue(cfg)           # warm-up outside the capture (module load); its optimiser update is undone
            torch.cuda.current_stream(gb.device).wait_stream(st)
            torch.cuda.synchronize(gb.device)
            for t, saved in zip((gb.params, gb.adam_m, gb.adam_v, gb.adam_t, gb.flags), state):
                t.copy_(saved)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                self.enqueue(cfg)
            self._graphs[key] = g
        return g

    def run(self, sdf, cfg, use_graph=True):
        """Forward + backward of one iteration for the given SDF values; leaves loss in gb.losses, dL/dSDF in grad_sdf."""
        self.sdf.copy_(sdf.detach().reshape(self.sdf.shape))
        if use_graph:
            self.graph(cfg).replay()
        else:
            self.enqueue(cfg)

    def __call__(self, sdf, cfg, use_graph=True):
        return _SdfObjectiveFn.apply(sdf, self, cfg, use_graph)

    def status(self):
        """[(n_verts, n_faces, flags)] per image of the last run (one host synchronisation)."""
        c = self.gb.obj_counts.cpu().numpy()
        f = self.gb.flags.cpu().numpy()
        return [(int(c[b, 0]), int(c[b, 1]), int(f[b])) for b in range(self.gb.B)]

    def active_rows(self):
        """Grid points per image that carry a gradient after the last run (non-zero entries of dL/dSDF: the end points of the grid edges
        the iso-surface crosses) -- the exact row count the geometry decoder's active-row backward will meet (one host sync)."""
        return torch.count_nonzero(self.grad_sdf, dim=1).cpu().tolist()

    def mesh(self, b=0):
        """(verts (Vo,3), faces (Fo,3) int64 mesh-local) of image b's current object (host sync for the counts)."""
        nv, nf, _ = self.status()[b]
        lo, _ = self.gb.obj_slot(b)
        return self.gb.verts_in[lo:lo + nv].clone(), self.gb.obj_faces64[b, :nf].clone()


class _SdfObjectiveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, obj, cfg, use_graph):
        obj.run(sdf, cfg, use_graph)
        ctx.obj, ctx.shape = obj, sdf.shape
        out = obj.gb.losses[:, 0].clone()
        return out[0] if sdf.dim() == 1 else out

    @staticmethod
    def backward(ctx, g):
        g = g.reshape(-1, 1)
        # an image whose upstream gradient is zero gets exact zeros (its dL/dSDF may hold a NaN: call_batch masks images that way)
        gs = torch.where(g != 0, ctx.obj.grad_sdf * g, torch.zeros((), device=g.device))
        return gs.reshape(ctx.shape), None, None, None


_STREAM_SETS = {}


def concurrent_streams(n, device="cuda", candidates=16):
    """`n` torch streams that really run side by side.  HIP multiplexes its streams onto a few hardware queues (4 by
    default) and two streams that share a queue serialise; which streams share one is not a simple function of their
    creation order (measured on ROCm 7.2: the pool's streams map to queues a b c d d c b a d c b a ..., so four
    CONSECUTIVE streams starting at the wrong offset sit on only two queues -- the same 8-image job then takes 13.5 ms per
    image instead of 8.6, scripts/dev_streams_queues.py).  Streams are therefore picked by measurement: a candidate is kept
    when a short spin kernel on it overlaps with one on every stream kept so far.  Asking for more streams than there are
    hardware queues returns the concurrent ones first and fills up with the rest.  Cached per (device, n)."""
    import time
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, int(n))
    if key in _STREAM_SETS:
        return _STREAM_SETS[key]
    cand = [torch.cuda.Stream(dev) for _ in range(max(int(candidates), int(n)))]
    if n <= 1 or not hasattr(torch.cuda, "_sleep"):
        _STREAM_SETS[key] = cand[:n]
        return _STREAM_SETS[key]

    def spin(ss, cycles=300_000):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for st in ss:
            with torch.cuda.stream(st):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    with torch.cuda.device(dev):
        spin(cand[:1])
        one = min(spin(cand[:1]) for _ in range(3))
        kept, rest = [cand[0]], []
        for st in cand[1:]:
            if len(kept) < n and all(min(spin([k, st]) for _ in range(2)) < 1.5 * one for k in kept):
                kept.append(st)
            else:
                rest.append(st)
    _STREAM_SETS[key] = (kept + rest)[:n]
    return _STREAM_SETS[key]


class GuidanceGroup:
    """Several independent GuidanceBatch loops, each on its own HIP stream with its own hipGraph.

    One image keeps only a fraction of the 256 CUs busy (the step is a chain of short, latency-bound launches), and
    one batched launch sequence still serialises on every kernel boundary.  Independent images have no such
    dependency: splitting them over a few streams lets the hardware queues overlap one group's kernel tails with
    another group's kernels.  Images are dealt out in contiguous chunks; results are per batch (`batches[i]`)."""

    def __init__(self, scenes, n_streams=1, device="cuda", **kw):
        n_streams = max(1, min(int(n_streams), len(scenes)))
        per = (len(scenes) + n_streams - 1) // n_streams
        chunks = [scenes[i:i + per] for i in range(0, len(scenes), per)]
        self.device = torch.device(device)
        self.batches = [GuidanceBatch(c, device=device, **kw) for c in chunks]
        self.streams = list(concurrent_streams(len(chunks), self.device))     # streams on DIFFERENT hardware queues
        self.graphs, self.joint, self.main, self.multi, self.steps_per_graph = [], None, None, [], 1

    @property
    def B(self):
        return sum(gb.B for gb in self.batches)

    def capture(self, cfg, joint=False, steps_per_graph=1):
        """joint=False (default): one hipGraph per stream, replayed one after the other from the host.  joint=True: ONE
        hipGraph whose branches are the batches (forked from / joined to the capture stream), a single replay per
        step -- measured slower on ROCm 7.2 (8 images: 33.7k vs 37.6k steps/s), the branches do not overlap as well."""
        self.graphs, self.joint, self.multi = [], None, []
        if joint and len(self.batches) > 1:
            main = torch.cuda.Stream(self.device)
            saved = [[t.clone() for t in (gb.params, gb.adam_m, gb.adam_v, gb.adam_t, gb.flags)] for gb in self.batches]
            for gb, st in zip(self.batches, self.streams):      # warm-up launches outside the capture (module load)
                with torch.cuda.stream(st):
                    gb.step(cfg)
            torch.cuda.synchronize(self.device)
            for gb, sv in zip(self.batches, saved):             # ... whose optimiser update is undone
                for t, s0 in zip((gb.params, gb.adam_m, gb.adam_v, gb.adam_t, gb.flags), sv):
                    t.copy_(s0)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main):
                for gb, st in zip(self.batches, self.streams):
                    st.wait_stream(main)                        # fork
                    with torch.cuda.stream(st):
                        gb.step(cfg)
                for st in self.streams:
                    main.wait_stream(st)                        # join
            self.joint, self.main = g, main
        else:
            for gb, st in zip(self.batches, self.streams):
                with torch.cuda.stream(st):
                    self.graphs.append(gb.capture(cfg))
                    if steps_per_graph > 1:   # a whole inner loop (or a slice of it) as one replay: no host work in between
                        self.multi.append(gb.capture(cfg, steps_per_graph=steps_per_graph))
            self.steps_per_graph = steps_per_graph if steps_per_graph > 1 else 1
        torch.cuda.synchronize(self.device)

    def run(self, cfg, n):
        """n iterations of every batch: multi-iteration graphs while they fit, single-iteration replays for the rest."""
        k = getattr(self, "steps_per_graph", 1)
        done = 0
        while k > 1 and self.multi and n - done >= k:
            for g, st in zip(self.multi, self.streams):
                with torch.cuda.stream(st):
                    g.replay()
            done += k
        for _ in range(n - done):
            self.step(cfg)

    def step(self, cfg):
        """One iteration of every batch: graph replay when captured, eager launches otherwise."""
        if getattr(self, "joint", None) is not None:
            with torch.cuda.stream(self.main):
                self.joint.replay()
            return
        for i, (gb, st) in enumerate(zip(self.batches, self.streams)):
            with torch.cuda.stream(st):
                if self.graphs:
                    self.graphs[i].replay()
                else:
                    gb.step(cfg)

    def restart(self, params):
        """New denoising step: the given (16,) parameter vector for every image and a fresh optimiser (PL:1478)."""
        joint = getattr(self, "joint", None) is not None
        for gb, st in zip(self.batches, self.streams):
            with torch.cuda.stream(self.main if joint else st):   # the joint graph replays on the capture stream
                gb.params.copy_(params.to(gb.params.device).expand_as(gb.params))
                gb.reset_optimizer()

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def _chunks(self, scenes):
        out, i = [], 0
        for gb in self.batches:
            out.append(scenes[i:i + gb.B])
            i += gb.B
        return out if i == len(scenes) else None

    def fits(self, scenes):
        ch = self._chunks(scenes)
        return ch is not None and all(gb.fits(c) for gb, c in zip(self.batches, ch))

    def load_scenes(self, scenes):
        """Capacity-mode groups: new images into the existing batches, in place (GuidanceBatch.load_scenes), each on its own
        stream; captured graphs stay valid."""
        if not self.fits(scenes):
            raise L.FohoError("GuidanceGroup.load_scenes: the scenes do not fit this group (count, image size, hand topology, capacity)")
        for gb, st, c in zip(self.batches, self.streams, self._chunks(scenes)):
            with torch.cuda.stream(st):
                gb.load_scenes(c)


def normal_map(gb, face_set, r=0, b=0, sigma=1e-8):
    """(H, W, 3) normal map of render r of image b as render_normal_and_disparity returns it (PL:272-289), read back from
    the G-buffer the last step left (face ids, edge distances, vertex normals) -- debug dumps only."""
    P = gb.H * gb.W
    R = gb.dims.n_renders
    m = gb.meta[b]
    p2f = gb.region("p2f", torch.int32, (R, gb.B, P))[r, b].long()
    sd = gb.region("sdist", torch.float32, (R, gb.B, P))[r, b]
    vn = gb.region("vn", torch.float32, (-1, 3))
    hit = p2f >= 0
    f0 = m["f_off"] + (m["Fh"] if face_set == L.FACES_OBJ else 0)
    f = gb.faces.long()[(p2f.clamp(min=0) + f0).clamp(max=gb.faces.shape[0] - 1)]
    col = vn[f].sum(1)
    p = torch.sigmoid(-torch.where(hit, sd, torch.zeros_like(sd)) / sigma)
    rgb = torch.where(hit[:, None], (p[:, None] * col + 1e-10) / (p[:, None] + 1e-10), torch.ones_like(col))
    nn = (rgb - rgb.min()) / (rgb.max() - rgb.min() + 1e-6)
    return torch.where(hit[:, None], nn, torch.zeros_like(nn)).reshape(gb.H, gb.W, 3).cpu().numpy()


def save_grid(img1, img2, path):
    """plot_in_grid of the reference (PL:189-201) without matplotlib: the two (H, W, 3) maps side by side, values clipped to
    [0, 1] like imshow does, at native resolution with an 8-pixel white gutter (no axes, no resampling)."""
    from PIL import Image
    a, b = (np.clip(np.asarray(x, np.float32), 0.0, 1.0) for x in (img1, img2))
    H = max(a.shape[0], b.shape[0])
    out = np.ones((H, a.shape[1] + 8 + b.shape[1], 3), np.float32)
    out[:a.shape[0], :a.shape[1]] = a
    out[:b.shape[0], a.shape[1] + 8:] = b
    Image.fromarray((out * 255.0 + 0.5).astype(np.uint8)).save(path)


class TargetRenderer:
    """Target maps of an image on the DEVICE: the reference renders the MoGe image mesh once per image with
    render_normal_and_disparity (PL:272-289, called at PL:1247-1256 on the mesh as it is) and masks the two maps with the
    hand-object mask (PL:1252-1253).  `render_into` does that straight into a GuidanceBatch's tgt_normal[b] / tgt_disp[b],
    asynchronously on the current stream: page-locked upload of the mesh (any vertex / face count within the capacity; the
    sizes of a render are its own, nothing is padded), incidence tables on the device, vertex + raster stages, the map
    arithmetic of `hip_render_fn` in torch.  No host synchronisation; `flags` (device, int32[2]) collects what the render
    reported: [0] the step's flag word, [1] foho_topology_tables' (non-zero: a valence above 48 -- not an image mesh)."""

    def __init__(self, H, W, vcap, fcap, device="cuda"):
        self.H, self.W, self.vcap, self.fcap = int(H), int(W), int(vcap), int(fcap)
        dummy = dict(obj_verts=np.zeros((self.vcap, 3), np.float32), obj_faces=np.zeros((self.fcap, 3), np.int64),     # sizes only
                     hand_verts=np.zeros((0, 3), np.float32),
                     hand_faces=np.zeros((0, 3), np.int64), T_h2m=np.eye(4, dtype=np.float32),
                     J_regressor=np.zeros((16, 1), np.float32), kps_2d=np.zeros((21, 2), np.float32),
                     moge_normal=np.zeros((H, W, 3), np.float32), moge_disp=np.zeros((H, W), np.float32),
                     hand_mask=np.zeros((H, W), bool), obj_mask=np.zeros((H, W), bool), fov=60.0, H=H, W=W)
        gb = self.gb = GuidanceBatch([dummy], device=device, n_renders=1, grid_res=2, topology="deferred")
        dev = gb.device
        gb.lib.foho_topology_workspace_bytes.restype = ctypes.c_size_t
        self.topo_ws = torch.empty(gb.lib.foho_topology_workspace_bytes(self.vcap), dtype=torch.uint8, device=dev)
        self.flags = torch.zeros(2, dtype=torch.int32, device=dev)
        self.cfg, _ = phase_cfg("B", do_update=False)
        self.cfg.world_space_input = 1
        self._mirrors = {}

    def fits(self, verts, faces, H, W):
        return len(verts) <= self.vcap and len(faces) <= self.fcap and (int(H), int(W)) == (self.H, self.W) and len(faces) > 0

    def render_into(self, key, verts, faces, fov, mask_u8, out_normal, out_disp):
        """`key` names the page-locked mirrors this call stages the mesh in (they must not be rewritten before the copies
        have run: one key per image the caller has in flight).  mask_u8: (H, W) device tensor, non-zero = inside the
        hand-object mask."""
        gb, H, W = self.gb, self.H, self.W
        v, f = len(verts), len(faces)
        if not self.fits(verts, faces, H, W):
            raise L.FohoError(f"TargetRenderer: image mesh of {v} vertices / {f} faces does not fit ({self.vcap} / {self.fcap})")
        mir = self._mirrors.get(key)
        if mir is None:
            mir = self._mirrors[key] = (torch.empty(self.vcap, 3, dtype=torch.float32, pin_memory=True),
                                        torch.empty(self.fcap, 3, dtype=torch.int32, pin_memory=True),
                                        torch.empty(gb.images.shape, dtype=torch.uint8, pin_memory=True))
        pv, pf, pim = mir
        pv.numpy()[:v] = verts
        pf.numpy()[:f] = faces
        # this render's own sizes: the step takes every bound from dims / the image record, the buffers keep the capacity
        m, im, d = gb.meta[0], gb._images_host[0], gb.dims
        m.update(Vo=v, Fo=f)
        im.Vo, im.Fo = v, f
        im.k00, im.k11 = fov_focal(float(fov))
        d.Vtot, d.Ftot, d.Vmax, d.Fmax, d.Vo_max, d.Fo_max = v, f, v, f, v, f
        gb.Vtot, gb.Ftot = v, f
        gb._desc = None
        pim.numpy()[...] = np.frombuffer(bytes(im), np.uint8)
        gb.verts_in[:v].copy_(pv[:v], non_blocking=True)
        gb.faces[:f].copy_(pf[:f], non_blocking=True)
        gb.images.copy_(pim, non_blocking=True)
        P = ctypes.c_void_p
        stream = torch.cuda.current_stream(gb.device).cuda_stream
        L.check(gb.lib.foho_topology_tables(P(gb.faces.data_ptr()), v, f, None, P(gb.inc_off.data_ptr()), P(gb.inc_fc.data_ptr()),
                                            P(gb.nbr_idx.data_ptr()), P(self.flags[1:].data_ptr()), P(self.topo_ws.data_ptr()),
                                            ctypes.c_size_t(self.topo_ws.numel()), P(stream)), "foho_topology_tables")
        gb.flags.zero_()
        gb._bbox_dirty = True        # new vertices, and the scatter planes of THIS layout start clean
        gb.step(self.cfg, stages=L.STAGE_VERTEX | L.STAGE_RASTER)
        self.flags[0:1].copy_(gb.flags)
        nn, disp, _ = _maps_from_gbuffer(gb, H * W)
        inside = mask_u8.reshape(-1) != 0
        out_normal.copy_(torch.where(inside[:, None], nn, torch.zeros_like(nn)).reshape(H, W, 3))
        out_disp.copy_(torch.where(inside, disp, torch.zeros_like(disp)).reshape(H, W))


def _maps_from_gbuffer(gb, P):
    """Normal map and disparity of render 0 of image 0 the way render_normal_and_disparity builds them (PL:272-289), from
    the G-buffer the raster stage left: (P, 3), (P,), pix_to_face (P,) -- device tensors."""
    p2f = gb.region("p2f", torch.int32, (P,)).long()
    z = gb.region("zbuf", torch.float32, (P,))
    sd = gb.region("sdist", torch.float32, (P,))
    vn = gb.region("vn", torch.float32, (-1, 3))
    hit = p2f >= 0
    f = gb.faces.long()[p2f.clamp(min=0)]
    col = vn[f].sum(1)
    p = torch.sigmoid(-sd / 1e-8)
    rgb = torch.where(hit[:, None], (p[:, None] * col + 1e-10) / (p[:, None] + 1e-10), torch.ones_like(col))
    nn = (rgb - rgb.min()) / (rgb.max() - rgb.min() + 1e-6)
    nn = torch.where(hit[:, None], nn, torch.zeros_like(nn))
    depth = torch.where(hit, z, torch.full_like(z, 10.0))
    disp = 1 / (depth + 1e-6)
    disp = (disp - disp.min()) / (disp.max() - disp.min() + 1e-6)
    return nn, disp, p2f


def hip_render_fn(device="cuda"):
    """Target-map renderer for synthetic scenes backed by the HIP rasteriser (data generation only):
    returns render_fn(verts, faces, H, W, fov) -> (normal (H,W,3), disp (H,W), pix_to_face (H,W)) following
    render_normal_and_disparity (PL:272-289)."""

    def render(verts, faces, H, W, fov):
        verts = np.asarray(verts, np.float32)
        faces = np.asarray(faces, np.int64)
        dummy = dict(obj_verts=verts, obj_faces=faces, hand_verts=np.zeros((0, 3), np.float32),
                     hand_faces=np.zeros((0, 3), np.int64), T_h2m=np.eye(4, dtype=np.float32),
                     J_regressor=np.zeros((16, 1), np.float32), kps_2d=np.zeros((21, 2), np.float32),
                     moge_normal=np.zeros((H, W, 3), np.float32), moge_disp=np.zeros((H, W), np.float32),
                     hand_mask=np.zeros((H, W), bool), obj_mask=np.zeros((H, W), bool), fov=fov, H=H, W=W)
        gb = GuidanceBatch([dummy], device=device, n_renders=1, grid_res=2, topology="render")
        cfg, _ = phase_cfg("B", do_update=False)
        cfg.world_space_input = 1     # the reference renders the target mesh as it is (PL:1247-1256)
        gb.step(cfg, stages=L.STAGE_VERTEX | L.STAGE_RASTER)
        torch.cuda.current_stream(gb.device).synchronize()     # this stream only: loader threads render next to a running job
        gb.raise_on_flags()
        nn, disp, p2f = _maps_from_gbuffer(gb, H * W)
        return (nn.reshape(H, W, 3).cpu().numpy(), disp.reshape(H, W).cpu().numpy(), p2f.reshape(H, W).cpu().numpy())

    return render
