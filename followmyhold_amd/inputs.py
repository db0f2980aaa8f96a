"""On-disk formats at the edges of the guidance path and the mesh-level guidance driver.

What the reference reads per image (third_party_patches/hy3dgen/shapegen/pipelines.py:1217-1256, file names from
src/foho/guidance/run.py:210-222) and how it is read here:

  {idx}_kps_for_guidance.npy      pickled dict, key 'mano_2d_kps' (21,2) in image space (hamer.py:275-279)
  {idx}_hamer_aligned_mano.ply    MANO mesh aligned to the Hunyuan space (778 V + wrist-closing faces)
  {idx}_cropped_{hand,obj}_mask.png   8-bit masks, > 0 means inside
  {idx}_hoi_mesh.npy              4x4 float64 Hunyuan -> MoGe similarity (h2m.py:38; np.save appends .npy)
  {idx}_hoi_mesh.ply              Hunyuan hand-object mesh
  {idx}_cropped_hoi/mesh.glb      MoGe image mesh (binary glTF 2.0), rendered ONCE per image into the target
                                  normal / disparity maps (render_normal_and_disparity, PL:272-289) and masked
  {idx}_cropped_hoi/fov.json      {"fov_x": degrees}

`run_mesh_guidance` drives phases A, B, C (PL:1320-1601) on a FIXED object mesh: the diffusion latent -> FlexiCubes
route that moves the object's vertices in the reference lives outside this repository's scope,
everything else -- inputs, targets, schedule of iterations, optimiser restarts, outputs -- follows the reference.
"""
import json
import os
import struct

import numpy as np

from . import meshio

# ------------------------------------------------------------------------------------------------ binary glTF 2.0
_GLB_MAGIC, _CHUNK_JSON, _CHUNK_BIN = 0x46546C67, 0x4E4F534A, 0x004E4942
_COMPONENT = {5120: ("b", 1), 5121: ("B", 1), 5122: ("h", 2), 5123: ("H", 2), 5125: ("I", 4), 5126: ("f", 4)}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _accessor(gltf, binary, idx):
    acc = gltf["accessors"][idx]
    fmt, size = _COMPONENT[acc["componentType"]]
    ncomp = _NCOMP[acc["type"]]
    count = acc["count"]
    if "bufferView" not in acc:
        return np.zeros((count, ncomp), dtype=np.dtype(fmt))
    bv = gltf["bufferViews"][acc["bufferView"]]
    if bv.get("buffer", 0) != 0:
        raise ValueError("only the embedded BIN buffer of a .glb is supported")
    start = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or size * ncomp
    dt = np.dtype(fmt).newbyteorder("<")
    if stride == size * ncomp:
        a = np.frombuffer(binary, dtype=dt, count=count * ncomp, offset=start).reshape(count, ncomp)
    else:
        a = np.stack([np.frombuffer(binary, dtype=dt, count=ncomp, offset=start + i * stride) for i in range(count)])
    return a


def _node_matrix(node):
    if "matrix" in node:
        return np.array(node["matrix"], dtype=np.float64).reshape(4, 4).T  # glTF stores column-major
    m = np.eye(4)
    if "scale" in node:
        m = np.diag(list(node["scale"]) + [1.0]) @ m
    if "rotation" in node:
        x, y, z, w = node["rotation"]
        r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        rm = np.eye(4)
        rm[:3, :3] = r
        m = rm @ m
    if "translation" in node:
        t = np.eye(4)
        t[:3, 3] = node["translation"]
        m = t @ m
    return m


def load_glb(path, index_dtype=np.int64):
    """Triangle geometry of a binary glTF 2.0 file: (verts (V,3) float32, faces (F,3) `index_dtype`), all primitives of all
    scene nodes concatenated with their node transforms applied (the MoGe mesh is a single primitive).  index_dtype=np.int32
    with 32-bit indices in the file (what MoGe writes) returns a VIEW of the file's bytes: no conversion pass over the
    half million faces."""
    with open(path, "rb") as f:
        data = f.read()
    magic, version, length = struct.unpack_from("<III", data, 0)
    if magic != _GLB_MAGIC or version != 2:
        raise ValueError(f"{path}: not a binary glTF 2.0 file")
    off, gltf, binary = 12, None, b""
    view = memoryview(data)          # chunk slices without copying the (megabytes of) geometry
    while off + 8 <= min(length, len(data)):
        clen, ctype = struct.unpack_from("<II", data, off)
        chunk = view[off + 8:off + 8 + clen]
        if ctype == _CHUNK_JSON:
            gltf = json.loads(bytes(chunk).decode("utf-8"))
        elif ctype == _CHUNK_BIN:
            binary = chunk
        off += 8 + clen
    if gltf is None:
        raise ValueError(f"{path}: no JSON chunk")
    verts, faces, voff = [], [], 0

    def visit(ni, parent):
        nonlocal voff
        node = gltf["nodes"][ni]
        m = parent @ _node_matrix(node)
        if "mesh" in node:
            for prim in gltf["meshes"][node["mesh"]]["primitives"]:
                if prim.get("mode", 4) != 4:
                    continue
                p = _accessor(gltf, binary, prim["attributes"]["POSITION"])
                if np.array_equal(m, np.eye(4)):     # the usual case (MoGe writes one untransformed primitive): no float64 round trip
                    p = p.astype(np.float32, copy=False)
                else:
                    p = (p.astype(np.float64) @ m[:3, :3].T + m[:3, 3]).astype(np.float32)
                if "indices" in prim:
                    idx = _accessor(gltf, binary, prim["indices"]).reshape(-1)
                    if idx.dtype == np.dtype("<u4") and index_dtype == np.int32 and len(p) < 2 ** 31:
                        idx = idx.view(np.int32)
                    else:
                        idx = idx.astype(index_dtype)
                else:
                    idx = np.arange(len(p), dtype=index_dtype)
                if voff:
                    idx = idx + voff
                verts.append(p)
                faces.append(idx.reshape(-1, 3))
                voff += len(p)
        for c in node.get("children", []):
            visit(c, m)

    scenes = gltf.get("scenes")
    roots = scenes[gltf.get("scene", 0)]["nodes"] if scenes else range(len(gltf.get("nodes", [])))
    for r in roots:
        visit(r, np.eye(4))
    if not verts:
        raise ValueError(f"{path}: no triangle primitive")
    if len(verts) == 1:
        return np.ascontiguousarray(verts[0]), faces[0]
    return np.concatenate(verts, 0), np.concatenate(faces, 0)


def save_glb(path, verts, faces):
    """Minimal single-primitive .glb writer (fixtures for the tests, same layout trimesh / utils3d emit)."""
    v = np.ascontiguousarray(verts, dtype="<f4")
    f = np.ascontiguousarray(faces, dtype="<u4").reshape(-1)
    vb, fb = v.tobytes(), f.tobytes()
    pad = (-len(vb)) % 4
    binary = vb + b"\0" * pad + fb
    gltf = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "mode": 4}]}],
        "buffers": [{"byteLength": len(binary)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": len(vb), "target": 34962},
                        {"buffer": 0, "byteOffset": len(vb) + pad, "byteLength": len(fb), "target": 34963}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": int(len(v)), "type": "VEC3",
                       "min": v.min(0).tolist(), "max": v.max(0).tolist()},
                      {"bufferView": 1, "componentType": 5125, "count": int(len(f)), "type": "SCALAR"}],
    }
    js = json.dumps(gltf, separators=(",", ":")).encode("utf-8")
    js += b" " * ((-len(js)) % 4)
    binary += b"\0" * ((-len(binary)) % 4)
    total = 12 + 8 + len(js) + 8 + len(binary)
    with open(path, "wb") as fh:
        fh.write(struct.pack("<III", _GLB_MAGIC, 2, total))
        fh.write(struct.pack("<II", len(js), _CHUNK_JSON) + js)
        fh.write(struct.pack("<II", len(binary), _CHUNK_BIN) + binary)


# ------------------------------------------------------------------------------------------------ small files
def load_mask(path):
    """8-bit PNG mask -> bool (H,W); the reference thresholds cv2.imread(..., GRAYSCALE) > 0 (PL:1230-1237)."""
    from PIL import Image
    return np.array(Image.open(path).convert("L")) > 0


def save_mask(path, mask):
    from PIL import Image
    Image.fromarray(np.asarray(mask).astype(np.uint8) * 255).save(path)


def load_kps_for_guidance(path):
    """{idx}_kps_for_guidance.npy: pickled dict written by the modified HaMeR demo (hamer.py:275-279)."""
    d = np.load(path, allow_pickle=True).item()
    return np.asarray(d["mano_2d_kps"], dtype=np.float32).reshape(21, 2)


def load_j_regressor(path=None):
    """(16, 778) MANO joint regressor.  The reference torch.load()s ./third_party/estimator/hamer/J_regressor_hamer.pt
    (PL:1218); $FOHO_J_REGRESSOR may point to a .pt or .npy copy."""
    import torch
    from foho.configs import third_party_root
    path = path or os.environ.get("FOHO_J_REGRESSOR") or os.path.join(third_party_root(), "estimator", "hamer",
                                                                       "J_regressor_hamer.pt")
    if path.endswith(".npy"):
        return np.load(path).astype(np.float32)
    jr = torch.load(path, map_location="cpu")
    return np.asarray(jr.to_dense() if hasattr(jr, "to_dense") and jr.is_sparse else jr, dtype=np.float32)


def load_fov(path):
    with open(path, "r", encoding="utf-8") as f:
        return float(json.load(f)["fov_x"])


# ------------------------------------------------------------------------------------------------ one image
def load_scene_from_files(p, J_regressor, render_fn, n_hand_verts=778, fov=None, with_object=True):
    """Scene dict for `engine.GuidanceBatch` from the reference's per-image files.  `p` = `foho.guidance.run.derive_paths`
    output; `render_fn(verts, faces, H, W, fov) -> (normal, disp, pix_to_face)` renders the MoGe mesh into the target
    maps (engine.hip_render_fn on the GPU); render_fn=None returns the mesh itself as scene["moge_mesh"] instead of the maps.  `fov` overrides fov.json (the pipeline gets it from the renderer's camera,
    RUN:90); with_object=False leaves the object empty (the pipeline decodes it from the latent every iteration)."""
    hand_mask, obj_mask = load_mask(p["cropped_hand_mask_path"]), load_mask(p["cropped_obj_mask_path"])
    H, W = hand_mask.shape
    fov = load_fov(p["moge_fov_path"]) if fov is None else float(fov)
    T = np.load(p["T_h2m_path"]).astype(np.float64).reshape(4, 4)
    mano_v, mano_f = meshio.load_ply(p["aligned_mano_mesh_path"])
    hand_moge = mano_v.astype(np.float64) @ T[:3, :3].T + T[:3, 3]          # transform_hunyuan2moge (PL:242-250, 1241)
    if with_object:
        obj_v, obj_f = meshio.load_ply(p["hunyuan_hoi_mesh_path"])
    else:
        obj_v, obj_f = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64)
    mv, mf = load_glb(p["moge_mesh_path"], np.int64 if render_fn is not None else np.int32)
    jr = np.asarray(J_regressor, dtype=np.float32)
    if jr.shape[1] != n_hand_verts:
        raise ValueError("J_regressor must be (16, number of MANO vertices)")
    scene = dict(
        hand_verts=hand_moge.astype(np.float32), hand_faces=mano_f.astype(np.int64),
        obj_verts=obj_v.astype(np.float32), obj_faces=obj_f.astype(np.int64), T_h2m=T.astype(np.float32),
        J_regressor=jr, kps_2d=load_kps_for_guidance(p["hamer_for_guid_path"]),
        hand_mask=hand_mask, obj_mask=obj_mask, fov=fov, H=int(H), W=int(W))
    if render_fn is None:      # host work only: the image mesh travels with the scene, MeshGuidanceRunner renders it on the device
        scene["moge_mesh"] = (np.ascontiguousarray(mv, np.float32), np.ascontiguousarray(mf, np.int32))
        return scene
    normal, disp, _ = render_fn(mv, mf, H, W, fov)
    hoi = (hand_mask | obj_mask).astype(np.float32)                         # PL:1243, 1252-1253
    scene.update(moge_normal=(normal * hoi[..., None]).astype(np.float32), moge_disp=(disp * hoi).astype(np.float32))
    return scene


def save_scene_files(scene, moge_verts, moge_faces, dirs, index, hand_verts_hunyuan=None):
    """Write one image's inputs in the reference's formats and names (fixture generator for the tests and a
    specification of the formats).  `dirs` holds mask_dir, moge_out_dir, hunyuan_hoi_mesh_dir, hamer_out_dir,
    h2m_rt_dir, aligned_mano_dir, cropped_obj_img_dir."""
    from PIL import Image
    j = os.path.join
    for d in dirs.values():
        os.makedirs(d, exist_ok=True)
    os.makedirs(j(dirs["moge_out_dir"], f"{index}_cropped_hoi"), exist_ok=True)
    H, W = scene["H"], scene["W"]
    Image.fromarray(np.zeros((H, W, 3), np.uint8)).save(j(dirs["cropped_obj_img_dir"], f"{index}_cropped_hoi_1.png"))
    save_mask(j(dirs["mask_dir"], f"{index}_cropped_hand_mask.png"), scene["hand_mask"])
    save_mask(j(dirs["mask_dir"], f"{index}_cropped_obj_mask.png"), scene["obj_mask"])
    save_glb(j(dirs["moge_out_dir"], f"{index}_cropped_hoi", "mesh.glb"), moge_verts, moge_faces)
    with open(j(dirs["moge_out_dir"], f"{index}_cropped_hoi", "fov.json"), "w", encoding="utf-8") as f:
        json.dump({"fov_x": scene["fov"]}, f)
    T = np.asarray(scene["T_h2m"], np.float64)
    np.save(j(dirs["h2m_rt_dir"], f"{index}_hoi_mesh"), T)                 # np.save appends .npy (h2m.py:38)
    if hand_verts_hunyuan is None:                                         # MoGe -> Hunyuan space
        Ti = np.linalg.inv(T)
        hand_verts_hunyuan = np.asarray(scene["hand_verts"], np.float64) @ Ti[:3, :3].T + Ti[:3, 3]
    meshio.save_ply(j(dirs["aligned_mano_dir"], f"{index}_hamer_aligned_mano.ply"), hand_verts_hunyuan.astype(np.float32),
                    scene["hand_faces"])
    meshio.save_ply(j(dirs["hunyuan_hoi_mesh_dir"], f"{index}_hoi_mesh.ply"), scene["obj_verts"], scene["obj_faces"])
    np.save(j(dirs["hamer_out_dir"], f"{index}_kps_for_guidance.npy"),
            {"mano_2d_kps": np.asarray(scene["kps_2d"], np.float32), "mano_3d_kps": np.zeros((21, 3), np.float32),
             "cam_t": np.zeros(3, np.float32)}, allow_pickle=True)


# ------------------------------------------------------------------------------------------------ driver
def job_schedule(config):
    """The reference's iteration schedule of one image (PL:1293-1610) as [(phase, iterations, denoising step)]: phase A at
    `handopt_start_step`, phase B at `guidance_start_step`, phase C at every later denoising step."""
    a_step, b_step, n_steps = int(config.handopt_start_step), int(config.guidance_start_step), int(config.num_inference_steps)
    sched = [("A", int(config.optimization_steps_hand), a_step), ("B", int(config.optimization_steps_scale), b_step)]
    sched += [("C", int(config.optimization_steps_joint), i) for i in range(b_step + 1, n_steps)]
    return sched


def _steps_per_graph(iters):
    # iterations per hipGraph replay.  Recording costs the host ~20 us per iteration (1.1 ms for 50; measured with
    # scripts/dev_replay_cost.py -- the 10 ms of earlier rounds were the device synchronisation of `torch.cuda.graph`, which
    # GuidanceBatch.capture no longer goes through), a replay ~1.5 us per iteration plus a release / re-acquisition of the
    # interpreter lock per call, a replay boundary ~10 us of device time: 50 per graph
    return max([d for d in range(1, 51) if iters % d == 0]) if iters > 0 else 1


def run_mesh_guidance(scenes, config=None, device="cuda", capture=True, log=None):
    """Phases A (hand), B (object), C (joint) of PL:1293-1610 for ONE exact-size batch of scenes on fixed object meshes
    (any mesh: open, non-manifold -- the topology tables come from the host builders when the device ones refuse).  The
    product driver goes through `MeshGuidanceRunner` (images in flight on several streams, graphs captured once per
    process) and falls back to this function for images the runner cannot take.

    Iteration counts and the guidance window come from the reference's OptimizationConfig, each loop with a fresh
    optimiser (PL:1318, 1384, 1478).  Returns the GuidanceBatch (parameters, losses, world-space vertices)."""
    import torch
    from . import engine as E
    cfg0 = config if config is not None else E.OptimizationConfig()
    gb = E.GuidanceBatch(scenes, device=device, n_renders=2)
    gb.prepare()
    graphs = {}     # the nine joint loops only differ in the intersection gate (denoising steps >= 17): two captures, not nine
    for phase, iters, denoise_i in job_schedule(cfg0):
        cfg, _ = E.phase_cfg(phase, cfg0, denoise_i=denoise_i, do_update=True)   # one workspace for all phases (n_active_renders)
        spg = _steps_per_graph(iters)
        graph = None
        if capture:
            key = (bytes(cfg), spg)
            if key not in graphs:
                graphs[key] = gb.capture(cfg, steps_per_graph=spg)
            graph = graphs[key]
        gb.reset_optimizer()
        for _ in range(iters // spg if graph is not None else iters):
            graph.replay() if graph is not None else gb.step(cfg)
        torch.cuda.synchronize(gb.device)
        gb.raise_on_flags(strict_k=False)
        if log is not None:
            log(phase, denoise_i, [gb.loss_dict(b)["total"] for b in range(gb.B)])
    gb.refresh_world()      # the output meshes follow from the FINAL parameters (PL:1614-1618, 1653-1657)
    torch.cuda.synchronize(gb.device)
    return gb


class GpuGate:
    """Many `shared()` holders OR one `exclusive()` holder.  hipGraph capture does not tolerate GPU work from other threads of
    the process (torch's capture_error_mode "global"): the runner captures under exclusive(), whoever prepares inputs on the
    GPU next to it (the driver's loader threads) works under shared() -- loaders do not exclude each other."""

    def __init__(self):
        import threading
        self._cv = threading.Condition()
        self._shared, self._excl = 0, False

    class _Ctx:
        def __init__(self, enter, leave):
            self._enter, self._leave = enter, leave

        def __enter__(self):
            self._enter()

        def __exit__(self, *a):
            self._leave()

    def shared(self):
        def enter():
            with self._cv:
                while self._excl:
                    self._cv.wait()
                self._shared += 1

        def leave():
            with self._cv:
                self._shared -= 1
                self._cv.notify_all()
        return GpuGate._Ctx(enter, leave)

    def exclusive(self):
        def enter():
            with self._cv:
                while self._excl:
                    self._cv.wait()
                self._excl = True            # new shared holders wait from here on
                while self._shared:
                    self._cv.wait()

        def leave():
            with self._cv:
                self._excl = False
                self._cv.notify_all()
        return GpuGate._Ctx(enter, leave)


class _Slot:
    """One capacity-mode GuidanceBatch of a MeshGuidanceRunner: its stream, its hipGraphs, the job queued on it and the
    page-locked buffers that job's results land in."""

    def __init__(self, gb, stream):
        import torch
        self.gb, self.stream = gb, stream
        self.graphs = {}
        self.job = None                 # [(tag, scene)] of the job queued on the stream, None when idle
        self.done = torch.cuda.Event()
        with torch.cuda.stream(stream):
            self.seen = torch.zeros_like(gb.flags)       # flags of every loop of the job, OR-ed
            self.nan_b = torch.zeros_like(gb.flags)      # the NaN bit as phase B left it
            self.render_flags = torch.zeros(gb.B, 2, dtype=torch.int32, device=gb.device)    # engine.TargetRenderer.flags per image
        src = dict(seen=self.seen, nan_b=self.nan_b, losses=gb.losses, params=gb.params, faces=gb.faces,
                   world=gb.region("world", torch.float32, (-1, 3)), render_flags=self.render_flags)
        self.src = src
        self.out = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in src.items()}


class MeshGuidanceRunner:
    """The mesh-level guidance job for MANY images per GPU ("within a GPU, batch the rank's images through
    each kernel launch"; the reference walks its list one image at a time with batch size 1, RUN:208-259, CFG:9).

    `in_flight` images are optimised at once: `n_streams` HIP streams on different hardware queues, each running a SLOT --
    a capacity-mode GuidanceBatch of in_flight / n_streams images with its own hipGraphs.  A job is one slot's images through
    the whole schedule (phase A, phase B, nine phase-C loops), queued on the slot's stream in one go: uploads from page-locked
    mirrors (GuidanceBatch.load_scenes: same buffers, objects installed on the device), graph replays, the final transform,
    read-back into page-locked result buffers, an event.  Nothing in it waits for the host -- NaN break, optimiser state and
    flags live on the device -- and the host waits for nothing but that event.  Long lists (run_stream) use TWO slots
    per stream, alternately: while one slot's job runs, the other's finished job is read, exported and replaced by the next
    images, and its new job is queued behind the running one -- a stream never idles between jobs.  The graphs of the
    schedule (phase A, phase B, phase C with and without the intersection gate) are captured once per slot and process, not
    once per image, and no workspace is re-allocated between phases (foho_step_cfg.n_active_renders).

    run(scenes) -> one result per scene: dict(ok, flags, losses (dict of the last step), params (16,), hand (verts, faces),
    obj (verts, faces), nan_in_phase_b); run_stream(items) is the same as a generator over (tag, scene) pairs that yields
    (tag, result) in submission order while later images are still on the GPU.  Images whose object is not a closed
    manifold (device-side edge tables refuse: flag bit 5) are reported with ok=False, reason="fallback": the caller runs them
    through `run_mesh_guidance`."""

    def __init__(self, config=None, device="cuda", in_flight=16, n_streams=None, grid_res=64):
        from . import engine as E
        self.E = E
        self.config = config if config is not None else E.OptimizationConfig()
        self.device = device
        self.in_flight = max(1, int(in_flight))
        # at most four streams (HIP has four hardware queues, docs/NOTEBOOK_r1-3.md section 6), at least two images per stream; measured for
        # whole jobs: 116 images/s at 8 in flight (4 x 2, the setting bench.py's `batched` record uses), 160 at 16 (4 x 4)
        self.n_streams = int(n_streams) if n_streams else max(1, min(4, (self.in_flight + 1) // 2))
        self.per_slot = (self.in_flight + self.n_streams - 1) // self.n_streams
        self.n_streams = (self.in_flight + self.per_slot - 1) // self.per_slot
        self.grid_res = grid_res
        self.slots = {}                 # ring position -> _Slot
        self.renderers = {}             # stream index -> engine.TargetRenderer
        self.streams = None
        self._turn = {}                 # stream index -> jobs its feeder has taken so far (which of its slots is next)
        self._lock = __import__("threading").Lock()
        self.stats = dict(captures=0, slots_built=0, jobs=0)
        self.gpu_gate = GpuGate()      # captures under exclusive(), loader threads under shared()

    # -------------------------------------------------------------------------------------------- slots
    @staticmethod
    def _capacity(scenes):
        rnd = lambda x, q: ((int(x) + q - 1) // q) * q
        return (rnd(max(len(s["obj_verts"]) for s in scenes) * 1.125 + 1, 1024), rnd(max(len(s["obj_faces"]) for s in scenes) * 1.125 + 2, 2048))

    def _slot_for(self, pos, scenes):
        """The slot at ring position `pos`, able to take `scenes` (exactly per_slot of them); rebuilt -- graphs and all --
        when they do not fit (other image size / hand topology, larger object than any before)."""
        import torch
        slot = self.slots.get(pos)
        if slot is not None and slot.gb.fits(scenes):
            return slot
        cap = self._capacity(scenes)
        for other in list(self.slots.values()):      # sizes drift, they do not alternate: never below what any slot already has
            old = other.gb.obj_capacity
            cap = (max(cap[0], old[0]), max(cap[1], old[1]))
        if self.streams is None:
            self.streams = list(self.E.concurrent_streams(self.n_streams, torch.device(self.device)))
        st = self.streams[pos % self.n_streams]
        with self.gpu_gate.shared(), torch.cuda.stream(st):
            gb = self.E.GuidanceBatch(scenes, device=self.device, grid_res=self.grid_res, n_renders=2, obj_capacity=cap)
            slot = _Slot(gb, st)
            slot.stream_index = pos % self.n_streams
            gb.load_scenes(scenes)          # real objects in the slot for the launch that precedes the process's first capture
        # the graphs of the whole schedule, now: a capture excludes every other thread's GPU work (gpu_gate), so it must not
        # come up in the middle of a job's queueing
        for phase, iters, denoise_i in job_schedule(self.config):
            cfg, _ = self.E.phase_cfg(phase, self.config, denoise_i=denoise_i, do_update=True)
            with torch.cuda.stream(st):
                self._graph_for(slot, cfg, _steps_per_graph(iters))
        self.slots[pos] = slot
        with self._lock:
            self.stats["slots_built"] += 1
        return slot

    def _renderer_for(self, slot, verts, faces, H, W):
        """The target-map renderer of the slot's stream (the two slots of a stream share it: their jobs run one after the
        other), sized for an image mesh with one vertex per pixel; rebuilt when a mesh does not fit."""
        k = slot.stream_index
        rd = self.renderers.get(k)
        if rd is None or not rd.fits(verts, faces, H, W):
            vcap = max(H * W, len(verts), rd.vcap if rd is not None else 0)
            fcap = max(2 * (H - 1) * (W - 1), len(faces), rd.fcap if rd is not None else 0)
            rd = self.renderers[k] = self.E.TargetRenderer(H, W, vcap, fcap, device=self.device)
        return rd

    def _graph_for(self, slot, cfg, spg):
        key = (bytes(cfg), spg)
        g = slot.graphs.get(key)
        if g is None:
            with self.gpu_gate.exclusive():
                g = slot.graphs[key] = slot.gb.capture(cfg, steps_per_graph=spg)
            with self._lock:
                self.stats["captures"] += 1
        return g

    # -------------------------------------------------------------------------------------------- one job
    def _enqueue(self, slot, chunk, scenes):
        """Queue the whole job of `scenes` on the slot's stream; returns at once."""
        import torch
        E = self.E
        gb = slot.gb
        with torch.cuda.stream(slot.stream):
            gb.load_scenes(scenes)
            slot.render_flags.zero_()
            if any("moge_normal" not in s for s in scenes):       # target maps rendered here, on the device, inside the job
                for b, s in enumerate(scenes):
                    if "moge_normal" in s:                         # a mixed job: this image brought its maps along
                        gb.tgt_normal[b].copy_(torch.from_numpy(np.ascontiguousarray(s["moge_normal"], np.float32)))
                        gb.tgt_disp[b].copy_(torch.from_numpy(np.ascontiguousarray(s["moge_disp"], np.float32)))
                        continue
                    mv, mf = s["moge_mesh"]
                    rd = self._renderer_for(slot, mv, mf, gb.H, gb.W)
                    rd.render_into((id(slot), b), mv, mf, s["fov"], gb.mask[b], gb.tgt_normal[b], gb.tgt_disp[b])
                    slot.render_flags[b].copy_(rd.flags)
                gb.prepare()
            slot.seen.copy_(gb.load_flags)
            slot.nan_b.zero_()
            for phase, iters, denoise_i in job_schedule(self.config):
                cfg, _ = E.phase_cfg(phase, self.config, denoise_i=denoise_i, do_update=True)
                spg = _steps_per_graph(iters)
                g = slot.graphs[(bytes(cfg), spg)]      # captured when the slot was built (_slot_for); never captured here, under the shared gate
                gb.reset_optimizer()
                for _ in range(iters // spg):
                    g.replay()
                slot.seen |= gb.flags
                if phase == "B":
                    slot.nan_b.copy_(gb.flags & 1)
            gb.refresh_world()      # output meshes from the FINAL parameters (PL:1614-1618, 1653-1657)
            for k, v in slot.src.items():
                slot.out[k].copy_(v, non_blocking=True)
            slot.done.record()
        slot.job = chunk
        slot.meta = [dict(m) for m in gb.meta]
        with self._lock:
            self.stats["jobs"] += 1

    def _retire(self, slot):
        """Wait for the slot's job and turn its result buffers into one result per submitted image."""
        E = self.E
        slot.done.synchronize()
        o = {k: v.numpy() for k, v in slot.out.items()}
        faces = o["faces"].astype(np.int64)
        out = []
        for b, (tag, _) in enumerate(slot.job):
            m = slot.meta[b]
            hv = o["world"][m["v_off"]:m["v_off"] + m["Vh"]].copy()
            ov = o["world"][m["v_off"] + m["Vh"]:m["v_off"] + m["Vh"] + m["Vo"]].copy()
            hf = faces[m["f_off"]:m["f_off"] + m["Fh"]] - m["v_off"]
            of = faces[m["f_off"] + m["Fh"]:m["f_off"] + m["Fh"] + m["Fo"]] - m["v_off"] - m["Vh"]
            f = int(o["seen"][b])
            rf = o["render_flags"][b]
            res = dict(ok=True, flags=f, losses=dict(zip(E.L.LOSS_NAMES, o["losses"][b].tolist())), losses_row=o["losses"][b].copy(),
                       params=o["params"][b].copy(), hand=(hv, hf), obj=(ov, of), nan_in_phase_b=bool(o["nan_b"][b]))
            if f & (16 | 32) or rf[1]:       # capacity (cannot happen: sized from the scenes) / not a closed manifold / odd image mesh
                res.update(ok=False, reason="fallback")
            elif f & 2 or rf[0] & 2:
                res.update(ok=False, reason="fractional-coverage fragment list overflowed")
            elif f & 64:
                res.update(ok=False, reason="empty object mesh")
            out.append((tag, res))
        slot.job = None
        return out

    # -------------------------------------------------------------------------------------------- feeding the streams
    def _feed_stream(self, k, slots_per_stream, inbox, outbox, dev_index):
        """Body of stream k's feeder thread: jobs from `inbox` ([(tag, scene)] lists; None = no more) onto the stream's slots
        in turn, one list of (tag, result) per job into `outbox`, in the order the jobs came.  The thread blocks where its
        OWN stream makes it wait -- a full hardware queue in a replay, the event of its oldest job -- and nowhere else, so one
        stream's backlog never keeps the others from being fed."""
        import torch
        torch.cuda.set_device(dev_index)              # the current device is a per-thread setting
        pending = []                                  # jobs queued on the stream, oldest first: _Slot, or a ready result list
        turn = self._turn.setdefault(k, 0)

        def retire_oldest():
            job = pending.pop(0)
            if isinstance(job, list):
                outbox.put(job)
                return
            tags = [tag for tag, _ in job.job]
            try:
                job.done.synchronize()                # outside the gate: this is where the thread spends its time
                with self.gpu_gate.shared():
                    outbox.put(self._retire(job))
            except Exception as e:  # noqa: BLE001
                job.job = None
                outbox.put([(tag, dict(ok=False, reason="fallback", error=e)) for tag in tags])

        while True:
            chunk = inbox.get()
            if chunk is None:
                break
            pos = k + self.n_streams * (turn % slots_per_stream)
            turn += 1
            while any(j is self.slots.get(pos) for j in pending if not isinstance(j, list)):
                retire_oldest()
            scenes = [s for _, s in chunk]
            scenes += [scenes[0]] * (self.per_slot - len(scenes))      # copies of the first image fill the slot: one shape, one set of graphs
            try:
                slot = self._slot_for(pos, scenes)
                if slots_per_stream > 1 and pos == k and pos + self.n_streams not in self.slots:
                    self._slot_for(pos + self.n_streams, scenes)      # the stream's second slot, built while the stream is still idle
                with self.gpu_gate.shared():
                    self._enqueue(slot, chunk, scenes)
                pending.append(slot)
            except Exception as e:  # noqa: BLE001 -- the job as a whole failed: its images go to the caller's one-by-one path
                bad = self.slots.pop(pos, None)
                if bad is not None:
                    bad.job = None
                pending.append([(tag, dict(ok=False, reason="fallback", error=e)) for tag, _ in chunk])
        while pending:
            retire_oldest()
        self._turn[k] = turn

    def run_stream(self, items, total=None):
        """Generator: (tag, scene) pairs in, (tag, result) pairs out in the same order, `in_flight` images on the GPU and --
        for long lists (`total`, when the caller knows it, above four times that; unknown = long) -- as many again queued
        behind them.  One feeder thread per stream for the duration of the call (`_feed_stream`); this thread forms the
        jobs, deals them out round robin and hands back the results."""
        import queue
        import threading
        import torch
        # a second slot per stream costs its construction and captures (~30 ms) and saves the stream's idle time while the
        # host reads a job back and loads the next (~20 ms per job): it pays from about five jobs per stream on
        slots_per_stream = 1 if total is not None and total <= 4 * self.in_flight else 2
        if self.streams is None:
            self.streams = list(self.E.concurrent_streams(self.n_streams, torch.device(self.device)))
        dev_index = torch.device(self.device).index
        if dev_index is None:
            dev_index = torch.cuda.current_device()
        feeders = []
        for k in range(self.n_streams):
            inbox, outbox = queue.Queue(maxsize=1), queue.Queue()
            th = threading.Thread(target=self._feed_stream, args=(k, slots_per_stream, inbox, outbox, dev_index), daemon=True)
            th.start()
            feeders.append((th, inbox, outbox))
        order = []                      # feeder of every job not yet handed back, in submission order
        n_jobs = 0

        def ready():
            while order and not feeders[order[0]][2].empty():
                yield from feeders[order.pop(0)][2].get()

        try:
            key = lambda s: (int(s["H"]), int(s["W"]), len(s["hand_verts"]), len(s["hand_faces"]))
            chunk, chunk_key = [], None

            def deal(chunk):
                nonlocal n_jobs
                k = n_jobs % self.n_streams
                n_jobs += 1
                feeders[k][1].put(chunk)            # blocks while the feeder has not taken its previous job: back-pressure
                order.append(k)

            for tag, scene in items:
                k = key(scene)
                if chunk and (k != chunk_key or len(chunk) == self.per_slot):
                    deal(chunk)
                    chunk = []
                    yield from ready()
                chunk.append((tag, scene))
                chunk_key = k
            if chunk:
                deal(chunk)
        finally:
            for _, inbox, _ in feeders:
                inbox.put(None)
        while order:
            yield from feeders[order.pop(0)][2].get()
        for th, _, _ in feeders:
            th.join()

    def run(self, scenes):
        """Results in the order of `scenes` (same-shape images are queued together, in list order)."""
        key = lambda s: (int(s["H"]), int(s["W"]), len(s["hand_verts"]), len(s["hand_faces"]))
        order = sorted(range(len(scenes)), key=lambda i: key(scenes[i]))        # stable
        results = [None] * len(scenes)
        for i, r in self.run_stream(((i, scenes[i]) for i in order), total=len(scenes)):
            results[i] = r
        return results


def export_result(res, save_path_obj, save_path_hand):
    """{idx}_obj.ply / {idx}_hand.ply (run.py:221-222) from a MeshGuidanceRunner result."""
    meshio.save_ply(save_path_obj, *res["obj"])
    meshio.save_ply(save_path_hand, *res["hand"])
    return res["obj"], res["hand"]


def export_meshes(gb, b, save_path_obj, save_path_hand):
    """{idx}_obj.ply / {idx}_hand.ply (run.py:221-222): the optimised meshes in the MoGe world."""
    import torch
    m = gb.meta[b]
    world = gb.region("world", torch.float32, (-1, 3)).detach().cpu().numpy()
    faces = gb.faces.detach().cpu().numpy().astype(np.int64)
    hv = world[m["v_off"]:m["v_off"] + m["Vh"]]
    ov = world[m["v_off"] + m["Vh"]:m["v_off"] + m["Vh"] + m["Vo"]]
    hf = faces[m["f_off"]:m["f_off"] + m["Fh"]] - m["v_off"]
    of = faces[m["f_off"] + m["Fh"]:m["f_off"] + m["Fh"] + m["Fo"]] - m["v_off"] - m["Vh"]
    meshio.save_ply(save_path_obj, ov, of)
    meshio.save_ply(save_path_hand, hv, hf)
    return (ov, of), (hv, hf)
