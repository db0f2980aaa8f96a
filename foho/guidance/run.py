"""`python3 -m foho.guidance.run` -- guidance stage driver (same flags, kwargs and file-name contract as the
reference's src/foho/guidance/run.py; RUN below), with the optimisation-in-the-loop arithmetic on MI355X.

What is kept from the reference: the nine required flags + --task_list_file (RUN:264-289), `run(**paths)`
(RUN:188-199), the per-image path derivation from `{index} = filename.split("_")[0]` (RUN:210-222), the skip rules
(outputs exist RUN:224-226, empty masks RUN:232-236), the per-image try/except-continue (RUN:257-259) and
`_load_task_list` (RUN:178-185).  Added (the MI355X counterpart of the reference's SLURM array, RUN:178-185): when
launched as one process per GPU (`python -m torch.distributed.run --nproc-per-node N -m foho.guidance.run ...`) every
rank binds to GPU LOCAL_RANK, takes its round-robin share of the image list, and at the end of the batch ONE
all-reduce(SUM) of the metrics vector (followmyhold_amd.sharding.METRIC_NAMES; RCCL over xGMI, "nccl") lets rank 0
print the batch totals.  There is no collective on the data path.  FOHO_DIST_BACKEND=gloo runs the same code with a
CPU-side reduce (several ranks on one GPU / no GPU: tests).

With FOHO_MESH_LEVEL_GUIDANCE=1 (fixed object meshes, no diffusion networks) the rank's images do not go one by one: `run()`
hands them to followmyhold_amd.inputs.MeshGuidanceRunner, FOHO_IMAGES_IN_FLIGHT (default 16) at a time -- several images per
kernel launch on several HIP streams, graphs captured once per process -- with the reference's per-image skip rules and
error isolation kept (an image that fails is reported and the others go on).

`run_hunyuan_w_guid` runs followmyhold_amd.pipeline.GuidedShapePipeline -- the patched Hunyuan pipeline's __call__ with
the guidance arithmetic on HIP -- over the Hunyuan3D-2 DiT + ShapeVAE (PyTorch networks outside the hot path, SURVEY.md
8(a) A20), which are looked up at call time; without hy3dgen a clear error is raised unless FOHO_STANDIN_NETWORKS=1
(random-initialised stand-ins) or FOHO_MESH_LEVEL_GUIDANCE=1 (fixed object mesh) is set.
"""
import argparse
import json
import os
from typing import Dict, List, Optional

import time

import numpy as np

from followmyhold_amd import sharding
from foho.configs import OptimizationConfig

# Per-process tally behind the end-of-batch all-reduce: run_hunyuan_w_guid adds the loss terms of every image it
# finishes (sharding.METRIC_NAMES order), run() adds wall time and failures.
_METRICS = np.zeros(len(sharding.METRIC_NAMES), np.float64)
_RUNNERS = {}      # (device, images in flight, configuration) -> followmyhold_amd.inputs.MeshGuidanceRunner of this process


def _tally(vec) -> None:
    global _METRICS
    _METRICS = _METRICS + np.asarray(vec, np.float64)


def _dist_setup():
    """Bind this process to its GPU and join the process group when launched under torch.distributed.
    Returns (rank, world_size, device string, dist module or None, whether this call created the group)."""
    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", rank if world > 1 else 0))
    has_gpu = torch.cuda.is_available()
    backend = os.environ.get("FOHO_DIST_BACKEND", "nccl" if has_gpu else "gloo")
    device = "cuda"
    if has_gpu:
        n = torch.cuda.device_count()
        if backend == "nccl" and world > 1 and local_rank >= n:
            raise RuntimeError(f"LOCAL_RANK {local_rank} but only {n} GPUs are visible: launch one process per GPU")
        local_rank %= n                      # gloo: several ranks may share one GPU (tests, development)
        torch.cuda.set_device(local_rank)
        device = f"cuda:{local_rank}"
    dist, created = None, False
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            created = True
    return rank, world, device, dist, created


def _reduce_and_report(rank, world, device, dist, created) -> Optional[Dict[str, float]]:
    """End of batch: all-reduce(SUM) of the metrics vector; rank 0 prints and returns the totals."""
    import torch
    on_gpu = dist is not None and dist.get_backend() == "nccl"
    vec = torch.tensor(_METRICS, dtype=torch.float64, device=device if on_gpu else "cpu")
    vec = sharding.all_reduce_metrics(vec, dist)
    out = None
    if rank == 0:
        out = dict(zip(sharding.METRIC_NAMES, vec.tolist()))
        out["world_size"] = world
        print("Batch metrics: " + json.dumps(out), flush=True)
    if dist is not None and created:
        dist.barrier()
        dist.destroy_process_group()
    return out


def _setup_sys_path(project_root: str) -> None:
    """RUN:57-64: the checkout's third_party trees (Hunyuan3D-2 / hy3dgen, the HaMeR estimator) and the project root
    become importable -- that is where the DiT / ShapeVAE networks come from when they are installed."""
    import sys
    from foho.configs import third_party_root
    tp = third_party_root()
    hy3dgen_root = os.path.join(tp, "Hunyuan3D-2")
    for p in (tp, os.path.join(tp, "estimator"), hy3dgen_root, os.path.join(hy3dgen_root, "hy3dgen"), project_root):
        if p and p not in sys.path:
            sys.path.append(p)


def derive_paths(cropped_obj_img: str, cropped_obj_img_dir: str, mask_dir: str, moge_out_dir: str,
                 hunyuan_hoi_mesh_dir: str, hamer_out_dir: str, h2m_rt_dir: str, aligned_mano_dir: str,
                 guidance_out_dir: str) -> Dict[str, str]:
    """File-name contract of RUN:210-222."""
    index = cropped_obj_img.split("_")[0]
    j = os.path.join
    return dict(
        index=index,
        is_right=cropped_obj_img.split("_")[-1].split(".")[0],
        cropped_obj_img_path=j(cropped_obj_img_dir, cropped_obj_img),
        cropped_hand_mask_path=j(mask_dir, f"{index}_cropped_hand_mask.png"),
        cropped_obj_mask_path=j(mask_dir, f"{index}_cropped_obj_mask.png"),
        moge_mesh_path=j(moge_out_dir, f"{index}_cropped_hoi/mesh.glb"),
        moge_fov_path=j(moge_out_dir, f"{index}_cropped_hoi/fov.json"),
        T_h2m_path=j(h2m_rt_dir, f"{index}_hoi_mesh.npy"),
        aligned_mano_mesh_path=j(aligned_mano_dir, f"{index}_hamer_aligned_mano.ply"),
        hunyuan_hoi_mesh_path=j(hunyuan_hoi_mesh_dir, f"{index}_hoi_mesh.ply"),
        hamer_for_guid_path=j(hamer_out_dir, f"{index}_kps_for_guidance.npy"),
        save_path_obj=j(guidance_out_dir, f"{index}_obj.ply"),
        save_path_hand=j(guidance_out_dir, f"{index}_hand.ply"),
    )


def _load_task_list(task_list_file: Optional[str], cropped_obj_img_dir: str) -> List[str]:
    """RUN:178-185, plus this rank's share when WORLD_SIZE > 1."""
    return sharding.load_task_list(task_list_file, cropped_obj_img_dir)


def _read_mask(path):
    from PIL import Image
    return np.array(Image.open(path))


def run_hunyuan_w_guid(cropped_obj_img_path, fovx, hamer_for_guid_path, aligned_mano_mesh_path, cropped_obj_mask_path,
                       cropped_hand_mask_path, moge_mesh_path, T_h2m_path, hunyuan_hoi_mesh_path, save_path_obj,
                       save_path_hand, config, device="cuda"):
    """RUN:65-175: camera + the two renderers, the patched Hunyuan pipeline call, post-processing and PLY export.
    Returns (obj_mesh, hand_mesh) or (None, None)."""
    from followmyhold_amd import facade as p3d  # pytorch3d-shaped operator facade backed by libfoho_hip.so
    from followmyhold_amd import meshio
    import torch

    H, W = _read_mask(cropped_hand_mask_path).shape[:2]
    R = torch.tensor([[-1.0, 0, 0], [0, 1.0, 0], [0, 0, -1.0]], device=device).unsqueeze(0)  # RUN:84-88
    cameras = p3d.FoVPerspectiveCameras(device=device, R=R, T=torch.zeros(1, 3, device=device), znear=0.01, zfar=100.0,
                                        fov=fovx)
    blend = p3d.BlendParams(sigma=1e-8, gamma=1e-8)
    blur = float(np.log(1.0 / 1e-4 - 1.0) * blend.sigma)
    renderer = p3d.MeshRenderer(
        rasterizer=p3d.MeshRasterizer(cameras=cameras, raster_settings=p3d.RasterizationSettings(
            image_size=(H, W), blur_radius=blur, faces_per_pixel=1, bin_size=-1)),
        shader=p3d.PhongNormalShader(cameras=cameras, blend_params=blend))
    sil_renderer = p3d.MeshRenderer(
        rasterizer=p3d.MeshRasterizer(cameras=cameras, raster_settings=p3d.RasterizationSettings(
            image_size=(H, W), blur_radius=blur, faces_per_pixel=100, bin_size=None)),
        shader=p3d.SoftSilhouetteShader(blend_params=blend))
    pipeline = _build_pipeline(device)
    if pipeline is None:        # FOHO_MESH_LEVEL_GUIDANCE=1 and no networks: phases A/B/C on the fixed Hunyuan mesh
        return _mesh_level_guidance(fovx, hamer_for_guid_path, aligned_mano_mesh_path, cropped_obj_mask_path,
                                    cropped_hand_mask_path, moge_mesh_path, T_h2m_path, hunyuan_hoi_mesh_path,
                                    save_path_obj, save_path_hand, config, device)
    out = pipeline(
        image=_load_object_image(cropped_obj_img_path), mc_algo="mc", generator=torch.manual_seed(2), config=config,
        renderer=renderer, sil_renderer=sil_renderer, cropped_obj_img_path=cropped_obj_img_path,
        hamer_for_guid_path=hamer_for_guid_path, aligned_mano_mesh_path=aligned_mano_mesh_path,
        obj_mask_path=cropped_obj_mask_path, hand_mask_path=cropped_hand_mask_path, moge_mesh_path=moge_mesh_path,
        h2m_rt_path=T_h2m_path, hunyuan_hoi_mesh_path=hunyuan_hoi_mesh_path)
    gb = getattr(pipeline, "guidance_batch", None)
    if out is not None and gb is not None:
        _tally(sharding.local_metrics(gb, n_steps=int(pipeline.stats["inner_iterations"]), wall_ms=0.0).cpu().numpy())
    obj_mesh, hand_mesh = out       # a None return (NaN in phase B, PL:1442-1444) raises here like in the reference
    return _postprocess_and_save(obj_mesh, hand_mesh, cropped_obj_img_path, save_path_obj, save_path_hand)


def _postprocess_and_save(obj_mesh, hand_mesh, cropped_obj_img_path, save_path_obj, save_path_hand):
    """RUN:158-175: floaters, degenerate faces, decimation to 40k faces, export of the two PLY files."""
    from followmyhold_amd import meshio
    try:    # RUN:159-166: floaters, degenerate faces, decimation to 40k faces, export
        from followmyhold_amd import postprocess as pp
        obj_mesh = pp.TriMesh(obj_mesh.verts_packed().cpu().numpy(), obj_mesh.faces_packed().cpu().numpy())
        obj_mesh = pp.FloaterRemover()(obj_mesh)
        obj_mesh = pp.DegenerateFaceRemover()(obj_mesh)
        obj_mesh = pp.FaceReducer()(obj_mesh)
        obj_mesh.export(save_path_obj)
        meshio.save_ply(save_path_hand, hand_mesh.verts_packed().cpu().numpy(), hand_mesh.faces_packed().cpu().numpy())
    except Exception:
        print(f"Error in saving mesh for {cropped_obj_img_path}")
        return None, None
    if len(obj_mesh.vertices) == 0:
        print(f"Empty mesh for {cropped_obj_img_path}")
        return None, None
    return obj_mesh, hand_mesh


def _load_object_image(path):
    """RUN:122-138: RGBA image of the object crop; pure white pixels become transparent.  The "inpaint" branch needs
    hy3dgen's BackgroundRemover (rembg), a network outside this repository."""
    from PIL import Image
    if "inpaint" in path:
        try:
            from hy3dgen.rembg import BackgroundRemover
        except ImportError as e:
            raise RuntimeError("inpainted crops need hy3dgen.rembg.BackgroundRemover (RUN:119-126)") from e
        return [BackgroundRemover()(Image.open(path).convert("RGB"))]
    a = np.array(Image.open(path).convert("RGBA"))
    white = (a[..., :3] == 255).all(-1)
    a[white] = (255, 255, 255, 0)
    return [Image.fromarray(a, "RGBA")]


_PIPELINE = None


def _build_pipeline(device):
    """The guided pipeline (followmyhold_amd.pipeline.GuidedShapePipeline) over the Hunyuan3D-2 networks when hy3dgen is
    installed (RUN:140), over random-initialised stand-ins when FOHO_STANDIN_NETWORKS=1 (plumbing runs without the
    weights), None when FOHO_MESH_LEVEL_GUIDANCE=1 asks for the fixed-mesh driver.  Built once per process."""
    global _PIPELINE
    if _PIPELINE is not None:
        return _PIPELINE
    from followmyhold_amd.pipeline import GuidedShapePipeline
    if os.environ.get("FOHO_STANDIN_NETWORKS") == "1":
        import torch
        from followmyhold_amd import standins
        _PIPELINE = standins.make_standin_pipeline(device=device, dtype=torch.float32)
        return _PIPELINE        # (the tiny stand-in decoder -- width 32 -- is outside the shapes of the HIP geometry decoder)
    try:
        from hy3dgen.shapegen import Hunyuan3DDiTFlowMatchingPipeline
    except ImportError as e:
        if os.environ.get("FOHO_MESH_LEVEL_GUIDANCE") == "1":
            return None
        raise RuntimeError("Hunyuan3D-2 (hy3dgen) is not installed: the DiT/VAE that produce the object latent are "
                           "PyTorch networks outside the MI355X hot path (SURVEY.md 8(a) A20). Set FOHO_STANDIN_NETWORKS=1 "
                           "to run the full guided pipeline on random-initialised stand-ins, FOHO_MESH_LEVEL_GUIDANCE=1 "
                           "to run phases A/B/C on the fixed Hunyuan mesh, or use followmyhold_amd.pipeline / "
                           "followmyhold_amd.engine directly.") from e
    # FOHO_HIP_GEO_DECODER=0 keeps the ShapeVAE's geometry decoder on its torch module (default: foho_geo_decode_fwd / _bwd),
    # FOHO_HIP_VAE_TRANSFORMER=0 its transformer (default: foho_vae_fwd / _bwd)
    _PIPELINE = GuidedShapePipeline.from_hy3dgen(Hunyuan3DDiTFlowMatchingPipeline.from_pretrained("tencent/Hunyuan3D-2"),
                                                 hip_geo_decoder=os.environ.get("FOHO_HIP_GEO_DECODER", "1") != "0",
                                                 hip_vae_transformer=os.environ.get("FOHO_HIP_VAE_TRANSFORMER", "1") != "0")
    return _PIPELINE


def _mesh_level_guidance(fovx, hamer_for_guid_path, aligned_mano_mesh_path, cropped_obj_mask_path, cropped_hand_mask_path,
                         moge_mesh_path, T_h2m_path, hunyuan_hoi_mesh_path, save_path_obj, save_path_hand, config, device):
    """Guidance on a fixed object mesh (no diffusion model): every input file of PL:1217-1256 is read, the MoGe mesh
    is rendered into the target maps on the GPU, phases A/B/C run with the reference's iteration schedule and the
    two output meshes are written.  Returns ((obj_verts, obj_faces), (hand_verts, hand_faces))."""
    from followmyhold_amd import engine as E
    from followmyhold_amd import inputs
    p = dict(cropped_hand_mask_path=cropped_hand_mask_path, cropped_obj_mask_path=cropped_obj_mask_path,
             moge_mesh_path=moge_mesh_path, moge_fov_path=os.path.join(os.path.dirname(moge_mesh_path), "fov.json"),
             T_h2m_path=T_h2m_path, aligned_mano_mesh_path=aligned_mano_mesh_path,
             hunyuan_hoi_mesh_path=hunyuan_hoi_mesh_path, hamer_for_guid_path=hamer_for_guid_path)
    scene = inputs.load_scene_from_files(p, inputs.load_j_regressor(), E.hip_render_fn(device))
    if fovx is not None:
        scene["fov"] = float(fovx)
    gb = inputs.run_mesh_guidance([scene], config, device=device)
    _tally(sharding.local_metrics(gb, n_steps=_n_iterations(config), wall_ms=0.0).cpu().numpy())
    return inputs.export_meshes(gb, 0, save_path_obj, save_path_hand)


def _n_iterations(config) -> int:
    return int(config.optimization_steps_hand) + int(config.optimization_steps_scale) + \
        int(config.optimization_steps_joint) * max(0, int(config.num_inference_steps) - int(config.guidance_start_step) - 1)


def _mesh_level_batched() -> bool:
    """True when run() should take the rank's images through MeshGuidanceRunner: mesh-level guidance was asked for and the
    networks are not there (with them, every image is a full pipeline call and stays one at a time like RUN:208-259)."""
    if os.environ.get("FOHO_MESH_LEVEL_GUIDANCE") != "1" or os.environ.get("FOHO_STANDIN_NETWORKS") == "1":
        return False
    if int(os.environ.get("FOHO_IMAGES_IN_FLIGHT", "16")) < 1:
        return False
    try:
        import hy3dgen.shapegen  # noqa: F401
        return False
    except ImportError:
        return True


def _screen(cropped_obj_img, dirs, say=print):
    """RUN:210-236 for one list entry: paths, skip rules, fov.  Returns (paths, fovx), or None when the image is skipped
    (say=print: the reference's message is printed; say=None: the message is returned instead, for callers that report in
    list order)."""
    p = derive_paths(cropped_obj_img, **dirs)
    index = p["index"]
    msg = None
    if os.path.exists(p["save_path_obj"]) and os.path.exists(p["save_path_hand"]):
        msg = f"{index} already exists, skipping"
    else:
        with open(p["moge_fov_path"], "r", encoding="utf-8") as f:
            fovx = float(json.load(f)["fov_x"])
        if _read_mask(p["cropped_hand_mask_path"]).max() == 0 or _read_mask(p["cropped_obj_mask_path"]).max() == 0:
            msg = f"Skipping {index} due to empty mask"
    if msg is not None:
        if say is None:
            return msg
        say(msg)
        return None
    return p, fovx


def _run_batched(assigned_imgs, dirs, config, device) -> None:
    """The per-image body of RUN:208-259 for the mesh-level path, FOHO_IMAGES_IN_FLIGHT images at a time.  Failures stay
    per image: a file that does not load, an image the runner hands back (object not a closed manifold -> the exact-size
    single-image driver), a NaN in phase B (the reference's pipeline returns None there, PL:1442-1444, which surfaces as
    "Error in processing")."""
    from followmyhold_amd import inputs
    # no more slots than list entries: a short list must not be padded up to the default with copies of its first image
    # default: 16 (four streams x four images per launch); long lists 32 -- 10-15 % more images per second in the steady state
    # (190 against 172 for inputs in host memory, 148-158 against 138 from files) for twice the memory and latency of a job and
    # a slower start (160 folders on a fresh process: 9.1 against 8.7 ms per image), hence only from 512 entries on
    default_in_flight = 32 if len(assigned_imgs) >= 512 else 16
    in_flight = max(1, min(int(os.environ.get("FOHO_IMAGES_IN_FLIGHT", default_in_flight)), len(assigned_imgs)))
    # slots, hipGraphs and target renderers belong to the PROCESS: a second run() call with the same settings finds them ready
    key = (str(device), in_flight, repr(sorted(vars(config).items())))
    runner = _RUNNERS.get(key)
    if runner is None:
        _RUNNERS.clear()             # one set of slots at a time: another configuration replaces it
        runner = _RUNNERS[key] = inputs.MeshGuidanceRunner(config, device=device, in_flight=in_flight)
    n_iter = _n_iterations(config)

    def fail(name, e):
        print(f"Error in processing {name} : {e}")
        _tally_named(n_failed=1)

    def finish(name, p, res):
        if not res["ok"] and res.get("reason") == "fallback":      # any mesh goes through the exact-size driver
            res = None
            with runner.gpu_gate.exclusive():      # it captures graphs of its own: not next to the loader threads' renders
                obj_mesh, hand_mesh = _mesh_level_guidance(
                    None, p["hamer_for_guid_path"], p["aligned_mano_mesh_path"], p["cropped_obj_mask_path"], p["cropped_hand_mask_path"],
                    p["moge_mesh_path"], p["T_h2m_path"], p["hunyuan_hoi_mesh_path"], p["save_path_obj"], p["save_path_hand"], config, device)
        else:
            if not res["ok"]:
                raise RuntimeError(res["reason"])
            if res["nan_in_phase_b"]:
                raise TypeError("cannot unpack non-iterable NoneType object")     # what RUN:141 raises on PL:1442-1444's `return None`
            obj_mesh, hand_mesh = inputs.export_result(res, p["save_path_obj"], p["save_path_hand"])
            _tally(sharding.image_metrics(res["losses_row"], res["flags"], n_iter))
        if len(obj_mesh[0]) == 0:
            print(f"Empty mesh for {p['cropped_obj_img_path']}")
            print(f"Error in reconstruction for {p['index']}")
            return
        print(f"Reconstructed object {p['index']}")

    # Reading an image's files (two masks, three meshes -- the MoGe image mesh has half a million faces --, key points) costs
    # the host 25-35 ms, five times the image's share of the GPU job: the list entries are prepared by a pool of threads, a
    # bounded number ahead of the images on the GPU.  The threads do HOST work only (file parsing releases the GIL); the
    # image mesh travels with the scene and is rendered into the target maps on the device, inside the image's job
    # (engine.TargetRenderer) -- GPU work from loader threads would queue behind whole jobs on the shared hardware queues.
    # Messages, results and failures are consumed in list order, so the log reads like the reference's sequential loop.
    jr = inputs.load_j_regressor()

    def prepare(cropped_obj_img):
        try:
            scr = _screen(cropped_obj_img, dirs, say=None)
            if isinstance(scr, str):
                return ("skip", scr)
            p, fovx = scr
            scene = inputs.load_scene_from_files(p, jr, None)
            scene["fov"] = float(fovx)
            return ("ok", p, scene)
        except Exception as e:  # noqa: BLE001 -- RUN:257-259
            return ("fail", e)

    workers = max(1, min(int(os.environ.get("FOHO_LOADER_THREADS", "8")), os.cpu_count() or 1))
    ahead = max(2 * in_flight, workers)
    # The main thread feeds the GPU with many short calls that release the interpreter lock (graph replays, copies, event
    # waits); with the default 5 ms switch interval every re-acquisition may wait that long for a loader thread in the middle
    # of Python code.  A short interval for the duration of the batch keeps the hand-overs in the 0.1 ms range.
    import sys
    old_interval = sys.getswitchinterval()
    sys.setswitchinterval(1e-4)
    try:
        _feed(runner, assigned_imgs, prepare, workers, ahead, fail, finish)
    finally:
        sys.setswitchinterval(old_interval)


def _feed(runner, assigned_imgs, prepare, workers, ahead, fail, finish) -> None:
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(max_workers=workers) as pool:
        def loaded():
            """(list entry, paths), scene of every entry that passes the skip rules and loads, in list order."""
            todo = iter(assigned_imgs)
            queue = []
            for name in todo:
                queue.append((name, pool.submit(prepare, name)))
                if len(queue) >= ahead:
                    break
            while queue:
                name, fut = queue.pop(0)
                nxt = next(todo, None)
                if nxt is not None:
                    queue.append((nxt, pool.submit(prepare, nxt)))
                out = fut.result()
                if out[0] == "skip":
                    print(out[1])
                elif out[0] == "fail":
                    fail(name, out[1])
                else:
                    print(f"Processing {out[1]['index']}")
                    yield (name, out[1]), out[2]

        # the runner keeps `in_flight` images on the GPU and as many queued behind them; a finished image is exported while
        # the next ones run (inputs.MeshGuidanceRunner.run_stream)
        for (name, p), res in runner.run_stream(loaded(), total=len(assigned_imgs)):
            try:
                if "error" in res:
                    print(f"The job of {p['index']} failed as a whole ({res['error']}); retrying the image on its own")
                finish(name, p, res)
            except Exception as e:  # noqa: BLE001 -- RUN:257-259
                fail(name, e)


def _tally_named(**kw) -> None:
    v = np.zeros(len(sharding.METRIC_NAMES), np.float64)
    for k, x in kw.items():
        v[sharding.IDX[k]] = x
    _tally(v)


def run(project_root: str, cropped_obj_img_dir: str, mask_dir: str, moge_out_dir: str, hunyuan_hoi_mesh_dir: str,
        hamer_out_dir: str, h2m_rt_dir: str, aligned_mano_dir: str, guidance_out_dir: str,
        task_list_file: Optional[str] = None) -> Optional[Dict[str, float]]:
    global _METRICS
    _METRICS = np.zeros(len(sharding.METRIC_NAMES), np.float64)
    _setup_sys_path(project_root)
    rank, world, device, dist, created = _dist_setup()
    # Every rank must reach the end-of-batch all-reduce, whatever happens to its own share of the list: the body below
    # runs under try/finally, a rank that failed outside the per-image handlers contributes its tally (n_failed > 0) and
    # re-raises after the collective instead of leaving the others waiting for the RCCL timeout.
    t_start = time.perf_counter()
    error = None
    try:
        config = OptimizationConfig()
        os.makedirs(guidance_out_dir, exist_ok=True)
        assigned_imgs = _load_task_list(task_list_file, cropped_obj_img_dir)   # this rank's share when WORLD_SIZE > 1
        dirs = dict(cropped_obj_img_dir=cropped_obj_img_dir, mask_dir=mask_dir, moge_out_dir=moge_out_dir,
                    hunyuan_hoi_mesh_dir=hunyuan_hoi_mesh_dir, hamer_out_dir=hamer_out_dir, h2m_rt_dir=h2m_rt_dir,
                    aligned_mano_dir=aligned_mano_dir, guidance_out_dir=guidance_out_dir)
        if _mesh_level_batched():
            _run_batched(assigned_imgs, dirs, config, device)
        elif (_pipeline_batch_size() > 1 and len(assigned_imgs) > 1 and not os.environ.get("FOHO_DEBUG_DIR") and _networks_available()
              and _build_pipeline(device) is not None):
            # (FOHO_DEBUG_DIR: the per-image dumps -- losses.txt, params.json, per-step renders and meshes, PL:1076-1091, 1664-1675 --
            # are `__call__`'s, so a debug run goes one image at a time like the reference)
            _run_pipeline_batched(assigned_imgs, dirs, config, device, _pipeline_batch_size())
        else:
            _run_one_by_one(assigned_imgs, dirs, config)
        print("Finished processing all images")
    except BaseException as e:  # noqa: BLE001 -- reported after the collective
        error = e
        _tally_named(n_failed=1)
    _tally_named(sum_wall_ms=(time.perf_counter() - t_start) * 1e3)
    try:
        out = _reduce_and_report(rank, world, device, dist, created)
    finally:
        if error is not None:
            raise error
    return out


def _networks_available() -> bool:
    """Can _build_pipeline succeed?  (Stand-ins requested, or hy3dgen importable.)  Without networks the one-by-one path keeps
    its behaviour: the per-image call raises the explanatory error, which the per-image handler reports (RUN:257-259)."""
    if os.environ.get("FOHO_STANDIN_NETWORKS") == "1":
        return True
    import importlib.util
    try:
        return importlib.util.find_spec("hy3dgen.shapegen") is not None
    except (ImportError, ValueError):
        return False


def _pipeline_batch_size() -> int:
    """Images per pass of the schedule when the networks are in the loop (GuidedShapePipeline.call_batch): FOHO_PIPELINE_BATCH,
    default 4; 1 = the reference's one image at a time (guid_config.py:9)."""
    return max(1, int(os.environ.get("FOHO_PIPELINE_BATCH", "4")))      # (a set FOHO_DEBUG_DIR overrides it: see run())


def _run_pipeline_batched(assigned_imgs, dirs, config, device, batch) -> None:
    """RUN:208-259 with the networks in the loop, `batch` images per pass of the 20-step schedule: DiT and ShapeVAE on `batch`
    latents, one capacity-mode GuidanceBatch of `batch` slots (GuidedShapePipeline.call_batch).  Skip rules, messages,
    post-processing and the per-image error isolation are the one-image loop's; an image that leaves the batch (its entry of
    call_batch's result is a BatchLeftFastPath: non-manifold / over-capacity iso-surface, NaN loss) is redone on its own through
    run_hunyuan_w_guid while the rest of its group keeps the batched result; a group that fails as a whole is redone image by image."""
    import torch
    from followmyhold_amd.pipeline import BatchLeftFastPath
    pipeline = _build_pipeline(device)

    def one(name, p, fovx):
        try:
            print(f"Processing {p['index']}")
            obj_mesh, hand_mesh = run_hunyuan_w_guid(
                cropped_obj_img_path=p["cropped_obj_img_path"], fovx=fovx, hamer_for_guid_path=p["hamer_for_guid_path"],
                aligned_mano_mesh_path=p["aligned_mano_mesh_path"], cropped_obj_mask_path=p["cropped_obj_mask_path"],
                cropped_hand_mask_path=p["cropped_hand_mask_path"], moge_mesh_path=p["moge_mesh_path"], T_h2m_path=p["T_h2m_path"],
                hunyuan_hoi_mesh_path=p["hunyuan_hoi_mesh_path"], save_path_obj=p["save_path_obj"], save_path_hand=p["save_path_hand"],
                config=config)
            print(f"Error in reconstruction for {p['index']}" if obj_mesh is None or hand_mesh is None else f"Reconstructed object {p['index']}")
        except Exception as e:  # noqa: BLE001 -- RUN:257-259
            print(f"Error in processing {name} : {e}")
            _tally_named(n_failed=1)

    def flush(group):
        if len(group) == 1:
            return one(*group[0])
        try:
            for _, p, _ in group:
                print(f"Processing {p['index']}")
            paths = [dict(cropped_obj_img_path=p["cropped_obj_img_path"], hamer_for_guid_path=p["hamer_for_guid_path"],
                          aligned_mano_mesh_path=p["aligned_mano_mesh_path"], obj_mask_path=p["cropped_obj_mask_path"],
                          hand_mask_path=p["cropped_hand_mask_path"], moge_mesh_path=p["moge_mesh_path"], h2m_rt_path=p["T_h2m_path"],
                          hunyuan_hoi_mesh_path=p["hunyuan_hoi_mesh_path"]) for _, p, _ in group]
            images = [_load_object_image(p["cropped_obj_img_path"])[0] for _, p, _ in group]
            out = pipeline.call_batch(images, paths, generators=[torch.Generator().manual_seed(2) for _ in group], config=config,
                                      fovs=[f for _, _, f in group])
        except BatchLeftFastPath as e:
            print(f"Batch of {[p['index'] for _, p, _ in group]} left the batched path ({e}); one image at a time")
            return [one(*g) for g in group]
        except Exception as e:  # noqa: BLE001 -- a failure of the whole pass must not cost the group its images
            print(f"Batch of {[p['index'] for _, p, _ in group]} failed as a whole ({e}); one image at a time")
            return [one(*g) for g in group]
        # the batch's images that stayed in it (an image that left is tallied by its own run below)
        gbt = pipeline.guidance_batch
        rows, fls = gbt.losses.detach().cpu().numpy(), gbt.flags.detach().cpu().numpy()
        for b_, res in enumerate(out):
            if not isinstance(res, BatchLeftFastPath):
                _tally(np.asarray(sharding.image_metrics(rows[b_], int(fls[b_]), int(pipeline.stats["inner_iterations"])), np.float64))
        for g, res in zip(group, out):
            (name, p, _) = g
            if isinstance(res, BatchLeftFastPath):
                print(f"Image {p['index']} left the batched path ({res}); on its own")
                one(*g)
                continue
            try:
                obj_mesh, hand_mesh = res       # (None, hand): no decode of this image ever gave a surface
                obj_mesh, hand_mesh = _postprocess_and_save(obj_mesh, hand_mesh, p["cropped_obj_img_path"], p["save_path_obj"], p["save_path_hand"])
                print(f"Error in reconstruction for {p['index']}" if obj_mesh is None or hand_mesh is None else f"Reconstructed object {p['index']}")
            except Exception as e:  # noqa: BLE001
                print(f"Error in processing {name} : {e}")
                _tally_named(n_failed=1)

    groups = {}      # images of one mask size travel together (a GuidanceBatch has one H x W)
    for cropped_obj_img in assigned_imgs:
        try:
            scr = _screen(cropped_obj_img, dirs)
            if scr is None:
                continue
            p, fovx = scr
            shape = _read_mask(p["cropped_hand_mask_path"]).shape[:2]
        except Exception as e:  # noqa: BLE001
            print(f"Error in processing {cropped_obj_img} : {e}")
            _tally_named(n_failed=1)
            continue
        g = groups.setdefault(shape, [])
        g.append((cropped_obj_img, p, fovx))
        if len(g) == batch:
            flush(groups.pop(shape))
    for g in groups.values():
        flush(g)


def _run_one_by_one(assigned_imgs, dirs, config) -> None:
    """RUN:208-259: one image at a time through run_hunyuan_w_guid (the full pipeline with the networks in the loop)."""
    for cropped_obj_img in assigned_imgs:
        try:
            scr = _screen(cropped_obj_img, dirs)
            if scr is None:
                continue
            p, fovx = scr
            index = p["index"]
            print(f"Processing {index}")
            obj_mesh, hand_mesh = run_hunyuan_w_guid(
                cropped_obj_img_path=p["cropped_obj_img_path"], fovx=fovx, hamer_for_guid_path=p["hamer_for_guid_path"],
                aligned_mano_mesh_path=p["aligned_mano_mesh_path"], cropped_obj_mask_path=p["cropped_obj_mask_path"],
                cropped_hand_mask_path=p["cropped_hand_mask_path"], moge_mesh_path=p["moge_mesh_path"],
                T_h2m_path=p["T_h2m_path"], hunyuan_hoi_mesh_path=p["hunyuan_hoi_mesh_path"],
                save_path_obj=p["save_path_obj"], save_path_hand=p["save_path_hand"], config=config)   # device: "cuda" = the GPU _dist_setup() bound
            if obj_mesh is None or hand_mesh is None:
                print(f"Error in reconstruction for {index}")
                continue
            print(f"Reconstructed object {index}")
        except Exception as e:  # noqa: BLE001 -- RUN:257-259
            print(f"Error in processing {cropped_obj_img} : {e}")
            _tally_named(n_failed=1)
            continue


def build_parser() -> argparse.ArgumentParser:
    """The reference's command line (RUN:264-289): nine required flags + --task_list_file."""
    parser = argparse.ArgumentParser(description="Hunyuan3D-2 guidance")
    for flag in ["project_root", "cropped_obj_img_dir", "mask_dir", "moge_out_dir", "hunyuan_hoi_mesh_dir",
                 "hamer_out_dir", "h2m_rt_dir", "aligned_mano_dir", "guidance_out_dir"]:
        parser.add_argument(f"--{flag}", required=True)
    parser.add_argument("--task_list_file", default=None)
    return parser


def main() -> None:
    a = build_parser().parse_args()
    run(**vars(a))


if __name__ == "__main__":
    main()
