"""Golden trajectory of the REFERENCE's guided pipeline loop.

Run in the build container only (needs /root/reference; the GPU box never runs this):
    python tests/golden/make_pipeline_golden.py
It imports the reference's patched pipelines.py and utilz/kaolin_sdf_ops.py (stub modules for what is not installed, as
in make_golden.py), replaces the pytorch3d / kaolin names they use by the CPU restatement (oracle/p3d_ref.py), builds a
`Hunyuan3DDiTFlowMatchingPipeline_main` around this repository's random-initialised stand-in networks and the
reference's own scheduler, and executes the reference's `__call__` (PL:1044-1679) on a small synthetic scene with a short
schedule.  Stored in tests/golden/ref_pipeline.npz: the returned hand / object meshes and the loss lines the loop
printed.  tests/test_pipeline.py replays the same inputs through followmyhold_amd.pipeline.GuidedShapePipeline on the GPU.

What this pins: the reference's orchestration -- CFG schedule, scheduler stepping, phases A / B / C and their optimisers
and learning rates, latent -> SDF -> mesh decode per iteration, detach/clone points, per-step and final (res 384) decode,
output transforms.  The operators underneath are this repository's restatements (parity unpinned).
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = sys.argv[1] if len(sys.argv) > 1 else HERE          # optional output directory (tests regenerate into a temp dir)
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

RADIUS = 0.35           # Hunyuan-space radius of the stand-in decoder's sphere prior
VAE_KW = dict(num_latents=16, embed_dim=4, width=16, heads=2, layers=1, num_freqs=3, radius=RADIUS, sharpness=4.0, gain=0.1)
# variant 0: one joint denoising step, intersection term on; variant 1: another scene and image size, two joint denoising
# steps (scheduler.step between them, CFG scale decaying), intersection term off
VARIANTS = [
    dict(tag="", scene=dict(obj_kind="ico2", H=64, W=64, seed=3), config={},
         schedule=dict(num_inference_steps=4, guidance_start_step=2, handopt_start_step=1, guidance_end_step=4,
                       optimization_steps_hand=3, optimization_steps_scale=2, optimization_steps_joint=2)),
    dict(tag="_v1", scene=dict(obj_kind="ico2", H=80, W=80, seed=8), config=dict(use_intersection_loss=False),
         schedule=dict(num_inference_steps=5, guidance_start_step=2, handopt_start_step=1, guidance_end_step=5,
                       optimization_steps_hand=2, optimization_steps_scale=2, optimization_steps_joint=2)),
]
# variant 2: variant 0's scene with phase A's learning rates divided by 500 and six hand iterations.  At the reference's rates
# (quaternion 0.5) Adam with eps = 1e-4 turns a 1e-6 difference in a small gradient component into a 1e-3 step, so phase A of
# variants 0 / 1 is only comparable to a few 1e-3; at these rates the trajectory is comparable to 1e-5.
VARIANTS.append(dict(tag="_tame", scene=dict(obj_kind="ico2", H=64, W=64, seed=3),
                     config=dict(phase1_hand_lrs={"scale": 2e-5, "trans": 2e-5, "rot": 1e-3}),
                     schedule=dict(num_inference_steps=4, guidance_start_step=2, handopt_start_step=1, guidance_end_step=4,
                                   optimization_steps_hand=6, optimization_steps_scale=2, optimization_steps_joint=2)))
# variant 3: variant 2 with a ShapeVAE the matrix-core kernels take -- width 128, 2 heads of 64, 128 latent tokens -- so that the replay on
# the GPU runs `vae(pred)` on foho_vae_fwd / _bwd and the decodes on foho_geo_decode_* INSIDE a trajectory that is compared with the
# reference's (the reference side is float32 torch on the CPU, as for the other variants; the kernels store fp16)
VAE_KW_HD64 = dict(num_latents=128, embed_dim=4, width=128, heads=2, layers=2, num_freqs=3, radius=RADIUS, sharpness=4.0, gain=0.1)
VARIANTS.append(dict(tag="_hd64", scene=dict(obj_kind="ico2", H=64, W=64, seed=3), vae_kw=VAE_KW_HD64,
                     config=dict(phase1_hand_lrs={"scale": 2e-5, "trans": 2e-5, "rot": 1e-3}),
                     schedule=dict(num_inference_steps=4, guidance_start_step=2, handopt_start_step=1, guidance_end_step=4,
                                   optimization_steps_hand=6, optimization_steps_scale=2, optimization_steps_joint=2)))
SCENE, SCHEDULE = VARIANTS[0]["scene"], VARIANTS[0]["schedule"]


def build_inputs(root, scene=None):
    """Scene files in the reference's formats + the RGBA object crop; shared with tests/test_pipeline.py."""
    from helpers import oracle_render_fn
    from followmyhold_amd import synthetic
    from test_pipeline import _write
    sc = synthetic.build_scene(oracle_render_fn, **(scene or SCENE))
    T = sc["T_h2m"].astype(np.float64)
    ov_moge = sc["obj_verts"].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    r_moge = np.linalg.norm(ov_moge - ov_moge.mean(0), axis=1).mean()
    T2 = T.copy()
    T2[:3, :3] *= (r_moge / RADIUS) / np.cbrt(np.linalg.det(T[:3, :3]))
    sc["T_h2m"] = T2.astype(np.float32)
    sc["obj_verts"] = ((ov_moge - T2[:3, 3]) @ np.linalg.inv(T2[:3, :3]).T).astype(np.float32)
    paths = _write(root, sc)
    return sc, paths


def object_stats(v, f):
    """Topology-independent summary of the decoded object: centroid, bounding box, mean / std radius, area, volume."""
    v = np.asarray(v, np.float64)
    t = v[np.asarray(f)]
    n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    c = v.mean(0)
    r = np.linalg.norm(v - c, axis=1)
    vol = np.einsum("ij,ij->i", t[:, 0] - c, np.cross(t[:, 1] - c, t[:, 2] - c)).sum() / 6.0
    return np.concatenate([c, v.min(0), v.max(0), [r.mean(), r.std(), 0.5 * np.linalg.norm(n, axis=1).sum(), vol]])


def main():
    import make_golden as MG
    PL, SCH, get_guidance_params, OptimizationConfig = MG.load_reference()
    for m in list(sys.modules.values()):       # stub modules answer every attribute: give inspect a real __file__
        if isinstance(m, types.ModuleType) and "__file__" not in m.__dict__ and "__getattr__" in m.__dict__:
            m.__file__ = "<stub>"
    import torch.distributed.tensor  # noqa: F401  (lazy torch import that walks sys.modules through inspect)
    from oracle import p3d_ref as P
    from followmyhold_amd import inputs, meshio, standins
    from PIL import Image

    # third-party names inside the reference modules -> CPU restatement
    import utilz.kaolin_sdf_ops as KS
    KS.mesh_ops = types.SimpleNamespace(index_vertices_by_faces=P.index_vertices_by_faces, check_sign=P.check_sign)
    KS.km = types.SimpleNamespace(trianglemesh=types.SimpleNamespace(point_to_mesh_distance=P.point_to_mesh_distance))
    KS.Meshes = P.Meshes

    class IO:
        def register_meshes_format(self, fmt):
            pass

        def load_mesh(self, path, **_):
            v, f = inputs.load_glb(path) if path.endswith(".glb") else meshio.load_mesh(path)
            return P.Meshes(torch.from_numpy(np.asarray(v, np.float32)), torch.from_numpy(np.asarray(f, np.int64)))

    def load_ply(path):
        v, f = meshio.load_ply(path)
        return torch.from_numpy(v), torch.from_numpy(f)

    def imread(path, flag=None):
        return np.array(Image.open(path).convert("L"))

    def randn_tensor(shape, generator=None, device=None, dtype=None):
        return torch.randn(shape, generator=generator, dtype=dtype).to(device)

    for k, v in dict(Meshes=P.Meshes, join_meshes_as_scene=P.join_meshes_as_scene, TexturesVertex=P.TexturesVertex,
                     load_ply=load_ply, knn_points=P.knn_points, IO=IO, mesh_edge_loss=P.mesh_edge_loss,
                     quaternion_to_matrix=P.quaternion_to_matrix, knc=types.SimpleNamespace(FlexiCubes=P.FlexiCubes),
                     kaolin_sdf=KS, randn_tensor=randn_tensor,
                     cv2=types.SimpleNamespace(imread=imread, IMREAD_GRAYSCALE=0)).items():
        setattr(PL, k, v)
    sys.modules["pytorch3d.io.experimental_gltf_io"] = types.SimpleNamespace(_read_header=None, MeshGlbFormat=lambda: None)

    only = os.environ.get("FOHO_GOLDEN_ONLY")          # e.g. "_tame" or "base,_hd64": regenerate these variants, leave the others' files alone
    only = None if only is None else [("" if t == "base" else t) for t in only.split(",")]
    for variant in VARIANTS:
        if only is None or variant["tag"] in only:
            run_variant(variant, PL, SCH, OptimizationConfig, P, standins)


def run_variant(variant, PL, SCH, OptimizationConfig, P, standins):
    from PIL import Image
    with tempfile.TemporaryDirectory() as root, contextlib.ExitStack() as stack:
        import pathlib
        sc, paths = build_inputs(pathlib.Path(root), variant["scene"])
        # the reference torch.load()s ./third_party/estimator/hamer/J_regressor_hamer.pt relative to the cwd (PL:1218)
        os.makedirs(os.path.join(root, "third_party/estimator/hamer"))
        torch.save(torch.from_numpy(sc["J_regressor"]), os.path.join(root, "third_party/estimator/hamer/J_regressor_hamer.pt"))
        cwd = os.getcwd()
        os.chdir(root)
        stack.callback(os.chdir, cwd)

        vae_kw = variant.get("vae_kw", VAE_KW)
        net = standins.make_standin_pipeline(device="cpu", dtype=torch.float32, seed=1, **vae_kw)
        # the reference holds its networks with requires_grad at torch's default (GuidedShapePipeline switches it off: the guidance
        # optimises no weight); autograd then also forms the weight gradients, through other kernels for LayerNorm / Linear whose
        # input gradients differ in the last bit -- the committed trajectories are the reference's, so run it the reference's way
        for m_ in (net.vae, net.model, net.conditioner):
            m_.requires_grad_(True)
        pipe = object.__new__(PL.Hunyuan3DDiTFlowMatchingPipeline_main)
        pipe.vae, pipe.model, pipe.conditioner, pipe.image_processor = net.vae, net.model, net.conditioner, net.image_processor
        pipe.scheduler = SCH.FlowMatchEulerDiscreteScheduler()
        pipe.device, pipe.dtype = torch.device("cpu"), torch.float32
        cfg = OptimizationConfig()
        for k, v in {**variant["schedule"], **variant["config"]}.items():
            setattr(cfg, k, v)
        renderer = P.NormalRenderer(sc["fov"], sc["H"], sc["W"])
        sil_renderer = P.SilhouetteRenderer(sc["fov"], sc["H"], sc["W"])
        img = Image.open(paths["cropped_obj_img_path"])
        # every phase builds a fresh optimiser over fresh leaf tensors (PL:1318, 1384, 1478): remember them to read the
        # parameter values each phase ended with
        made = []
        for name in ("Adam", "AdamW"):
            base = getattr(torch.optim, name)

            def factory(params, *a, _base=base, _name=name, **kw):
                params = list(params)
                made.append((_name, [g["params"][0] for g in params], [g["lr"] for g in params]))
                return _base(params, *a, **kw)

            stack.callback(setattr, torch.optim, name, base)
            setattr(torch.optim, name, factory)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = pipe(image=[img], mc_algo="mc", generator=torch.manual_seed(2), config=cfg, renderer=renderer,
                       sil_renderer=sil_renderer, enable_pbar=False, **paths)
        os.chdir(cwd)
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    print("\n".join(lines))
    obj, hand = out
    ov, of = obj.verts_packed().detach().numpy().astype(np.float64), obj.faces_packed().numpy()
    arrays = dict(hand_verts=hand.verts_packed().detach().numpy(), hand_faces=hand.faces_packed().numpy(),
                  obj_stats=object_stats(ov, of), obj_counts=np.array([len(ov), len(of)]),
                  scene_checksum=np.array([float(np.abs(sc[k].astype(np.float64)).sum()) for k in
                                           ("hand_verts", "obj_verts", "moge_normal", "moge_disp", "kps_2d", "T_h2m")]),
                  weights_checksum=np.array([float(sum(p.detach().double().abs().sum() for p in m.parameters()))
                                             for m in (net.vae, net.model, net.conditioner)]))
    for n, (kind, params, lrs) in enumerate(made):
        arrays[f"opt{n}_small"] = np.concatenate([p.detach().reshape(-1).float().numpy() for p in params if p.numel() <= 4])
        arrays[f"opt{n}_lrs"] = np.array(lrs)
        big = [p for p in params if p.numel() > 4]
        if big:
            arrays[f"opt{n}_noise"] = big[0].detach().float().numpy()
    np.savez_compressed(os.path.join(OUT, f"ref_pipeline{variant['tag']}.npz"), **arrays)
    with open(os.path.join(OUT, f"ref_pipeline{variant['tag']}.json"), "w") as f:
        json.dump(dict(scene=variant["scene"], radius=RADIUS, schedule=variant["schedule"], config=variant["config"], vae_kw=variant.get("vae_kw", VAE_KW), log=lines, optimizers=[m[0] for m in made],
                       torch=torch.__version__), f, indent=1)
    print("hand", arrays["hand_verts"].shape, "object", arrays["obj_counts"], arrays["obj_stats"])


if __name__ == "__main__":
    main()
