"""Generate golden vectors by importing the REFERENCE's own pure-torch/numpy helpers.

Run in the build container only (needs /root/reference; the GPU box never runs this):
    python tests/golden/make_golden.py
Writes tests/golden/ref_helpers.npz (+ ref_meta.json).  The un-vendored dependencies of the
reference (pytorch3d, kaolin, diffusers, trimesh, cv2, ...) are replaced by empty stub modules,
so only functions whose arithmetic lives in /root/reference itself are exercised (SURVEY.md 8c,
fixtures F1-F9; F10-F12 added for the pipeline helpers).  Nothing from the reference is copied: the outputs are data.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = sys.argv[1] if len(sys.argv) > 1 else HERE          # optional output directory (tests regenerate into a temp dir)


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, n):
        return _Anything()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__getattr__ = lambda n: _Anything  # any other attribute resolves to a dummy class
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        if parent in sys.modules:
            setattr(sys.modules[parent], child, m)
    return m


def install_stubs():
    for n in ["torchvision", "torchvision.transforms", "trimesh", "diffusers", "diffusers.utils",
              "diffusers.utils.torch_utils", "skimage", "skimage.measure", "kiui", "kiui.vis", "cv2",
              "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "pytorch3d.renderer.mesh",
              "pytorch3d.renderer.mesh.textures", "pytorch3d.renderer.mesh.shader", "pytorch3d.ops",
              "pytorch3d.renderer.blending", "pytorch3d.io", "pytorch3d.loss", "pytorch3d.transforms",
              "kaolin", "kaolin.non_commercial", "kaolin.ops", "kaolin.ops.mesh", "kaolin.metrics"]:
        _stub(n)
    sys.modules["pytorch3d.renderer.mesh.shader"].ShaderBase = object

    # minimal stand-ins for the diffusers mixins the scheduler subclasses (SCH:23-25)
    class ConfigMixin:
        pass

    class SchedulerMixin:
        pass

    class BaseOutput(dict):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.__dict__.update(kw)

        def __init_subclass__(cls, **kw):
            pass

    def register_to_config(init):
        def wrapper(self, *a, **kw):
            import inspect
            sig = inspect.signature(init)
            ba = sig.bind(self, *a, **kw)
            ba.apply_defaults()
            cfg = {k: v for k, v in ba.arguments.items() if k != "self"}
            self.config = types.SimpleNamespace(**cfg)
            init(self, *a, **kw)
        return wrapper

    cu = _stub("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    su = _stub("diffusers.schedulers", )
    _stub("diffusers.schedulers.scheduling_utils", SchedulerMixin=SchedulerMixin)
    lg = types.SimpleNamespace(get_logger=lambda n: types.SimpleNamespace(warning=print, info=print))
    sys.modules["diffusers.utils"].BaseOutput = BaseOutput
    sys.modules["diffusers.utils"].logging = lg


def load_reference():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "third_party"))
    sys.path.insert(0, os.path.join(REF, "src"))
    spec = importlib.util.spec_from_file_location(
        "ref_pipelines", os.path.join(REF, "third_party_patches/hy3dgen/shapegen/pipelines.py"))
    PL = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(PL)
    spec = importlib.util.spec_from_file_location(
        "ref_schedulers", os.path.join(REF, "third_party_patches/hy3dgen/shapegen/schedulers.py"))
    SCH = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(SCH)
    from utilz.code_utils import get_guidance_params
    from foho.configs.guid_config import OptimizationConfig
    return PL, SCH, get_guidance_params, OptimizationConfig


class FakeMesh:
    """Exposes the handful of Meshes methods the reference helpers touch."""

    def __init__(self, verts):
        self.v = verts

    def verts_padded(self):
        return self.v.unsqueeze(0)

    def verts_packed(self):
        return self.v

    def update_padded(self, v):
        return FakeMesh(v.squeeze(0))


class FakeRenderer:
    def __init__(self, rgba, zbuf):
        self.rgba, self.z = rgba, zbuf
        self.rasterizer = lambda mesh: types.SimpleNamespace(zbuf=self.z.clone())

    def __call__(self, mesh):
        return self.rgba.clone()


def main():
    PL, SCH, get_guidance_params, OptimizationConfig = load_reference()
    g = torch.Generator().manual_seed(1234)
    out, meta = {}, {}

    # F1 normal_alignment_loss (PL:178-186)
    a = torch.randn(1, 16, 16, 3, generator=g)
    b = torch.randn(1, 16, 16, 3, generator=g)
    m = torch.rand(1, 16, 16, generator=g) > 0.4
    out.update(f1_a=a, f1_b=b, f1_mask=m, f1_loss_mask=PL.normal_alignment_loss(a, b, m),
               f1_loss_nomask=PL.normal_alignment_loss(a, b))

    # F2 intersection losses (PL:204-239)
    sh = torch.randn(4096, generator=g)
    so = torch.randn(4096, generator=g)
    out.update(f2_sdf_hand=sh, f2_sdf_obj=so, f2_honerf=PL.honerf_intersection_loss(sh, so),
               f2_safe=PL.safe_intersection_loss(sh, so))
    meta["f2_honerf_requires_grad"] = bool(PL.honerf_intersection_loss(sh.requires_grad_(True), so).requires_grad)

    # F3 dense grid (PL:341-360)
    xyz, gs, length = PL.generate_dense_grid_points(np.array([-1.1] * 3), np.array([1.1] * 3), 5, "ij", 64)
    out.update(f3_first=xyz[:70], f3_last=xyz[-70:], f3_grid_size=np.array(gs), f3_length=length,
               f3_shape=np.array(xyz.shape), f3_checksum=np.array([xyz.astype(np.float64).sum(), (xyz.astype(np.float64) ** 2).sum()]))
    xyz2, _, _ = PL.generate_dense_grid_points(np.array([-0.3, 0.1, -0.7], np.float32), np.array([0.4, 0.35, -0.2], np.float32), 5, "ij", 8)
    out.update(f3_small=xyz2)

    # F4 render_normal_and_disparity with a fake renderer (PL:272-289)
    rgba = torch.rand(1, 32, 32, 4, generator=g) * 2 - 0.5
    hit = torch.rand(1, 32, 32, generator=g) > 0.5
    rgba[..., 3] = hit.float()
    rgba[..., :3] = torch.where(hit[..., None], rgba[..., :3], torch.ones_like(rgba[..., :3]))
    z = torch.where(hit, 0.3 + torch.rand(1, 32, 32, generator=g), torch.full((1, 32, 32), -1.0)).unsqueeze(-1)
    nn_, dd_ = PL.render_normal_and_disparity(FakeRenderer(rgba, z), None)
    out.update(f4_rgba=rgba, f4_zbuf=z, f4_normal=nn_, f4_disp=dd_)

    # F5 transforms + keypoints (PL:95-135, PL:242-250)
    v = torch.randn(778, 3, generator=g) * 0.05
    T = torch.eye(4)
    q = torch.randn(4, generator=g)
    q = q / q.norm()
    r, i, j, k = q.tolist()
    T[:3, :3] = torch.tensor([[1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r)],
                              [2 * (i * j + k * r), 1 - 2 * (i * i + k * k), 2 * (j * k - i * r)],
                              [2 * (i * k - j * r), 2 * (j * k + i * r), 1 - 2 * (i * i + j * j)]])
    T[:3, 3] = torch.randn(3, generator=g) * 0.1
    scale = torch.tensor([1.3])
    jr = torch.rand(16, 778, generator=g)
    jr = jr / jr.sum(1, keepdim=True)
    out.update(f5_verts=v, f5_T=T, f5_scale=scale, f5_jreg=jr,
               f5_center=PL.transform_mesh_around_center(FakeMesh(v), T).v,
               f5_center_scale=PL.transform_mesh_around_center_w_scale(FakeMesh(v), T, scale).v,
               f5_h2m=PL.transform_hunyuan2moge(FakeMesh(v), T).v,
               f5_kps=PL.mano_vert_to_3dkps(FakeMesh(v), jr, "cpu"))

    # F6 get_guidance_params (GP:3-83)
    cfg = OptimizationConfig()
    noise = torch.randn(1, 8, 4, generator=g).half()
    base = dict(noise_pred_obj=noise, scale_hand=torch.tensor([1.0]), trans_hand=torch.zeros(3),
                rotation_hand=torch.tensor([1.0, 0, 0, 0]), device="cpu", phase1_hand_lrs=cfg.phase1_hand_lrs,
                phase2_hand_lrs=cfg.phase2_hand_lrs, noise_obj_lr1=cfg.noise_obj_lr1, noise_obj_lr2=cfg.noise_obj_lr2,
                obj_lrs=cfg.obj_lrs, obj_2half_lrs=cfg.obj_2half_lrs, scale_obj=torch.tensor([1.0]),
                trans_obj=torch.zeros(3), rotation_obj=torch.tensor([1.0, 0, 0, 0]))
    f6 = {}
    for phase in (1, 1.5, 2):
        res = get_guidance_params(phase, **base)
        groups = res[0]
        names = ["noise_pred_obj", "scale_hand", "trans_hand", "rotation_hand", "scale_obj", "trans_obj", "rotation_obj"]
        f6[str(phase)] = dict(
            lrs=[gp["lr"] for gp in groups], shapes=[list(gp["params"][0].shape) for gp in groups],
            dtypes=[str(gp["params"][0].dtype) for gp in groups],
            requires_grad={n: bool(t.requires_grad) for n, t in zip(names, res[1:])},
            is_same_object={n: bool(t is base[n]) for n, t in zip(names, res[1:])})
    try:
        get_guidance_params(3, **base)
        f6["bad_phase_raises"] = False
    except ValueError:
        f6["bad_phase_raises"] = True
    meta["f6"] = f6

    # F7 OptimizationConfig (CFG:6-32)
    meta["f7"] = {k: v for k, v in vars(cfg).items()}
    meta["f7_call_returns_self"] = bool(cfg() is cfg)

    # F8 scheduler (SCH:171-211, 235-318, 411-493) via retrieve_timesteps (PL:363-419)
    sch = SCH.FlowMatchEulerDiscreteScheduler()
    sig_in = np.linspace(0, 1, 20)
    ts, n = PL.retrieve_timesteps(sch, 20, "cpu", sigmas=sig_in)
    lat = torch.randn(1, 8, 4, generator=g).half()
    vel = [torch.randn(1, 8, 4, generator=g).half() for _ in range(3)]
    f8 = dict(f8_timesteps=ts, f8_sigmas=sch.sigmas, f8_lat0=lat, f8_vel=torch.stack(vel))
    meta["f8_config"] = {k: v for k, v in vars(sch.config).items()}
    x = lat
    prevs, finals_before, finals_after = [], [], []
    for s in range(3):
        t = ts[s]
        finals_before.append(sch.step_final(vel[s], t, x))  # what the inner loops see (PL:1391, PL:1507)
        x = sch.step(vel[s], t, x).prev_sample            # PL:1612
        prevs.append(x)
        finals_after.append(sch.step_final(vel[s], t, x))  # the "debug" decode after step (PL:1621)
    f8.update(f8_prev=torch.stack(prevs), f8_final_before=torch.stack(finals_before), f8_final_after=torch.stack(finals_after))
    out.update(f8)

    # compute_loss_stable_fp32 (PL:1001-1018)
    lt = {"a": torch.tensor(1.5), "b": torch.tensor(float("nan")), "c": torch.tensor(0.25, dtype=torch.float16), "d": None}
    out["f10_stable_sum"] = PL.compute_loss_stable_fp32(lt)

    # F11 latent2sdf (PL:292-313) with an arithmetic stand-in for the VAE (the test rebuilds it from f11_w): rescaling by
    # 1/scale_factor, the vae() call, fp16 queries in chunks of 8000, concatenation, (1,G,G,G) view, float32, negation
    class FakeVAE:
        scale_factor = 0.7

        def __init__(self, w):
            self.w = w

        def __call__(self, x):
            return x * 2 + 1

        def geo_decoder(self, queries, latents):
            return (queries.float() @ self.w + latents.float().mean())[..., :1].to(latents.dtype)

    w11 = torch.randn(3, 2, generator=g)
    lat11 = torch.randn(1, 16, 4, generator=g).half()
    xyz11, gs11, _ = PL.generate_dense_grid_points(np.array([-1.1] * 3), np.array([1.1] * 3), 5, "ij", 24)   # 15625 points: 2 chunks
    out.update(f11_w=w11, f11_latent=lat11,
               f11_sdf=PL.latent2sdf(lat11, torch.FloatTensor(xyz11), gs11, FakeVAE(w11), "cpu"))

    # F12 encode_cond (PL:599-639): classifier-free-guidance batches are [cond, uncond] (and [cond, uncond-with-additional,
    # uncond] for dual guidance), cast to the pipeline dtype
    class FakeCond:
        def __call__(self, image=None, mask=None):
            return {"main": image.mean(dim=(2, 3)).unsqueeze(1).float(), "additional": {"x": mask.float().sum(dim=(1, 2, 3)).reshape(-1, 1)}}

        def unconditional_embedding(self, bsz):
            return {"main": torch.zeros(bsz, 1, 3), "additional": {"x": -torch.ones(bsz, 1)}}

    refpipe = object.__new__(PL.Hunyuan3DDiTPipeline)
    refpipe.conditioner, refpipe.dtype = FakeCond(), torch.float16
    img12 = torch.rand(1, 3, 8, 8, generator=g)
    msk12 = (torch.rand(1, 1, 8, 8, generator=g) > 0.5).float()
    c_plain = refpipe.encode_cond(img12, msk12, False, False)
    c_cfg = refpipe.encode_cond(img12, msk12, True, False)
    c_dual = refpipe.encode_cond(img12, msk12, True, True)
    out.update(f12_image=img12, f12_mask=msk12, f12_plain_main=c_plain["main"], f12_cfg_main=c_cfg["main"],
               f12_cfg_add=c_cfg["additional"]["x"], f12_dual_main=c_dual["main"], f12_dual_add=c_dual["additional"]["x"])
    meta["f12_dtypes"] = dict(plain=str(c_plain["main"].dtype), cfg=str(c_cfg["main"].dtype), dual=str(c_dual["additional"]["x"].dtype))

    # F13 alignment drivers (src/foho/alignment/h2m.py:12-55, mano.py:12-44): which files are paired and with which
    # align_meshes_impl arguments, on a small directory tree (the call itself is captured, trimesh is not needed)
    import tempfile
    calls = []
    fake = types.ModuleType("foho.alignment.mesh_align")
    fake.align_meshes_impl = lambda **kw: calls.append(kw)
    sys.modules["foho.alignment.mesh_align"] = fake
    mods = {}
    for name in ("h2m", "mano"):
        spec = importlib.util.spec_from_file_location(f"ref_{name}", os.path.join(REF, "src/foho/alignment", f"{name}.py"))
        mods[name] = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mods[name])
    with tempfile.TemporaryDirectory() as root:
        tree = ["hy/3_hoi_mesh.ply", "hy/12_hoi_mesh.ply", "hy/7_hoi_mesh.ply", "hy/9_hoi_mesh.ply", "hy/notes.txt",
                "moge/3_cropped_hoi/mesh.ply", "moge/3_cropped_hoi/pointcloud.ply", "moge/12_cropped_hoi/pointcloud.ply",
                "moge/12_cropped_hoi/mesh.glb", "moge/7_cropped_hoi/mesh.glb", "moge/9_cropped_hoi/other.txt",
                "hamer/3_hamer.obj", "hamer/12_hamer.obj", "hamer/3_hamer.ply"]
        for rel in tree:
            os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True)
            open(os.path.join(root, rel), "w").close()
        rel_ = lambda v: os.path.relpath(v, root) if isinstance(v, str) else v
        mods["h2m"].run(os.path.join(root, "hy"), os.path.join(root, "moge"), os.path.join(root, "rt"))
        h2m_calls = sorted(({k: rel_(v) for k, v in c.items()} for c in calls), key=lambda c: c["source_mesh_path"])
        calls.clear()
        mods["mano"].run(os.path.join(root, "hamer"), os.path.join(root, "hy"), os.path.join(root, "aligned"))
        mano_calls = sorted(({k: rel_(v) for k, v in c.items()} for c in calls), key=lambda c: c["source_mesh_path"])
    meta["f13"] = dict(tree=tree, h2m=h2m_calls, mano=mano_calls)

    # F14 the mesh_align command line (ICP:219-262, a click command): option names, short forms, types and defaults as seen
    # by align_meshes_impl (captured; trimesh / pyvista stubbed)
    for n in ("pyvista", "trimesh.registration", "trimesh.proximity"):
        _stub(n)
    del sys.modules["foho.alignment.mesh_align"]
    spec = importlib.util.spec_from_file_location("ref_mesh_align", os.path.join(REF, "src/foho/alignment/mesh_align.py"))
    MA = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(MA)
    from click.testing import CliRunner
    got = []
    MA.align_meshes_impl = lambda *a: got.append(list(a))
    with tempfile.TemporaryDirectory() as root:
        a_, b_ = os.path.join(root, "a.ply"), os.path.join(root, "b.ply")
        open(a_, "w").close(), open(b_, "w").close()
        argvs = [["A", "B"],
                 ["A", "B", "-tp", "T", "-tmp", "M.ply", "-fs", "-o", "0.35", "-trot", "-tref", "-os", "-ir", "7", "-csr", "11", "-ctr",
                  "13", "-if", "17", "-csf", "19", "-ctf", "23", "-mis", "0.9", "-mas", "1.5"],
                 ["A", "B", "--transform_path", "T2", "--outliers", "0.1", "--iterations_fine", "3", "--max_scale", "2.0"]]
        for av in argvs:
            r = CliRunner().invoke(MA.align_meshes, [a_ if x == "A" else b_ if x == "B" else x for x in av])
            assert r.exit_code == 0, r.output
        got = [["A" if x == a_ else "B" if x == b_ else x for x in g_] for g_ in got]
    meta["f14"] = dict(argv=argvs, args=got)

    # F15 the guidance stage driver (RUN:178-261): task list, per-image file names, skip rules (outputs exist, empty mask),
    # fov.json, the keyword arguments handed to run_hunyuan_w_guid (captured; cv2.imread stood in by PIL)
    from PIL import Image

    def imread(path, flag=None):
        return np.array(Image.open(path)) if os.path.exists(path) else None

    for n in ("hy3dgen", "hy3dgen.rembg", "hy3dgen.shapegen", "hy3dgen.shapegen.pipelines", "hy3dgen.shapegen.postprocessors"):
        _stub(n)
    sys.modules["cv2"].imread = imread
    sys.modules["cv2"].IMREAD_UNCHANGED = -1
    spec = importlib.util.spec_from_file_location("ref_run", os.path.join(REF, "src/foho/guidance/run.py"))
    RUN = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(RUN)
    seen = []
    with tempfile.TemporaryDirectory() as root:
        d = {k: os.path.join(root, k) for k in ["cropped_obj_img_dir", "mask_dir", "moge_out_dir", "hunyuan_hoi_mesh_dir",
                                                "hamer_out_dir", "h2m_rt_dir", "aligned_mano_dir", "guidance_out_dir"]}
        for v in d.values():
            os.makedirs(v)
        imgs = ["12_cropped_hoi_1.png", "3_cropped_hoi_0.png", "7_cropped_hoi_1.png", "9_cropped_hoi_1.png", "21_cropped_hoi_0.png"]
        for name in imgs:
            idx = name.split("_")[0]
            Image.fromarray(np.zeros((4, 4, 3), np.uint8)).save(os.path.join(d["cropped_obj_img_dir"], name))
            os.makedirs(os.path.join(d["moge_out_dir"], f"{idx}_cropped_hoi"))
            if idx != "21":     # image 21 has no fov.json: the reference's try/except moves on
                with open(os.path.join(d["moge_out_dir"], f"{idx}_cropped_hoi", "fov.json"), "w") as f:
                    json.dump({"fov_x": 40.0 + int(idx)}, f)
            hand = np.full((4, 4), 255, np.uint8) if idx != "7" else np.zeros((4, 4), np.uint8)   # image 7: empty hand mask
            Image.fromarray(hand).save(os.path.join(d["mask_dir"], f"{idx}_cropped_hand_mask.png"))
            Image.fromarray(np.full((4, 4), 255, np.uint8)).save(os.path.join(d["mask_dir"], f"{idx}_cropped_obj_mask.png"))
        for tag in ("obj", "hand"):                                                          # image 3: outputs exist
            open(os.path.join(d["guidance_out_dir"], f"3_{tag}.ply"), "w").close()
        rel_ = lambda v: os.path.relpath(v, root) if isinstance(v, str) else v

        def fake_guid(**kw):
            seen.append({k: rel_(v) for k, v in kw.items() if k != "config"})
            return (None, None) if "9_" in kw["cropped_obj_img_path"] else (1, 1)

        orig_guid = RUN.run_hunyuan_w_guid
        RUN.run_hunyuan_w_guid = fake_guid
        os.environ.pop("SLURM_ARRAY_TASK_ID", None)
        RUN.run(project_root=root, task_list_file=None, **d)
        order_all = [c["cropped_obj_img_path"] for c in seen]
        tl = os.path.join(root, "tasks.json")
        with open(tl, "w") as f:
            json.dump([["12_cropped_hoi_1.png"], ["9_cropped_hoi_1.png", "3_cropped_hoi_0.png"]], f)
        os.environ["SLURM_ARRAY_TASK_ID"] = "1"
        n0 = len(seen)
        RUN.run(project_root=root, task_list_file=tl, **d)
        os.environ.pop("SLURM_ARRAY_TASK_ID")
        listing = RUN._load_task_list(None, d["cropped_obj_img_dir"])
    meta["f15"] = dict(images=imgs, calls=sorted(seen[:n0], key=lambda c: c["cropped_obj_img_path"]), n_calls=n0,
                       task_list_calls=[c["cropped_obj_img_path"] for c in seen[n0:]],
                       listing_is_sorted=bool(listing == sorted(listing)), listing_set=sorted(listing))

    # F16 run_hunyuan_w_guid (RUN:65-175) with recording stand-ins for the pytorch3d / hy3dgen classes: camera, blend and
    # raster settings of the two renderers, the RGBA image handed to the pipeline (white made transparent), the pipeline's
    # keyword arguments, the post-processing order and the export targets
    log = []

    class Rec:
        def __init__(self, *a, **kw):
            self.args, self.kw = a, kw
            log.append((type(self).__name__, kw))

    def rec(name):
        return type(name, (Rec,), {})

    names = ["FoVPerspectiveCameras", "BlendParams", "RasterizationSettings", "MeshRenderer", "MeshRasterizer",
             "SoftSilhouetteShader", "PhongNormalShader"]
    K = {n: rec(n) for n in names}
    K["BlendParams"] = type("BlendParams", (Rec,), {"sigma": property(lambda self: self.kw["sigma"])})
    for n in names:
        setattr(RUN, n, K[n])

    class FakeMeshes:
        def __init__(self, v, f):
            self.v, self.f = v, f

        def verts_packed(self):
            return self.v

        def faces_packed(self):
            return self.f

    pipe_calls = []

    class FakePipe:
        @classmethod
        def from_pretrained(cls, path):
            log.append(("from_pretrained", {"path": path}))
            return cls()

        def __call__(self, **kw):
            pipe_calls.append(kw)
            return FakeMeshes(torch.rand(5, 3), torch.tensor([[0, 1, 2]])), FakeMeshes(torch.rand(4, 3), torch.tensor([[0, 1, 3]]))

    class FakeTrimesh:
        def __init__(self, vertices=None, faces=None):
            self.vertices, self.faces = vertices, faces

        def export(self, path):
            log.append(("export", {"path": path, "n_vertices": len(self.vertices)}))

    def post(name):
        class P:
            def __call__(self, mesh):
                log.append((name, {}))
                return mesh
        return P

    class FakeIO:
        def save_mesh(self, mesh, path):
            log.append(("IO.save_mesh", {"path": path}))

    RUN.Hunyuan3DDiTFlowMatchingPipeline_main = FakePipe
    RUN.trimesh = types.SimpleNamespace(Trimesh=FakeTrimesh)
    RUN.FloaterRemover, RUN.DegenerateFaceRemover, RUN.FaceReducer = post("FloaterRemover"), post("DegenerateFaceRemover"), post("FaceReducer")
    RUN.IO = FakeIO
    RUN.BackgroundRemover = lambda: (lambda im: im)
    sys.modules["cv2"].IMREAD_GRAYSCALE = 0
    with tempfile.TemporaryDirectory() as root:
        rgb = (np.arange(6 * 5 * 3).reshape(6, 5, 3) * 3 % 256).astype(np.uint8)
        rgb[1, 2] = 255
        rgb[4, 0] = 255
        rgb[3, 3] = (255, 255, 254)
        img_path = os.path.join(root, "12_cropped_hoi_1.png")
        Image.fromarray(rgb).save(img_path)
        hm = os.path.join(root, "hand.png")
        Image.fromarray(np.full((6, 5), 255, np.uint8)).save(hm)
        res = orig_guid(
            cropped_obj_img_path=img_path, fovx=41.5, hamer_for_guid_path="K", aligned_mano_mesh_path="MANO",
            cropped_obj_mask_path="OM", cropped_hand_mask_path=hm, moge_mesh_path="MOGE", T_h2m_path="T",
            hunyuan_hoi_mesh_path="HY", save_path_obj=os.path.join(root, "o.ply"), save_path_hand=os.path.join(root, "h.ply"),
            config="CFG", device="cpu")
        kw = pipe_calls[0]
        image16 = np.array(kw["image"][0])
        rel_ = lambda v: os.path.relpath(v, root) if isinstance(v, str) and v.startswith(root) else v

        def plain(v):
            if isinstance(v, torch.Tensor):
                return v.tolist()
            if isinstance(v, (np.floating, np.integer)):
                return v.item()
            if isinstance(v, Rec):
                return type(v).__name__
            if isinstance(v, tuple):
                return [plain(x) for x in v]
            return rel_(v)

        meta["f16"] = dict(
            log=[[n, {k: plain(v) for k, v in kw_.items()}] for n, kw_ in log],
            pipeline_kwargs={k: plain(v) for k, v in kw.items() if k not in ("image", "generator", "renderer", "sil_renderer")},
            pipeline_kwarg_names=sorted(kw.keys()), image_mode=kw["image"][0].mode, n_images=len(kw["image"]),
            generator_seed=int(kw["generator"].initial_seed()), returns_pair=bool(isinstance(res, tuple) and len(res) == 2))
        out.update(f16_rgb=rgb, f16_image=image16)

    np.savez_compressed(os.path.join(OUT, "ref_helpers.npz"),
                        **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    with open(os.path.join(OUT, "ref_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True, default=str)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
