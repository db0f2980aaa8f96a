"""Timing of the guided pipeline (followmyhold_amd/pipeline.py) with stand-in networks: the reference's full schedule
(20 denoising steps, 200 + 100 + 9 x 50 inner iterations, final decode at res 384) at 512 x 512, and the cost split of one
phase-C iteration between the PyTorch networks (latent -> SDF, fwd + bwd) and the HIP part (FlexiCubes, tables, step)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from PIL import Image
from followmyhold_amd import engine as E, ops, pipeline as PLN, standins
from test_pipeline import _scene_for_pipeline, _write, _renderer
import pathlib

sc = _scene_for_pipeline(512, 512)
with tempfile.TemporaryDirectory() as d:
    paths = _write(pathlib.Path(d), sc)
    img = Image.open(paths["cropped_obj_img_path"])
    for name, kw, dt in (("tiny stand-in VAE (64 x 8 latents, width 32)", {}, torch.float32),
                         ("Hunyuan-sized stand-in VAE (3072 x 64 latents, width 1024, 16 heads, 16 layers, fp16)",
                          dict(num_latents=3072, embed_dim=64, width=1024, heads=16, layers=16, num_freqs=8), torch.float16)):
        pipe = standins.make_standin_pipeline(device="cuda", dtype=dt, seed=1, **kw)
        cfg = E.OptimizationConfig()
        if kw:      # keep the big one short: 3 joint steps instead of 9
            cfg.num_inference_steps, cfg.guidance_start_step, cfg.handopt_start_step, cfg.guidance_end_step = 6, 2, 1, 6
            cfg.optimization_steps_scale, cfg.optimization_steps_joint = 20, 20
        for rep in range(2 if not kw else 1):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            obj, hand = pipe(image=[img], generator=torch.manual_seed(2), config=cfg, renderer=_renderer(sc["fov"]),
                             J_regressor=sc["J_regressor"], final_octree_resolution=384 if not kw else 128, **paths)
            torch.cuda.synchronize(); dtm = time.perf_counter() - t0
            print(f"{name}: pipeline {dtm*1e3:.0f} ms, {pipe.stats}, final object {tuple(obj.verts_packed().shape)} verts", flush=True)
        # one phase-C iteration split
        gb = pipe.guidance_batch
        c, _ = E.phase_cfg("C", cfg, denoise_i=5, do_update=True)
        xyz, gsz, _ = PLN.generate_dense_grid_points(np.full(3, -1.1), np.full(3, 1.1), 5, octree_resolution=64)
        xyz = torch.as_tensor(xyz, device="cuda")
        lat = torch.randn(1, *pipe.vae.latent_shape, device="cuda", dtype=dt)
        noise = torch.randn_like(lat).requires_grad_(True)
        def it(split):
            t = {}
            torch.cuda.synchronize(); a = time.perf_counter()
            sdf = PLN.latent2sdf(lat + 0.5 * noise, xyz, gsz, pipe.vae, "cuda")
            torch.cuda.synchronize(); t["latent2sdf fwd"] = time.perf_counter() - a; a = time.perf_counter()
            v, f, _ = ops.flexicubes(xyz, sdf[0].flatten(), 64)
            torch.cuda.synchronize(); t["flexicubes"] = time.perf_counter() - a; a = time.perf_counter()
            loss = gb.objective(v, f, c)
            torch.cuda.synchronize(); t["tables + step"] = time.perf_counter() - a; a = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize(); t["backward (flexi bwd + latent2sdf bwd)"] = time.perf_counter() - a
            return t
        it(0); it(0)
        ts = [it(0) for _ in range(5)]
        print("  one phase-C iteration: " + ", ".join(f"{k} {np.median([x[k] for x in ts])*1e3:.2f} ms" for k in ts[0]), flush=True)
        del pipe; torch.cuda.empty_cache()
