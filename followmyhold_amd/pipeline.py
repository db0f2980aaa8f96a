"""Guided flow-matching shape pipeline: the `__call__` of the reference's patched
`Hunyuan3DDiTFlowMatchingPipeline_main` (third_party_patches/hy3dgen/shapegen/pipelines.py:1041-1679; PL below) with the
optimisation-in-the-loop arithmetic on the MI355X kernels.

What stays PyTorch-ROCm ("host code stays Python"): the DiT (`model`), the ShapeVAE (`vae`: latent ->
transformer -> cross-attention geometry decoder), the image conditioner and the flow-matching scheduler.  They are
constructor arguments with the interfaces the reference uses (PL:563-742); `from_hy3dgen` adopts the components of an
installed Hunyuan3D-2 pipeline object.  What runs on hand-written HIP: FlexiCubes (`foho_flexi_fwd/_bwd`), the topology
tables of the mesh it emits (`foho_topology_tables`), and the whole guidance iteration -- transforms, three renders,
keypoints, nearest neighbours, edge loss, inside count, every loss head, their backward and the Adam/AdamW update of the
14 similarity parameters (`foho_step_run`).  Per inner iteration the only torch autograd left is
latent -> VAE -> SDF, which receives dL/dSDF from the FlexiCubes backward kernel and carries it to `noise_pred_obj`
(PL:1507-1509, PL:1600-1601).

Differences from the reference, all outside the arithmetic:
  * `renderer` / `sil_renderer` only supply the camera (fov); the fused step renders itself.
  * debug dumps (`FOHO_DEBUG_DIR`, PL:1076-1091, 1664-1675) write losses.txt / params.json and the final meshes; the
    matplotlib grids are not produced.
  * the 14 similarity parameters live in the GuidanceBatch (device memory) instead of seven leaf tensors; each phase gets
    a fresh optimiser state exactly like the reference's per-step `torch.optim.Adam/AdamW(...)` (PL:1318, 1384, 1478).
    `noise_pred_obj` keeps its own torch AdamW with the same hyper-parameters (Adam is element-wise, so splitting one
    optimiser into two changes nothing).
"""
import datetime
import json
import os
from typing import Optional

import numpy as np
import torch

from . import engine as E
from . import inputs, ops
from .facade import Meshes, TexturesVertex, generate_dense_grid_points, quaternion_to_matrix
from .scheduler import retrieve_timesteps


def vae_attention_backend():
    """Context for the ShapeVAE transformer's forward (and the backward recorded under it).  At the transformer's shape -- (1, 16 heads,
    3072 tokens, 64) fp16, sixteen layers, run and back-propagated in every inner iteration (PL:295, 1391-1393, 1507-1509) -- torch's
    scaled_dot_product_attention is a quarter of an iteration: forward + backward per layer on an MI355X 509-548 us with ROCm's default
    (flash, AOTriton) backend, 350 us with the memory-efficient one (139 forward, 212 backward: `vae_attention` bench record).
    This is the FALLBACK route: with `vae_transformer.install(vae)` (what `from_hy3dgen` does) the whole transformer runs on foho_vae_fwd / _bwd
    and no torch attention is called.  FOHO_VAE_SDPA selects, for a VAE that stays on its torch module:
      hip (default)  forward AND backward on this package's attention kernels (followmyhold_amd.sdpa: 71 us forward, operands read where the
                     projections left them; backward 190 us, dK / dV written directly); calls the kernels do not take (a mask, another head size)
                     and everything outside this context stay torch's, with the memory-efficient backend preferred;
      hip_torch_bwd  the HIP forward with torch's memory-efficient backward fed from it (a private torch operator: 181 us; round 5's default);
      hip_bwd        = hip (the name of round 5);
      efficient      torch's memory-efficient backend first, the others allowed behind it;
      default        torch's own choice, as the reference runs."""
    import contextlib
    mode = os.environ.get("FOHO_VAE_SDPA", "hip")
    if mode not in ("efficient", "hip", "hip_bwd", "hip_torch_bwd") or not torch.cuda.is_available():
        return contextlib.nullcontext()
    stack = contextlib.ExitStack()
    try:
        from torch.nn.attention import SDPBackend, sdpa_kernel
        stack.enter_context(sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION, SDPBackend.FLASH_ATTENTION, SDPBackend.MATH], set_priority=True))
    except (ImportError, TypeError):       # an older torch without the priority form: its own choice
        pass
    if mode != "efficient":
        from . import sdpa
        stack.enter_context(sdpa.hip_sdpa(backward="torch" if mode == "hip_torch_bwd" else "hip"))
    return stack


def vae_tokens(vae, latents):
    """`vae(latents)` of PL:295 (ShapeVAE.forward: post_kl -> transformer).  With `vae_transformer.install(vae)` the transformer -- and, under
    autograd, its backward to the latents (PL:1391-1393, 1507-1509) -- runs on this package's kernels (`foho_vae_fwd/_bwd`); inputs they do
    not take (another dtype, a token count that is not a multiple of 128) and VAEs without it go through the torch module, with the
    attention backend of `vae_attention_backend()`."""
    tr = getattr(vae, "hip_transformer", None)
    if tr is not None:
        x0 = vae.post_kl(latents)
        if tr.accepts(x0):
            return tr(x0)
    with vae_attention_backend():
        return vae(latents)


def latent2sdf(pred, xyz_samples, grid_size, vae, device, num_chunks=8000):
    """PL:292-338 (return_mesh=False): rescale the latent, run the VAE transformer, query the geometry decoder in chunks of
    8000 grid points, negate the logits so that the field is negative inside.  -> (1, G, G, G) float32."""
    pred = 1 / vae.scale_factor * pred
    pred = vae_tokens(vae, pred)
    hip = getattr(vae, "hip_geo", None)          # geo_decode.install(vae): the decoder on the matrix cores
    if hip is not None:
        # all grid points in one call, no 8000-query chunks; under autograd (PL:1391-1393, 1507-1509) the gradient reaches
        # `pred` through foho_geo_decode_bwd
        grid_logits = hip(hip.grid_queries(xyz_samples), pred)      # fp16 query points like PL:303; the query side is cached per grid
        return -grid_logits.view((1, grid_size[0], grid_size[1], grid_size[2])).float()
    logits = []
    for start in range(0, xyz_samples.shape[0], num_chunks):
        queries = xyz_samples[start:start + num_chunks].to(device).half()      # fp16 whatever the VAE's dtype (PL:303)
        logits.append(vae.geo_decoder(queries.unsqueeze(0), pred))
    grid_logits = torch.cat(logits, dim=1)
    return -grid_logits.view((1, grid_size[0], grid_size[1], grid_size[2])).float()


def _bound_active_rows(vae, n_rows):
    """The decoder's active-row backward (foho_geo_decode_bwd_rows) launches row blocks up to an upper bound of the rows that carry a
    gradient; every block beyond the actual count is fourteen empty launches (4.5 us each).  In the guidance loop the gradient comes
    out of the FlexiCubes backward, which has run by the time the iteration's flags are read back: `n_rows` = the exact count
    (SdfObjective.active_rows(), or a surface's face count on the exact-size path: one quad per crossed edge).  Returns the decoder
    (or None) so that the caller can check `rows_dropped` when the phase is over -- a safety net that never fires with an exact count."""
    hip = getattr(vae, "hip_geo", None)
    if hip is not None:
        hip.row_cap = max(int(n_rows), 1)
    return hip


def _check_rows_dropped(hip):
    """End of a phase: did ANY backward of it (every iteration, every image of a batch) drop rows?  Also lifts the bound again -- `row_cap`
    is state on the shared decoder, and a backward outside the loop (a dense gradient from user code) must not inherit the last
    iteration's count."""
    if hip is not None:
        cap, hip.row_cap = hip.row_cap, None
        dropped = hip.take_rows_dropped()
        if dropped:
            raise E.L.FohoError(f"geometry decoder backward: {dropped} active rows exceeded row_cap (last: {cap}) during this phase: its gradient "
                                "is incomplete (raise obj_capacity or leave vae.hip_geo.row_cap = None)")


def similarity_about_center(verts, scale, quat, trans):
    """transform_mesh_around_center_w_scale (PL:108-118): scale and rotate about the bounding-box centre, then shift."""
    center = (verts.min(dim=0)[0] + verts.max(dim=0)[0]) / 2.0
    R = quaternion_to_matrix(quat).float().reshape(3, 3)
    return (scale * (verts - center)) @ R.T + center + trans


def _cat_recursive(*parts, dtype):
    if isinstance(parts[0], torch.Tensor):
        return torch.cat(parts, dim=0).to(dtype)
    return {k: _cat_recursive(*[p[k] for p in parts], dtype=dtype) for k in parts[0].keys()}


class BatchLeftFastPath(RuntimeError):
    """An image of a `call_batch` batch needs the per-image handling of `__call__` (a non-manifold or over-capacity iso-surface, a
    NaN loss).  `call_batch` does not raise it: the image's entry of the result list IS an instance (with `.phase`, `.step`,
    `.iteration`, `.flags`), the other images of the batch carry on, and the caller re-runs that image through `__call__`."""

    def __init__(self, msg, phase=None, step=None, iteration=None, flags=0):
        super().__init__(msg)
        self.phase, self.step, self.iteration, self.flags = phase, step, iteration, flags


class GuidedShapePipeline:
    """Drop-in for `Hunyuan3DDiTFlowMatchingPipeline_main`: same constructor (PL:563-585), same `__call__` signature and
    return value (PL:1044-1072, 1679)."""

    def __init__(self, vae, model, scheduler, conditioner, image_processor, device="cuda", dtype=torch.float16, **kwargs):
        self.vae, self.model, self.scheduler = vae, model, scheduler
        self.conditioner, self.image_processor = conditioner, image_processor
        # The guidance optimises the noise prediction and fourteen pose parameters (PL:1318, 1384, 1478), never a network weight: with
        # the weights' requires_grad left at torch's default the backward of every inner iteration would also form d loss / d W of
        # the whole ShapeVAE transformer -- as much matrix work again as the gradient that is wanted -- into .grad buffers nobody reads.
        for net in (vae, model, conditioner):
            if isinstance(net, torch.nn.Module):
                net.requires_grad_(False)
        self.to(device, dtype)

    @classmethod
    def from_hy3dgen(cls, pipe, hip_geo_decoder=True, hip_vae_transformer=True):
        """Adopt the networks of an (unpatched) hy3dgen `Hunyuan3DDiTFlowMatchingPipeline` object.  The ShapeVAE's geometry
        decoder -- the 65^3-point decode and its backward inside every inner iteration, PL:292-313, 1391-1393, 1507-1509 -- is
        taken over by the matrix-core kernels (`geo_decode.install`), and so is the transformer in front of it (`vae(pred)`, PL:295:
        `vae_transformer.install`); a module outside the shapes / layouts they take raises `FohoError` here,
        `hip_geo_decoder=False` / `hip_vae_transformer=False` keep the torch modules."""
        from .scheduler import FlowMatchEulerDiscreteScheduler
        sch = FlowMatchEulerDiscreteScheduler(num_train_timesteps=pipe.scheduler.config.num_train_timesteps,
                                              shift=getattr(pipe.scheduler.config, "shift", 1.0))
        self = cls(pipe.vae, pipe.model, sch, pipe.conditioner, pipe.image_processor, device=pipe.device, dtype=pipe.dtype)
        if hip_geo_decoder:
            from . import geo_decode
            geo_decode.install(self.vae, device=self.device)
        if hip_vae_transformer:
            from . import vae_transformer
            vae_transformer.install(self.vae, device=self.device)
        return self

    def to(self, device=None, dtype=None):
        if device is not None:
            self.device = torch.device(device)
            for m in (self.vae, self.model, self.conditioner):
                m.to(device)
        if dtype is not None:
            self.dtype = dtype
            for m in (self.vae, self.model, self.conditioner):
                m.to(dtype=dtype)

    # ------------------------------------------------------------------ PL:599-742
    def encode_cond(self, image, mask, do_classifier_free_guidance, dual_guidance, to_cpu=False):
        bsz = image.shape[0]
        cond = self.conditioner(image=image, mask=mask)
        if do_classifier_free_guidance:
            un_cond = self.conditioner.unconditional_embedding(bsz)
            if dual_guidance:
                drop_main = dict(un_cond)
                drop_main["additional"] = cond["additional"]
                cond = _cat_recursive(cond, drop_main, un_cond, dtype=self.dtype)
            else:
                cond = _cat_recursive(cond, un_cond, dtype=self.dtype)
        if to_cpu:
            self.conditioner.to("cpu")
        return cond

    def prepare_latents(self, batch_size, dtype, device, generator, latents=None):
        shape = (batch_size, *self.vae.latent_shape)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective "
                             f"batch size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            gdev = generator.device if isinstance(generator, torch.Generator) else device
            latents = torch.randn(shape, generator=generator if isinstance(generator, torch.Generator) else None,
                                  device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device)
        return latents * getattr(self.scheduler, "init_noise_sigma", 1.0)

    def prepare_image(self, image, hand_mask=None):
        if isinstance(image, str) and not os.path.exists(image):
            raise FileNotFoundError(f"Couldn't find image at path {image}")
        images, masks = [], []
        for img in image if isinstance(image, list) else [image]:
            out = self.image_processor(img, return_mask=True)
            im, mk = (out.get("image"), out.get("mask")) if isinstance(out, dict) else out
            images.append(im)
            masks.append(mk)
        images = torch.cat(images, dim=0).to(self.device, dtype=self.dtype)
        masks = torch.cat(masks, dim=0).to(self.device, dtype=self.dtype) if masks[0] is not None else None
        return images, masks

    # ------------------------------------------------------------------ B images through one pass of the schedule
    @torch.no_grad()
    def call_batch(self, images, paths, generators=None, guidance_scale=7.5, num_chunks=8000, config=None, renderer=None,
                   J_regressor=None, guidance_octree_resolution=64, final_octree_resolution=384, obj_capacity=None, fovs=None):
        """`__call__` for B images at once ("within a GPU, batch the rank's images through each kernel
        launch").  The reference runs its images one after the other (RUN:208-259, batch_size = 1, guid_config.py:9); here
        one pass of the 20-step schedule serves all of them: the DiT and the ShapeVAE transformer run on B latents, the
        optimisation-in-the-loop iterations run as ONE capacity-mode GuidanceBatch of B slots -- per iteration one hipGraph
        replay of iso-surfacing, object installation, fused step and iso-surface backward for all B (engine.SdfObjective) --
        and one AdamW over the (B, L, D) noise prediction (element-wise: B independent optimisers).

        images: list of B images (as `__call__` takes one); paths: list of B dicts with `__call__`'s eight path arguments;
        generators: list of B torch.Generators (default: every image seeded with 2, RUN:120, 144); fovs: list of B fields of
        view in degrees (default: the renderer's camera for all, else each image's fov.json -- what RUN:228-230 hands to its
        camera).  All images must share H x W.  -> list of B (object Meshes, hand Meshes) in the MoGe world.

        Events the reference handles per image are handled per image here, too, with ONE read-back of the B flag words per
        iteration (the reference's NaN test is one as well): an EMPTY iso-surface skips that image's iteration -- no optimiser
        step for its noise prediction, none for its pose (PL:1394-1397, 1511-1513; `stats["skipped_empty"]`); an image whose
        surface is not a closed manifold or exceeds the capacity, or whose loss is NaN (PL:1442-1444, 1590-1592), LEAVES the batch:
        its slot is frozen, the other images carry on, and its entry of the result list is a `BatchLeftFastPath` instance -- the
        caller re-runs that image through `__call__`, which handles all of these the reference's way."""
        B = len(images)
        device, dtype = self.device, self.dtype
        cfg0 = config() if config is not None else E.OptimizationConfig()
        self.stats = stats = {"inner_iterations": 0, "images": B, "skipped_empty": 0}
        left = {}                 # image -> BatchLeftFastPath: frozen slots
        do_cfg = guidance_scale >= 0 and not (getattr(self.model, "guidance_embed", False) is True)
        img, msk = self.prepare_image(list(images))
        cond = self.encode_cond(image=img, mask=msk, do_classifier_free_guidance=do_cfg, dual_guidance=False)
        guid_res = int(guidance_octree_resolution)
        bmin, bmax = np.full(3, -1.10), np.full(3, 1.10)

        def grid(res):
            xyz_np, gsz, _ = generate_dense_grid_points(bmin, bmax, octree_depth=5, octree_resolution=res, indexing="ij")
            return torch.as_tensor(xyz_np, dtype=torch.float32, device=device), gsz

        xyz_samples, grid_size = grid(guid_res)
        n_steps = cfg0.num_inference_steps
        timesteps, n_steps_obj = retrieve_timesteps(self.scheduler, n_steps, device, sigmas=np.linspace(0, 1, n_steps))
        guidance = None
        if getattr(self.model, "guidance_embed", False) is True:
            guidance = torch.tensor([guidance_scale] * B, device=device, dtype=dtype)
        self.model.eval()
        self.vae.eval()
        if generators is None:
            generators = [torch.Generator().manual_seed(2) for _ in range(B)]
        latents = torch.cat([self.prepare_latents(1, dtype, device, g) for g in generators], 0).clone()

        if fovs is None:
            fovs = [float(renderer.rasterizer.cameras.fov) if renderer is not None else None] * B
        jr = inputs.load_j_regressor() if J_regressor is None else np.asarray(J_regressor, np.float32)
        scenes = []
        for p_, fov in zip(paths, fovs):
            q = dict(cropped_hand_mask_path=p_["hand_mask_path"], cropped_obj_mask_path=p_["obj_mask_path"], moge_mesh_path=p_["moge_mesh_path"],
                     moge_fov_path=os.path.join(os.path.dirname(p_["moge_mesh_path"]), "fov.json"), T_h2m_path=p_["h2m_rt_path"],
                     aligned_mano_mesh_path=p_["aligned_mano_mesh_path"], hamer_for_guid_path=p_["hamer_for_guid_path"])
            scenes.append(inputs.load_scene_from_files(q, jr, E.hip_render_fn(device), fov=fov, with_object=False))
        cap = tuple(obj_capacity) if obj_capacity else (8 * guid_res * guid_res, 16 * guid_res * guid_res)
        gb = E.GuidanceBatch(scenes, device=device, grid_res=guid_res, n_renders=2, obj_capacity=cap)
        fobj = E.SdfObjective(gb, xyz_samples, guid_res)
        hip_dec = getattr(self.vae, "hip_geo", None)
        T_h2m = [torch.as_tensor(sc["T_h2m"], dtype=torch.float32, device=device) for sc in scenes]
        hand_moge = [torch.as_tensor(sc["hand_verts"], dtype=torch.float32, device=device) for sc in scenes]
        hand_faces = [torch.as_tensor(sc["hand_faces"], dtype=torch.int64, device=device) for sc in scenes]

        def sdf_of(x1, xyz, gsz):
            """latent2sdf (PL:292-313) for B latents: the VAE transformer on all of them, the geometry decoder per image."""
            pred = vae_tokens(self.vae, 1 / self.vae.scale_factor * x1)      # all images through ONE foho_vae_fwd (rows = images x tokens)
            out = []
            hip = getattr(self.vae, "hip_geo", None)
            for b in range(x1.shape[0]):
                if b in left:                # a frozen slot is not decoded any more (an all-positive field: no surface)
                    out.append(torch.ones(xyz.shape[0], device=device))
                    continue
                if hip is not None:          # geo_decode.install(vae): all points in one call, gradients through foho_geo_decode_bwd
                    out.append(-hip(hip.grid_queries(xyz), pred[b:b + 1]).view(-1).float())
                    continue
                logits = [self.vae.geo_decoder(xyz[s0:s0 + num_chunks].half().unsqueeze(0), pred[b:b + 1]) for s0 in range(0, xyz.shape[0], num_chunks)]
                out.append(-torch.cat(logits, dim=1).view(-1).float())
            return torch.stack(out, 0)

        LEAVE, EMPTY = 1 | 16 | 32, 64      # flag bits: NaN loss, capacity overflow, not a closed manifold | empty iso-surface

        def latent_phase(phase, iters, i, t, noise_pred, lr):
            try:
                return latent_phase_body(phase, iters, i, t, noise_pred, lr)
            finally:          # however the phase ends (an exception included), the shared decoder's active-row bound does not outlive it
                if hip_dec is not None:
                    hip_dec.row_cap = None

        def latent_phase_body(phase, iters, i, t, noise_pred, lr):
            cfg, n_renders = E.phase_cfg(phase, cfg0, denoise_i=i, do_update=True)
            gb.set_n_renders(n_renders)
            gb.reset_optimizer()
            for b in left:                # reset_optimizer cleared the sticky bits: a slot that left stays frozen
                gb.flags[b] |= 16
            # one leaf and one parameter group per image: torch.optim.AdamW then keeps a step count per image and skips an image
            # whose gradient is None -- the reference's `continue` in front of its optimiser step, image by image
            noise = [noise_pred[b:b + 1].clone().detach().requires_grad_(True) for b in range(B)]
            opt = torch.optim.AdamW([{"params": [n], "lr": lr} for n in noise], eps=1e-4)
            for k in range(int(iters)):
                opt.zero_grad(set_to_none=True)
                x1 = self.scheduler.step_final(torch.cat(noise, 0), t, latents)
                sdf = sdf_of(x1, xyz_samples, grid_size)
                loss = fobj(sdf, cfg)                                   # (B,): one replay for all images
                fl = gb.flags.cpu().tolist()                            # the iteration's read-back (NaN is bit 0 of the flags) ...
                if hip_dec is not None:                                 # ... with the rows the decoder's backward will find a gradient on
                    _bound_active_rows(self.vae, max(fobj.active_rows()))
                go_host = [0.0] * B                                     # (host list, uploaded once: no per-image device synchronisation)
                for b in range(B):
                    if b in left:
                        continue
                    if fl[b] & LEAVE:
                        what = "a NaN loss" if fl[b] & 1 else ("an iso-surface beyond the capacity" if fl[b] & 16 else "an iso-surface that is not a closed manifold")
                        left[b] = BatchLeftFastPath(f"phase {phase}, denoising step {i}, iteration {k}: {what}", phase, i, k, fl[b])
                        gb.flags[b] |= 16                               # frozen from here on (the step leaves flagged slots alone)
                    elif fl[b] & EMPTY:
                        print("Invalid mesh detected, aborting step!")  # PL:1396, 1512
                        stats["skipped_empty"] += 1
                        gb.flags[b] &= ~EMPTY
                    else:
                        go_host[b] = 1.0
                if len(left) == B:
                    break
                go = torch.tensor(go_host, device=device)
                (loss * go).sum().backward()         # _SdfObjectiveFn.backward returns exact zeros for go = 0 (also past a NaN)
                for b in range(B):
                    if go_host[b] == 0:
                        noise[b].grad = None
                opt.step()
                stats["inner_iterations"] += 1
            gb.raise_on_flags(strict_k=False, ignore_images=list(left))
            _check_rows_dropped(hip_dec)
            return torch.cat([n.detach() for n in noise], 0).clone()

        results = [None] * B
        for i, t in enumerate(timesteps):
            latent_in = torch.cat([latents] * 2) if do_cfg else latents
            timestep = t.expand(latent_in.shape[0]).to(latents.dtype) / self.scheduler.config.num_train_timesteps
            noise_pred = self.model(latent_in, timestep, cond, guidance=guidance)
            if do_cfg:
                scale_i = cfg0.obj_guidance_scale * (1 - i / n_steps_obj) if i >= cfg0.guidance_start_step + 1 else cfg0.obj_guidance_scale
                c, u = noise_pred.chunk(2)
                noise_pred = u + scale_i * (c - u)
            if i >= cfg0.handopt_start_step:
                with torch.enable_grad():
                    if i == cfg0.handopt_start_step:                     # phase A: hands only (PL:1296-1358)
                        cfg, n_renders = E.phase_cfg("A", cfg0, denoise_i=i, do_update=True)
                        gb.set_n_renders(n_renders)
                        gb.reset_optimizer()
                        n = int(cfg0.optimization_steps_hand)
                        spg = max([d for d in range(1, 51) if n % d == 0]) if n > 0 else 1
                        graph = gb.capture(cfg, steps_per_graph=spg)
                        gb.reset_optimizer()
                        for _ in range(n // spg):
                            graph.replay()
                        stats["inner_iterations"] += n
                        torch.cuda.synchronize(device)
                        gb.raise_on_flags(strict_k=False)
                    elif len(left) == B:
                        pass                                             # nobody left to optimise: the caller re-runs them all
                    elif i == cfg0.handopt_start_step + 1:               # phase B (PL:1361-1453)
                        noise_pred = latent_phase("B", cfg0.optimization_steps_scale, i, t, noise_pred, cfg0.noise_obj_lr1)
                    else:                                                # phase C (PL:1455-1601)
                        noise_pred = latent_phase("C", cfg0.optimization_steps_joint, i, t, noise_pred, cfg0.noise_obj_lr2)
                noise_pred = noise_pred.detach().clone()
            latents = self.scheduler.step(noise_pred, t, latents).prev_sample
            # the clean-sample estimate as a mesh in the MoGe world (PL:1612-1661): the last one is the result; earlier ones
            # only matter as the fall-back of a later empty decode, so they are taken on the guidance grid
            res = final_octree_resolution if i == n_steps - 1 else guid_res
            xyz_d, gsz_d = grid(res) if res != guid_res else (xyz_samples, grid_size)
            sdf = sdf_of(self.scheduler.step_final(noise_pred, t, latents), xyz_d, gsz_d)
            for b in range(B):
                if b in left:
                    continue
                p = gb.params[b]
                obj, hand = results[b] if results[b] is not None else (None, None)
                if i >= cfg0.handopt_start_step:         # the hand of THIS step, whatever the decode gives (PL:1615-1619 come before the
                    hv = similarity_about_center(hand_moge[b], p[0], p[4:8], p[1:4])       # empty-mesh test of PL:1644-1646)
                    tex_h = torch.zeros_like(hv)
                    tex_h[:, 1] = 1.0
                    hand = Meshes(verts=[hv], faces=[hand_faces[b]], textures=TexturesVertex(verts_features=[tex_h]))
                verts, faces, _ = ops.flexicubes(xyz_d, sdf[b], res)
                if verts.shape[0] == 0:
                    print("Invalid mesh detected, aborting step!")
                else:
                    obj_world = similarity_about_center(verts @ T_h2m[b][:3, :3].T + T_h2m[b][:3, 3], p[8], p[12:16], p[9:12])
                    tex = torch.zeros_like(obj_world)
                    tex[:, 2] = 1.0
                    obj = Meshes(verts=[obj_world], faces=[faces], textures=TexturesVertex(verts_features=[tex]))
                results[b] = (obj, hand)
        for b, why in left.items():
            results[b] = why
        self.guidance_batch = gb
        stats["left_batch"] = sorted(left)
        return results

    # ------------------------------------------------------------------ PL:1044-1679
    @torch.no_grad()
    def __call__(self, image=None, num_inference_steps: int = 30, timesteps=None, sigmas=None, eta: float = 0.0,
                 guidance_scale: float = 7.5, generator=None, box_v=1.10, octree_resolution=64, mc_level=0.0, mc_algo="mc",
                 num_chunks=8000, output_type: Optional[str] = "trimesh", enable_pbar=True, config=None, renderer=None,
                 sil_renderer=None, cropped_obj_img_path=None, hamer_for_guid_path=None, aligned_mano_mesh_path=None,
                 obj_mask_path=None, hand_mask_path=None, moge_mesh_path=None, h2m_rt_path=None, hunyuan_hoi_mesh_path=None,
                 **kwargs):
        kwargs.pop("callback", None)            # popped and ignored, as in PL:1073-1074
        kwargs.pop("callback_steps", None)
        final_res = int(kwargs.pop("final_octree_resolution", 384))          # PL:1627 (tests use a smaller grid)
        J_regressor = kwargs.pop("J_regressor", None)                        # default: the file of PL:1218
        on_phase_end = kwargs.pop("on_phase_end", None)      # hook(phase, denoising step, GuidanceBatch): inspection / tests
        self.stats = stats = {"inner_iterations": 0, "skipped_empty": 0}
        self.loss_log = loss_log = []        # (phase, denoising step, iteration, loss terms) every 10th iteration, like the prints
        self.param_log = param_log = []      # (phase, denoising step, the 16 similarity parameters, noise prediction) at phase end
        device, dtype = self.device, self.dtype
        cfg0 = config() if config is not None else E.OptimizationConfig()

        debug_root = os.environ.get("FOHO_DEBUG_DIR")
        index = cropped_obj_img_path.split("/")[-1].split("_")[0]
        log = None
        if debug_root:
            save_dir = os.path.join(debug_root, f"{datetime.datetime.now().strftime('%Y%m%d_%H%M%S')}_exp_obj{index}_inpainted")
            os.makedirs(save_dir, exist_ok=True)
            log = open(os.path.join(save_dir, "losses.txt"), "w")

        def say(msg):
            if log:
                log.write(msg + "\n")
            print(msg)

        do_cfg = guidance_scale >= 0 and not (getattr(self.model, "guidance_embed", False) is True)
        obj_img, obj_mask = self.prepare_image(image)
        cond_obj = self.encode_cond(image=obj_img, mask=obj_mask, do_classifier_free_guidance=do_cfg, dual_guidance=False)

        # dense grid the latent is decoded on (PL:1126-1143); FlexiCubes needs no cube index table here
        octree_res = 64 if "guidance_octree_resolution" not in kwargs else int(kwargs.pop("guidance_octree_resolution"))
        guid_res = octree_res
        bmin, bmax = np.full(3, -1.10), np.full(3, 1.10)
        xyz_np, grid_size, _ = generate_dense_grid_points(bmin, bmax, octree_depth=5, octree_resolution=octree_res, indexing="ij")
        xyz_samples = torch.as_tensor(xyz_np, dtype=torch.float32, device=device)

        obj_guidance_scale = cfg0.obj_guidance_scale
        batch_size = cfg0.batch_size
        num_inference_steps = cfg0.num_inference_steps
        handopt_start_step = cfg0.handopt_start_step
        guidance_start_step = cfg0.guidance_start_step
        guidance_end_step = num_inference_steps
        if debug_root:
            keys = ["obj_guidance_scale", "optimization_steps_hand", "optimization_steps_joint", "optimization_steps_scale",
                    "num_inference_steps", "guidance_start_step", "handopt_start_step", "phase1_hand_lrs", "phase2_hand_lrs",
                    "obj_lrs", "obj_2half_lrs", "noise_obj_lr1", "noise_obj_lr2", "use_intersection_loss"]
            with open(os.path.join(save_dir, "params.json"), "w") as f:
                json.dump({**{k: getattr(cfg0, k) for k in keys}, "guidance_end_step": guidance_end_step}, f, indent=4)

        sigmas = np.linspace(0, 1, num_inference_steps) if sigmas is None else sigmas     # starts from 0 (PL:1186-1193)
        timesteps_obj, n_steps_obj = retrieve_timesteps(self.scheduler, num_inference_steps, device, sigmas=sigmas)
        guidance = None
        if getattr(self.model, "guidance_embed", False) is True:
            guidance = torch.tensor([guidance_scale] * batch_size, device=device, dtype=dtype)
        self.model.eval()
        self.vae.eval()
        obj_latents = self.prepare_latents(batch_size, dtype, device, generator).clone()

        # per-image inputs (PL:1217-1256): masks, keypoints, aligned MANO mesh, Hunyuan -> MoGe transform, MoGe target maps
        fov = float(renderer.rasterizer.cameras.fov) if renderer is not None else None
        paths = dict(cropped_hand_mask_path=hand_mask_path, cropped_obj_mask_path=obj_mask_path, moge_mesh_path=moge_mesh_path,
                     moge_fov_path=os.path.join(os.path.dirname(moge_mesh_path), "fov.json"), T_h2m_path=h2m_rt_path,
                     aligned_mano_mesh_path=aligned_mano_mesh_path, hamer_for_guid_path=hamer_for_guid_path)
        jr = inputs.load_j_regressor() if J_regressor is None else np.asarray(J_regressor, np.float32)
        scene = inputs.load_scene_from_files(paths, jr, E.hip_render_fn(device), fov=fov, with_object=False)
        gb = E.GuidanceBatch([scene], device=device, grid_res=guid_res, n_renders=2)
        T_h2m = torch.as_tensor(scene["T_h2m"], dtype=torch.float32, device=device)
        hand_moge = torch.as_tensor(scene["hand_verts"], dtype=torch.float32, device=device)
        hand_faces = torch.as_tensor(scene["hand_faces"], dtype=torch.int64, device=device)

        def decode_mesh(noise_pred, t, latents, res, xyz, gsz):
            x1 = self.scheduler.step_final(noise_pred, t, latents)
            sdf = latent2sdf(x1, xyz, gsz, self.vae, device, num_chunks)
            return ops.flexicubes(xyz, sdf[0].flatten(), res)

        # Phases B and C re-extract the object from the latent in every iteration (PL:1391-1393, 1507-1509): vertex count,
        # face count and connectivity change each time.  Fast path: engine.SdfObjective -- iso-surfacing, installing the new
        # object, the fused step and the backward to the SDF as one hipGraph replay over capacity-sized buffers, the counts
        # staying on the device.  Iterations it cannot serve (capacity exceeded, a surface that is not a closed manifold)
        # are redone on the exact-size path (ops.flexicubes + GuidanceBatch.objective), which handles any mesh.
        fast = {"gb": None, "obj": None, "cap": (0, 0)}

        def fast_objective(cap):
            if fast["gb"] is None or fast["cap"] != cap:
                g2 = E.GuidanceBatch([scene], device=device, grid_res=guid_res, n_renders=2, obj_capacity=cap)
                fast.update(gb=g2, obj=E.SdfObjective(g2, xyz_samples, guid_res), cap=cap)
            return fast["gb"], fast["obj"]

        def sync_state(src, dst):       # pose parameters and optimiser moments travel with the iteration
            for name in ("params", "adam_m", "adam_v", "adam_t", "flags"):
                getattr(dst, name).copy_(getattr(src, name))

        def latent_phase(phase, iters, i, t, noise_pred, lr, nan_returns_none):
            try:
                return latent_phase_body(phase, iters, i, t, noise_pred, lr, nan_returns_none)
            finally:          # however the phase ends (a NaN return, an exception), the shared decoder's active-row bound does not outlive it
                hip_ = getattr(self.vae, "hip_geo", None)
                if hip_ is not None:
                    hip_.row_cap = None

        def latent_phase_body(phase, iters, i, t, noise_pred, lr, nan_returns_none):
            """Phases B / C (PL:1362-1453 / 1456-1601): `iters` iterations of decode -> fused step -> AdamW."""
            cfg, n_renders = E.phase_cfg(phase, cfg0, denoise_i=i, do_update=True)
            gb.set_n_renders(n_renders)
            gb.reset_optimizer()
            noise_pred = noise_pred.clone().detach().requires_grad_(True)
            opt = torch.optim.AdamW([{"params": [noise_pred], "lr": lr}], eps=1e-4)
            use_fast = os.environ.get("FOHO_EXACT_SIZE_OBJECTIVE") != "1"
            cap = fast["cap"] if fast["cap"][0] else (8 * guid_res * guid_res, 16 * guid_res * guid_res)
            if not fast["cap"][0] and os.environ.get("FOHO_OBJ_CAPACITY"):      # "verts,faces": start value (tests: force the growth path)
                cap = tuple(int(x) for x in os.environ["FOHO_OBJ_CAPACITY"].split(","))
            for k in range(int(iters)):
                opt.zero_grad()
                x1 = self.scheduler.step_final(noise_pred, t, obj_latents)
                sdf = latent2sdf(x1, xyz_samples, grid_size, self.vae, device, num_chunks)[0].flatten()
                loss, cur = None, gb
                if use_fast:
                    g2, fobj = fast_objective(cap)
                    g2.set_n_renders(n_renders)
                    sync_state(gb, g2)
                    loss = fobj(sdf, cfg)
                    nv, nf, fl = fobj.status()[0]          # one read-back per iteration (the reference's NaN test is one, too)
                    if getattr(self.vae, "hip_geo", None) is not None:
                        _bound_active_rows(self.vae, fobj.active_rows()[0])
                    if fl & 64:                            # empty iso-surface (PL:1394-1397, 1511-1513)
                        print("Invalid mesh detected, aborting step!")
                        stats["skipped_empty"] += 1
                        continue
                    if fl & 16:                            # larger than the capacity: grow it, this iteration goes the exact way
                        cap = (max(cap[0], 2 * nv), max(cap[1], 2 * nf))
                        stats["capacity_grown"] = stats.get("capacity_grown", 0) + 1
                    if fl & 48:
                        loss = None
                    else:
                        cur = g2
                if loss is None:
                    verts, faces, _ = ops.flexicubes(xyz_samples, sdf, guid_res)
                    if verts.shape[0] == 0:
                        print("Invalid mesh detected, aborting step!")
                        stats["skipped_empty"] += 1
                        continue
                    _bound_active_rows(self.vae, xyz_samples.shape[0])     # the exact-size path: no count known before the backward -- all rows
                    loss = gb.objective(verts, faces, cfg)
                    stats["exact_size_iterations"] = stats.get("exact_size_iterations", 0) + 1
                stats["inner_iterations"] += 1
                if torch.isnan(loss):
                    print("Total loss is NaN")
                    if cur is not gb:
                        sync_state(cur, gb)
                    if nan_returns_none:
                        return None
                    break
                if k % 10 == 0:
                    if debug_root and phase == "B":     # PL:1417-1419
                        E.save_grid(E.normal_map(cur, E.L.FACES_OBJ), scene["moge_normal"] * np.asarray(scene["obj_mask"], np.float32)[..., None],
                                    os.path.join(save_dir, f"rendered_obj_normal_t{i}_opt{k}.png"))
                    l = cur.loss_dict(0)
                    loss_log.append((phase, i, k, l))
                    say(f"Opt step {k}, object loss: {l['edge']}, loss_intersection: {l.get('intersection', 0.0)}, "
                        f"total: {l['total']}")
                loss.backward()
                opt.step()
                if cur is not gb:
                    sync_state(cur, gb)
            gb.raise_on_flags(strict_k=False)
            _check_rows_dropped(getattr(self.vae, "hip_geo", None))
            param_log.append((phase, i, gb.params[0].detach().clone(), noise_pred.detach().clone()))
            if on_phase_end is not None:
                on_phase_end(phase, i, gb)
            return noise_pred.detach().clone()

        obj_out = hand_out = None
        for i, t in enumerate(timesteps_obj):
            latent_in = torch.cat([obj_latents] * 2) if do_cfg else obj_latents
            timestep = t.expand(latent_in.shape[0]).to(obj_latents.dtype) / self.scheduler.config.num_train_timesteps
            noise_pred_obj = self.model(latent_in, timestep, cond_obj, guidance=guidance)
            if do_cfg:      # trust the learned prior early, the guidance later (PL:1284-1293)
                scale_i = obj_guidance_scale * (1 - i / n_steps_obj) if i >= guidance_start_step + 1 else obj_guidance_scale
                c, u = noise_pred_obj.chunk(2)
                noise_pred_obj = u + scale_i * (c - u)

            if i >= handopt_start_step:
                with torch.enable_grad():
                    if i == handopt_start_step:                      # phase A: hand only, no latent involved (PL:1296-1358)
                        say(f"Pre-guidance step {i}, optimizing hands only")
                        cfg, n_renders = E.phase_cfg("A", cfg0, denoise_i=i, do_update=True)
                        gb.set_n_renders(n_renders)
                        gb.reset_optimizer()
                        n = int(cfg0.optimization_steps_hand)
                        spg = max([d for d in range(1, 51) if n % d == 0]) if n > 0 else 1
                        if debug_root and n % 10 == 0:
                            spg = 10        # the reference plots the rendered hand normals every 10 iterations (PL:1331-1333)
                        graph = gb.capture(cfg, steps_per_graph=spg)
                        # the k = 0 losses the reference prints (PL:1351-1355): one evaluation at the current parameters, no update
                        cfg_eval, _ = E.phase_cfg("A", cfg0, denoise_i=i, do_update=False)
                        gb.step(cfg_eval)
                        l0 = gb.loss_dict(0)
                        loss_log.append(("A", i, 0, l0))
                        say(f"Opt step 0, loss_2d_kps: {l0['kps']}, loss_normal_hand: {l0['normal0']}, loss_disp_hand: {l0['disp0']}")
                        gb.reset_optimizer()
                        for rep in range(n // spg):
                            if debug_root and spg == 10:    # the render iteration k = 10 rep starts from
                                gb.step(cfg_eval)
                                E.save_grid(E.normal_map(gb, E.L.FACES_HAND), scene["moge_normal"],
                                            os.path.join(save_dir, f"rendered_normal_hand_t{i}_opt{10 * rep}.png"))
                            graph.replay()
                        stats["inner_iterations"] += n
                        torch.cuda.synchronize(device)
                        # the reference has no NaN guard in phase A (PL:1320-1358): a NaN loss there poisons the hand pose;
                        # here the sticky flag freezes the parameters instead -- say so rather than continue silently
                        if int(gb.raise_on_flags(strict_k=False)[0]) & 1:
                            say("Total loss is NaN in the hand-only phase: hand parameters frozen at their last finite value")
                        l = gb.loss_dict(0)
                        say(f"Opt step {n - 1}, loss_2d_kps: {l.get('kps', 0.0)}, total: {l['total']}")
                        param_log.append(("A", i, gb.params[0].detach().clone(), None))
                        if on_phase_end is not None:
                            on_phase_end("A", i, gb)
                    elif i == handopt_start_step + 1:                # phase B: object transform + latent (PL:1361-1453)
                        say(f"Object optimization step {i}, optimizing object transformation")
                        noise_pred_obj = latent_phase("B", cfg0.optimization_steps_scale, i, t, noise_pred_obj,
                                                      cfg0.noise_obj_lr1, nan_returns_none=True)
                        if noise_pred_obj is None:
                            return None
                    elif handopt_start_step + 2 <= i <= guidance_end_step:   # phase C: joint (PL:1455-1601)
                        say(f"Joint optimization step {i}, optimizing hands and object together")
                        noise_pred_obj = latent_phase("C", cfg0.optimization_steps_joint, i, t, noise_pred_obj,
                                                      cfg0.noise_obj_lr2, nan_returns_none=False)
                noise_pred_obj = noise_pred_obj.detach().clone()

            obj_latents = self.scheduler.step(noise_pred_obj, t, obj_latents).prev_sample

            # current clean-sample estimate as a mesh in the MoGe world (PL:1612-1661); the last one is the result
            p = gb.params[0]
            if i >= handopt_start_step:
                hand_now = similarity_about_center(hand_moge, p[0], p[4:8], p[1:4])
            if i == num_inference_steps - 1 and final_res != octree_res:     # final decode on the fine grid (PL:1626-1642)
                octree_res = final_res
                xyz_np, grid_size, _ = generate_dense_grid_points(bmin, bmax, octree_depth=5, octree_resolution=octree_res,
                                                                  indexing="ij")
                xyz_samples = torch.as_tensor(xyz_np, dtype=torch.float32, device=device)
            verts, faces, _ = decode_mesh(noise_pred_obj, t, obj_latents, octree_res, xyz_samples, grid_size)
            if verts.shape[0] == 0:
                print("Invalid mesh detected, aborting step!")
                continue
            obj_world = similarity_about_center(verts @ T_h2m[:3, :3].T + T_h2m[:3, 3], p[8], p[12:16], p[9:12])
            tex = torch.zeros_like(obj_world)
            tex[:, 2] = 1.0
            obj_out = Meshes(verts=[obj_world], faces=[faces], textures=TexturesVertex(verts_features=[tex]))
            if i >= handopt_start_step:
                tex_h = torch.zeros_like(hand_now)
                tex_h[:, 1] = 1.0
                hand_out = Meshes(verts=[hand_now], faces=[hand_faces], textures=TexturesVertex(verts_features=[tex_h]))
            if debug_root:      # PL:1664-1667: the scene of this denoising step against the MoGe normals
                if i >= handopt_start_step:
                    dv = torch.cat([hand_now, obj_world], 0)
                    df = torch.cat([hand_faces, faces + hand_now.shape[0]], 0)
                else:
                    dv, df = obj_world, faces
                dn, _, _ = E.hip_render_fn(device)(dv.detach().cpu().numpy(), df.cpu().numpy(), gb.H, gb.W, scene["fov"])
                E.save_grid(dn, scene["moge_normal"], os.path.join(save_dir, f"rendered_normal_t{i}.png"))
            if debug_root and i in (14, num_inference_steps - 1):
                from . import meshio
                tag = "final" if i == num_inference_steps - 1 else f"guidance_step_{i}"
                meshio.save_ply(os.path.join(save_dir, f"{tag}_hand_mesh.ply"), hand_out.verts_packed().cpu().numpy(),
                                hand_faces.cpu().numpy())
                meshio.save_ply(os.path.join(save_dir, f"{tag}_obj_mesh.ply"), obj_world.cpu().numpy(), faces.cpu().numpy())
        if log:
            log.close()
        self.guidance_batch = gb
        return obj_out, hand_out
