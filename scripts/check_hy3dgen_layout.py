#!/usr/bin/env python
"""Real-module readiness check of the HIP geometry decoder and VAE transformer (run it once on a machine where Hunyuan3D-2 is installed).

`followmyhold_amd.geo_decode._parts` reads hy3dgen's `CrossAttentionDecoder` by attribute name (query_proj, cross_attn_decoder.{ln_1,
ln_2, ln_3, attn.{c_q, c_kv, c_proj, attention.{heads, q_norm, k_norm}}, mlp.{c_fc, c_proj}}, ln_post, output_proj, fourier_embedder.
frequencies; c_kv's rows interleave K and V per head).  hy3dgen is not part of the reference tree, so in this repository that layout
is a restatement (`standins.Hy3dgenLayoutDecoder`); this script meets the REAL module:

    python scripts/check_hy3dgen_layout.py [--config path/to/config.yaml | --random] [--queries 20000]

  * hy3dgen not importable            -> prints "hy3dgen not installed", exit 0 (nothing to check here)
  * builds hy3dgen's ShapeVAE (random weights suffice: the check is about layout and arithmetic, not about a checkpoint; --config
    instantiates the released configuration's `vae` section), adopts its geo_decoder with HipGeoDecoder.from_module, decodes random
    queries against random latents and compares with the module itself, forward and latent gradient; exit 1 on a mismatch, with the
    first attribute that was not where `_parts` expects it when the adoption itself fails.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_vae(args):
    from hy3dgen.shapegen.models.autoencoders import ShapeVAE
    if args.config:
        import yaml
        with open(args.config) as f:
            cfg = yaml.safe_load(f)
        params = cfg["vae"]["params"] if "vae" in cfg else cfg["params"]
        return ShapeVAE(**params)
    # the released Hunyuan3D-2 shape (hy3dgen configs: 3072 latents of 64, width 1024, 16 heads, 16 decoder layers, 8 frequencies, qk_norm)
    return ShapeVAE(num_latents=3072, embed_dim=64, width=1024, heads=16, num_decoder_layers=16, num_freqs=8, include_pi=False, qkv_bias=False,
                    qk_norm=True, scale_factor=1.0188137142395404)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None, help="yaml of a Hunyuan3D-2 shape model (its `vae` section is instantiated)")
    ap.add_argument("--queries", type=int, default=20000)
    args = ap.parse_args()
    try:
        import hy3dgen.shapegen  # noqa: F401
    except Exception as e:  # noqa: BLE001
        print(f"hy3dgen not installed ({type(e).__name__}: {e}); nothing to check on this machine")
        return 0
    import torch
    from followmyhold_amd import _lib as L
    from followmyhold_amd.geo_decode import HipGeoDecoder, _parts
    if not torch.cuda.is_available():
        print("hy3dgen is importable but there is no GPU: the HIP decoder cannot run here")
        return 1
    torch.manual_seed(0)
    vae = build_vae(args).cuda().eval()
    dec = vae.geo_decoder
    with torch.no_grad():
        for p in dec.parameters():
            p.copy_(p.half().float())          # weights the fp16 kernels represent exactly
    try:
        parts = _parts(dec)
    except (AttributeError, L.FohoError) as e:
        print(f"FAIL: geo_decode._parts does not fit this hy3dgen's CrossAttentionDecoder: {type(e).__name__}: {e}")
        print("      attributes of the module:", [n for n, _ in dec.named_children()])
        return 1
    print("adopted:", {k: (type(v).__name__ if not isinstance(v, (int, float, bool, tuple)) else v) for k, v in parts.items()})
    hip = HipGeoDecoder.from_module(dec)
    n_lat, width = vae.latent_shape[0] if hasattr(vae, "latent_shape") else 3072, parts["q"].weight.shape[0]
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(1, n_lat, width, generator=g).half().cuda()
    q = (torch.rand(1, args.queries, 3, generator=g) * 2.0 - 1.0).half().cuda()
    go = torch.randn(1, args.queries, 1, generator=g).cuda()
    lat_h = lat.float().clone().requires_grad_(True)
    out = hip(q.float(), lat_h)
    (out.float() * go).sum().backward()
    lat_r = lat.float().requires_grad_(True)
    ref = dec(q, lat_r)
    (ref * go).sum().backward()
    scale = ref.abs().max().item()
    err = (out.float() - ref).abs().max().item()
    gh, gr = lat_h.grad, lat_r.grad
    gerr = (gh - gr).abs().max().item() / max(gr.abs().max().item(), 1e-30)
    cos = torch.nn.functional.cosine_similarity(gh.flatten(), gr.flatten(), dim=0).item()
    print(f"forward: max |diff| {err:.3e} at a logit scale of {scale:.3e};  latent gradient: rel {gerr:.3e}, cosine {cos:.7f}")
    ok = err <= 3e-3 * max(scale, 1.0) and gerr <= 1e-2 and cos >= 1 - 1e-4
    print("OK: the HIP decoder reproduces hy3dgen's CrossAttentionDecoder" if ok else "FAIL: mismatch against hy3dgen's module")
    return (0 if ok else 1) | check_transformer(vae, n_lat)


def check_transformer(vae, n_lat):
    """... and the transformer in front of it (`vae(latents)` = post_kl -> transformer, PL:295): followmyhold_amd.vae_transformer reads
    `vae.transformer.resblocks[i].{ln_1, attn.{c_qkv, c_proj, attention.{heads, q_norm, k_norm}}, ln_2, mlp.{c_fc, c_proj}}` (c_qkv's rows
    interleave q | k | v per head).  Tokens and latent gradient of the kernels against the module in float32; then the fallback route:
    does `sdpa.hip_sdpa()` reach the module's attention calls at all (hy3dgen binds `scaled_dot_product_attention` at import time)?"""
    import torch
    from followmyhold_amd import _lib as L, sdpa
    from followmyhold_amd.vae_transformer import HipVaeTransformer
    with torch.no_grad():
        for p in vae.parameters():
            p.copy_(p.half().float())
    try:
        tr = HipVaeTransformer.from_module(vae)
    except (AttributeError, L.FohoError) as e:
        print(f"FAIL: vae_transformer._blocks does not fit this hy3dgen's ShapeVAE.transformer: {type(e).__name__}: {e}")
        print("      attributes of a block:", [n for n, _ in vae.transformer.resblocks[0].named_children()] if hasattr(vae.transformer, "resblocks") else dir(vae.transformer))
        return 1
    embed = vae.post_kl.in_features
    g = torch.Generator().manual_seed(2)
    lat = torch.randn(1, n_lat, embed, generator=g).cuda()
    go = torch.randn(1, n_lat, tr.width, generator=g).cuda()
    lr = lat.clone().requires_grad_(True)
    ref = vae(lr)
    (ref * go).sum().backward()
    lh = lat.clone().requires_grad_(True)
    n0 = tr.calls
    out = tr(vae.post_kl(lh).half())
    (out.float() * go).sum().backward()
    assert tr.calls == n0 + 1
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    gerr = (lh.grad - lr.grad).abs().max().item() / max(lr.grad.abs().max().item(), 1e-30)
    cos = torch.nn.functional.cosine_similarity(lh.grad.flatten(), lr.grad.flatten(), dim=0).item()
    print(f"transformer: tokens rel {err:.3e};  latent gradient: rel {gerr:.3e}, cosine {cos:.7f}")
    ok = err <= 1e-2 and gerr <= 3e-2 and cos >= 1 - 2e-4
    print("OK: foho_vae_fwd / _bwd reproduce hy3dgen's transformer" if ok else "FAIL: transformer mismatch against hy3dgen's module")
    vae16 = vae.half()
    h0 = sdpa.hits
    with torch.no_grad(), sdpa.hip_sdpa():
        vae16(lat.half())
    print(f"fallback route: sdpa.hip_sdpa() served {sdpa.hits - h0} of the module's {len(vae.transformer.resblocks)} attention calls"
          + ("" if sdpa.hits - h0 else "  <-- the patch does not reach this hy3dgen build (only the backend priority applies)"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
