"""Random-initialised stand-ins with the interfaces of the Hunyuan3D-2 networks the guided pipeline consumes.

Hunyuan3D-2 (hy3dgen @ e664e74, README.md:39-41 of the reference) and its weights are not available on the MI355X
image and there is no network, so tests, the demo and the chain benchmark drive `pipeline.GuidedShapePipeline` with
these small torch modules instead.  They reproduce the INTERFACES the pipeline touches (PL:292-338, 563-742, 1270-1293):

  vae.scale_factor, vae.latent_shape, vae(latents) -> (B, L, width), vae.geo_decoder(queries (B,N,3), latents) -> (B,N,1)
  model(latents, timestep in [0,1], cond dict, guidance=None) -> velocity of the latents' shape; model.guidance_embed
  conditioner(image=, mask=) -> {"main": (B, T, C)}, conditioner.unconditional_embedding(B)
  image_processor(img, return_mask=True) -> {"image": (1,3,S,S), "mask": (1,1,S,S)}

and the STRUCTURE of the ShapeVAE decoder route (latent -> linear -> self-attention blocks -> Fourier-embedded query
cross-attention -> linear), with sizes as constructor arguments so that the full-size shape (3072 x 64 latents, width
1024, 16 heads, 16 layers) can be instantiated for timing.  The occupancy logits get an analytic sphere prior so that a
random-initialised decoder still yields a closed surface inside the +-1.1 box.  Nothing here is a trained model.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Block(nn.Module):
    """Pre-norm attention + MLP block; kv=None -> self-attention."""

    def __init__(self, width, heads):
        super().__init__()
        self.heads = heads
        self.ln_q, self.ln_kv, self.ln_2 = nn.LayerNorm(width), nn.LayerNorm(width), nn.LayerNorm(width)
        self.q, self.kv, self.proj = nn.Linear(width, width), nn.Linear(width, 2 * width), nn.Linear(width, width)
        self.fc1, self.fc2 = nn.Linear(width, 4 * width), nn.Linear(4 * width, width)

    def forward(self, x, kv=None):
        B, N, C = x.shape
        src = self.ln_kv(x if kv is None else kv)
        q = self.q(self.ln_q(x)).view(B, N, self.heads, -1).transpose(1, 2)
        k, v = self.kv(src).view(B, src.shape[1], 2, self.heads, -1).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, C)
        x = x + self.proj(a)
        return x + self.fc2(F.gelu(self.fc1(self.ln_2(x))))


class _GeoDecoder(nn.Module):
    def __init__(self, width, heads, num_freqs, radius, sharpness, gain):
        super().__init__()
        self.register_buffer("freqs", 2.0 ** torch.arange(num_freqs, dtype=torch.float32) * math.pi, persistent=False)
        self.query_proj = nn.Linear(3 * (2 * num_freqs + 1), width)
        self.block = _Block(width, heads)
        self.ln_post, self.out = nn.LayerNorm(width), nn.Linear(width, 1)
        self.radius, self.sharpness, self.gain = radius, sharpness, gain

    def forward(self, queries, latents):
        q32 = queries.float()
        emb = (q32[..., None] * self.freqs).flatten(-2)
        emb = torch.cat([q32, emb.sin(), emb.cos()], -1).to(latents.dtype)
        x = self.block(self.query_proj(emb), kv=latents)
        learned = self.out(self.ln_post(x))
        prior = (self.radius - q32.norm(dim=-1, keepdim=True)) * self.sharpness       # > 0 inside (occupancy logits)
        return (prior + self.gain * learned.float()).to(latents.dtype)


class Hy3dgenLayoutDecoder(nn.Module):
    """A module laid out like hy3dgen's CrossAttentionDecoder (hy3dgen/shapegen/models/autoencoders/attention_blocks.py, not in
    the reference tree; restated from its published structure): FourierEmbedder.frequencies, query_proj,
    cross_attn_decoder = ResidualCrossAttentionBlock{ln_1 (queries), ln_2 (latents), ln_3, attn{c_q, c_kv, c_proj,
    attention{heads, q_norm, k_norm}}, mlp{c_fc, c_proj}}, ln_post, output_proj -- c_kv's output is viewed as (tokens, heads, 2 d)
    and split into K and V per head, i.e. K and V rows INTERLEAVE head by head."""

    def __init__(self, width, heads, num_freqs=8, qk_norm=False):
        super().__init__()
        self.fourier_embedder = nn.Module()
        self.fourier_embedder.register_buffer("frequencies", 2.0 ** torch.arange(num_freqs, dtype=torch.float32))      # include_pi=False
        self.query_proj = nn.Linear(3 * (2 * num_freqs + 1), width)
        blk = nn.Module()
        blk.ln_1, blk.ln_2, blk.ln_3 = nn.LayerNorm(width, eps=1e-6), nn.LayerNorm(width, eps=1e-6), nn.LayerNorm(width, eps=1e-6)   # hy3dgen's eps
        blk.attn = nn.Module()
        blk.attn.c_q, blk.attn.c_kv, blk.attn.c_proj = nn.Linear(width, width, bias=False), nn.Linear(width, 2 * width, bias=False), nn.Linear(width, width)
        blk.attn.attention = nn.Module()
        blk.attn.attention.heads = heads
        d = width // heads     # qk_norm: LayerNorm over the head dimension on q and k (the released ShapeVAE config switches it on)
        blk.attn.attention.q_norm = nn.LayerNorm(d, elementwise_affine=True, eps=1e-6) if qk_norm else nn.Identity()
        blk.attn.attention.k_norm = nn.LayerNorm(d, elementwise_affine=True, eps=1e-6) if qk_norm else nn.Identity()
        blk.mlp = nn.Module()
        blk.mlp.c_fc, blk.mlp.c_proj = nn.Linear(width, 4 * width), nn.Linear(4 * width, width)
        self.cross_attn_decoder = blk
        self.ln_post, self.output_proj = nn.LayerNorm(width), nn.Linear(width, 1)
        self.heads = heads

    def forward(self, queries, latents):
        q32 = queries.float()
        emb = (q32[..., None] * self.fourier_embedder.frequencies).flatten(-2)
        x = self.query_proj(torch.cat([q32, emb.sin(), emb.cos()], -1).to(latents.dtype))
        b = self.cross_attn_decoder
        B, N, C = x.shape
        q = b.attn.c_q(b.ln_1(x)).view(B, N, self.heads, -1)
        kv = b.attn.c_kv(b.ln_2(latents)).view(B, latents.shape[1], self.heads, -1)
        k, v = torch.split(kv, C // self.heads, dim=-1)
        q, k = b.attn.attention.q_norm(q), b.attn.attention.k_norm(k)
        a = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).reshape(B, N, C)
        x = x + b.attn.c_proj(a)
        x = x + b.mlp.c_proj(F.gelu(b.mlp.c_fc(b.ln_3(x))))
        return self.output_proj(self.ln_post(x))


class Hy3dgenLayoutShapeVAE(nn.Module):
    """post_kl + transformer laid out like hy3dgen's ShapeVAE (hy3dgen/shapegen/models/autoencoders/{model,attention_blocks}.py, not in the
    reference tree; restated from its published structure): transformer.resblocks[i] = ResidualAttentionBlock{ln_1, attn{c_qkv (bias per
    qkv_bias), c_proj, attention{heads, q_norm, k_norm}}, ln_2, mlp{c_fc, c_proj}} -- c_qkv's output is viewed as (tokens, heads, 3 d) and
    split into q | k | v per head (rows INTERLEAVE head by head), LayerNorms built with eps 1e-6, qk_norm = LayerNorm over the head
    dimension of q and of k (the released config: qkv_bias False, qk_norm True).  `geo_decoder` is a Hy3dgenLayoutDecoder."""

    def __init__(self, num_latents=3072, embed_dim=64, width=1024, heads=16, layers=16, num_freqs=8, qk_norm=True, qkv_bias=False,
                 scale_factor=1.0):
        super().__init__()
        self.latent_shape = (num_latents, embed_dim)
        self.scale_factor = scale_factor
        self.post_kl = nn.Linear(embed_dim, width)
        self.transformer = nn.Module()
        blocks = []
        d = width // heads
        for _ in range(layers):
            blk = nn.Module()
            blk.ln_1, blk.ln_2 = nn.LayerNorm(width, eps=1e-6), nn.LayerNorm(width, eps=1e-6)
            blk.attn = nn.Module()
            blk.attn.c_qkv, blk.attn.c_proj = nn.Linear(width, 3 * width, bias=qkv_bias), nn.Linear(width, width)
            blk.attn.attention = nn.Module()
            blk.attn.attention.heads = heads
            blk.attn.attention.q_norm = nn.LayerNorm(d, elementwise_affine=True, eps=1e-6) if qk_norm else nn.Identity()
            blk.attn.attention.k_norm = nn.LayerNorm(d, elementwise_affine=True, eps=1e-6) if qk_norm else nn.Identity()
            blk.mlp = nn.Module()
            blk.mlp.c_fc, blk.mlp.c_proj = nn.Linear(width, 4 * width), nn.Linear(4 * width, width)
            blocks.append(blk)
        self.transformer.resblocks = nn.ModuleList(blocks)
        self.geo_decoder = Hy3dgenLayoutDecoder(width, heads, num_freqs=num_freqs, qk_norm=qk_norm)
        self.heads = heads

    def block_forward(self, blk, x):
        B, N, C = x.shape
        qkv = blk.attn.c_qkv(blk.ln_1(x)).view(B, N, self.heads, -1)
        q, k, v = torch.split(qkv, C // self.heads, dim=-1)
        q, k = blk.attn.attention.q_norm(q), blk.attn.attention.k_norm(k)
        a = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).reshape(B, N, C)
        x = x + blk.attn.c_proj(a)
        return x + blk.mlp.c_proj(F.gelu(blk.mlp.c_fc(blk.ln_2(x))))

    def forward(self, latents):
        x = self.post_kl(latents)
        for blk in self.transformer.resblocks:
            x = self.block_forward(blk, x)
        return x


class StandInShapeVAE(nn.Module):
    def __init__(self, num_latents=64, embed_dim=8, width=32, heads=2, layers=1, num_freqs=4, radius=0.8, sharpness=4.0,
                 gain=0.15, scale_factor=1.0):
        super().__init__()
        self.latent_shape = (num_latents, embed_dim)
        self.scale_factor = scale_factor
        self.post_kl = nn.Linear(embed_dim, width)
        self.transformer = nn.ModuleList([_Block(width, heads) for _ in range(layers)])
        self.geo_decoder = _GeoDecoder(width, heads, num_freqs, radius, sharpness, gain)

    def forward(self, latents):
        x = self.post_kl(latents)
        for blk in self.transformer:
            x = blk(x)
        return x


class StandInDiT(nn.Module):
    guidance_embed = False

    def __init__(self, embed_dim=8, width=32, cond_dim=16):
        super().__init__()
        self.inp, self.cond, self.time = nn.Linear(embed_dim, width), nn.Linear(cond_dim, width), nn.Linear(1, width)
        self.out = nn.Linear(width, embed_dim)

    def forward(self, latents, timestep, cond, guidance=None):
        c = self.cond(cond["main"]).mean(dim=1, keepdim=True)
        t = self.time(timestep.reshape(-1, 1, 1).to(latents.dtype))
        return self.out(F.gelu(self.inp(latents) + c + t))


class StandInConditioner(nn.Module):
    def __init__(self, tokens=4, cond_dim=16):
        super().__init__()
        self.tokens, self.cond_dim = tokens, cond_dim
        self.proj = nn.Linear(3, cond_dim)

    def forward(self, image=None, mask=None):
        B = image.shape[0]
        feat = F.adaptive_avg_pool2d(image, (self.tokens, 1)).reshape(B, 3, self.tokens).transpose(1, 2)
        return {"main": self.proj(feat)}

    def unconditional_embedding(self, batch_size):
        p = self.proj.weight
        return {"main": torch.zeros(batch_size, self.tokens, self.cond_dim, device=p.device, dtype=p.dtype)}


class StandInImageProcessor:
    def __init__(self, size=32):
        self.size = size

    def __call__(self, image, return_mask=True, **_):
        from PIL import Image
        if isinstance(image, str):
            image = Image.open(image)
        a = np.asarray(image.convert("RGBA").resize((self.size, self.size)), np.float32) / 255.0
        img = torch.from_numpy(a[..., :3] * 2 - 1).permute(2, 0, 1)[None]
        mask = torch.from_numpy(a[..., 3:]).permute(2, 0, 1)[None]
        return {"image": img, "mask": mask} if return_mask else img


def make_standin_pipeline(device="cuda", dtype=torch.float32, seed=0, **vae_kw):
    """GuidedShapePipeline over the stand-in networks (seeded initialisation)."""
    from .pipeline import GuidedShapePipeline
    from .scheduler import FlowMatchEulerDiscreteScheduler
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    vae = StandInShapeVAE(**vae_kw)
    dit = StandInDiT(embed_dim=vae.latent_shape[1])
    cond = StandInConditioner()
    torch.random.set_rng_state(g)
    return GuidedShapePipeline(vae, dit, FlowMatchEulerDiscreteScheduler(), cond, StandInImageProcessor(), device=device,
                               dtype=dtype)
