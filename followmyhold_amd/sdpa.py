"""Scaled-dot-product attention, forward and backward, on the geometry decoder's attention kernels (`foho_sdpa_fwd/_bwd`,
csrc/foho_geo.hip) -- for the self-attention layers of the ShapeVAE transformer inside `latent2sdf`.

Reference: third_party_patches/hy3dgen/shapegen/pipelines.py:295 (`pred = vae(pred)`: hy3dgen's ShapeVAE runs sixteen transformer
layers over the 3072 latent tokens, 16 heads of 64, through `torch.nn.functional.scaled_dot_product_attention`) -- executed in every
one of the 550 inner iterations per image, forward AND backward (PL:1391-1393, 1507-1509: the guidance gradient reaches the noise
prediction through it).  On an MI355X torch's default (flash) backend takes 509 us per layer forward + backward at that shape, its
memory-efficient one 350; these kernels -- the decoder's `k_geo_attn` forward, `k_geo_attn_bwd` for dK / dV, `k_geo_attn_dq` -- are
measured in `bench.py`'s `vae_attention` record.

    out = sdpa.attention(q, k, v)                    # (B, H, N, 64) fp16 CUDA tensors, like F.scaled_dot_product_attention(q, k, v)
    with sdpa.hip_sdpa():                            # ... or: every eligible F.scaled_dot_product_attention call inside the context
        tokens = vae(latents)

Eligible: 4-D fp16 CUDA tensors, head dimension 64, 1..16 heads, key count a multiple of 128, no mask, no dropout, not causal, the
default scale.  Everything else goes to torch's own implementation; there is no CPU path (`_lib.lib()` raises without the library).
"""
import contextlib
import ctypes
import math

import torch
import torch.nn.functional as F

from . import _lib as L

_QSCALE = math.log2(math.e) / 8.0     # the kernels exponentiate with exp2: log2(e) / sqrt(64) folded into q


def _stream(t):
    return L.vp(torch.cuda.current_stream(t.device).cuda_stream)


class _HipSdpaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v):
        lib = L.lib()
        B, H, M, _ = q.shape
        Lk, W = k.shape[2], H * 64
        qs = (q.transpose(1, 2) * _QSCALE).to(torch.float16).reshape(B, M, W).contiguous()
        kv = torch.cat([k.transpose(1, 2).reshape(B, Lk, W), v.transpose(1, 2).reshape(B, Lk, W)], dim=-1).to(torch.float16).contiguous()
        out = torch.empty(B, M, W, dtype=torch.float16, device=q.device)
        nlse = torch.empty(B, (M + 63) // 64 * 64, H, dtype=torch.float32, device=q.device)
        lib.foho_sdpa_workspace_bytes.restype = ctypes.c_size_t
        nws = int(lib.foho_sdpa_workspace_bytes(M, Lk, H))
        if nws == 0:
            raise L.FohoError(f"foho_sdpa: shape (M={M}, L={Lk}, heads={H}) is outside what the kernels take")
        ws = torch.empty(nws, dtype=torch.uint8, device=q.device)
        lib.foho_geo_last_error.restype = ctypes.c_char_p
        for b in range(B):
            rc = lib.foho_sdpa_fwd(L.vp(qs[b].data_ptr()), L.vp(kv[b].data_ptr()), L.vp(out[b].data_ptr()), L.vp(nlse[b].data_ptr()), M, Lk, H,
                                   L.vp(ws.data_ptr()), ctypes.c_size_t(nws), _stream(q))
            if rc != 0:
                raise L.FohoError(f"foho_sdpa_fwd failed ({rc}): {lib.foho_geo_last_error().decode()}")
        ctx.save_for_backward(qs, kv, out, nlse)
        ctx.ws, ctx.dims, ctx.dtypes = ws, (B, H, M, Lk), (q.dtype, k.dtype, v.dtype)
        return out.view(B, M, H, 64).transpose(1, 2).to(q.dtype)

    @staticmethod
    def backward(ctx, g):
        lib = L.lib()
        qs, kv, out, nlse = ctx.saved_tensors
        B, H, M, Lk = ctx.dims
        W = H * 64
        go = g.transpose(1, 2).reshape(B, M, W).to(torch.float16).contiguous()
        gq = torch.empty(B, M, W, dtype=torch.float16, device=g.device)
        gkv = torch.empty(B, Lk, 2 * W, dtype=torch.float32, device=g.device)
        ws = ctx.ws
        for b in range(B):
            rc = lib.foho_sdpa_bwd(L.vp(qs[b].data_ptr()), L.vp(kv[b].data_ptr()), L.vp(out[b].data_ptr()), L.vp(nlse[b].data_ptr()), L.vp(go[b].data_ptr()),
                                   L.vp(gq[b].data_ptr()), L.vp(gkv[b].data_ptr()), M, Lk, H, L.vp(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream(g))
            if rc != 0:
                raise L.FohoError(f"foho_sdpa_bwd failed ({rc}): {lib.foho_geo_last_error().decode()}")
        dq = gq.view(B, M, H, 64).transpose(1, 2).to(ctx.dtypes[0])
        dk = gkv[..., :W].reshape(B, Lk, H, 64).transpose(1, 2).to(ctx.dtypes[1])
        dv = gkv[..., W:].reshape(B, Lk, H, 64).transpose(1, 2).to(ctx.dtypes[2])
        return dq, dk, dv


def eligible(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
    """Can `attention` serve this scaled_dot_product_attention call?"""
    if attn_mask is not None or dropout_p != 0.0 or is_causal or kw.get("enable_gqa"):
        return False
    if not (torch.is_tensor(q) and q.is_cuda and q.dim() == 4 and q.dtype == k.dtype == v.dtype == torch.float16):
        return False
    B, H, M, D = q.shape
    if D != 64 or not 1 <= H <= 16 or k.shape != v.shape or k.shape[0] != B or k.shape[1] != H or k.shape[3] != 64:
        return False
    if k.shape[2] % 128 or k.shape[2] < 128 or M < 1:
        return False
    return scale is None or abs(float(scale) - 0.125) < 1e-12


def attention(q, k, v):
    """softmax(q k^T / 8) v per head: q (B, H, M, 64), k / v (B, H, L, 64), fp16 on the GPU -> (B, H, M, 64); differentiable w.r.t. all three."""
    if not eligible(q, k, v):
        raise L.FohoError("sdpa.attention: fp16 CUDA tensors (B, H, N, 64) with 1..16 heads and a multiple of 128 keys are supported")
    return _HipSdpaFn.apply(q, k, v)


@contextlib.contextmanager
def hip_sdpa():
    """Inside the context `torch.nn.functional.scaled_dot_product_attention` sends eligible calls to the HIP kernels and everything
    else to torch's own implementation.  (A module that bound the function at import time -- `from torch.nn.functional import
    scaled_dot_product_attention` -- keeps torch's.)"""
    orig = F.scaled_dot_product_attention

    def dispatch(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
        if eligible(query, key, value, attn_mask, dropout_p, is_causal, scale, **kw):
            return _HipSdpaFn.apply(query, key, value)
        return orig(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale, **kw)

    F.scaled_dot_product_attention = dispatch
    try:
        yield
    finally:
        F.scaled_dot_product_attention = orig
