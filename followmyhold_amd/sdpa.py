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

import torch
import torch.nn.functional as F

from . import _lib as L


class FohoSdpaDesc(ctypes.Structure):
    """include/foho_hip.h: foho_sdpa_desc"""
    _fields_ = [("M", ctypes.c_int32), ("L", ctypes.c_int32), ("heads", ctypes.c_int32), ("batch", ctypes.c_int32),
                ("q_batch", ctypes.c_int64), ("q_row", ctypes.c_int64), ("q_head", ctypes.c_int64),
                ("kv_batch", ctypes.c_int64), ("kv_row", ctypes.c_int64), ("kv_head", ctypes.c_int64)]


_workspaces = {}     # (device index, M, L, heads) -> uint8 tensor; stream-ordered reuse, like torch's own workspaces


def _workspace(lib, dev, M, Lk, H):
    key = (dev.index, M, Lk, H)
    ws = _workspaces.get(key)
    if ws is None:
        lib.foho_sdpa_workspace_bytes.restype = ctypes.c_size_t
        n = int(lib.foho_sdpa_workspace_bytes(M, Lk, H))
        if n == 0:
            raise L.FohoError(f"foho_sdpa: shape (M={M}, L={Lk}, heads={H}) is outside what the kernels take")
        if len(_workspaces) >= 4:
            _workspaces.clear()
        ws = _workspaces[key] = torch.empty(n, dtype=torch.uint8, device=dev)
    return ws


def _in_place(q, k, v):
    """(B, H, N, 64) views whose memory the kernels can read directly: unit stride in d, 16-byte aligned strides, k and v laid out alike."""
    def ok(t):      # (strides below 64 -- expand()ed views -- are refused by the C side: those operands are copied)
        span = (t.shape[2] - 1) * t.stride(2) + (t.shape[1] - 1) * t.stride(1) + 64
        return (t.stride(3) == 1 and all(st % 8 == 0 for st in t.stride()[:3]) and t.stride(1) >= 64 and t.stride(2) >= 64 and t.data_ptr() % 16 == 0
                and span * 2 < 2 ** 31)
    return ok(q) and ok(k) and ok(v) and k.stride() == v.stride()


def _desc(q, k):
    B, H, M, _ = q.shape
    return FohoSdpaDesc(M, k.shape[2], H, B, q.stride(0), q.stride(2), q.stride(1), k.stride(0), k.stride(2), k.stride(1))


def _stream(t):
    return L.vp(torch.cuda.current_stream(t.device).cuda_stream)


# Which backward follows the HIP forward.  "hip" (default): foho_sdpa_bwd -- k_geo_attn_bwd (dK / dV written directly, one workgroup per
# (head, key block)) + k_geo_attn_dq: this package's own kernels, bitwise repeatable.  "torch" (opt-in, FOHO_VAE_SDPA=hip_torch_bwd): torch's
# memory-efficient attention backward fed with this forward's output and log-sum-exp -- a PRIVATE operator
# (aten::_scaled_dot_product_efficient_attention_backward) whose signature may change with a torch upgrade; measured 181 against 190 us per
# layer at 16 x 3072 x 64 and 7 % ahead for a batch of four (it takes the batch in one launch).  If the operator is missing or refuses the
# call, a warning is printed ONCE and the HIP backward takes over for the rest of the process.  (With `vae_transformer.install` -- the
# product's default -- none of this runs: the whole transformer is foho_vae_fwd / _bwd.)
backward_route = "hip"
_torch_route_refused = False      # set by the first backward the torch operator refuses


def _torch_backward_op():
    return getattr(getattr(torch.ops, "aten", None), "_scaled_dot_product_efficient_attention_backward", None)


class _HipSdpaFn(torch.autograd.Function):
    """One library call per direction: the operands are read where they lie (strided views of the projections' outputs), the
    scale is applied inside, the gradients come back as fp16 (B, N, H, 64) tensors viewed as (B, H, N, 64)."""

    @staticmethod
    def forward(ctx, q, k, v):
        lib = L.lib()
        if not _in_place(q, k, v):
            q, k, v = (t.transpose(1, 2).contiguous().transpose(1, 2) for t in (q, k, v))
        B, H, M, _ = q.shape
        Lk, W = k.shape[2], H * 64
        ctx.route = "torch" if backward_route == "torch" and not _torch_route_refused and _torch_backward_op() is not None else "hip"
        out = torch.empty(B, M, W, dtype=torch.float16, device=q.device)
        nlse = torch.empty(B, (M + 63) // 64 * 64, H, dtype=torch.float32, device=q.device)
        lse = torch.empty(B, H, M, dtype=torch.float32, device=q.device) if ctx.route == "torch" else None
        ws = _workspace(lib, q.device, M, Lk, H)
        d = _desc(q, k)
        rc = lib.foho_sdpa_fwd(ctypes.byref(d), L.vp(q.data_ptr()), L.vp(k.data_ptr()), L.vp(v.data_ptr()), L.vp(out.data_ptr()), L.vp(nlse.data_ptr()),
                               L.vp(lse.data_ptr()) if lse is not None else None, L.vp(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream(q))
        if rc != 0:
            lib.foho_geo_last_error.restype = ctypes.c_char_p
            raise L.FohoError(f"foho_sdpa_fwd failed ({rc}): {lib.foho_geo_last_error().decode()}")
        if lse is not None:
            ctx.save_for_backward(q, k, v, out, nlse, lse)
        else:
            ctx.save_for_backward(q, k, v, out, nlse)
        return out.view(B, M, H, 64).transpose(1, 2)

    @staticmethod
    def backward(ctx, g):
        global _torch_route_refused
        lib = L.lib()
        q, k, v, out, nlse = ctx.saved_tensors[:5]
        B, H, M, _ = q.shape
        if ctx.route == "torch":
            try:
                zero = torch.zeros((), dtype=torch.int64, device=g.device)
                gq, gk, gv, _ = _torch_backward_op()(g, q, k, v, None, out.view(B, M, H, 64).transpose(1, 2), ctx.saved_tensors[5], zero, zero, 0.0,
                                                     [True, True, True, False], False)
                return gq, gk, gv
            except (RuntimeError, TypeError) as e:
                if not _torch_route_refused:
                    import warnings
                    warnings.warn(f"followmyhold_amd.sdpa: torch's private attention backward refused the call ({type(e).__name__}: {e}); "
                                  "the HIP backward (foho_sdpa_bwd) serves this process from here on")
                _torch_route_refused = True        # this build's operator does not take the call: the HIP backward from here on
        Lk, W = k.shape[2], H * 64
        go = g.transpose(1, 2).reshape(B, M, W)
        if go.dtype != torch.float16 or not go.is_contiguous():
            go = go.to(torch.float16).contiguous()
        gq = torch.empty(B, M, W, dtype=torch.float16, device=g.device)
        gk = torch.empty(B, Lk, W, dtype=torch.float16, device=g.device)
        gv = torch.empty(B, Lk, W, dtype=torch.float16, device=g.device)
        ws = _workspace(lib, q.device, M, Lk, H)
        d = _desc(q, k)
        rc = lib.foho_sdpa_bwd(ctypes.byref(d), L.vp(q.data_ptr()), L.vp(k.data_ptr()), L.vp(v.data_ptr()), L.vp(out.data_ptr()), L.vp(nlse.data_ptr()),
                               L.vp(go.data_ptr()), L.vp(gq.data_ptr()), L.vp(gk.data_ptr()), L.vp(gv.data_ptr()), L.vp(ws.data_ptr()),
                               ctypes.c_size_t(ws.numel()), _stream(g))
        if rc != 0:
            lib.foho_geo_last_error.restype = ctypes.c_char_p
            raise L.FohoError(f"foho_sdpa_bwd failed ({rc}): {lib.foho_geo_last_error().decode()}")
        return gq.view(B, M, H, 64).transpose(1, 2), gk.view(B, Lk, H, 64).transpose(1, 2), gv.view(B, Lk, H, 64).transpose(1, 2)


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
    if not (torch.is_tensor(k) and torch.is_tensor(v) and k.is_cuda and v.is_cuda and k.device == q.device == v.device and k.dim() == 4 and v.dim() == 4):
        return False
    # the limits of sdpa_args (csrc/foho_geo.hip): operands within 32-bit buffer offsets; views with a zero / tiny stride (expand()) are
    # copied by _HipSdpaFn.forward (not _in_place), which makes them (B, N, H, 64) contiguous -- that copy must fit, too
    for t in (q, k):
        if t.shape[1] * t.shape[2] * 64 * 2 >= 2 ** 31:
            return False
    return scale is None or abs(float(scale) - 0.125) < 1e-12


hits = 0          # eligible calls served by the HIP kernels through hip_sdpa() (diagnostics: was the patch reached at all?)
# Modules that bind the function at import time (`scaled_dot_product_attention = nn.functional.scaled_dot_product_attention` at module
# level -- what hy3dgen's attention_blocks.py / attention_processors.py are believed to do; the files are not in the reference tree) never
# look F.scaled_dot_product_attention up again: hip_sdpa() rebinds these module-level names too, when the modules are importable.
PATCH_MODULES = ("hy3dgen.shapegen.models.autoencoders.attention_blocks", "hy3dgen.shapegen.models.autoencoders.attention_processors")


def attention(q, k, v):
    """softmax(q k^T / 8) v per head: q (B, H, M, 64), k / v (B, H, L, 64), fp16 on the GPU -> (B, H, M, 64); differentiable w.r.t. all three."""
    if not eligible(q, k, v):
        raise L.FohoError("sdpa.attention: fp16 CUDA tensors (B, H, N, 64) with 1..16 heads and a multiple of 128 keys are supported")
    return _HipSdpaFn.apply(q, k, v)


@contextlib.contextmanager
def hip_sdpa(backward=None):
    """Inside the context `torch.nn.functional.scaled_dot_product_attention` sends eligible calls to the HIP kernels and everything
    else to torch's own implementation.  (A module that bound the function at import time -- `from torch.nn.functional import
    scaled_dot_product_attention` -- keeps torch's.)  backward: "torch" / "hip" sets `backward_route` for the forwards recorded inside.
    The patch is process-wide while the context is open (other threads calling the function meanwhile get the same dispatch: eligible
    calls to the HIP kernels, the rest to torch -- harmless, but the context is not meant to be nested or entered concurrently)."""
    global backward_route
    orig = F.scaled_dot_product_attention
    saved_route = backward_route
    if backward is not None:
        backward_route = backward

    def dispatch(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
        global hits
        if eligible(query, key, value, attn_mask, dropout_p, is_causal, scale, **kw):
            hits += 1
            return _HipSdpaFn.apply(query, key, value)
        return orig(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale, **kw)

    import sys
    rebound = []          # (module, previous value) of the import-time bindings of the SAME function
    for name in PATCH_MODULES:
        mod = sys.modules.get(name)
        if mod is not None and getattr(mod, "scaled_dot_product_attention", None) is orig:
            rebound.append((mod, orig))
            mod.scaled_dot_product_attention = dispatch
    F.scaled_dot_product_attention = dispatch
    try:
        yield
    finally:
        F.scaled_dot_product_attention = orig
        for mod, prev in rebound:
            mod.scaled_dot_product_attention = prev
        backward_route = saved_route
