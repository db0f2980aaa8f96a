"""The geometry decoder of latent2sdf on the matrix cores (foho_geo_decode_fwd, csrc/foho_geo.hip) against the torch module
of the same shape (standins._GeoDecoder, float32 arithmetic on the same fp16-representable weights): the GEMM with each
epilogue, the attention kernel, and the whole chain at a reduced and at the full Hunyuan3D-2 shape (3072 x 1024 latent
tokens, 16 heads, hidden 4096, 65^3 queries).  fp16 storage / fp32 accumulation: tolerance 2e-3 of the output's scale."""
import ctypes
import os
import math

import numpy as np
import pytest
import torch

gpu = pytest.mark.gpu


def _lib():
    from followmyhold_amd import _lib as L
    lib = L.lib()
    lib.foho_geo_last_error.restype = ctypes.c_char_p
    return L, lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def test_geo_workspace_query_and_argument_checks_run_without_a_gpu():
    from followmyhold_amd.geo_decode import FohoGeoWeights
    L, lib = _lib()
    lib.foho_geo_workspace_bytes.restype = ctypes.c_size_t
    lib.foho_geo_workspace_bytes.argtypes = [ctypes.POINTER(FohoGeoWeights), ctypes.c_int32]
    w = FohoGeoWeights()
    w.width, w.heads, w.n_latents, w.hidden, w.n_freqs = 1024, 16, 3072, 4096, 8
    for name, typ in FohoGeoWeights._fields_:
        if typ is L.vp:
            setattr(w, name, 1)       # non-null: the size query only looks at the shape
    n = lib.foho_geo_workspace_bytes(ctypes.byref(w), 16384)
    # per row: the block's activations + the folded LayerNorms' statistics (16 parts x 4 floats + 2 floats); per weight set: K / V / V^T
    # of the tokens, fc1's weights with ln_2's gain folded in and the folded vectors
    per_row = 2 * (64 + 3 * 1024 + 4096) + 16 * 16 + 8
    fixed = 3072 * 1024 * 2 * 3 + 4096 * 1024 * 2 + 4096 * 8 + 1024 * 8 + 8
    assert 16384 * per_row + fixed <= n <= 16384 * per_row + fixed + 16384
    w.heads = 8                        # head dimension 128: refused
    assert lib.foho_geo_workspace_bytes(ctypes.byref(w), 16384) == 0 and b"head dimension" in lib.foho_geo_last_error()
    w.heads, w.n_latents = 16, 100
    assert lib.foho_geo_workspace_bytes(ctypes.byref(w), 16384) == 0 and b"n_latents" in lib.foho_geo_last_error()
    lib.foho_geo_decode_fwd.restype = ctypes.c_int
    assert lib.foho_geo_decode_fwd(None, None, ctypes.c_int64(0), None, 0, None, ctypes.c_size_t(0), None) == -1
    # the backward's sizes: 18.1 KB of kept activations per query row (rounded up to whole row blocks), a workspace that holds one
    # row block's scratch + the partial sums of the K / V gradient; argument errors come back as codes, not as faults
    w.n_latents = 3072
    for fn in (lib.foho_geo_bwd_workspace_bytes, lib.foho_geo_saved_bytes):
        fn.restype = ctypes.c_size_t
    lib.foho_geo_bwd_workspace_bytes.argtypes = [ctypes.POINTER(FohoGeoWeights), ctypes.c_int32]
    lib.foho_geo_saved_bytes.argtypes = [ctypes.POINTER(FohoGeoWeights), ctypes.c_int32, ctypes.c_int64]
    per_row_saved = 2 * (5 * 1024 + 4096) + 16 * 4
    n = lib.foho_geo_saved_bytes(ctypes.byref(w), 16384, 65 ** 3)
    assert 17 * 16384 * per_row_saved <= n <= 17 * 16384 * (per_row_saved + 64) + 17 * 8192
    nb = lib.foho_geo_bwd_workspace_bytes(ctypes.byref(w), 16384)
    part = 3072 * 2048 * 4
    assert nb >= 16384 * 2 * (64 + 7 * 1024 + 2 * 4096) + part and (nb - 16384 * 2 * (64 + 9 * 1024 + 2 * 4096)) % part < part
    lib.foho_geo_decode_bwd.restype = ctypes.c_int
    assert lib.foho_geo_decode_bwd(None, None, ctypes.c_int64(0), None, None, 0, None, ctypes.c_size_t(0), None, ctypes.c_size_t(0), None,
                                   ctypes.c_size_t(0), None) == -1
    w.n_latents = 3072 + 64              # fine for the forward, not for the backward's 128-key blocks
    assert lib.foho_geo_workspace_bytes(ctypes.byref(w), 16384) > 0
    one = ctypes.c_void_p(1)
    assert lib.foho_geo_decode_bwd(ctypes.byref(w), one, ctypes.c_int64(8), one, one, 16384, one, ctypes.c_size_t(1 << 40), one, ctypes.c_size_t(1 << 40), None,
                                   ctypes.c_size_t(0), None) != 0 and b"multiple of 128" in lib.foho_geo_last_error()


@gpu
@pytest.mark.parametrize("M,N,K,mode", [(300, 256, 64, "plain"), (1000, 1024, 1024, "scale"), (777, 512, 256, "gelu"), (4097, 256, 1024, "resid")])
def test_gemm_epilogues_against_torch(M, N, K, mode):
    L, lib = _lib()
    g = torch.Generator().manual_seed(M + N)
    A = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).half().cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).half().cuda() if mode == "resid" else None
    C = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
    scale = 0.18 if mode == "scale" else 1.0
    lib.foho_geo_gemm.restype = ctypes.c_int
    rc = lib.foho_geo_gemm(_p(A), _p(W), _p(b), _p(R) if R is not None else None, _p(C), M, N, K, int(mode == "gelu"), ctypes.c_float(scale),
                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.foho_geo_last_error()
    torch.cuda.synchronize()
    ref = A.float() @ W.float().T + b
    if mode == "gelu":
        ref = torch.nn.functional.gelu(ref)
    ref = ref * scale
    if R is not None:
        ref = ref.half().float() + R.float()
    assert torch.isfinite(C.float()).all()
    err = (C.float() - ref).abs().max().item()
    assert err <= 2e-3 * max(ref.abs().max().item(), 1.0), (err, ref.abs().max().item())


@gpu
@pytest.mark.parametrize("M,N,K,resid", [(3072, 1024, 1024, True), (3072, 1024, 4096, False), (1000, 256, 256, True), (95, 128, 512, False), (2304, 512, 512, True)])
def test_gemm_kernel_variants_give_the_same_bits(M, N, K, resid):
    """Every kernel variant of foho_geo_gemm (`gelu | 2` 128 x 128, `| 4` lock-step 256 x 256, `| 8` deep ring, `| 16` phased, `| 32` fill + matrix
    waves, `| 64` phased on 192-row tiles -- ragged last tiles included) accumulates a K tile after the other in fp32 on the same matrix
    instruction and shares one epilogue: the outputs are equal bit for bit, and right against torch."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).half().cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).half().cuda() if resid else None
    lib.foho_geo_gemm.restype = ctypes.c_int
    outs = {}
    for flag in (0, 2, 4, 8, 16, 32, 64):
        C = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
        rc = lib.foho_geo_gemm(_p(A), _p(W), _p(b), _p(R) if resid else None, _p(C), M, N, K, flag, ctypes.c_float(1.0),
                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.foho_geo_last_error()
        torch.cuda.synchronize()
        outs[flag] = C
    ref = A.float() @ W.float().T + b
    if resid:
        ref = ref.half().float() + R.float()
    for flag, C in outs.items():
        assert torch.isfinite(C.float()).all(), flag
        err = (C.float() - ref).abs().max().item()
        assert err <= 2e-3 * max(ref.abs().max().item(), 1.0), (flag, err)
        assert torch.equal(C, outs[2]), flag


@gpu
@pytest.mark.parametrize("M,Lk,heads", [(256, 64, 2), (1000, 256, 4), (700, 3072, 16)])
def test_attention_against_torch(M, Lk, heads):
    L, lib = _lib()
    W = heads * 64
    g = torch.Generator().manual_seed(Lk)
    q = torch.randn(M, W, generator=g)
    kv = torch.randn(Lk, 2 * W, generator=g)
    kv[5, :64] *= 4.0                                   # one key that dominates some rows: the deferred rescale must fire late, too
    kv[Lk - 3, 64 * (heads - 1):64 * heads] *= 4.0
    qs = (q * (math.log2(math.e) / 8.0)).half().cuda()
    kvh = kv.half().cuda()
    O = torch.full((M, W), float("nan"), dtype=torch.float16, device="cuda")
    vt = torch.empty(W * Lk, dtype=torch.float16, device="cuda")
    lib.foho_geo_attention.restype = ctypes.c_int
    rc = lib.foho_geo_attention(_p(qs), _p(kvh), _p(vt), _p(O), M, Lk, heads, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.foho_geo_last_error()
    torch.cuda.synchronize()
    qf = (qs.float() * (8.0 / math.log2(math.e))).view(M, heads, 64).transpose(0, 1)          # the values the kernel saw
    kf = kvh.float()[:, :W].view(Lk, heads, 64).transpose(0, 1)
    vf = kvh.float()[:, W:].view(Lk, heads, 64).transpose(0, 1)
    ref = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf).transpose(0, 1).reshape(M, W)
    assert torch.isfinite(O.float()).all()
    err = (O.float() - ref).abs().max().item()
    assert err <= 2e-3 * max(ref.abs().max().item(), 1.0), (err, ref.abs().max().item())


def _decoder(width, heads, n_lat, hidden_ratio=4, seed=0):
    from followmyhold_amd import standins
    torch.manual_seed(seed)
    vae = standins.StandInShapeVAE(num_latents=n_lat, embed_dim=16, width=width, heads=heads, layers=1, num_freqs=8)
    dec = vae.geo_decoder
    with torch.no_grad():          # weights the fp16 kernels can represent exactly; LayerNorm parameters off their defaults
        for p in dec.parameters():
            p.copy_(p.half().float())
        for ln in (dec.block.ln_q, dec.block.ln_kv, dec.block.ln_2, dec.ln_post):
            ln.weight.add_(0.1 * torch.randn_like(ln.weight))
            ln.bias.add_(0.1 * torch.randn_like(ln.bias))
    return dec.cuda().eval()


@gpu
# (8192-row blocks at width 1024: c_proj and fc2 -- EP_RESID | EP_STATS, EP_RESID | EP_LOGIT -- are one round of 256-row tiles on half the chip and
# go to k_geo_gemm8p's 192-row tiles, whose last epilogue part is 32 rows; 3000 rows: fc1 + GELU behind the folded LayerNorm on them)
@pytest.mark.parametrize("width,heads,n_lat,n_q,chunk", [(256, 4, 256, 5000, 2048), (1024, 16, 3072, 20000, 16384), (1024, 16, 3072, 11000, 8192),
                                                         (1024, 16, 3072, 3000, 16384)])
def test_decoder_chain_against_the_torch_module(width, heads, n_lat, n_q, chunk):
    """Fourier embedding -> query projection -> cross attention -> MLP -> LayerNorm -> logit, all queries in one call (row
    blocks of `chunk`, the last one ragged), against the float32 torch module on the same fp16-rounded inputs."""
    from followmyhold_amd.geo_decode import HipGeoDecoder
    dec = _decoder(width, heads, n_lat)
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(1, n_lat, width, generator=g).half().cuda()
    q = (torch.rand(1, n_q, 3, generator=g) * 2.2 - 1.1).half().cuda()
    hip = HipGeoDecoder.from_module(dec, chunk_rows=chunk)
    out = hip(q.float(), lat)
    torch.cuda.synchronize()
    assert out.shape == (1, n_q, 1) and out.dtype == torch.float16
    with torch.no_grad():
        ref = dec(q, lat.float())
        learned = (ref - (dec.radius - q.float().norm(dim=-1, keepdim=True)) * dec.sharpness) / dec.gain
    err = (out.float() - ref).abs().max().item()
    # the tolerance refers to the LEARNED part (the analytic prior is exact on both sides)
    assert learned.abs().max().item() > 0.1
    assert err <= 2e-3 * max(learned.abs().max().item(), 1.0) * dec.gain + 1e-3 * ref.abs().max().item(), (err, learned.abs().max().item())
    # a second call on the same tokens reuses K / V; new tokens are projected again
    out2 = hip(q.float(), lat)
    assert torch.equal(out, out2)
    # the forward folds ln_2 into fc1 and ln_post + output_proj into fc2's epilogue; the chain with LayerNorm KERNELS (what the
    # backward routes recompute) gives the same logits up to one fp16 rounding of an intermediate, and is as close to torch
    hip.ln_fuse = False
    try:
        out_ln = hip(q.float(), lat)
        torch.cuda.synchronize()
    finally:
        hip.ln_fuse = True
    scale = max(learned.abs().max().item(), 1.0) * dec.gain
    assert not torch.equal(out, out_ln) or width < 0          # (they are different computations)
    # (the second term: the result is handed back in fp16, the analytic prior included)
    assert (out.float() - out_ln.float()).abs().max().item() <= 1e-3 * scale + 1e-3 * ref.abs().max().item(), ((out.float() - out_ln.float()).abs().max().item(), scale)
    assert (out_ln.float() - ref).abs().max().item() <= 2e-3 * scale + 1e-3 * ref.abs().max().item()
    # ... also when they live where the old ones did (the caching allocator reuses a freed latent's block: no reuse by address)
    lat_b = lat.clone()
    out_b = hip(q.float(), lat_b)
    ptr = lat_b.data_ptr()
    del lat_b
    lat_c = (lat.float() * 0.5).half()
    if lat_c.data_ptr() == ptr:                       # what the allocator usually does
        with torch.no_grad():
            ref_c = dec(q, lat_c.float())
        assert (hip(q.float(), lat_c).float() - ref_c).abs().max().item() <= 2e-3 * max(learned.abs().max().item(), 1.0) * dec.gain + 1e-3 * ref_c.abs().max().item()
    lat_d = lat.clone()
    hip(q.float(), lat_d)
    lat_d.mul_(1.25)                                  # same object, new version: projected again
    with torch.no_grad():
        ref_d = dec(q, lat_d.float())
    assert (hip(q.float(), lat_d).float() - ref_d).abs().max().item() <= 2e-3 * max(learned.abs().max().item(), 1.0) * dec.gain + 1e-3 * ref_d.abs().max().item()
    lat2 = (lat.float() * 1.5).half()
    out3 = hip(q.float(), lat2)
    with torch.no_grad():
        ref3 = dec(q, lat2.float())
    assert (out3.float() - ref3).abs().max().item() <= 2e-3 * max(learned.abs().max().item(), 1.0) * dec.gain + 1e-3 * ref3.abs().max().item()


@gpu
def test_folded_layernorms_do_not_cancel_on_rows_with_a_large_common_offset():
    """ln_2 inside fc1 is rstd (x (W gamma)^T) - rstd mean s + b': two terms of the size of the row's MEAN.  s is the row sum of the
    ROUNDED folded weights, so a common offset of a row cancels exactly whatever the rounding of W gamma was; with the residual stream
    30 standard deviations off zero (query_proj's bias) the folded chain still agrees with the chain that runs the LayerNorm kernels --
    both see the same fp16 x1 -- and with the float32 module as closely as that chain does."""
    from followmyhold_amd.geo_decode import HipGeoDecoder
    dec = _decoder(256, 4, 256)
    with torch.no_grad():
        dec.query_proj.bias.add_(30.0 * dec.query_proj.weight.std() * (dec.query_proj.in_features ** 0.5))
        dec.query_proj.bias.copy_(dec.query_proj.bias.half().float())
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, 256, 256, generator=g).half().cuda()
    q = (torch.rand(1, 4000, 3, generator=g) * 2.2 - 1.1).half().cuda()
    hip = HipGeoDecoder.from_module(dec, chunk_rows=2048)
    out = hip(q.float(), lat).float()
    hip.ln_fuse = False
    try:
        out_ln = hip(q.float(), lat).float()
    finally:
        hip.ln_fuse = True
    with torch.no_grad():
        q32 = q.float()
        emb = (q32[..., None] * dec.freqs).flatten(-2)
        x0 = dec.query_proj(torch.cat([q32, emb.sin(), emb.cos()], -1))
        ref = dec(q, lat.float())
    assert (x0.mean(-1).abs() / x0.std(-1)).median().item() > 5              # the rows really sit far off zero
    scale = max((ref - (dec.radius - q.float().norm(dim=-1, keepdim=True)) * dec.sharpness).abs().max().item() / dec.gain, 1.0) * dec.gain
    d_fold, d_ln, d_between = (out - ref).abs().max().item(), (out_ln - ref).abs().max().item(), (out - out_ln).abs().max().item()
    assert torch.isfinite(out).all() and d_between <= 2e-3 * scale + 1e-3 * ref.abs().max().item(), (d_between, scale)
    assert d_fold <= 1.5 * d_ln + 2e-3 * scale, (d_fold, d_ln, scale)


@gpu
@pytest.mark.parametrize("width,heads,n_lat,n_q,chunk", [(256, 4, 256, 5000, 2048), (1024, 16, 3072, 20000, 16384), (1024, 16, 3072, 3000, 16384),
                                                         (256, 4, 128, 70, 2048), (256, 4, 256, 4097, 4096)])
def test_decoder_backward_against_torch_autograd(width, heads, n_lat, n_q, chunk):
    """foho_geo_decode_bwd: d sum(g . logits) / d latents through the HIP chain (forward recomputed per row block, K / V gradients
    accumulated over the blocks, LayerNorm + K/V projection of the tokens by torch autograd) against float32 autograd through the
    torch module.  fp16 storage of every activation and of dS / dP: 1 % of the gradient's scale, direction to 1e-4."""
    from followmyhold_amd.geo_decode import HipGeoDecoder
    dec = _decoder(width, heads, n_lat)
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, n_lat, width, generator=g).half().cuda()
    q = (torch.rand(1, n_q, 3, generator=g) * 2.2 - 1.1).half().cuda()
    go = torch.randn(1, n_q, 1, generator=g).cuda()
    hip = HipGeoDecoder.from_module(dec, chunk_rows=chunk)
    lat_h = lat.clone().requires_grad_(True)
    out = hip(q.float(), lat_h)
    assert out.requires_grad and out.shape == (1, n_q, 1)
    (out.float() * go).sum().backward()
    lat_r = lat.float().requires_grad_(True)
    ref = dec(q, lat_r)
    (ref * go).sum().backward()
    torch.cuda.synchronize()
    assert (out.float() - ref).abs().max().item() <= 2e-3 * max(ref.abs().max().item(), 1.0) + 2e-3 * dec.gain * 8
    gh, gr = lat_h.grad.float(), lat_r.grad
    assert torch.isfinite(gh).all() and gr.abs().max().item() > 0
    err = (gh - gr).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(gh.flatten(), gr.flatten(), dim=0).item()
    assert err <= 1e-2 * gr.abs().max().item() and cos >= 1 - 1e-4, (err, gr.abs().max().item(), cos)
    # the K / V gradient on its own: from kept activations and from the recomputed forward -- the same kernels on the same
    # numbers, no atomics anywhere (partial sums per (split, key block), fixed order): bitwise equal, bitwise repeatable
    kv = hip.kv_of(lat).detach()
    hip.set_kv(kv)
    gkv = hip.decode_bwd(q.float(), go)
    gkv2 = hip.decode_bwd(q.float(), go)
    assert torch.isfinite(gkv).all() and torch.equal(gkv, gkv2)
    out_k, saved = hip.decode_keep(q.float())
    # (the keeping forward is another instantiation of the GEMM epilogue: where the compiler folds "times scale, to fp16" into
    # one v_fma_mixlo_f16 it rounds once, elsewhere twice -- a handful of logits differ in the last fp16 bit of an activation)
    hip.ln_fuse = False          # ... against the plain forward in the same form (LayerNorm kernels) ...
    try:
        out_plain = hip.decode(q.float())
    finally:
        hip.ln_fuse = True
    assert (out_k - out_plain).abs().max().item() <= 1e-3 * dec.gain
    # ... and against the forward the loop runs (LayerNorms folded into the GEMMs: another rounding of one intermediate)
    assert (out_k - hip.decode(q.float())).abs().max().item() <= 2e-3 * dec.gain * 8
    assert torch.equal(hip.decode_bwd(q.float(), go, saved), gkv)
    hip.keep_activations = False
    lat_n = lat.clone().requires_grad_(True)
    (hip(q.float(), lat_n).float() * go).sum().backward()
    assert torch.equal(lat_n.grad, lat_h.grad)


@gpu
def test_latent2sdf_uses_the_hip_decoder_with_and_without_gradients():
    """pipeline.latent2sdf: with geo_decode.install(vae) the decodes run on the HIP decoder -- same grid to fp16 accuracy --,
    and a decode under autograd (PL:1391-1393, 1507-1509) carries the gradient to the latent through foho_geo_decode_bwd."""
    from followmyhold_amd import geo_decode, pipeline, standins
    from followmyhold_amd.facade import generate_dense_grid_points
    torch.manual_seed(0)
    vae = standins.StandInShapeVAE(num_latents=128, embed_dim=8, width=128, heads=2, layers=1, num_freqs=8).cuda().eval()
    xyz, gsz, _ = generate_dense_grid_points(np.array([-1.1] * 3), np.array([1.1] * 3), 4, "ij")       # 17^3 points
    xyz = torch.as_tensor(xyz, dtype=torch.float32)
    lat = torch.randn(1, 128, 8, device="cuda")
    wgt = torch.randn(1, *[int(v) for v in gsz], device="cuda")
    with torch.no_grad():
        ref = pipeline.latent2sdf(lat, xyz, gsz, vae, "cuda")
    lat_r = lat.clone().requires_grad_(True)
    (pipeline.latent2sdf(lat_r, xyz, gsz, vae, "cuda") * wgt).sum().backward()
    geo_decode.install(vae)
    with torch.no_grad():
        got = pipeline.latent2sdf(lat, xyz, gsz, vae, "cuda")
    assert got.shape == ref.shape and (got - ref).abs().max().item() <= 5e-3 * ref.abs().max().item()
    assert not torch.equal(got, ref)                                    # it did take the other path
    lat_h = lat.clone().requires_grad_(True)
    sdf = pipeline.latent2sdf(lat_h, xyz, gsz, vae, "cuda")
    assert sdf.requires_grad and (sdf.detach() - ref).abs().max().item() <= 5e-3 * ref.abs().max().item()
    (sdf * wgt).sum().backward()
    err = (lat_h.grad - lat_r.grad).abs().max().item()
    assert err <= 2e-2 * lat_r.grad.abs().max().item(), (err, lat_r.grad.abs().max().item())


@gpu
def test_backward_reinstalls_its_tokens_only_when_the_workspace_served_others_in_between():
    """_GeoDecodeFn.backward: the workspace still holds this forward's K / V^T when nothing else was decoded since (the guidance loop:
    four launches less per iteration); a decode of OTHER tokens in between bumps the epoch and the backward installs its own again --
    either way the gradient is the one of the undisturbed call, bit for bit."""
    from followmyhold_amd import geo_decode, pipeline, standins
    from followmyhold_amd.facade import generate_dense_grid_points
    torch.manual_seed(1)
    vae = standins.StandInShapeVAE(num_latents=128, embed_dim=8, width=128, heads=2, layers=1, num_freqs=8).cuda().eval()
    xyz, gsz, _ = generate_dense_grid_points(np.array([-1.1] * 3), np.array([1.1] * 3), 4, "ij")
    xyz = torch.as_tensor(xyz, dtype=torch.float32)
    lat, other = torch.randn(1, 128, 8, device="cuda"), torch.randn(1, 128, 8, device="cuda")
    wgt = torch.randn(1, *[int(v) for v in gsz], device="cuda")
    dec = geo_decode.install(vae)
    calls = []
    real = dec.set_kv
    dec.set_kv = lambda kv: (calls.append(1), real(kv))[1]

    def grad_of(disturb):
        calls.clear()
        a = lat.clone().requires_grad_(True)
        sdf = pipeline.latent2sdf(a, xyz, gsz, vae, "cuda")
        if disturb:
            with torch.no_grad():
                pipeline.latent2sdf(other, xyz, gsz, vae, "cuda")       # prepare() of other tokens: the workspace is theirs now
        (sdf * wgt).sum().backward()
        return a.grad.clone(), len(calls)

    g0, n0 = grad_of(False)
    g1, n1 = grad_of(True)
    assert n0 == 1 and n1 == 2, (n0, n1)            # forward only | forward + the backward's re-install
    assert torch.isfinite(g0).all() and g0.abs().max().item() > 0 and torch.equal(g0, g1)


from followmyhold_amd.standins import Hy3dgenLayoutDecoder as _Hy3dLikeDecoder   # the hy3dgen attribute layout, restated


@gpu
@pytest.mark.parametrize("qk_norm", [False, True])
def test_decoder_adopts_a_module_laid_out_like_hy3dgen(qk_norm):
    """geo_decode._parts on the hy3dgen layout: bias-free c_q / c_kv, K and V interleaved per head in c_kv's rows, frequencies
    without pi, no analytic prior, with and without qk_norm (LayerNorm over the head dimension of q -- in the q GEMM's epilogue --
    and of k) -- forward and latent gradient against the module itself."""
    from followmyhold_amd.geo_decode import HipGeoDecoder
    from followmyhold_amd import _lib as L
    torch.manual_seed(4)
    dec = _Hy3dLikeDecoder(256, 4, qk_norm=qk_norm).cuda().eval()
    with torch.no_grad():
        for p in dec.parameters():
            p.copy_(p.half().float())
        if qk_norm:
            for nrm in (dec.cross_attn_decoder.attn.attention.q_norm, dec.cross_attn_decoder.attn.attention.k_norm):
                nrm.weight.add_(0.2 * torch.randn_like(nrm.weight))
                nrm.bias.add_(0.2 * torch.randn_like(nrm.bias))
    # one eps per LayerNorm (hy3dgen: 1e-6 in the block, torch's default on ln_post); made large and distinct here so that a decoder
    # which applied one eps to all four would miss the tolerance
    blk = dec.cross_attn_decoder
    blk.ln_1.eps, blk.ln_2.eps, blk.ln_3.eps, dec.ln_post.eps = 3e-2, 1e-1, 5e-2, 2e-1
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(1, 384, 256, generator=g).half().cuda()
    q = (torch.rand(1, 3000, 3, generator=g) * 2.0 - 1.0).half().cuda()
    go = torch.randn(1, 3000, 1, generator=g).cuda()
    hip = HipGeoDecoder.from_module(dec, chunk_rows=2048)
    lat_h = lat.clone().requires_grad_(True)
    out = hip(q.float(), lat_h)
    (out.float() * go).sum().backward()
    lat_r = lat.float().requires_grad_(True)
    ref = dec(q, lat_r)
    (ref * go).sum().backward()
    scale = ref.abs().max().item()
    assert scale > 0.1 and (out.float() - ref).abs().max().item() <= 3e-3 * max(scale, 1.0), ((out.float() - ref).abs().max().item(), scale)
    gh, gr = lat_h.grad.float(), lat_r.grad
    cos = torch.nn.functional.cosine_similarity(gh.flatten(), gr.flatten(), dim=0).item()
    assert (gh - gr).abs().max().item() <= 1e-2 * gr.abs().max().item() and cos >= 1 - 1e-4, ((gh - gr).abs().max().item(), gr.abs().max().item(), cos)
    # the no-gradient route (foho_geo_prepare normalises K itself) gives the same logits as the autograd route
    with torch.no_grad():
        out_ng = hip(q.float(), lat)
    assert (out_ng.float() - out.detach().float()).abs().max().item() <= 2e-3 * max(scale, 1.0)
    dec.cross_attn_decoder.attn.attention.q_norm = torch.nn.GroupNorm(4, 64).cuda()             # another kind of norm: refused, not mis-decoded
    with pytest.raises(L.FohoError):
        HipGeoDecoder.from_module(dec)
    # ... and so are the decoder variants these kernels do not implement
    for spoil in ("latents_proj", "no_ln_post", "two_channels"):
        d2 = _Hy3dLikeDecoder(256, 4, qk_norm=False).cuda().eval()
        if spoil == "latents_proj":
            d2.latents_proj = torch.nn.Linear(256, 256).cuda()
        elif spoil == "no_ln_post":
            d2.ln_post = None
        else:
            d2.output_proj = torch.nn.Linear(256, 2).cuda()
        with pytest.raises(L.FohoError):
            HipGeoDecoder.from_module(d2)


@gpu
def test_from_hy3dgen_installs_the_hip_decoder():
    """GuidedShapePipeline.from_hy3dgen (what the guidance stage builds from an installed Hunyuan3D-2 pipeline) hands the ShapeVAE's
    geometry decoder to the HIP kernels by default, keeps the torch module on request, and fails loudly for a decoder outside the
    kernels' shapes instead of quietly staying on torch."""
    import types
    from followmyhold_amd import _lib as L, standins
    from followmyhold_amd.pipeline import GuidedShapePipeline

    def fake(width, heads):
        vae = standins.StandInShapeVAE(num_latents=128, embed_dim=8, width=width, heads=heads, layers=1, num_freqs=8)
        sch = types.SimpleNamespace(config=types.SimpleNamespace(num_train_timesteps=1000, shift=1.0))
        return types.SimpleNamespace(vae=vae, model=standins.StandInDiT(embed_dim=8), scheduler=sch, conditioner=standins.StandInConditioner(),
                                     image_processor=standins.StandInImageProcessor(), device="cuda", dtype=torch.float32)

    pipe = GuidedShapePipeline.from_hy3dgen(fake(128, 2))
    assert getattr(pipe.vae, "hip_geo", None) is not None
    assert getattr(GuidedShapePipeline.from_hy3dgen(fake(128, 2), hip_geo_decoder=False).vae, "hip_geo", None) is None
    with pytest.raises(L.FohoError):
        GuidedShapePipeline.from_hy3dgen(fake(96, 2))          # head dimension 48


@gpu
@pytest.mark.parametrize("width,heads,n_lat,n_q,chunk,fill", [(256, 4, 256, 5000, 2048, 0.07), (256, 4, 128, 4097, 4096, 0.5), (256, 4, 256, 9000, 2048, 1.0),
                                                              (256, 4, 128, 3000, 2048, 0.0), (1024, 16, 3072, 30000, 16384, 0.08)])
def test_active_row_backward_equals_the_dense_backward(width, heads, n_lat, n_q, chunk, fill):
    """foho_geo_decode_bwd_rows (device-side compaction of the rows with grad != 0, chain recomputed and back-propagated for them
    only) against foho_geo_decode_bwd over all rows: a zero row adds exactly zero, so the two agree to the accumulation order of
    the partial sums; with every row active the compaction is the identity and the results are bitwise equal.  No atomics in the
    compaction or the chain: bitwise repeatable.  The device-resident statistics report the number of active rows."""
    from followmyhold_amd.geo_decode import HipGeoDecoder
    dec = _decoder(width, heads, n_lat)
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, n_lat, width, generator=g).half().cuda()
    q = (torch.rand(1, n_q, 3, generator=g) * 2.2 - 1.1).half().float().cuda()
    go = torch.randn(n_q, generator=g)
    keep = torch.rand(n_q, generator=g) < fill
    if 0.0 < fill < 1.0:
        keep[-1] = True                       # the very last row of a ragged block is active
        keep[:chunk // 2] = False             # ... and a long stretch of leading zeros
    go = torch.where(keep, go, torch.zeros(())).cuda()
    hip = HipGeoDecoder.from_module(dec, chunk_rows=chunk)
    hip.set_kv(hip.kv_of(lat).detach())
    dense = hip.decode_bwd(q, go)
    rows = hip.decode_bwd_rows(q, go)
    stats = hip.last_row_stats.cpu().tolist()
    assert stats == [int(keep.sum()), 0]
    assert torch.isfinite(rows).all()
    assert torch.equal(rows, hip.decode_bwd_rows(q, go))
    if fill == 1.0:
        assert torch.equal(rows, dense)
    elif fill == 0.0:
        assert not rows.any() and not dense.any()
    else:
        scale = dense.abs().max().item()
        cos = torch.nn.functional.cosine_similarity(rows.flatten().double(), dense.flatten().double(), dim=0).item()
        assert scale > 0 and (rows - dense).abs().max().item() <= 1e-3 * scale and cos >= 1 - 1e-6, ((rows - dense).abs().max().item(), scale, cos)
        # a capacity below the number of active rows drops the rows beyond it and says so (never silently)
        n_act = int(keep.sum())
        cap = n_act - 5
        short = hip.decode_bwd_rows(q, go, row_cap=cap)
        assert hip.last_row_stats.cpu().tolist() == [n_act, 5]
        last5 = torch.nonzero(go).flatten()[-5:]
        go2 = go.clone()
        go2[last5] = 0.0
        assert torch.equal(short, hip.decode_bwd_rows(q, go2))
    # through autograd: "rows" is the default route of a decode whose latents require grad; "keep" and "recompute" give the same gradient
    grads = {}
    for mode in ("rows", "keep", "recompute"):
        hip.backward_mode = mode
        lat_m = lat.clone().requires_grad_(True)
        (hip(q.unsqueeze(0) if q.dim() == 2 else q, lat_m).float().reshape(-1) * go).sum().backward()
        grads[mode] = lat_m.grad.float() if lat_m.grad is not None else torch.zeros_like(lat).float()
    sc = grads["recompute"].abs().max().item()
    for mode in ("rows", "keep"):
        assert (grads[mode] - grads["recompute"]).abs().max().item() <= 2e-3 * sc + 1e-12, mode


@gpu
def test_cached_query_side_gives_bitwise_equal_logits():
    """foho_geo_prepare_queries + foho_geo_decode_fwd_cached: the latent-independent half of the chain (embedding -> query_proj -> ln_1 ->
    c_q) computed once per grid; the same kernels on the same numbers => logits bitwise equal to foho_geo_decode_fwd, for several
    latents, with a ragged last block; a query tensor that changes (new version) or another tensor is not served from the cache."""
    from followmyhold_amd.geo_decode import HipGeoDecoder
    dec = _decoder(256, 4, 256)
    g = torch.Generator().manual_seed(2)
    q = (torch.rand(1, 5001, 3, generator=g) * 2.2 - 1.1).half().float().cuda()
    hip = HipGeoDecoder.from_module(dec, chunk_rows=2048)
    lats = [torch.randn(1, 256, 256, generator=g).half().cuda() for _ in range(2)]
    plain = [hip(q, lat).clone() for lat in lats]
    plain_g = hip(q, lats[0].clone().requires_grad_(True)).detach().clone()      # the autograd route (K / V by torch: not bitwise the no-grad route's)
    assert hip._cached_queries(q) is None
    hip.prepare_queries(q)
    assert hip._cached_queries(q) is not None and hip._cached_queries(q.clone()) is None
    for lat, ref in zip(lats, plain):
        assert torch.equal(hip(q, lat), ref)
    # under autograd, too (the "rows" route's forward is the cached forward)
    lat_g = lats[0].clone().requires_grad_(True)
    assert torch.equal(hip(q, lat_g).detach(), plain_g)
    q.mul_(0.5)                                    # same object, new version: the cache no longer applies
    assert hip._cached_queries(q) is None
    with torch.no_grad():
        ref = dec(q.half(), lats[0].float())
    assert (hip(q, lats[0]).float() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 2e-3
    # grid_queries: latent2sdf's way in -- one fp16-rounded device copy and one cache per grid tensor
    xyz = (torch.rand(3000, 3, generator=g) * 2.2 - 1.1)
    q1 = hip.grid_queries(xyz)
    assert hip.grid_queries(xyz) is q1 and hip._cached_queries(q1) is not None
    assert torch.equal(q1.reshape(-1, 3).cpu(), xyz.half().float())
    hip.query_cache_limit = 1000                   # a grid whose cache would not fit: decoded without one
    q2 = hip.grid_queries(xyz.clone())
    assert hip._cached_queries(q2) is None and torch.equal(hip(q2, lats[0]), hip(q1.clone(), lats[0]))


@gpu
def test_full_grid_forward_and_backward_against_fp32_torch_and_sparse_against_dense():
    """The Hunyuan3D-2 shape on the whole 65^3 grid (274 625 queries, 3072 x 1024 tokens, 16 heads, hidden 4096):
    (a) forward and dense backward against the float32 torch module (autograd chunk by chunk, PL:298-308's chunks of 8000);
    (b) with the gradient the guidance loop really produces -- dL/dSDF out of the FlexiCubes backward (PL:1507-1509, 1600), non-zero at
        the end points of the crossed grid edges only -- the active-row backward against the dense one and against fp32 autograd."""
    from followmyhold_amd import ops
    from followmyhold_amd.facade import generate_dense_grid_points
    from followmyhold_amd.geo_decode import HipGeoDecoder
    dec = _decoder(1024, 16, 3072)
    g = torch.Generator().manual_seed(7)
    lat = torch.randn(1, 3072, 1024, generator=g).half().cuda()
    xyz_np, gsz, _ = generate_dense_grid_points(np.full(3, -1.10), np.full(3, 1.10), octree_depth=5, octree_resolution=64, indexing="ij")
    xyz = torch.as_tensor(xyz_np, dtype=torch.float32, device="cuda")
    N = xyz.shape[0]
    assert N == 65 ** 3
    hip = HipGeoDecoder.from_module(dec)
    q = hip.grid_queries(xyz)

    def torch_fwd_bwd(go):
        lat_r = lat.float().requires_grad_(True)
        outs = []
        for s0 in range(0, N, 8000):
            o = dec(q[:, s0:s0 + 8000].half(), lat_r)
            (o.reshape(-1) * go[s0:s0 + 8000]).sum().backward()
            outs.append(o.detach().reshape(-1))
        return torch.cat(outs), lat_r.grad

    def hip_grad(go, mode):
        hip.backward_mode = mode
        lat_h = lat.float().clone().requires_grad_(True)      # a float32 leaf: its gradient is not rounded to fp16 (the values are ~1e-5)
        out = hip(q, lat_h)
        (out.float().reshape(-1) * go).sum().backward()
        return out.detach().float().reshape(-1), lat_h.grad.float()

    # (a) dense random gradient
    go = torch.randn(N, generator=g).cuda()
    ref, gref = torch_fwd_bwd(go)
    out, gh = hip_grad(go, "keep")
    learned = (ref - (dec.radius - q.reshape(-1, 3).norm(dim=-1)) * dec.sharpness) / dec.gain
    assert (out - ref).abs().max().item() <= 2.5e-3 * max(learned.abs().max().item(), 1.0) * dec.gain + 1e-3 * ref.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(gh.flatten(), gref.flatten(), dim=0).item()
    assert (gh - gref).abs().max().item() <= 1e-2 * gref.abs().max().item() and cos >= 1 - 1e-4, ((gh - gref).abs().max().item(), gref.abs().max().item(), cos)
    # (b) the gradient of the loop: SDF = -logits -> FlexiCubes -> a loss on the vertices -> dL/dSDF
    sdf = (-out).clone().requires_grad_(True)
    verts, faces, _ = ops.flexicubes(xyz, sdf, 64)
    assert verts.shape[0] > 1000
    (verts * torch.randn(verts.shape, generator=g).cuda()).sum().backward()
    go_s = -sdf.grad
    n_act = int((go_s != 0).sum())
    assert 0 < n_act < 0.2 * N                               # the surface touches a small part of the grid
    _, g_rows = hip_grad(go_s, "rows")
    assert hip.last_row_stats.cpu().tolist() == [n_act, 0]
    _, g_dense = hip_grad(go_s, "recompute")
    _, gref_s = torch_fwd_bwd(go_s)
    sc = g_dense.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(g_rows.flatten().double(), g_dense.flatten().double(), dim=0).item()
    assert (g_rows - g_dense).abs().max().item() <= 1e-3 * sc and cos >= 1 - 1e-6, ((g_rows - g_dense).abs().max().item(), sc, cos)
    # ... and the K / V gradients themselves (float32, what the two routes compute): same rows' contributions, another order of partial sums
    hip.set_kv(hip.kv_of(lat).detach())
    kv_rows, kv_dense = hip.decode_bwd_rows(q, go_s), hip.decode_bwd(q, go_s)
    sc = kv_dense.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(kv_rows.flatten().double(), kv_dense.flatten().double(), dim=0).item()
    assert sc > 0 and (kv_rows - kv_dense).abs().max().item() <= 1e-3 * sc and cos >= 1 - 1e-6, ((kv_rows - kv_dense).abs().max().item(), sc, cos)
    cos = torch.nn.functional.cosine_similarity(g_rows.flatten(), gref_s.flatten(), dim=0).item()
    assert (g_rows - gref_s).abs().max().item() <= 1e-2 * gref_s.abs().max().item() and cos >= 1 - 1e-4


@gpu
def test_active_row_backward_is_graph_capturable():
    """foho_geo_decode_bwd_rows has no host synchronisation -- the number of active rows never leaves the device --, so one captured
    hipGraph serves gradients with ANY number of active rows: captured once, replayed with three different sparsity patterns (and an
    all-zero gradient), each replay equal to the eager call on the same gradient, bit for bit."""
    from followmyhold_amd.geo_decode import HipGeoDecoder
    dec = _decoder(256, 4, 256)
    g = torch.Generator().manual_seed(21)
    lat = torch.randn(1, 256, 256, generator=g).half().cuda()
    n_q = 9000
    q = (torch.rand(1, n_q, 3, generator=g) * 2.2 - 1.1).half().float().cuda()
    hip = HipGeoDecoder.from_module(dec, chunk_rows=2048)
    hip.set_kv(hip.kv_of(lat).detach())
    grads = []
    for fill in (0.03, 0.4, 0.0, 1.0):
        go = torch.randn(n_q, generator=g)
        grads.append(torch.where(torch.rand(n_q, generator=g) < fill, go, torch.zeros(())).cuda())
    eager = [hip.decode_bwd_rows(q, gg).clone() for gg in grads]
    counts = [int((gg != 0).sum()) for gg in grads]
    g_static = grads[0].clone()
    hip.decode_bwd_rows(q, g_static)                    # warm-up outside the capture (workspaces allocated)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_static = hip.decode_bwd_rows(q, g_static)
        stats_static = hip.last_row_stats
    for gg, ref, n_act in zip(grads, eager, counts):
        g_static.copy_(gg)
        graph.replay()
        torch.cuda.synchronize()
        assert stats_static.tolist() == [n_act, 0]
        assert torch.equal(out_static, ref), n_act


@gpu
def test_unit_gemm_variants_beside_a_decode_on_another_thread_leave_its_bits_alone():
    """The library keeps no mode state (re-entrant): while one thread hammers `foho_geo_gemm` with every kernel-variant bit
    (`gelu | 2` = 128 x 128 tiles, `| 4` lock-step, `| 8` deep ring, `| 16` phased, `| 32` fill + matrix waves, `| 64` phased on 192-row tiles) on its own stream, decodes on another thread give the
    bits they give alone.  (Until round 5 the variant was a process-global the unit entry point set and reset around its launch.)"""
    import threading
    from followmyhold_amd.geo_decode import HipGeoDecoder
    L, lib = _lib()
    dec = _decoder(256, 4, 256)
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, 256, 256, generator=g).half().cuda()
    q = (torch.rand(1, 6000, 3, generator=g) * 2.2 - 1.1).half().cuda()
    hip = HipGeoDecoder.from_module(dec, chunk_rows=2048)      # 2048-row blocks: the chain's GEMMs take the 256 x 256 route by shape
    alone = hip(q.float(), lat).clone()
    torch.cuda.synchronize()
    M, N, K = 2304, 512, 512
    A = torch.randn(M, K, generator=g).half().cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    b = torch.zeros(N).cuda()
    ref = (A.float() @ W.float().t()).half()
    stop, errors, count = threading.Event(), [], [0]

    def hammer():
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            C = torch.empty(M, N, dtype=torch.float16, device="cuda")
            with torch.cuda.stream(st):
                while not stop.is_set():
                    for flag in (2, 4, 8, 16, 32, 64, 0):
                        rc = lib.foho_geo_gemm(_p(A), _p(W), _p(b), None, _p(C), M, N, K, flag, ctypes.c_float(1.0), ctypes.c_void_p(st.cuda_stream))
                        assert rc == 0, lib.foho_geo_last_error()
                        count[0] += 1
                    st.synchronize()
                    assert (C.float() - ref.float()).abs().max().item() <= 2e-2 * ref.float().abs().max().item()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    t = threading.Thread(target=hammer)
    t.start()
    try:
        for _ in range(30):
            lat2 = lat.clone()                       # a new tensor object: K / V are projected again, the whole chain runs
            out = hip(q.float(), lat2)
            torch.cuda.synchronize()
            assert torch.equal(out, alone)
    finally:
        stop.set()
        t.join()
    assert not errors, errors
    assert count[0] >= 10
