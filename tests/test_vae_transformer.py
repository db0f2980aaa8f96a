"""The ShapeVAE transformer of latent2sdf on the matrix cores (foho_vae_fwd / foho_vae_bwd, csrc/foho_vae.inc) against the torch module of
the same shape in float32 on the same fp16-representable weights (PL:295 `pred = vae(pred)`; PL:1391-1393, 1507-1509: its backward to the
latent): one layer and a stack, both module layouts (the stand-in's q / kv pair, hy3dgen's interleaved c_qkv with qk_norm), one image and
a batch, forward tokens and the gradient of the input.  fp16 storage / fp32 accumulation: tolerances of tests/test_geo_decode.py."""
import ctypes

import pytest
import torch

gpu = pytest.mark.gpu


def _standin(width, heads, layers, latents=256, embed=16, seed=0):
    from followmyhold_amd import standins
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    vae = standins.StandInShapeVAE(num_latents=latents, embed_dim=embed, width=width, heads=heads, layers=layers, num_freqs=4)
    torch.random.set_rng_state(g)
    return _round_weights(vae)


def _hy3d(width, heads, layers, latents=256, embed=16, qk_norm=True, qkv_bias=False, seed=0):
    from followmyhold_amd import standins
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    vae = standins.Hy3dgenLayoutShapeVAE(num_latents=latents, embed_dim=embed, width=width, heads=heads, layers=layers, num_freqs=4, qk_norm=qk_norm,
                                         qkv_bias=qkv_bias)
    with torch.no_grad():      # LayerNorm gains / biases away from their (1, 0) initialisation: the folding has something to fold
        for m in vae.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(0.3 * torch.randn_like(m.weight))
                m.bias.add_(0.2 * torch.randn_like(m.bias))
    torch.random.set_rng_state(g)
    return _round_weights(vae)


def _round_weights(vae):
    with torch.no_grad():
        for p in vae.parameters():
            p.copy_(p.half().float())
    return vae.cuda().eval().requires_grad_(False)


def _transformer_only(vae, x):
    """the module's transformer on tokens x (B, L, width), float32"""
    if hasattr(vae.transformer, "resblocks"):
        for blk in vae.transformer.resblocks:
            x = vae.block_forward(blk, x)
        return x
    for blk in vae.transformer:
        x = blk(x)
    return x


def _compare(vae, B, L, seed=1, tol_fwd=4e-3, tol_grad=1.5e-2):
    from followmyhold_amd.vae_transformer import HipVaeTransformer
    tr = HipVaeTransformer.from_module(vae)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, L, tr.width, generator=g).half().cuda()
    go = torch.randn(B, L, tr.width, generator=g).half().cuda()
    # reference: float32 module, float32 autograd
    xr = x.float().requires_grad_(True)
    ref = _transformer_only(vae, xr)
    (ref * go.float()).sum().backward()
    # inference route (nothing kept) and the autograd route (activations kept) give the same tokens
    out_plain = tr(x)
    xh = x.clone().requires_grad_(True)
    out = tr(xh)
    assert out.dtype == torch.float16 and out.shape == x.shape
    (out.float() * go.float()).sum().backward()
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    e_fwd = (out.float() - ref).abs().max().item()
    assert torch.isfinite(out).all() and e_fwd <= tol_fwd * scale, (e_fwd, scale)
    assert (out_plain.float() - out.float()).abs().max().item() <= 4e-3 * scale      # (another epilogue instantiation: differences of an fp16 ulp or two)
    gr, gh = xr.grad, xh.grad.float()
    gs = gr.abs().max().item()
    e_g = (gh - gr).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(gh.reshape(-1), gr.reshape(-1), dim=0).item()
    assert torch.isfinite(gh).all() and e_g <= tol_grad * gs and cos >= 1 - 2e-4, (e_g, gs, cos)
    # bitwise repeatable (no atomics anywhere: partial sums per split in a fixed order)
    xh2 = x.clone().requires_grad_(True)
    out2 = tr(xh2)
    (out2.float() * go.float()).sum().backward()
    assert torch.equal(out2, out) and torch.equal(xh2.grad, xh.grad)
    return e_fwd / scale, e_g / gs, cos


def test_vae_sizes_and_argument_checks_run_without_a_gpu():
    from followmyhold_amd import _lib as L
    from followmyhold_amd.vae_transformer import FohoVaeDesc, FohoVaeLayer
    lib = L.lib()
    lib.foho_vae_abi_size.restype = ctypes.c_int64
    assert lib.foho_vae_abi_size() == ctypes.sizeof(FohoVaeLayer) * 1000 + ctypes.sizeof(FohoVaeDesc)
    for fn in (lib.foho_vae_workspace_bytes, lib.foho_vae_saved_bytes):
        fn.restype = ctypes.c_size_t
        fn.argtypes = [ctypes.POINTER(FohoVaeDesc)]
    lib.foho_geo_last_error.restype = ctypes.c_char_p
    layers = (FohoVaeLayer * 16)()
    for y in layers:
        for name, typ in FohoVaeLayer._fields_:
            if typ is L.vp:
                setattr(y, name, 1)       # non-null: the size queries only look at the shape
        y.eps1 = y.eps2 = 1e-6
        y.qk_norm = 1
    d = FohoVaeDesc()
    d.width, d.heads, d.hidden, d.n_layers, d.n_tokens, d.batch = 1024, 16, 4096, 16, 3072, 1
    d.layers = ctypes.cast(layers, ctypes.POINTER(FohoVaeLayer))
    d.zeros = 1
    M = 3072
    # x, o, x1, scaled q, q^T, k^T (the last three: what the backward attention streams, written by the projection's epilogue) | q k v and
    # its un-normalised copy | z | lse
    per_layer = M * 2 * (1024 * 6 + 2 * 3072 + 4096) + M * 16 * 4
    n = lib.foho_vae_saved_bytes(ctypes.byref(d))
    assert 16 * per_layer <= n <= 16 * (per_layer + 11 * 256)
    assert lib.foho_vae_workspace_bytes(ctypes.byref(d)) > M * 2 * (4096 + 4 * 1024 + 2 * 3072)
    d.heads = 8
    assert lib.foho_vae_workspace_bytes(ctypes.byref(d)) == 0 and b"head dimension" in lib.foho_geo_last_error()
    d.heads, d.n_tokens = 16, 3000
    assert lib.foho_vae_saved_bytes(ctypes.byref(d)) == 0 and b"n_tokens" in lib.foho_geo_last_error()
    lib.foho_vae_fwd.restype = ctypes.c_int
    assert lib.foho_vae_fwd(None, None, None, None, ctypes.c_size_t(0), None, ctypes.c_size_t(0), None) == -1


def test_weight_folding_reorders_hy3dgen_rows_and_cancels_a_row_offset():
    """CPU: the packed q | k | v weights of both layouts reproduce LayerNorm -> Linear on un-normalised rows (the algebra the epilogue applies)."""
    from followmyhold_amd import standins
    from followmyhold_amd.vae_transformer import _blocks, _fold
    torch.manual_seed(0)
    vae = standins.Hy3dgenLayoutShapeVAE(num_latents=128, embed_dim=8, width=128, heads=2, layers=1, num_freqs=4, qkv_bias=True)
    with torch.no_grad():
        for m in vae.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(0.3 * torch.randn_like(m.weight)), m.bias.add_(0.2 * torch.randn_like(m.bias))
    p = _blocks(vae)[0]
    (ln, lin, perm), = p["qkv"]
    wf, b, s = _fold(ln, lin, "cpu", perm)
    x = torch.randn(5, 128) + 30.0
    mean, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
    rstd = (var + ln.eps).rsqrt()
    got = rstd * (x @ wf.float().t()) - rstd * mean * s + b
    want = lin(ln(x)).view(5, 2, 3, 64).permute(0, 2, 1, 3).reshape(5, -1)      # [q | k | v][head][d]
    assert (got - want).abs().max().item() <= 2e-2 * want.abs().max().item()        # (fp16 rounding of W gamma against a row offset of 30 sigma)


@gpu
@pytest.mark.parametrize("layout", ["standin", "hy3dgen", "hy3dgen_plain"])
def test_one_layer_matches_the_float32_module(layout):
    if layout == "standin":
        vae = _standin(256, 4, 1)
    else:
        vae = _hy3d(256, 4, 1, qk_norm=layout == "hy3dgen", qkv_bias=layout == "hy3dgen_plain")
    _compare(vae, 1, 256)


@gpu
def test_stack_and_batch_match_the_float32_module():
    vae = _hy3d(256, 4, 4)
    _compare(vae, 3, 384, tol_fwd=6e-3, tol_grad=2e-2)
    vae = _standin(128, 2, 3)
    _compare(vae, 2, 128, tol_fwd=6e-3, tol_grad=2e-2)
    # eight images x 16 heads x 2 key blocks = 256 (image, head, key block) workgroups: the batched DIRECT backward (blockIdx.y = image up to 7,
    # dK / dV written by the workgroup itself) at a small token count; and a width / head count between the two shapes above
    _compare(_hy3d(1024, 16, 1, latents=256, embed=16), 8, 256, seed=5)
    _compare(_hy3d(512, 8, 2, latents=640, embed=16, qk_norm=False, qkv_bias=True), 2, 640, seed=6, tol_fwd=6e-3, tol_grad=2e-2)


@gpu
def test_full_hunyuan_shape_one_layer_and_sixteen():
    """3072 tokens x 1024, 16 heads, hidden 4096, qk_norm -- one layer, then the sixteen-layer stack, tokens and latent gradient."""
    vae = _hy3d(1024, 16, 1, latents=3072, embed=64)
    _compare(vae, 1, 3072)
    _compare(vae, 2, 3072, seed=4)       # two images: ONE attention launch per kernel for both (blockIdx.y = image), dK / dV written directly
    del vae
    vae = _hy3d(1024, 16, 16, latents=3072, embed=64)
    e_f, e_g, cos = _compare(vae, 1, 3072, tol_fwd=1e-2, tol_grad=3e-2)
    print("sixteen layers: forward", e_f, "gradient", e_g, "cosine", cos)


@gpu
def test_latent2sdf_routes_through_the_hip_transformer_and_falls_back():
    """pipeline.vae_tokens: with install() the tokens (and the gradient of the latents, through post_kl in torch) come from the kernels and
    agree with the module; an input the kernels do not take goes through the module."""
    from followmyhold_amd import pipeline as PLN, vae_transformer
    vae = _hy3d(256, 4, 2, latents=256, embed=16).half()
    tr = vae_transformer.install(vae)
    lat = torch.randn(1, 256, 16, device="cuda").half().requires_grad_(True)
    n0 = tr.calls
    tok = PLN.vae_tokens(vae, lat)
    assert tr.calls == n0 + 1
    tok.float().square().sum().backward()
    g_hip = lat.grad.clone()
    lat.grad = None
    vae32 = _hy3d(256, 4, 2, latents=256, embed=16)
    ref = vae32(lat.float())
    ref.square().sum().backward()
    assert (tok.float() - ref).abs().max().item() <= 6e-3 * ref.abs().max().item()
    assert (g_hip.float() - lat.grad.float()).abs().max().item() <= 2e-2 * lat.grad.float().abs().max().item()
    short = torch.randn(1, 100, 16, device="cuda").half()       # 100 tokens: not a multiple of 128 -> the torch module
    n1 = tr.calls
    with torch.no_grad():
        out = PLN.vae_tokens(vae, short)
    assert tr.calls == n1 and out.shape == (1, 100, 256)


@gpu
def test_forward_and_backward_are_capturable_in_a_hip_graph():
    """foho_vae_fwd / foho_vae_bwd make no host synchronisation and allocate nothing: both directions of a two-layer stack captured into ONE
    hipGraph and replayed give the bits of the eager calls (the guidance loop replays whole iterations as graphs)."""
    from followmyhold_amd.vae_transformer import HipVaeTransformer
    vae = _hy3d(256, 4, 2)
    tr = HipVaeTransformer.from_module(vae)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 256, 256, generator=g).half().cuda()
    go = torch.randn(2, 256, 256, generator=g).half().cuda()
    out_e, saved_e = tr.forward_raw(x, keep=True)
    gx_e = tr.backward_raw(go, saved_e, tuple(x.shape))
    torch.cuda.synchronize()
    import ctypes
    from followmyhold_amd import _lib as L
    d = tr._desc(2, 256)
    ws = tr._workspace(d)
    saved = torch.empty_like(saved_e)
    out, gx = torch.empty_like(x), torch.empty_like(x)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph.capture_begin()
        st = L.vp(side.cuda_stream)
        rc1 = tr.lib.foho_vae_fwd(ctypes.byref(d), L.vp(x.data_ptr()), L.vp(out.data_ptr()), L.vp(ws.data_ptr()), ctypes.c_size_t(ws.numel()), L.vp(saved.data_ptr()),
                                  ctypes.c_size_t(saved.numel()), st)
        rc2 = tr.lib.foho_vae_bwd(ctypes.byref(d), L.vp(go.data_ptr()), L.vp(gx.data_ptr()), L.vp(ws.data_ptr()), ctypes.c_size_t(ws.numel()), L.vp(saved.data_ptr()),
                                  ctypes.c_size_t(saved.numel()), st)
        graph.capture_end()
    assert rc1 == 0 and rc2 == 0
    for _ in range(2):
        out.zero_(), gx.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, out_e) and torch.equal(gx, gx_e)
