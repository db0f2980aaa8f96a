"""followmyhold_amd.sdpa: scaled-dot-product attention forward and backward on the geometry decoder's attention kernels (the
self-attention of the ShapeVAE transformer inside latent2sdf, PL:295) against torch's float32 math implementation."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

gpu = pytest.mark.gpu


def test_sdpa_argument_checks_run_without_a_gpu():
    from followmyhold_amd import _lib as L
    lib = L.lib()
    lib.foho_sdpa_workspace_bytes.restype = ctypes.c_size_t
    assert lib.foho_sdpa_workspace_bytes(3072, 3072, 16) > 2 * 1024 * 3072 * 2
    assert lib.foho_sdpa_workspace_bytes(3072, 3000, 16) == 0            # keys not a multiple of 64
    assert lib.foho_sdpa_workspace_bytes(3072, 3072, 17) == 0
    lib.foho_sdpa_fwd.restype = lib.foho_sdpa_bwd.restype = ctypes.c_int
    from followmyhold_amd import sdpa
    one = ctypes.c_void_p(1)
    d = sdpa.FohoSdpaDesc(64, 128, 2, 1, 64 * 128, 128, 64, 128 * 256, 256, 64)
    assert lib.foho_sdpa_fwd(None, one, one, one, one, None, None, one, ctypes.c_size_t(1 << 40), None) != 0                 # no descriptor
    assert lib.foho_sdpa_fwd(ctypes.byref(d), None, None, None, None, None, None, None, ctypes.c_size_t(0), None) != 0      # null operands
    assert lib.foho_sdpa_fwd(ctypes.byref(d), one, one, one, one, None, None, one, ctypes.c_size_t(16), None) != 0           # workspace too small
    d.L = 192                                                                                                               # 192 keys: forward only
    assert lib.foho_sdpa_bwd(ctypes.byref(d), one, one, one, one, one, one, one, one, one, one, ctypes.c_size_t(1 << 40), None) != 0
    d.L, d.q_head = 128, 60                                                                                                 # strides: multiples of 8 halfs
    assert lib.foho_sdpa_bwd(ctypes.byref(d), one, one, one, one, one, one, one, one, one, one, ctypes.c_size_t(1 << 40), None) != 0
    q = torch.zeros(1, 2, 8, 64)
    assert not sdpa.eligible(q, q, q)                                    # CPU / fp32: torch's business


@gpu
@pytest.mark.parametrize("route", ["hip", "torch"])
@pytest.mark.parametrize("B,H,M,Lk", [(1, 2, 256, 256), (2, 4, 300, 384), (1, 16, 3072, 3072), (1, 16, 70, 128)])
def test_attention_forward_and_backward_against_torch_math(B, H, M, Lk, route, monkeypatch):
    """route: the backward behind the HIP forward -- the package's own kernels, or torch's memory-efficient attention backward fed with
    the HIP forward's output and log-sum-exp (the default)."""
    from followmyhold_amd import sdpa
    monkeypatch.setattr(sdpa, "backward_route", route)
    g = torch.Generator().manual_seed(H + M)
    q, k, v = (torch.randn(B, H, n, 64, generator=g).half().cuda() for n in (M, Lk, Lk))
    k[:, 0, 5] *= 3.0                                                   # a key that dominates some rows
    go = torch.randn(B, H, M, 64, generator=g).half().cuda()
    qh, kh, vh = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = sdpa.attention(qh, kh, vh)
    assert out.shape == (B, H, M, 64) and out.dtype == torch.float16
    out.backward(go)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    att = torch.softmax(qr @ kr.transpose(-1, -2) / 8.0, dim=-1)
    ref = att @ vr
    ref.backward(go.float())
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    assert (out.float() - ref).abs().max().item() <= 3e-3 * max(ref.abs().max().item(), 1.0)
    for name, a, b in (("dq", qh.grad, qr.grad), ("dk", kh.grad, kr.grad), ("dv", vh.grad, vr.grad)):
        a = a.float()
        cos = F.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
        assert torch.isfinite(a).all() and (a - b).abs().max().item() <= 1.5e-2 * b.abs().max().item() and cos >= 1 - 2e-4, (name, (a - b).abs().max().item(), b.abs().max().item(), cos)
    # the operands are read where they lie: k and v as the two halves of one projection's output (row 2 x 64 H), q / k / v interleaved
    # per head as hy3dgen's c_qkv leaves them (row 3 x 64 H, head 192) -- same numbers, same results
    if B == 1:
        kv = torch.cat([k.transpose(1, 2).reshape(B, Lk, H * 64), v.transpose(1, 2).reshape(B, Lk, H * 64)], dim=-1).contiguous()
        ks, vs = (kv[..., i * H * 64:(i + 1) * H * 64].view(B, Lk, H, 64).transpose(1, 2) for i in (0, 1))
        assert ks.stride() == vs.stride() and ks.stride(2) == 2 * H * 64
        q3 = q.detach().clone().requires_grad_(True)
        kv3 = kv.detach().clone().requires_grad_(True)
        k3, v3 = (kv3[..., i * H * 64:(i + 1) * H * 64].view(B, Lk, H, 64).transpose(1, 2) for i in (0, 1))
        out3 = sdpa.attention(q3, k3, v3)
        out3.backward(go)
        assert torch.equal(out3, out)
        if route == "hip":
            assert torch.equal(q3.grad, qh.grad)
            assert torch.equal(kv3.grad[..., :H * 64].view(B, Lk, H, 64).transpose(1, 2), kh.grad)
            assert torch.equal(kv3.grad[..., H * 64:].view(B, Lk, H, 64).transpose(1, 2), vh.grad)
        else:
            for a, b in ((q3.grad, qh.grad), (kv3.grad[..., :H * 64].view(B, Lk, H, 64).transpose(1, 2), kh.grad), (kv3.grad[..., H * 64:].view(B, Lk, H, 64).transpose(1, 2), vh.grad)):
                assert (a.float() - b.float()).abs().max().item() <= 2e-3 * b.float().abs().max().item()
        if M == Lk:
            qkv = torch.stack([t.transpose(1, 2) for t in (q, k, v)], dim=3).reshape(B, M, H, 192).contiguous()   # (B, N, H, [q | k | v])
            q4, k4, v4 = (qkv[..., i * 64:(i + 1) * 64].transpose(1, 2) for i in (0, 1, 2))
            assert q4.stride(1) == 192 and torch.equal(q4, q)
            assert torch.equal(sdpa.attention(q4, k4, v4), out)
    assert not sdpa._torch_route_refused                                # (the torch route did not fall back)
    # repeatable: no atomics anywhere in the three kernels
    if route == "hip":
        q2, k2, v2 = (t.clone().requires_grad_(True) for t in (q, k, v))
        sdpa.attention(q2, k2, v2).backward(go)
        assert torch.equal(q2.grad, qh.grad) and torch.equal(k2.grad, kh.grad) and torch.equal(v2.grad, vh.grad)


@gpu
def test_backward_falls_back_to_the_hip_kernels_when_torch_refuses(monkeypatch):
    """The default backward behind the HIP forward is a private torch operator; a build that lacks it, or one that refuses the call,
    must not cost a run: the HIP backward takes over (from that call on) with the same gradients."""
    from followmyhold_amd import sdpa
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(1, 256, 4, 64, generator=g).half().cuda().transpose(1, 2) for _ in range(3))
    go = torch.randn(1, 256, 4, 64, generator=g).half().cuda().transpose(1, 2)

    def grads():
        leaves = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
        sdpa.attention(*leaves).backward(go)
        return [t.grad for t in leaves]

    monkeypatch.setattr(sdpa, "backward_route", "hip")
    want = grads()
    monkeypatch.setattr(sdpa, "backward_route", "torch")
    monkeypatch.setattr(sdpa, "_torch_route_refused", False)

    def refuse(*a, **kw):
        raise RuntimeError("no such kernel on this build")

    monkeypatch.setattr(sdpa, "_torch_backward_op", lambda: refuse)
    got = grads()
    assert sdpa._torch_route_refused
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    got2 = grads()                                   # ... and the following calls do not even try
    assert all(torch.equal(a, b) for a, b in zip(got2, want))
    monkeypatch.setattr(sdpa, "_torch_backward_op", lambda: None)      # an absent operator: the HIP route from the forward on
    monkeypatch.setattr(sdpa, "_torch_route_refused", False)
    assert all(torch.equal(a, b) for a, b in zip(grads(), want)) and not sdpa._torch_route_refused


@gpu
def test_hip_sdpa_context_serves_eligible_calls_and_leaves_the_rest_to_torch():
    """Inside `with sdpa.hip_sdpa():` a module's F.scaled_dot_product_attention goes to the HIP kernels when the call is eligible (the
    stand-in ShapeVAE transformer in fp16: forward and the gradient to its input agree with torch's) and to torch otherwise (fp32, a
    mask); outside the context nothing is patched."""
    from followmyhold_amd import sdpa, standins
    torch.manual_seed(0)
    vae = standins.StandInShapeVAE(num_latents=256, embed_dim=8, width=128, heads=2, layers=2, num_freqs=8).cuda().half().eval()
    vae.requires_grad_(False)
    lat = torch.randn(1, 256, 8, device="cuda").half()
    calls = {"n": 0}
    orig_apply = sdpa._HipSdpaFn.apply
    a = lat.clone().requires_grad_(True)
    ref = vae(a)
    ref.float().square().sum().backward()
    b = lat.clone().requires_grad_(True)
    with sdpa.hip_sdpa():
        assert F.scaled_dot_product_attention is not None
        import unittest.mock as um
        with um.patch.object(sdpa._HipSdpaFn, "apply", side_effect=lambda *x: (calls.__setitem__("n", calls["n"] + 1), orig_apply(*x))[1]):
            got = vae(b)
            got.float().square().sum().backward()
            # not eligible: float32 tensors, a mask -> torch's implementation, no error
            x = torch.randn(1, 2, 128, 64, device="cuda")
            F.scaled_dot_product_attention(x, x, x)
            xh = x.half()
            F.scaled_dot_product_attention(xh, xh, xh, attn_mask=torch.ones(128, 128, dtype=torch.bool, device="cuda"))
    assert calls["n"] == 2                                              # the two self-attention layers, nothing else
    assert F.scaled_dot_product_attention.__module__ != "followmyhold_amd.sdpa"
    scale = ref.float().abs().max().item()
    assert (got.float() - ref.float()).abs().max().item() <= 1e-2 * scale
    ga, gb_ = a.grad.float(), b.grad.float()
    assert F.cosine_similarity(ga.flatten(), gb_.flatten(), dim=0).item() >= 1 - 1e-3
