"""Probe: torch's efficient-attention op on this build -- logsumexp shape / units, backward op signature."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, H, M, L = 1, 4, 300, 384
q, k, v = (torch.randn(B, n, H, 64, device=dev, dtype=torch.float16).transpose(1, 2) for n in (M, L, L))
out, lse, seed, off = torch.ops.aten._scaled_dot_product_efficient_attention(q, k, v, None, True)
print("out", out.shape, out.stride(), "lse", lse.shape, lse.dtype, lse.stride(), "seed", seed, off)
s = (q.float() @ k.float().transpose(-1, -2)) / 8.0
ref = torch.logsumexp(s, dim=-1)
print("lse vs natural-log reference:", (lse[..., :M] - ref).abs().max().item(), " vs log2:", (lse[..., :M] - ref / math.log(2)).abs().max().item())
print(torch.ops.aten._scaled_dot_product_efficient_attention_backward.default._schema)
go = torch.randn_like(out)
g = torch.ops.aten._scaled_dot_product_efficient_attention_backward(go, q, k, v, None, out, lse, seed, off, 0.0, [True, True, True, False], False)
print([None if t is None else (t.shape, t.stride()) for t in g])
# the same backward fed with OUR forward's out / lse
from followmyhold_amd import sdpa, _lib as L_
import ctypes
lib = L_.lib()
o2 = torch.empty(B, M, H * 64, dtype=torch.float16, device=dev)
nlse = torch.empty(B, (M + 63) // 64 * 64, H, dtype=torch.float32, device=dev)
ws = sdpa._workspace(lib, dev, M, L, H)
d = sdpa._desc(q, k)
rc = lib.foho_sdpa_fwd(ctypes.byref(d), L_.vp(q.data_ptr()), L_.vp(k.data_ptr()), L_.vp(v.data_ptr()), L_.vp(o2.data_ptr()), L_.vp(nlse.data_ptr()), L_.vp(ws.data_ptr()),
                       ctypes.c_size_t(ws.numel()), L_.vp(torch.cuda.current_stream().cuda_stream))
assert rc == 0
o2v = o2.view(B, M, H, 64).transpose(1, 2)
lse2 = (-nlse[:, :M] * math.log(2)).transpose(1, 2).contiguous()
print("our lse vs theirs:", (lse2 - lse[..., :M]).abs().max().item(), "out diff", (o2v.float() - out.float()).abs().max().item())
lse2p = torch.zeros_like(lse); lse2p[..., :M] = lse2
g2 = torch.ops.aten._scaled_dot_product_efficient_attention_backward(go, q, k, v, None, o2v, lse2p, seed, off, 0.0, [True, True, True, False], False)
for a, b, n in zip(g[:3], g2[:3], "qkv"):
    print("d" + n, (a.float() - b.float()).abs().max().item(), a.float().abs().max().item())
