"""k_geo_attn against k_geo_attn2 (FOHO_GEO_ATTN=2): bitwise comparison and interleaved timing at the decoder's shape."""
import ctypes, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import _lib as L
lib = L.lib() if not os.environ.get("FOHO_HIP_SO") else ctypes.CDLL(os.path.join(ROOT, os.environ["FOHO_HIP_SO"]))
dev = torch.device("cuda", 0)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, Lk, H) in ((49152, 3072, 16),):
    W = H * 64
    g = torch.Generator().manual_seed(1)
    q = (torch.randn(M, W, generator=g) * (math.log2(math.e) / 8.0)).half().to(dev)
    kv = torch.randn(Lk, 2 * W, generator=g)
    kv[5, :64] *= 4.0
    kv = kv.half().to(dev)
    vt = torch.empty(W * Lk, dtype=torch.float16, device=dev)
    outs, ts = {}, {"1": [], "2": []}
    def run(v, n):
        os.environ["FOHO_GEO_ATTN"] = v
        O = torch.full((M, W), float("nan"), dtype=torch.float16, device=dev)
        for _ in range(2):
            lib.foho_geo_attention(P(q), P(kv), P(vt), P(O), M, Lk, H, st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            lib.foho_geo_attention(P(q), P(kv), P(vt), P(O), M, Lk, H, st)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6, O
    for _ in range(4):
        for v in ("1", "2"):
            t, O = run(v, 10)
            ts[v].append(t); outs[v] = O
    fl = 4.0 * M * Lk * W
    print(f"M={M} L={Lk} heads={H}: k_geo_attn {min(ts['1']):.1f} us = {fl / min(ts['1']) / 1e6:.0f} TFLOP/s | k_geo_attn2 {min(ts['2']):.1f} us = {fl / min(ts['2']) / 1e6:.0f} | bitwise equal: {bool(torch.equal(outs['1'], outs['2']))}"
          f" (max |diff| {(outs['1'].float() - outs['2'].float()).abs().max().item():.3g}, finite {bool(torch.isfinite(outs['2'].float()).all())})", flush=True)
