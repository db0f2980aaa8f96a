"""torch SDPA backends at the VAE transformer's self-attention shape (1, 16, 3072, 64) fp16: forward and forward + backward."""
import time, torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel
dev = torch.device("cuda", 0)
q, k, v = (torch.randn(1, 16, 3072, 64, device=dev, dtype=torch.float16, requires_grad=True) for _ in range(3))
go = torch.randn(1, 16, 3072, 64, device=dev, dtype=torch.float16)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH), ("default", None)):
    try:
        def fwd():
            with torch.no_grad():
                if be is None: return F.scaled_dot_product_attention(q, k, v)
                with sdpa_kernel(be): return F.scaled_dot_product_attention(q, k, v)
        def fb():
            for t in (q, k, v): t.grad = None
            if be is None: o = F.scaled_dot_product_attention(q, k, v)
            else:
                with sdpa_kernel(be): o = F.scaled_dot_product_attention(q, k, v)
            o.backward(go)
        tf, tb = timed(fwd), timed(fb)
        fl = 4 * 16 * 3072 * 3072 * 64
        print(f"{name:10s}: forward {tf:7.1f} us = {fl / tf / 1e6:5.0f} TFLOP/s   forward + backward {tb:7.1f} us = {3.5 * fl / tb / 1e6:5.0f} TFLOP/s", flush=True)
    except Exception as e:
        print(name, "failed:", type(e).__name__, str(e)[:200], flush=True)
# ... and the HIP kernels (followmyhold_amd.sdpa) at the same shape
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from followmyhold_amd import sdpa
def hfwd():
    with torch.no_grad():
        return sdpa.attention(q, k, v)
def hfb():
    for t in (q, k, v): t.grad = None
    sdpa.attention(q, k, v).backward(go)
tf, tb = timed(hfwd), timed(hfb)
fl = 4 * 16 * 3072 * 3072 * 64
print(f"{'hip':10s}: forward {tf:7.1f} us = {fl / tf / 1e6:5.0f} TFLOP/s   forward + backward {tb:7.1f} us = {3.5 * fl / tb / 1e6:5.0f} TFLOP/s", flush=True)
