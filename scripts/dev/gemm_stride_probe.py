"""Does the power-of-two row stride of the GEMM operands cost anything (L2 channel hot spots)?  foho_geo_gemm at M = 3072 on K = 4096 / 1024
against K a tile more or less (row stride 8192 bytes against 8064 / 8320), per kernel variant: us per launch and ns per K tile.
python scripts/dev/gemm_stride_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import _lib as L

lib = L.lib()
lib.foho_geo_gemm.restype = ctypes.c_int
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


M = 3072
for N, Ks in ((1024, (4096, 4032, 4160, 3072, 3008, 1024, 960, 1088)), (4096, (1024, 960, 1088)), (3072, (1024, 960, 1088))):
    for K in Ks:
        A = torch.randn(M, K, device="cuda").half()
        W = (torch.randn(N, K, device="cuda") * 0.02).half()
        b = torch.zeros(N, device="cuda")
        C = torch.empty(M, N, device="cuda", dtype=torch.float16)
        out = []
        for name, flag in (("auto", 0), ("fill+mma128", 32), ("phased192", 64), ("phased256", 16)):
            t = timed(lambda: lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, flag, ctypes.c_float(1.0), st))
            out.append(f"{name} {t:6.1f} us ({t * 1e3 / (K // 64):5.0f} ns / K tile)")
        print(f"N={N} K={K} (row stride {2 * K} B): " + ", ".join(out))
