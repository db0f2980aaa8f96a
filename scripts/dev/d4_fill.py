"""k_geo_gemm_d4 with its matrix work compiled out (make -C followmyhold_amd/csrc VARIANT=fill EXTRA=-DD4_FILL_ONLY): how fast does a CU
fill its LDS ring?  python scripts/dev/d4_fill.py"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda", 0)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name in ("libfoho_hip.so", "libfoho_hip_fill.so"):
    lib = ctypes.CDLL(os.path.join(ROOT, "followmyhold_amd", name))
    for (M, N, K) in ((3072, 1024, 4096), (2048, 1024, 4096), (1024, 1024, 4096), (4096, 1024, 4096)):
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).half(); b = torch.randn(N, device=dev)
        C = torch.empty(M, N, dtype=torch.float16, device=dev)
        for _ in range(3):
            lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, 8, ctypes.c_float(1.0), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, 8, ctypes.c_float(1.0), st)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
        tiles = (M // 128) * (N // 128)
        print(f"{name}: M={M} ({tiles} workgroups): {t:.1f} us = {t / (K // 64) * 1e3:.0f} ns per K tile; fill {tiles * (K // 64) * 32768 / t / 1e6:.2f} TB/s chip, {32768 / (t / (K // 64)) / 1e3:.1f} GB/s per CU", flush=True)
