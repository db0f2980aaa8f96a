"""Which Tensile kernels does hipBLASLt run for the geometry decoder's two big GEMM shapes, and how fast?  (Run under rocprofv3
--kernel-trace to get the names.)  python scripts/dev/dev_blaslt_trace.py"""
import time
import torch
import torch.nn.functional as F
dev = torch.device("cuda", 0)
for M in (16384, 49152):
    for (N, K) in ((4096, 1024), (1024, 4096), (1024, 1024)):
        A = torch.randn(M, K, device=dev).half()
        W = (torch.randn(N, K, device=dev) / K ** 0.5).half()
        b = torch.randn(N, device=dev).half()
        for _ in range(3):
            C = F.linear(A, W, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            C = F.linear(A, W, b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"F.linear M={M} N={N} K={K}: {dt * 1e6:.1f} us = {2 * M * N * K / dt / 1e12:.0f} TFLOP/s", flush=True)
