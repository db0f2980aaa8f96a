"""Segment durations of one K tile of k_geo_gemm_d4 (development build: make -C followmyhold_amd/csrc VARIANT=p8st EXTRA=-DP8_STAMPS).
python scripts/dev/d4_stamps.py"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lib = ctypes.CDLL(os.path.join(ROOT, "followmyhold_amd", "libfoho_hip_p8st.so"))
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in ((3072, 1024, 4096), (3072, 1024, 1024)):
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).half(); b = torch.randn(N, device=dev)
    C = torch.empty(M, N, dtype=torch.float16, device=dev)
    for _ in range(3):
        lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, 8, ctypes.c_float(1.0), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, 8, ctypes.c_float(1.0), st)
    e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.foho_geo_p8_stamps(buf)
    names = ["vmcnt wait (own tile landed)", "barrier", "issue of 8 DMA pieces", "4 reads + wait (k step 0)", "k steps 0-1: 8 MFMAs (+ reads)", "k steps 2-3: 8 MFMAs (+ reads)"]
    nk = buf[6]
    print(f"M={M} N={N} K={K}: {e0.elapsed_time(e1) * 100:.1f} us per launch WITH the stamps; cycles per K tile (mean over {nk} tiles), waves 0-3 of workgroup 0")
    for q, nm in enumerate(names):
        print(f"  {nm:32s} " + "  ".join(f"w{w}: {buf[w * 8 + q] / nk:7.1f}" for w in range(4)))
    print("  total per K tile                 " + "  ".join(f"w{w}: {sum(buf[w * 8 + q] for q in range(6)) / nk:7.1f}" for w in range(4)))
