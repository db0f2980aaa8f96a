"""Segment durations of the phased GEMM loop (development build libfoho_hip_p8st.so: make VARIANT=p8st EXTRA=-DP8_STAMPS)."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lib = ctypes.CDLL(os.path.join(ROOT, "followmyhold_amd", "libfoho_hip_p8st.so"))
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in ((49152, 1024, 4096), (49152, 4096, 1024)):
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).half(); b = torch.randn(N, device=dev)
    C = torch.empty(M, N, dtype=torch.float16, device=dev)
    for _ in range(3):
        lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, 0, ctypes.c_float(1.0), st)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.foho_geo_p8_stamps(buf)
    names = ["load seg (reads + DMA issue)", "barrier after load", "lgkmcnt wait", "8 MFMAs", "vmcnt wait", "barrier after compute"]
    print(f"M={M} N={N} K={K}: cycles per PHASE (mean over {buf[6]} K tiles x 4 phases), waves 0 (group 0) / 4 (group 1) of workgroup 0")
    for q, nm in enumerate(names):
        print(f"  {nm:30s} " + "  ".join(f"w{w}: {buf[w * 8 + q] / (4 * buf[w * 8 + 6]):7.1f}" for w in (0, 1, 4, 5)))
    print("  total per phase               " + "  ".join(f"w{w}: {sum(buf[w * 8 + q] for q in range(6)) / (4 * buf[w * 8 + 6]):7.1f}" for w in (0, 1, 4, 5)))
