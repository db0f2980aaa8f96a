"""A/B of builds of the geometry decoder's GEMM: python scripts/dev_gemm_ab.py [so files...]  (each .so in its own process)"""
import ctypes, math, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    lib = ctypes.CDLL(sys.argv[2])
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    for (M, N, K, gelu) in ((16384, 1024, 4096, 0), (49152, 1024, 4096, 0), (49152, 4096, 1024, 0), (49152, 4096, 1024, 1), (49152, 1024, 1024, 0)):
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).half(); b = torch.randn(N, device=dev)
        C = torch.empty(M, N, dtype=torch.float16, device=dev)
        best = 1e9
        for flag in (int(sys.argv[3]) if len(sys.argv) > 3 else 0,):
            for rep in range(6):
                for _ in range(2):
                    lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, gelu | flag, ctypes.c_float(1.0), st)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10):
                    lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, gelu | flag, ctypes.c_float(1.0), st)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
        out.append(f"{2 * M * N * K / best / 1e12:6.0f}")
    print(f"{os.path.basename(sys.argv[2]):28s} flag {sys.argv[3] if len(sys.argv) > 3 else 0}: TFLOP/s  " + " ".join(out), flush=True)
    sys.exit(0)
print("shapes (M,N,K,gelu): (16384,1024,4096,0) (49152,1024,4096,0) (49152,4096,1024,0) (49152,4096,1024,1) (49152,1024,1024,0)")
sos = sys.argv[1:] or [os.path.join(ROOT, "followmyhold_amd", "libfoho_hip.so")]
for rnd in range(2):
    for so in sos:
        flag = "0"
        if so.endswith(":4"):
            so, flag = so[:-2], "4"
        subprocess.run([sys.executable, __file__, "--one", so, flag])
