"""A/B of builds: the forward attention kernel alone (16384 x 3072 x 16 heads) and the whole cached forward.  python scripts/dev_attn_ab.py so..."""
import ctypes, math, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    lib = ctypes.CDLL(sys.argv[2])
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    M, Lk, H = 49152, 3072, 16
    q = torch.randn(M, 1024, device=dev).half(); kv = torch.randn(Lk, 2048, device=dev).half()
    O = torch.empty(M, 1024, dtype=torch.float16, device=dev); vt = torch.empty(1024 * Lk, dtype=torch.float16, device=dev)
    best = 1e9
    for rep in range(6):
        for _ in range(2):
            lib.foho_geo_attention(P(q), P(kv), P(vt), P(O), M, Lk, H, st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            lib.foho_geo_attention(P(q), P(kv), P(vt), P(O), M, Lk, H, st)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 5)
    print(f"{os.path.basename(sys.argv[2]):28s}: attention {best * 1e6:7.1f} us = {4 * M * Lk * 1024 / best / 1e12:5.0f} TFLOP/s", flush=True)
    sys.exit(0)
for rnd in range(2):
    for so in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "--one", so])
