"""Times foho_geo_decode_fwd at the Hunyuan3D-2 decoder shape against the torch module (bench.py's geo_decode record alone),
and each GEMM / the attention kernel on their own.  python scripts/geo_bench.py [--parts | --fb]
(--fb: three forward + backward passes of the HIP decoder and nothing else -- the run to put under rocprofv3)"""
import ctypes, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from followmyhold_amd import _lib as L

dev = torch.device("cuda", 0)
if "--fb" in sys.argv:
    from followmyhold_amd import standins
    from followmyhold_amd.geo_decode import HipGeoDecoder
    torch.manual_seed(0)
    vae = standins.StandInShapeVAE(num_latents=3072, embed_dim=64, width=1024, heads=16, layers=1, num_freqs=8)
    hip = HipGeoDecoder.from_module(vae.geo_decoder.to(dev).eval(), device=dev)
    n = 65 ** 3
    q = (torch.rand(1, n, 3, device=dev) * 2.2 - 1.1).half().float()
    lat = torch.randn(1, 3072, 1024, device=dev).half()
    go = torch.randn(1, n, 1, device=dev)
    for _ in range(3):
        l = lat.clone().requires_grad_(True)
        t0 = time.perf_counter()
        (hip(q, l).float() * go).sum().backward()
        torch.cuda.synchronize()
        print(f"forward + backward: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
    for _ in range(3):                       # ... and the no-gradient forward (K / V projection of the tokens included)
        hip._prepared = None
        t0 = time.perf_counter()
        with torch.no_grad():
            hip(q, lat)
        torch.cuda.synchronize()
        print(f"forward: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
    sys.exit(0)
if "--parts" in sys.argv:
    lib = L.lib()
    lib.foho_geo_last_error.restype = ctypes.c_char_p
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for M in (16384, 49152):
      for (N, K, gelu) in ((1024, 1024, 0), (4096, 1024, 1), (4096, 1024, 0), (1024, 4096, 0)):
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).half(); b = torch.randn(N, device=dev)
        C = torch.empty(M, N, dtype=torch.float16, device=dev)
        def run(flag, n=20):
            for _ in range(3):
                lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, gelu | flag, ctypes.c_float(1.0), st)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, gelu | flag, ctypes.c_float(1.0), st)
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
        # interleaved A/B rounds in one process: phased (default) vs lock-step (flag 4)
        ts = {0: [], 4: []}
        for _ in range(5):
            for flag in (0, 4):
                ts[flag].append(run(flag))
        dt, dl = min(ts[0]), min(ts[4])
        Cp = C.clone()
        lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, gelu | 4, ctypes.c_float(1.0), st)
        torch.cuda.synchronize()
        same = bool(torch.equal(Cp, C))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            Ct = torch.nn.functional.linear(A, W, b.half())
        torch.cuda.synchronize(); dtt = (time.perf_counter() - t0) / 20
        fl = 2 * M * N * K
        print(f"gemm M={M} N={N} K={K} gelu={gelu}: phased {dt * 1e6:.1f} us = {fl / dt / 1e12:.0f} TFLOP/s | lock-step {dl * 1e6:.1f} us = {fl / dl / 1e12:.0f} | "
              f"torch linear {dtt * 1e6:.1f} us = {fl / dtt / 1e12:.0f} | phased == lock-step bitwise: {same}", flush=True)
    M = 16384
    Lk, H = 3072, 16
    q = torch.randn(M, 1024, device=dev).half(); kv = torch.randn(Lk, 2048, device=dev).half()
    O = torch.empty(M, 1024, dtype=torch.float16, device=dev); vt = torch.empty(1024 * Lk, dtype=torch.float16, device=dev)
    for _ in range(3):
        lib.foho_geo_attention(P(q), P(kv), P(vt), P(O), M, Lk, H, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        lib.foho_geo_attention(P(q), P(kv), P(vt), P(O), M, Lk, H, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    qf = q.view(1, M, H, 64).transpose(1, 2); kf = kv[:, :1024].reshape(1, Lk, H, 64).transpose(1, 2); vf = kv[:, 1024:].reshape(1, Lk, H, 64).transpose(1, 2)
    for _ in range(2):
        torch.nn.functional.scaled_dot_product_attention(qf, kf, vf)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        torch.nn.functional.scaled_dot_product_attention(qf, kf, vf)
    torch.cuda.synchronize(); dtt = (time.perf_counter() - t0) / 10
    fl = 4 * M * Lk * 1024
    print(f"attention M={M} L={Lk} heads={H}: {dt * 1e6:.1f} us = {fl / dt / 1e12:.0f} TFLOP/s (torch SDPA {dtt * 1e6:.1f} us = {fl / dtt / 1e12:.0f})", flush=True)
print(json.dumps(bench.geo_decode_record(torch, dev)), flush=True)
