"""The ShapeVAE transformer at the Hunyuan3D-2 shape (3072 tokens x 1024, 16 heads, 16 layers, qk_norm): torch module (fp16, memory-efficient
attention; and with this package's attention forward) against foho_vae_fwd / foho_vae_bwd, forward and forward + backward to the input, one
image and four; then every GEMM shape of a layer at M = 3072 on each kernel variant.
python scripts/dev/vae_bench.py [--layers 16] [--batch 1,4] [--gemms]"""
import argparse, ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import _lib as L, pipeline as PLN, standins, vae_transformer

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=16)
ap.add_argument("--batch", default="1,4")
ap.add_argument("--gemms", action="store_true")
ap.add_argument("--layout", default="hy3dgen")
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)


def timed(fn, reps=a.reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


torch.manual_seed(0)
if a.layout == "hy3dgen":
    vae = standins.Hy3dgenLayoutShapeVAE(layers=a.layers).to(dev).half().eval().requires_grad_(False)
else:
    vae = standins.StandInShapeVAE(num_latents=3072, embed_dim=64, width=1024, heads=16, layers=a.layers, num_freqs=8).to(dev).half().eval().requires_grad_(False)
tr = vae_transformer.HipVaeTransformer.from_module(vae)
flop_f = a.layers * (2 * 3072 * 1024 * (3072 + 1024 + 4096 + 4096) + 4 * 3072 * 3072 * 1024)
flop_b = a.layers * (2 * 3072 * 1024 * (3072 + 1024 + 4096 + 4096) + 10 * 3072 * 3072 * 1024)
for B in [int(b) for b in a.batch.split(",")]:
    lat = torch.randn(B, 3072, 64, device=dev, dtype=torch.float16)
    x0 = vae.post_kl(lat).detach()
    go = torch.randn_like(x0)

    def torch_fwd():
        with torch.no_grad(), PLN.vae_attention_backend():
            return vae(lat)

    def torch_fb():
        l = lat.clone().requires_grad_(True)
        with PLN.vae_attention_backend():
            out = vae(l)
        out.backward(go)

    def hip_fwd():
        return tr(x0)

    def hip_fb():
        x = x0.clone().requires_grad_(True)
        tr(x).backward(go)

    os.environ["FOHO_VAE_SDPA"] = "efficient"
    t_tf, t_tfb = timed(torch_fwd), timed(torch_fb)
    os.environ["FOHO_VAE_SDPA"] = "hip"
    t_hf0, t_hfb0 = timed(torch_fwd), timed(torch_fb)
    t_f, t_fb = timed(hip_fwd), timed(hip_fb)
    print(f"B={B} layers={a.layers}: torch(efficient sdpa) fwd {t_tf:.2f} ms, fwd+bwd {t_tfb:.2f}; torch + hip sdpa fwd {t_hf0:.2f}, fwd+bwd {t_hfb0:.2f}; "
          f"foho_vae fwd {t_f:.2f} ms ({B * flop_f / t_f / 1e9:.0f} TFLOP/s), fwd+bwd {t_fb:.2f} ms ({B * (flop_f + flop_b) / t_fb / 1e9:.0f} TFLOP/s)", flush=True)
    with torch.no_grad():
        ref, got = vae(lat), tr(x0)
    print(f"   max |hip - torch fp16| / max|torch| = {((got.float() - ref.float()).abs().max() / ref.float().abs().max()).item():.2e}", flush=True)

if a.gemms:
    lib = L.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for M in (3072, 6144, 12288):
        for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096), (1024, 3072)):
            A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half() * 0.03
            b = torch.zeros(N, device=dev); C = torch.empty(M, N, device=dev).half()
            res = []
            ref_c = torch.nn.functional.linear(A.float(), W.float())
            for name, flag in (("auto", 0), ("128", 2), ("deep128", 8), ("fill+mma128", 32), ("phased256", 16), ("phased192", 64)):
                t = timed(lambda: lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, flag, ctypes.c_float(1.0), st), reps=20)
                err = ((C.float() - ref_c).abs().max() / ref_c.abs().max()).item()
                res.append(f"{name} {t * 1e3:.1f} us ({2 * M * N * K / t / 1e9:.0f} TF{'' if err < 2e-3 else f' WRONG {err:.2e}'})")
            t = timed(lambda: torch.nn.functional.linear(A, W), reps=20)
            res.append(f"hipBLASLt {t * 1e3:.1f} us ({2 * M * N * K / t / 1e9:.0f} TF)")
            print(f"GEMM M={M} N={N} K={K}: " + ", ".join(res), flush=True)
