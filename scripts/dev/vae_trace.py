"""foho_vae_fwd + foho_vae_bwd at the Hunyuan3D-2 shape, a few iterations, nothing else -- for rocprofv3 --kernel-trace
(scripts/dev/trace_summary.py prints the per-kernel table of one iteration).  python scripts/dev/vae_trace.py [batch] [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import standins, vae_transformer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
torch.manual_seed(0)
vae = standins.Hy3dgenLayoutShapeVAE().to(dev).half().eval().requires_grad_(False)
tr = vae_transformer.HipVaeTransformer.from_module(vae)
x0 = vae.post_kl(torch.randn(B, 3072, 64, device=dev, dtype=torch.float16)).detach()
go = torch.randn_like(x0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(n):
    if i == 2:
        torch.cuda.synchronize(); e0.record()
    out, saved = tr.forward_raw(x0, keep=True)
    tr.backward_raw(go, saved, tuple(x0.shape))
e1.record(); torch.cuda.synchronize()
print(f"B={B}: forward + backward {e0.elapsed_time(e1) / (n - 2):.2f} ms per iteration", flush=True)
