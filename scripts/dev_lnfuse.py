"""Cached 65^3 forward of the geometry decoder with the LayerNorms folded into the GEMMs (default) against the chain with LayerNorm
kernels (FOHO_GEO_LNFUSE=0), interleaved in one process; and the difference of the logits.  python scripts/dev_lnfuse.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import standins
from followmyhold_amd.geo_decode import HipGeoDecoder

dev = torch.device("cuda", 0)
torch.manual_seed(0)
vae = standins.StandInShapeVAE(num_latents=3072, embed_dim=64, width=1024, heads=16, layers=1, num_freqs=8)
dec = vae.geo_decoder.to(dev).eval()
with torch.no_grad():
    for ln in (dec.block.ln_2, dec.ln_post):
        ln.weight.add_(0.1 * torch.randn_like(ln.weight)); ln.bias.add_(0.1 * torch.randn_like(ln.bias))
hip = HipGeoDecoder.from_module(dec, device=dev)
n = 65 ** 3
q = (torch.rand(1, n, 3, device=dev) * 2.2 - 1.1).half().float()
lat = torch.randn(1, 3072, 1024, device=dev).half()
xyz = q.reshape(-1, 3)
qc = hip.grid_queries(xyz)

def run(mode, reps=5):
    if mode: os.environ.pop("FOHO_GEO_LNFUSE", None)
    else: os.environ["FOHO_GEO_LNFUSE"] = "0"
    with torch.no_grad():
        out = hip(qc, lat)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            hip._prepared = None
            out = hip(qc, lat)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out

ts = {0: [], 1: []}
outs = {}
for _ in range(4):
    for m in (1, 0):
        t, o = run(m); ts[m].append(t); outs[m] = o
print("folded  ms:", ["%.3f" % t for t in ts[1]])
print("kernels ms:", ["%.3f" % t for t in ts[0]])
d = (outs[1].float() - outs[0].float()).abs()
print("max |diff|", d.max().item(), "mean", d.mean().item(), "logit abs max", outs[0].float().abs().max().item())
with torch.no_grad():
    idx = torch.randperm(n, device=dev)[:20000]
    ref = dec(qc[:, idx].half(), lat.float())
for m in (1, 0):
    print("mode", m, "max |hip - torch fp32| on 20k rows:", (outs[m][:, idx].float() - ref).abs().max().item())
