"""Cached 65^3 forward (+ rows backward) of the geometry decoder by row-block size, interleaved in one process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import standins
from followmyhold_amd.geo_decode import HipGeoDecoder
dev = torch.device("cuda", 0)
torch.manual_seed(0)
vae = standins.StandInShapeVAE(num_latents=3072, embed_dim=64, width=1024, heads=16, layers=1, num_freqs=8)
dec = vae.geo_decoder.to(dev).eval()
n = 65 ** 3
q = (torch.rand(1, n, 3, device=dev) * 2.2 - 1.1).half().float()
lat = torch.randn(1, 3072, 1024, device=dev).half()
chunks = [int(c) for c in (sys.argv[1:] or ["49152", "65536", "98304", "147456"])]
hips = {c: HipGeoDecoder.from_module(dec, device=dev, chunk_rows=c) for c in chunks}
qcs = {c: hips[c].grid_queries(q.reshape(-1, 3)) for c in chunks}
def run(c, reps=5):
    hip, qc = hips[c], qcs[c]
    with torch.no_grad():
        out = hip(qc, lat)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            hip._prepared = None
            out = hip(qc, lat)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out
ts = {c: [] for c in chunks}; outs = {}
for _ in range(4):
    for c in chunks:
        t, o = run(c); ts[c].append(t); outs[c] = o
for c in chunks:
    print(f"chunk {c}: {min(ts[c]):.3f} ms (runs {['%.3f' % t for t in ts[c]]}), equal to chunk {chunks[0]}: {bool(torch.equal(outs[c], outs[chunks[0]]))}, workspace {hips[c].workspace.numel() / 2**20:.0f} MiB", flush=True)
