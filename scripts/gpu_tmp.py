import sys, os, time, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from followmyhold_amd import standins
from followmyhold_amd.geo_decode import HipGeoDecoder
dev = torch.device("cuda", 0)
torch.manual_seed(0)
vae = standins.StandInShapeVAE(num_latents=3072, embed_dim=64, width=1024, heads=16, layers=1, num_freqs=8)
mod = vae.geo_decoder.to(dev).eval()
n = 65 ** 3
q = (torch.rand(n, 3, device=dev) * 2.2 - 1.1).half().float()
lat = torch.randn(1, 3072, 1024, device=dev).half()
def bench(nstreams, chunk):
    decs = [HipGeoDecoder.from_module(mod, device=dev, chunk_rows=chunk) for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    per = (n + nstreams - 1) // nstreams
    per = (per + chunk - 1) // chunk * chunk if nstreams > 1 else n
    parts = [q[i * per:(i + 1) * per] for i in range(nstreams)]
    for d in decs:
        d.prepare(lat)
    torch.cuda.synchronize()
    def run():
        cur = torch.cuda.current_stream()
        outs = []
        for d, s, p in zip(decs, streams, parts):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(d.decode(p))
        for s in streams:
            cur.wait_stream(s)
        return outs
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); o = run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, torch.cat(o)
ref_t, ref = bench(1, 16384)
print(f"1 stream chunk 16384: {ref_t:.2f} ms")
for ns, ch in ((1, 8192), (1, 32768), (2, 8192), (2, 16384), (3, 8192), (4, 4096), (4, 8192)):
    t, o = bench(ns, ch)
    print(f"{ns} stream(s) chunk {ch}: {t:.2f} ms  equal {torch.equal(o, ref)}", flush=True)
