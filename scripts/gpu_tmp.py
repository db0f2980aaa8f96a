import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_geo_decode import _decoder
from followmyhold_amd.geo_decode import HipGeoDecoder
for (width, heads, n_lat, n_q, chunk) in [(256, 4, 256, 5000, 2048), (1024, 16, 3072, 20000, 16384)]:
    dec = _decoder(width, heads, n_lat)
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, n_lat, width, generator=g).half().cuda()
    q = (torch.rand(1, n_q, 3, generator=g) * 2.2 - 1.1).half().cuda()
    hip = HipGeoDecoder.from_module(dec, chunk_rows=chunk)
    hip.set_kv(hip.kv_of(lat).detach())
    a = hip.decode(q.float()); b, saved = hip.decode_keep(q.float()); c = hip.decode(q.float()); d, _ = hip.decode_keep(q.float())
    torch.cuda.synchronize()
    print(width, "decode vs decode", (a - c).abs().max().item(), "keep vs keep", (b - d).abs().max().item(), "decode vs keep", (a - b).abs().max().item(),
          "rows differing", (a != b).nonzero().flatten()[:10].tolist(), int((a != b).sum()))
