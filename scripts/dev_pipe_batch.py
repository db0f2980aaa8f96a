"""bench.py's `batch_of_4` measurement for other batch sizes: one inner iteration of B images the way call_batch runs them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import engine as E, synthetic, geo_decode, pipeline as PLN, standins
dev = torch.device("cuda", 0)
scene = dict(synthetic.build_scene(E.hip_render_fn(dev), obj_kind="20k", H=512, W=512, seed=0))
res = 64
g = np.linspace(-1.1, 1.1, res + 1, dtype=np.float32)
xyz = torch.from_numpy(np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)).to(dev)
T = np.array(scene["T_h2m"], np.float32); T[:3, :3] *= 0.9 * 0.06; scene["T_h2m"] = T
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
torch.manual_seed(0)
vae = standins.StandInShapeVAE(num_latents=3072, embed_dim=64, width=1024, heads=16, layers=16, num_freqs=8).to(dev).half().eval()
vae.requires_grad_(False)
geo_decode.install(vae, device=dev)
hip = vae.hip_geo
for B in (1, 2, 4, 8, 16):
    gb = E.GuidanceBatch([scene] * B, device=dev, obj_capacity=(32768, 65536))
    obj = E.SdfObjective(gb, xyz, res)
    lat = torch.randn(B, 3072, 64, device=dev, dtype=torch.float16)
    noise = torch.zeros_like(lat).requires_grad_(True)
    def one():
        noise.grad = None
        torch.cuda.synchronize(dev); a = time.perf_counter()
        with PLN.vae_attention_backend():
            pred = vae((1 / vae.scale_factor) * (lat + 0.1 * noise))
        sdf = torch.stack([-hip(hip.grid_queries(xyz), pred[b:b + 1]).view(-1).float() for b in range(B)], 0)
        loss = obj(sdf, cfg)
        gb.flags.cpu()
        PLN._bound_active_rows(vae, max(obj.active_rows()))
        loss.sum().backward()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - a) * 1e3
    one(); one()
    t = float(np.median([one() for _ in range(5)]))
    print(f"B={B}: {t:.2f} ms per iteration, {t / B:.2f} ms per image, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del gb, obj
