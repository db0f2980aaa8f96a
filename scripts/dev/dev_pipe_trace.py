"""One pipeline iteration (stand-in networks of the Hunyuan3D-2 shape, HIP decoder) five times -- for rocprofv3 --kernel-trace: how much of
the iteration's wall time is kernel time?  python scripts/dev/dev_pipe_trace.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import engine as E, geo_decode, pipeline as PLN, standins, synthetic, vae_transformer
dev = torch.device("cuda", 0)
scene = dict(synthetic.build_scene(E.hip_render_fn(dev), obj_kind="20k", H=512, W=512, seed=0))
res = 64
g = np.linspace(-1.1, 1.1, res + 1, dtype=np.float32)
xyz = torch.from_numpy(np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)).to(dev)
T = np.array(scene["T_h2m"], np.float32); T[:3, :3] *= 0.9 * 0.06; scene["T_h2m"] = T
gb = E.GuidanceBatch([scene], device=dev, obj_capacity=(32768, 65536))
obj = E.SdfObjective(gb, xyz, res)
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
torch.manual_seed(0)
vae = standins.StandInShapeVAE(num_latents=3072, embed_dim=64, width=1024, heads=16, layers=16, num_freqs=8).to(dev).half().eval()
vae.requires_grad_(False)
geo_decode.install(vae, device=dev)
if os.environ.get("FOHO_TORCH_VAE") != "1":
    vae_transformer.install(vae, device=dev)
lat = torch.randn(1, 3072, 64, device=dev, dtype=torch.float16)
noise = torch.zeros_like(lat).requires_grad_(True)
gsz = (res + 1,) * 3
def one():
    noise.grad = None
    sdf = PLN.latent2sdf(lat + 0.1 * noise, xyz, gsz, vae, dev)
    loss = obj(sdf.reshape(1, -1), cfg)
    PLN._bound_active_rows(vae, obj.active_rows()[0])
    loss.sum().backward()
for _ in range(3):
    one()
torch.cuda.synchronize()
print("MARK start", flush=True)
t0 = time.perf_counter()
for _ in range(5):
    one()
torch.cuda.synchronize()
print(f"iteration: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
