"""The pipeline's final decode (PL:1626-1642): latent -> VAE transformer -> geometry decoder on the 385^3 grid -> iso-surface at
resolution 384, with stand-in networks of the Hunyuan3D-2 shape -- does it run, how long does it take, how much memory."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import standins, geo_decode, ops, pipeline as PLN
from followmyhold_amd.facade import generate_dense_grid_points
dev = torch.device("cuda", 0)
torch.manual_seed(0)
vae = standins.StandInShapeVAE(num_latents=3072, embed_dim=64, width=1024, heads=16, layers=16, num_freqs=8).to(dev).half().eval()
vae.requires_grad_(False)
geo_decode.install(vae)
lat = torch.randn(1, 3072, 64, device=dev).half()
for res in (64, int(os.environ.get("RES", "384"))):
    bmin, bmax = np.array([-1.1] * 3, dtype=np.float32), np.array([1.1] * 3, dtype=np.float32)
    xyz_np, gsz, _ = generate_dense_grid_points(bmin, bmax, octree_depth=5, octree_resolution=res, indexing="ij")
    xyz = torch.as_tensor(xyz_np, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    for rep in range(2):
        t0 = time.perf_counter()
        with torch.no_grad():
            sdf = PLN.latent2sdf(lat, xyz, gsz, vae, dev)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        v, f, _ = ops.flexicubes(xyz, sdf[0].flatten(), res)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"res {res}: {xyz.shape[0]} points, latent2sdf {(t1 - t0) * 1e3:.1f} ms, flexicubes {(t2 - t1) * 1e3:.1f} ms, {v.shape[0]} vertices / {f.shape[0]} faces, "
              f"sdf finite {bool(torch.isfinite(sdf).all())}, inside fraction {float((sdf < 0).float().mean()):.4f}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
