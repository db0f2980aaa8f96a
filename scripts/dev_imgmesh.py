import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from followmyhold_amd import engine as E, ops
from oracle import ref_ops as R, clib
from test_inputs import _image_mesh
v, f = _image_mesh(96)
H = W = 128
cam = R.Camera(60.0, H, W)
ndc = R.world_to_ndc(torch.from_numpy(v), cam)
fv = ndc[torch.from_numpy(f)].numpy()
blur = R.blur_radius_from_sigma()
p2f, zb, ba, di = clib.rasterize(fv, H, W, blur, K=4)
out = ops.raster_fwd(ndc.cuda(), torch.from_numpy(f).int().cuda(), H, W, blur, 1e-8)
hp = out["pix_to_face"].cpu().numpy(); hz = out["zbuf"].cpu().numpy(); hd = out["dists"].cpu().numpy()
bad = np.argwhere(hp != p2f[..., 0])
print("mismatches", len(bad))
for y, x in bad[:8]:
    print((y, x), "oracle K=4 faces", p2f[y, x].tolist(), "z", [float.hex(float(z)) for z in zb[y, x]], "d", di[y, x].tolist())
    print("      hip face", hp[y, x], "z", float.hex(float(hz[y, x])), "d", hd[y, x])
