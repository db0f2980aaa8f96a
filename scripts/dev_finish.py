import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, numpy as np
from followmyhold_amd import engine as E, synthetic
dev = torch.device("cuda", 0)
sc = synthetic.build_scene(E.hip_render_fn(dev), obj_kind="20k", H=512, W=512, seed=0)
T = np.array(sc["T_h2m"], np.float32); T[:3, :3] *= 0.9 * 0.06; sc["T_h2m"] = T
res = 64
g = np.linspace(-1.1, 1.1, res + 1, dtype=np.float32)
xyz = torch.from_numpy(np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)).to(dev)
gb = E.GuidanceBatch([sc], device=dev, obj_capacity=(24576, 49152))
obj = E.SdfObjective(gb, xyz, res)
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
obj.run((torch.linalg.norm(xyz, dim=1) - 0.85).contiguous(), cfg, use_graph=False)
torch.cuda.synchronize()
for _ in range(50):
    gb.adopt_objects()
torch.cuda.synchronize()
print(obj.status())
