import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
gb = E.GuidanceBatch([sc]); cfg, _ = E.phase_cfg("C")
for k in range(130):
    gb.step(cfg); torch.cuda.synchronize()
    if k % 10 == 0 or int(gb.flags[0]) != 0:
        l = gb.loss_dict(0)
        print(k, "flags", int(gb.flags[0]), "frac", gb.region("frac_count", torch.int32).tolist(), "total %.4f sil %.4f" % (l["total"], l["sil1"]),
              "p", [round(x, 4) for x in gb.params[0].tolist()])
    if int(gb.flags[0]) != 0: break
