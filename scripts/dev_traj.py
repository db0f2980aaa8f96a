"""Trajectory of the bench scene: loss terms, parameters and per-kernel time every 20 steps (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
gb = E.GuidanceBatch([sc]); cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
for k in range(321):
    gb.step(cfg)
    if k % 20 == 0 or int(gb.flags[0]) != 0:
        torch.cuda.synchronize()
        l = gb.loss_dict(0)
        cfg0, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
        t = gb.step_profiled(cfg0)
        hits = [(gb.region("p2f", torch.int32, (2, -1))[r] >= 0).sum().item() for r in range(2)]
        print(k, "total %.3f n0 %.3f d0 %.4f n1 %.3f d1 %.4f sil %.4f cont %.5f" % (l["total"], l["normal0"], l["disp0"], l["normal1"], l["disp1"], l["sil1"], l["contact"]),
              "hits", hits, "stage2 %.1f resolve %.1f" % (t["k_stage2"] * 1e3, t["k_resolve"] * 1e3),
              "p", [round(x, 3) for x in gb.params[0].tolist()])
    if int(gb.flags[0]) != 0: break
