"""Hit pixels and per-kernel time along one 50-iteration window of the bench scene (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind=sys.argv[1] if len(sys.argv) > 1 else "20k", H=512, W=512, seed=0)
gb = E.GuidanceBatch([sc]); cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
cfg0, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
for k in range(51):
    if k % 5 == 0:
        torch.cuda.synchronize()
        t = gb.step_profiled(cfg0)
        hits = [(gb.region("p2f", torch.int32, (2, -1))[r] >= 0).sum().item() for r in range(2)]
        print(k, "hits", hits, "total %.2f" % gb.loss_dict(0)["total"], " ".join("%s %.1f" % (n[2:], v * 1e3) for n, v in t.items()), "scale_obj %.3f" % gb.params[0, 8].item())
    gb.step(cfg)
