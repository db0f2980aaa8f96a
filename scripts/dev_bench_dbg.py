import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
gb = E.GuidanceBatch([sc]); cfg, _ = E.phase_cfg("C")
gb.reset_optimizer()
g = gb.capture(cfg)
def show(tag):
    torch.cuda.synchronize()
    print(tag, "flags", gb.flags.tolist(), "frac", gb.region("frac_count", torch.int32).tolist(), "t", gb.adam_t.tolist(),
          "total %.3f" % gb.loss_dict(0)["total"], "p", [round(x, 3) for x in gb.params[0].tolist()])
show("after capture")
for k in range(50):
    g.replay()
    if k % 10 == 0: show("settle %d" % k)
show("after settle")
gb.set_params(0, scale_hand=[1.0], trans_hand=[0, 0, 0], rot_hand=[1, 0, 0, 0], scale_obj=[1.0], trans_obj=[0, 0, 0], rot_obj=[1, 0, 0, 0])
gb.reset_optimizer()
show("after reset")
for k in range(60):
    g.replay()
    if k % 10 == 0 or int(gb.flags[0]): show("run %d" % k)
    if int(gb.flags[0]): break
