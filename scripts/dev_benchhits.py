import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from followmyhold_amd import engine as E, synthetic
dev = torch.device("cuda", 0)
sc = synthetic.build_scene(E.hip_render_fn(dev), obj_kind="20k", H=512, W=512, seed=0)
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
group, run_steps = bench.make_runner(E, torch, [sc], 1, dev, cfg, 50)
gb = group.batches[0]
def hits(tag):
    torch.cuda.synchronize()
    p = gb.region("p2f", torch.int32, (2, -1))
    print(tag, [(p[r] >= 0).sum().item() for r in range(2)], "params", [round(x, 3) for x in gb.params[0].tolist()], "t", int(gb.adam_t[0]), "total", gb.loss_dict(0)["total"])
run_steps(50); hits("after 50 (one multi graph)")
run_steps(25); hits("after restart + 25 single")
cfg0, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
gb.step_profiled(cfg0); hits("after profiled")
run_steps(5); hits("after restart + 5")
run_steps(1); hits("after restart + 1")
