import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from followmyhold_amd import engine as E, synthetic
dev = torch.device("cuda", 0)
rf = E.hip_render_fn(dev)
scene = synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, seed=0)
for rep in range(3):
    r = bench.pipeline_iteration_record(E, torch, scene, dev, iters=3)
    print(rep, {k: (round(v["iteration_ms"], 1), v["grad_abs_max"]) for k, v in r.items() if isinstance(v, dict)}, r["faces"], r["sdf_max_abs_diff_between_decoders"], flush=True)
