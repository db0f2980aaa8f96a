import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from followmyhold_amd import engine as E, synthetic
dev = torch.device("cuda", 0)
rf = E.hip_render_fn(dev)
scene = synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, seed=0)
print(json.dumps(bench.pipeline_iteration_record(E, torch, scene, dev), indent=1))
