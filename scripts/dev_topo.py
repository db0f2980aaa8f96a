"""topology_changing sub-record of bench.py alone (run on the GPU box)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
from followmyhold_amd import engine as E, synthetic
dev = torch.device("cuda", 0)
sc = synthetic.build_scene(E.hip_render_fn(dev), obj_kind="20k", H=512, W=512, seed=0)
print(json.dumps(bench.topology_record(E, torch, sc, dev)))
