"""bench.py's vae_transformer and pipeline_iteration records alone.  python scripts/dev/pipe_iter.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from followmyhold_amd import engine as E, synthetic
dev = torch.device("cuda", 0)
scene = synthetic.build_scene(E.hip_render_fn(dev), obj_kind="20k", H=512, W=512, seed=0)
print(json.dumps(bench.vae_transformer_record(torch, dev)), flush=True)
print(json.dumps(bench.pipeline_iteration_record(E, torch, scene, dev)), flush=True)
