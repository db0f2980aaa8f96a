"""Timing experiment: cost of each k_stage2 role (run on the GPU box)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import torch
from followmyhold_amd import engine as E, synthetic
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
gb = E.GuidanceBatch([sc]); cfg, _ = E.phase_cfg("C")
for _ in range(5): gb.step(cfg)
acc = {}
for _ in range(20):
    for k, v in gb.step_profiled(cfg).items(): acc[k] = acc.get(k, 0) + v / 20
print(os.environ.get("FOHO_DEBUG_SKIP_ROLES", "0"), {k: round(v * 1e3, 1) for k, v in acc.items() if k in ("k_stage2", "k_resolve", "k_xform", "k_loss")})
''' % ROOT
for m in [0, 1, 2, 4, 8, 16, 32, 64, 96]:
    env = dict(os.environ, FOHO_DEBUG_SKIP_ROLES=str(m))
    print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
