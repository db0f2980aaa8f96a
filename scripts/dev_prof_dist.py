import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from followmyhold_amd import engine as E, synthetic
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
gb = E.GuidanceBatch([sc]); cfgu, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
for _ in range(25): gb.step(cfgu)
cfg_frozen, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
rows = []
for _ in range(60):
    rows.append(gb.step_profiled(cfg_frozen, deferred=True))
for k in rows[0]:
    v = np.array([r[k] for r in rows]) * 1e3
    print(k, "mean %.1f median %.1f min %.1f max %.1f" % (v.mean(), np.median(v), v.min(), v.max()), np.round(np.sort(v)[-5:], 1))
