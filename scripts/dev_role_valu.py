"""VALU instructions of k_stage2's roles: NB images, 20 eager steps with FOHO_DEBUG_SKIP_ROLES from the environment (STAMPS
build); run under `rocprofv3 --pmc SQ_INSTS_VALU ...` (scripts/dev_r03_s.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import _lib as L_
L_.SO_PATH = os.path.join(ROOT, "followmyhold_amd", "libfoho_hip_stamps.so")
from followmyhold_amd import engine as E, synthetic
NB = int(os.environ.get("NB", "8"))
mask = os.environ.pop("ROLE_MASK", "0")
rf = E.hip_render_fn("cuda")
scs = [synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, seed=s) for s in range(NB)]
gb = E.GuidanceBatch(scs); cfgu, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
for _ in range(25): gb.step(cfgu)
torch.cuda.synchronize()
os.environ["FOHO_DEBUG_SKIP_ROLES"] = mask
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
for _ in range(20): gb.step(cfg)
torch.cuda.synchronize()
