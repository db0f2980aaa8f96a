"""In-kernel time stamps of the raster role of k_stage2 for a batch of NB images (needs `make -C followmyhold_amd/csrc STAMPS=1`)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import _lib as L
L.SO_PATH = os.path.join(ROOT, "followmyhold_amd", "libfoho_hip_stamps.so")
from followmyhold_amd import engine as E, synthetic
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0, crop=os.environ.get("CROP"))
NB = int(os.environ.get("NB", "16"))
gb = E.GuidanceBatch([sc] * NB); cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
for _ in range(5): gb.step(cfg)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 1024)()
for m in os.environ.get("MASKS", "0,61").split(","):
    os.environ["FOHO_DEBUG_SKIP_ROLES"] = m
    for rep in range(2):
        gb.lib.foho_debug_clear()
        t = gb.step_profiled(cfg)
        torch.cuda.synchronize()
        gb.lib.foho_debug_stamps(out)
        a = np.array(out[:], dtype=np.int64)
        d = lambda i, j: (a[j] - a[i]) / 100.0
        print("mask %s stage2 %.1f us" % (m, t["k_stage2"] * 1e3))
        for nm, o in (("hand", 50), ("obj", 60)):
            print("   raster %s blk: setup %.2f barrier %.2f enumerate %.2f barrier %.2f evaluate %.2f [loop %.2f, barrier %.2f, reserve + stores %.2f] (T=%d candidates)" % (
                nm, d(o, o + 1), d(o + 1, o + 2), d(o + 2, o + 3), d(o + 3, o + 4), d(o + 4, o + 5), d(o + 4, o + 6), d(o + 6, o + 7), d(o + 7, o + 5), a[o + 8]))
