"""Device-side timeline of one step inside a 50-iteration hipGraph replay: per kernel, first workgroup start -> last workgroup
end, and the gaps in between (needs `make -C followmyhold_amd/csrc STAMPS=1`; run on the GPU box)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import _lib as L
L.SO_PATH = os.path.join(ROOT, "followmyhold_amd", "libfoho_hip_stamps.so")
from followmyhold_amd import engine as E, synthetic
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind=sys.argv[1] if len(sys.argv) > 1 else "20k", H=512, W=512, seed=0, crop=os.environ.get("CROP"))
phase = sys.argv[2] if len(sys.argv) > 2 else "C"
cfgu, nr = E.phase_cfg(phase, denoise_i=19, do_update=True)
gb = E.GuidanceBatch([sc], n_renders=nr)
names = ["xform", "stage2", "resolve", "loss", "pix_bwd", "vert_bwd"]
KS_K, KS_WG = 6, 8192
out = (ctypes.c_ulonglong * (KS_K * KS_WG * 2))()
g1 = gb.capture(cfgu)
g3 = gb.capture(cfgu, steps_per_graph=3)       # deferred update: the stamps that survive are those of the LAST iteration
for mode in ("eager", "graph1", "graph3-deferred"):
    rows = []
    for rep in range(6):
        gb.reset_optimizer(); gb.params.copy_(torch.tensor([1, 0, 0, 0, 1, 0, 0, 0] * 2, dtype=torch.float32, device="cuda"))
        for _ in range(10): gb.step(cfgu)
        torch.cuda.synchronize()
        gb.lib.foho_debug_spans_clear()
        torch.cuda.synchronize()
        gb.step(cfgu) if mode == "eager" else (g1.replay() if mode == "graph1" else g3.replay())
        torch.cuda.synchronize()
        gb.lib.foho_debug_spans(out)
        a = np.frombuffer(out, dtype=np.uint64).reshape(KS_K, KS_WG, 2).astype(np.int64)
        st = np.array([a[k, :, 0][a[k, :, 0] > 0].min() for k in range(KS_K)]); en = a[:, :, 1].max(1)
        if rep == 0 and mode == "graph1":
            for k in range(KS_K):     # how the workgroups of a launch are spread in time
                s_, e_ = a[k, :, 0], a[k, :, 1]; m = s_ > 0
                print("   %-8s %5d WGs: starts within %.1f us, WG life median %.1f max %.1f us, 90%% of WGs done after %.1f us" % (
                    names[k], m.sum(), (s_[m].max() - s_[m].min()) / 100.0, np.median(e_[m] - s_[m]) / 100.0, (e_[m] - s_[m]).max() / 100.0,
                    (np.percentile(e_[m], 90) - s_[m].min()) / 100.0))
                if k == 1:
                    o = 0
                    for nm, cnt in (("knn", 273), ("raster hand", 194), ("raster obj", 320), ("inside", 344), ("normals", 44), ("kps", 6), ("edge", 41)):
                        sl = slice(o, o + cnt); o += cnt
                        lf = (e_[sl] - s_[sl]) / 100.0
                        print("            role %-12s life median %.1f max %.1f us, last end +%.1f us" % (nm, np.median(lf), lf.max(), (e_[sl].max() - s_[m].min()) / 100.0))
                        if nm == "inside":
                            print("              by block:", " ".join("%.0f" % v for v in lf[::8]), "| start offsets:", " ".join("%.1f" % ((v - s_[m].min()) / 100.0) for v in s_[sl][::40]))
                life = np.where(m, e_ - s_, 0); top = np.argsort(-life)[:6]
                print("            longest:", ", ".join("wg %d: %.1f us (start +%.1f)" % (i, life[i] / 100.0, (s_[i] - s_[m].min()) / 100.0) for i in top),
                      "| last to end:", ", ".join("wg %d" % i for i in np.argsort(-np.where(m, e_, 0))[:4]))
        rows.append(np.concatenate([(en - st) / 100.0, (st[1:] - en[:-1]) / 100.0, [(en[-1] - st[0]) / 100.0]]))
    r = np.median(np.array(rows), 0)
    print(mode, " ".join("%s %.1f" % (n, v) for n, v in zip(names, r[:6])), "| gaps", " ".join("%.1f" % v for v in r[6:11]), "| total %.1f us" % r[11])
