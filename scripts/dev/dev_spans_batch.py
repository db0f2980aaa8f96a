"""Device-side timeline of one k_stage2 launch at NB images per batch (one stream): per role, when its workgroups start, how
long they live, when the last one ends (needs `make -C followmyhold_amd/csrc STAMPS=1`; run on the GPU box)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import _lib as L
L.SO_PATH = os.path.join(ROOT, "followmyhold_amd", "libfoho_hip_stamps.so")
from followmyhold_amd import engine as E, synthetic
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rf = E.hip_render_fn("cuda")
scs = [synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, seed=s, crop=os.environ.get("CROP")) for s in range(NB)]
cfgu, nr = E.phase_cfg("C", denoise_i=19, do_update=True)
gb = E.GuidanceBatch(scs)
cd = lambda a, b: (a + b - 1) // b
Vh, Vo, Fh, Fo = 778, 10242, 1552, 20480
def rfpb(F, B):
    r = 8
    while r < 64 and r * 256 < F * B: r <<= 1
    return r
rf_h, rf_o = max(2, rfpb(Fh, NB) // 2), rfpb(Fo, NB)
roles = [("knn", cd(Vh, 64) * cd(Vo, 1024)), ("raster hand", cd(Fh, rf_h)), ("raster obj", cd(Fo, rf_o)), ("inside hand", cd(Fh, 16)),
         ("inside obj", cd(Fo, 64)), ("normals", cd(Vh + Vo, 256)), ("kps", 6), ("edge", cd(Vo, 256))]
nx = sum(c for _, c in roles)
print("NB", NB, "rf_h", rf_h, "rf_o", rf_o, "grid.x", nx, "total WGs", nx * NB)
KS_K, KS_WG = 6, 8192
out = (ctypes.c_ulonglong * (KS_K * KS_WG * 2))()
g1 = gb.capture(cfgu)
names = ["xform", "stage2", "resolve", "loss", "pix_bwd", "vert_bwd"]
for rep in range(3):
    gb.reset_optimizer(); gb.params.copy_(torch.tensor([1, 0, 0, 0, 1, 0, 0, 0] * 2, dtype=torch.float32, device="cuda").expand_as(gb.params))
    for _ in range(10): gb.step(cfgu)
    torch.cuda.synchronize(); gb.lib.foho_debug_spans_clear(); torch.cuda.synchronize()
    g1.replay(); torch.cuda.synchronize()
    gb.lib.foho_debug_spans(out)
    a = np.frombuffer(out, dtype=np.uint64).reshape(KS_K, KS_WG, 2).astype(np.int64)
    st = np.array([a[k, :, 0][a[k, :, 0] > 0].min() for k in range(KS_K)]); en = a[:, :, 1].max(1)
    print("rep", rep, " ".join("%s %.1f" % (n, v) for n, v in zip(names, (en - st) / 100.0)), "| total %.1f us" % ((en[-1] - st[0]) / 100.0))
    if rep < 2: continue
    s_, e_ = a[1, :nx * NB, 0].reshape(NB, nx), a[1, :nx * NB, 1].reshape(NB, nx)
    t0 = s_[s_ > 0].min()
    o = 0
    for nm, cnt in roles:
        ss, ee = s_[:, o:o + cnt], e_[:, o:o + cnt]; o += cnt
        m = ss > 0
        if not m.any(): continue
        lf = (ee - ss)[m] / 100.0
        print("   %-12s %5d WGs/img: start %.1f..%.1f us, life median %.1f p90 %.1f max %.1f, last end +%.1f us | per image last end: %s" % (
            nm, cnt, (ss[m].min() - t0) / 100.0, (ss[m].max() - t0) / 100.0, np.median(lf), np.percentile(lf, 90), lf.max(),
            (ee[m].max() - t0) / 100.0, " ".join("%.0f" % ((ee[b][m[b]].max() - t0) / 100.0) for b in range(NB))))
    # concurrency: resident workgroups over time
    ev = np.concatenate([np.stack([s_[s_ > 0], np.ones((s_ > 0).sum(), np.int64)], 1), np.stack([e_[s_ > 0], -np.ones((s_ > 0).sum(), np.int64)], 1)])
    ev = ev[np.argsort(ev[:, 0])]; conc = np.cumsum(ev[:, 1]); tt = (ev[:, 0] - t0) / 100.0
    print("   resident WGs at t =", " ".join("%d:%d" % (t, conc[np.searchsorted(tt, t) - 1]) for t in range(2, int(tt.max()), 4)))
