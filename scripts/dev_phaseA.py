"""Phase A at 512x512, teacher-forced: where do the HIP gradients leave the oracle's?  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from followmyhold_amd import engine as E, synthetic
from oracle import clib, step_ref as S, ref_ops as R
import test_fullsize_gpu as T
T._threads()
sc = T._scene("20k"); p = T._perturbed()
st = S.PhaseStepper("A", T._t(sc), p)
gb = E.GuidanceBatch([sc], n_renders=1)
cfg0, _ = E.phase_cfg("A", do_update=False)
t = sc["hand_mask"].reshape(-1)
for k in range(5):
    gb.set_params(0, **{kk: v.detach().numpy() for kk, v in st.p.items()})
    total, terms, aux, grads = st.step(update=True)
    gb.step(cfg0); torch.cuda.synchronize()
    l = gb.loss_dict(0); g = gb.grad_params[0].cpu().numpy()
    print(k, "total", l["total"], float(total), "sil", l["sil0"], float(terms["sil_hand"]), "flags", int(gb.flags[0]))
    for kk in ["scale_hand", "trans_hand", "rot_hand"]:
        print("    ", kk, g[E.PARAM_SLICES[kk]], grads[kk].numpy())
    prod = gb.region("prod", torch.float32, (1, -1))[0].cpu().numpy(); p2f = gb.region("p2f", torch.int32, (1, -1))[0].cpu().numpy()
    fragc = gb.region("frag_count", torch.int32, (1, -1))[0].cpu().numpy()
    a = np.where(p2f >= 0, np.float32(1) - prod, np.float32(0)).astype(np.float32)
    sil_ref = aux["render"]["sil"].detach().numpy().reshape(-1)
    hit = p2f >= 0
    print("     alpha != ref:", int((a != sil_ref).sum()), "| px with alpha==1, prod>0:", int(((a == 1) & (prod > 0) & hit).sum()), "of which t==0:", int(((a == 1) & (prod > 0) & hit & ~t).sum()),
          "| 0<prod<1e-30:", int(((prod > 0) & (prod < 1e-30) & hit).sum()), "| ref alpha==1 & t==0:", int(((sil_ref == 1) & ~t).sum()))
    pairs = aux["render"]["sel"]["pairs"]; pd = aux["render"]["sel"]["pair_dist"]
    cnt = np.bincount(pairs[:, 0], minlength=len(t))
    print("     max fragments per pixel (oracle K-buffer)", cnt.max(), " frac list count", int(gb.region("frac_count", torch.int32)[0]))
    bad = np.flatnonzero(a != sil_ref)
    for px in bad[:5]:
        s_ = pairs[:, 0] == px
        print("       px", int(px), "t", int(t[px]), "alpha hip %.9g ref %.9g prod %.6g" % (a[px], sil_ref[px], prod[px]), [(int(f), float(s)) for f, s in zip(pairs[s_, 1], pd[s_])][:6])
