"""configs[1], 50 joint steps: free-running HIP vs oracle trajectories, and the teacher-forced comparison (HIP evaluated at
the oracle's parameters of every step: loss and gradient errors along the whole trajectory).  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import engine as E, synthetic
from oracle import clib, step_ref as S
n = min(32, len(os.sched_getaffinity(0))); clib.set_threads(n); torch.set_num_threads(n)
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
tsc = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
st = S.JointStepper(tsc, S.make_params(), denoise_i=19, grid_res=64)
free = E.GuidanceBatch([sc]); forced = E.GuidanceBatch([sc])
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
cfg0, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / max(np.linalg.norm(b), 1e-30))
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    p_now = {kk: v.detach().clone().numpy() for kk, v in st.p.items()}
    total, terms, aux, grads = st.step(update=True)
    forced.set_params(0, **p_now); forced.step(cfg0); free.step(cfg); torch.cuda.synchronize()
    g = forced.grad_params[0].cpu().numpy(); gref = np.concatenate([grads[kk].numpy().reshape(-1) for kk in E.PARAM_NAMES])
    mism = int((forced.region("p2f", torch.int32, (2, -1))[1].cpu().numpy() != aux["render"]["sel"]["pix_to_face"].reshape(-1)).sum())
    pf = free.params[0].cpu().numpy(); po = np.concatenate([st.p[kk].detach().numpy().reshape(-1) for kk in E.PARAM_NAMES])
    print(k, "oracle %.5f forced %.5f (rel %.1e, grad rel %.1e, gv rel %.1e, p2f mism %d) free %.5f (rel %.1e) |dp|max %.1e at %d  gref %s" % (
        float(total), forced.loss_dict(0)["total"], abs(forced.loss_dict(0)["total"] - float(total)) / abs(float(total)), rel(g, gref),
        rel(forced.grad_obj_verts(0).cpu().numpy(), grads["obj_verts"].numpy()), mism, free.loss_dict(0)["total"],
        abs(free.loss_dict(0)["total"] - float(total)) / abs(float(total)), np.abs(pf - po).max(), int(np.abs(pf - po).argmax()),
        np.array2string(gref, precision=2, max_line_width=250)), flush=True)
    if abs(forced.loss_dict(0)["total"] - float(total)) > 1e-5 * abs(float(total)):
        l = forced.loss_dict(0)
        print("   terms hip", {kk: round(v, 6) for kk, v in l.items()})
        print("   terms ref", {kk: round(float(v), 6) for kk, v in terms.items()}, "n_int", aux["n_int"], "w_int", aux["w_int"])
        for r, ren in enumerate([aux["hand"]["render"], aux["render"]]):
            sel = ren["sel"]
            p2f = forced.region("p2f", torch.int32, (2, -1))[r].cpu().numpy(); ref = sel["pix_to_face"].reshape(-1)
            hit = ref >= 0
            zb = forced.region("zbuf", torch.float32, (2, -1))[r].cpu().numpy(); sd = forced.region("sdist", torch.float32, (2, -1))[r].cpu().numpy()
            print("   render", r, "p2f mism", int((p2f != ref).sum()), "z mism", int((zb[hit] != sel["zbuf"].reshape(-1)[hit]).sum()),
                  "sd mism", int((sd[hit] != sel["dists"].reshape(-1)[hit]).sum()), "frac px (sd > -1e-6)", int((sel["dists"].reshape(-1)[hit] > -1e-6).sum()),
                  "rgb min/max ref", float(ren["rgba"][..., :3].min()), float(ren["rgba"][..., :3].max()))
        print("   stats", forced.region("stats", torch.float32, (2, -1)).cpu().numpy()[:, :12])
        sil_ref = aux["render"]["sil"].detach().numpy().reshape(-1)
        prod = forced.region("prod", torch.float32, (2, -1))[1].cpu().numpy()
        p2f = forced.region("p2f", torch.int32, (2, -1))[1].cpu().numpy()
        a_hip = np.where(p2f >= 0, 1.0 - prod, 0.0).astype(np.float32)
        t = (sc["hand_mask"] | sc["obj_mask"]).reshape(-1)
        bce = lambda a: -(t * np.maximum(np.log(np.maximum(a, 0)), -100) + (1 - t) * np.maximum(np.log(np.maximum(1 - a, 0)), -100))
        with np.errstate(divide="ignore"):
            d = bce(a_hip.astype(np.float32)) - bce(sil_ref.astype(np.float32))
        bad = np.flatnonzero(np.abs(d) > 1e-3)
        print("   sil pixels that differ:", len(bad), "sum of BCE differences", float(d.sum()))
        pairs = aux["render"]["sel"]["pairs"]; pd = aux["render"]["sel"]["pair_dist"]
        fragc = forced.region("frag_count", torch.int32, (2, -1))[1].cpu().numpy()
        for px in bad[:8]:
            sel_ = pairs[:, 0] == px
            print("     px", int(px), "t", int(t[px]), "alpha hip %.9g ref %.9g" % (a_hip[px], sil_ref[px]), "1-a hip %.6g ref %.6g" % (1 - a_hip[px], 1 - sil_ref[px]),
                  "oracle frags (face, sdist):", [(int(f), float(s)) for f, s in zip(pairs[sel_, 1], pd[sel_])], "x = -sd/sigma", [float(np.float32(-s) / np.float32(1e-8)) for s in pd[sel_]])
