"""Is the forward pass deterministic at the oracle's step-1 parameters of the configs[1] scene?  Repeats the HIP step and
dumps the silhouette pixels that differ from the oracle whenever the total changes.  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import engine as E, synthetic
from oracle import clib, step_ref as S
n = min(32, len(os.sched_getaffinity(0))); clib.set_threads(n); torch.set_num_threads(n)
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
tsc = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
st = S.JointStepper(tsc, S.make_params(), denoise_i=19, grid_res=64)
st.step(update=True)
p_now = {kk: v.detach().clone().numpy() for kk, v in st.p.items()}
total, terms, aux, grads = st.step(update=False)
sil_ref = aux["render"]["sil"].detach().numpy().reshape(-1)
pairs = aux["render"]["sel"]["pairs"]; pd = aux["render"]["sel"]["pair_dist"]
t = (sc["hand_mask"] | sc["obj_mask"]).reshape(-1)
cfg0, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
print("oracle total", float(total), "sil", float(terms["sil_hoi"]))
seen = {}
for rep in range(40):
    gb = E.GuidanceBatch([sc]) if rep % 2 == 0 else gb
    gb.set_params(0, **p_now); gb.step(cfg0); torch.cuda.synchronize()
    tot = gb.loss_dict(0)["total"]; sil = gb.loss_dict(0)["sil1"]
    prod = gb.region("prod", torch.float32, (2, -1))[1].cpu().numpy(); p2f = gb.region("p2f", torch.int32, (2, -1))[1].cpu().numpy()
    a_hip = np.where(p2f >= 0, np.float32(1.0) - prod, np.float32(0.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        bce = lambda a: -(t * np.maximum(np.log(a), -100) + (1 - t) * np.maximum(np.log(np.float32(1) - a), -100))
        d = bce(a_hip) - bce(sil_ref)
    bad = np.flatnonzero(np.abs(d) > 1e-2)
    key = (tot, len(bad))
    if key not in seen:
        seen[key] = 0
        print("rep", rep, "total %.6f sil %.6f" % (tot, sil), "pixels whose BCE differs by > 0.01:", len(bad), "fragc", [int(x) for x in gb.region("frag_count", torch.int32, (2, -1))[1].cpu().numpy()[bad[:6]]])
        for px in bad[:6]:
            s_ = pairs[:, 0] == px
            print("   px", int(px), "t", int(t[px]), "alpha hip %.9g ref %.9g  1-a hip %.6g ref %.6g" % (a_hip[px], sil_ref[px], 1 - a_hip[px], 1 - sil_ref[px]),
                  "oracle frags (face, sd, x):", [(int(f), float(s), float(np.float32(-s) / np.float32(1e-8))) for f, s in zip(pairs[s_, 1], pd[s_])])
    seen[key] += 1
print(seen)
