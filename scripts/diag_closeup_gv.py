"""Diagnostic: teacher-forced joint steps on the crop scene; where does the vertex gradient differ from the oracle's?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from followmyhold_amd import engine as E, synthetic
from oracle import clib, step_ref as S, ref_ops as R
clib.set_threads(32); torch.set_num_threads(32)
H = W = 512; P = H * W
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=H, W=W, seed=0, crop="hoi")
sct = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
st = S.JointStepper(sct, S.make_params(), denoise_i=19, grid_res=64)
gb = E.GuidanceBatch([sc])
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
order = [st.p[k] for k in E.PARAM_NAMES]
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    p_k = {kk: v.detach().clone() for kk, v in st.p.items()}
    gb.set_params(0, **{kk: v.numpy() for kk, v in p_k.items()})
    if k > 0:
        gb.adam_m[0].copy_(torch.cat([st.opt.state[p_]["exp_avg"].reshape(-1) for p_ in order]))
        gb.adam_v[0].copy_(torch.cat([st.opt.state[p_]["exp_avg_sq"].reshape(-1) for p_ in order]))
    gb.adam_t.fill_(k)
    total, terms, aux, grads = st.step(update=True)
    gb.step(cfg); torch.cuda.synchronize()
    g_h = gb.grad_obj_verts(0).cpu().numpy().astype(np.float64); g_r = grads["obj_verts"].numpy().astype(np.float64)
    d = np.linalg.norm(g_h - g_r, axis=1)
    rel = np.linalg.norm(g_h - g_r) / np.linalg.norm(g_r)
    prod = gb.region("prod", torch.float32, (2, 1, P))[1, 0].cpu().numpy()
    p2f = gb.region("p2f", torch.int32, (2, 1, P))[1, 0].cpu().numpy()
    a_h = np.where(p2f >= 0, np.float32(1) - prod, np.float32(0)).astype(np.float32)
    a_r = aux["render"]["sil"].detach().numpy().reshape(-1).astype(np.float32)
    diff = np.nonzero(a_h != a_r)[0]
    print(f"step {k}: gv rel {rel:.2e} |g_ref| {np.linalg.norm(g_r):.3e}; alpha differs on {len(diff)} pixels; loss rel {abs(gb.loss_dict(0)['total'] - float(total)) / abs(float(total)):.1e}", flush=True)
    top = np.argsort(-d)[:4]
    for v in top:
        print(f"   vertex {v}: |diff| {d[v]:.3e} hip {g_h[v]} ref {g_r[v]}")
    tgt = (sc["hand_mask"] | sc["obj_mask"]).reshape(-1)
    for px in diff[:12]:
        print(f"   pixel {px} ({px // W},{px % W}): alpha hip {a_h[px]!r} (1-a = {1 - np.float64(a_h[px]):.3e}) ref {a_r[px]!r} (1-a = {1 - np.float64(a_r[px]):.3e}) target {int(tgt[px])} face {p2f[px]}")
