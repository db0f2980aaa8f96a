"""Diagnostic: which loss term carries the vertex-gradient difference on the crop scene late in the loop?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from followmyhold_amd import engine as E, synthetic, _lib as L
from oracle import clib, step_ref as S, ref_ops as R
clib.set_threads(32); torch.set_num_threads(32)
H = W = 512; P = H * W
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=H, W=W, seed=0, crop="hoi")
sct = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
st = S.JointStepper(sct, S.make_params(), denoise_i=19, grid_res=64)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 9
for k in range(K):
    st.step(update=True)
p_k = {kk: v.detach().clone() for kk, v in st.p.items()}
p = S.leafify(p_k, S.PARAM_KEYS)
ov = sct["obj_verts"].detach().clone().requires_grad_(True)
edges = R.unique_edges(sct["obj_faces"])
total, terms, aux = S.phase_c_loss(sct, p, ov, edges, 19, 20, grid_res=64)
gb = E.GuidanceBatch([sc])
gb.set_params(0, **{kk: v.numpy() for kk, v in p_k.items()})
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
specs = {"normal_hoi": ("r1.w_normal", 10.0), "disp_hoi": ("r1.w_disp", 10.0), "sil_hoi": ("r1.w_sil", 10.0), "contact": ("w_contact", 10.0),
         "edge": ("w_edge", 1.0), "verts_obj": ("w_verts_obj", 1e-3)}
sum_h = np.zeros((ov.shape[0], 3)); sum_r = np.zeros((ov.shape[0], 3))
WATCH = [126, 7170, 7148]
for name, (field, wgt) in specs.items():
    g_ref = torch.autograd.grad(wgt * terms[name], ov, retain_graph=True, allow_unused=True)[0]
    g_ref = np.zeros((ov.shape[0], 3)) if g_ref is None else g_ref.numpy().astype(np.float64)
    cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
    keep = {f: None for f in ("w_kps", "w_trans_hand", "w_trans_obj", "w_verts_obj", "w_edge", "w_contact")}
    vals = {}
    for f in keep:
        vals[f] = getattr(cfg, f); setattr(cfg, f, 0.0)
    rv = {}
    for r in range(2):
        for f in ("w_normal", "w_disp", "w_sil"):
            rv[(r, f)] = getattr(cfg.render[r], f); setattr(cfg.render[r], f, 0.0)
    cfg.use_intersection = 0
    if field.startswith("r1."):
        setattr(cfg.render[1], field[3:], rv[(1, field[3:])])
    else:
        setattr(cfg, field, vals[field])
    gb.step(cfg); torch.cuda.synchronize()
    g_h = gb.grad_obj_verts(0).cpu().numpy().astype(np.float64)
    d = np.linalg.norm(g_h - g_ref, axis=1)
    print(f"{name:12s}: |g_ref| {np.linalg.norm(g_ref):.4e} |g_hip| {np.linalg.norm(g_h):.4e} rel {rel(g_h, g_ref):.2e} loss hip {gb.loss_dict(0)['total']:.6e} ref {float(wgt * terms[name]):.6e}", flush=True)
    for v in np.argsort(-d)[:3]:
        print(f"      vertex {v}: |diff| {d[v]:.3e} hip {g_h[v]} ref {g_ref[v]}")
    sum_h += g_h; sum_r += g_ref
    for v in WATCH:
        print(f"      watch {v}: hip {g_h[v]} ref {g_ref[v]}")
# totals
g_tot_ref = torch.autograd.grad(total, ov, retain_graph=True)[0].numpy().astype(np.float64)
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
gb.step(cfg); torch.cuda.synchronize()
g_tot_hip = gb.grad_obj_verts(0).cpu().numpy().astype(np.float64)
print(f"TOTAL: hip vs ref {rel(g_tot_hip, g_tot_ref):.2e}; hip total vs sum of hip terms {rel(g_tot_hip, sum_h):.2e}; ref total vs sum of ref terms {rel(g_tot_ref, sum_r):.2e}")
for v in WATCH:
    print(f"      watch {v}: hip total {g_tot_hip[v]} sum {sum_h[v]} | ref total {g_tot_ref[v]} sum {sum_r[v]}")
d = np.linalg.norm(g_tot_hip - g_tot_ref, axis=1)
for v in np.argsort(-d)[:5]:
    print(f"      vertex {v}: |diff| {d[v]:.3e} hip {g_tot_hip[v]} ref {g_tot_ref[v]}")
