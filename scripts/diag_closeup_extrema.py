"""Diagnostic: the gradient through the extrema of the normal map's min-max normalisation, HIP statistics vs the oracle."""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 9
for k in range(K):
    st.step(update=True)
p_k = {kk: v.detach().clone() for kk, v in st.p.items()}
p = S.leafify(p_k, S.PARAM_KEYS)
ov = sct["obj_verts"].detach().clone().requires_grad_(True)
edges = R.unique_edges(sct["obj_faces"])
total, terms, aux = S.phase_c_loss(sct, p, ov, edges, 19, 20, grid_res=64)
r = aux["render"]
rgba = r["rgba"].detach()
n = rgba[..., :3].clone().requires_grad_(True)
mask = rgba[..., 3] > 0
mn, mx = n.min(), n.max()
D = mx - mn + 1e-6
nn = torch.where(mask[..., None], (n - mn) / D, torch.zeros_like(n))
nn.retain_grad()
hoi = sct["hand_mask"] | sct["obj_mask"]
loss = 10.0 * R.normal_alignment_loss(nn, sct["moge_normal"], valid_mask=hoi)
loss.backward()
g = nn.grad.double(); nd = nn.detach().double(); Dd = float(D)
g_mn64 = float((g * (nd - 1.0) * mask[..., None]).sum() / Dd); g_mx64 = float(-(g * nd * mask[..., None]).sum() / Dd)
g32 = nn.grad; n32 = nn.detach()
g_mn32 = float((g32 * (n32 - 1.0) * mask[..., None]).sum() / D); g_mx32 = float(-(g32 * n32 * mask[..., None]).sum() / D)
is_mn = (n.detach() == mn); is_mx = (n.detach() == mx)
tot_at_mn = float(n.grad[is_mn].double().sum()); direct_at_mn = float((g / Dd)[is_mn].sum())
print(f"oracle: mn {float(mn)!r} mx {float(mx)!r} D {Dd!r} cnt_mn {int(is_mn.sum())} cnt_mx {int(is_mx.sum())}")
print(f"oracle: g_mn (double sum) {g_mn64:.9e} (float sum) {g_mn32:.9e} | autograd: total at min pixels {tot_at_mn:.9e} - direct {direct_at_mn:.9e} = {tot_at_mn - direct_at_mn:.9e}")
print(f"oracle: g_mx (double sum) {g_mx64:.9e} (float sum) {g_mx32:.9e}; sum |terms| for g_mn {float((g * (nd - 1.0)).abs().sum() / Dd):.4e}")
gb = E.GuidanceBatch([sc])
gb.set_params(0, **{kk: v.numpy() for kk, v in p_k.items()})
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
gb.step(cfg); torch.cuda.synchronize()
stt = gb.region("stats", torch.float32, (2, 32))[1].cpu().numpy()
names = "S_MN S_MX S_DN S_DMN S_DMX S_DD S_M S_CNT_MX S_CNT_MN S_CNT_DMX S_CNT_DMN S_G_MX S_G_MN S_G_DMX S_G_DMN S_L_NORMAL S_L_DISP S_L_SIL S_KN S_KD".split()
print("hip   :", {nm: float(stt[i]) for i, nm in enumerate(names)})
print(f"ratio hip / oracle: g_mn {stt[12] / g_mn64:.8f} g_mx {stt[11] / g_mx64:.8f}")
