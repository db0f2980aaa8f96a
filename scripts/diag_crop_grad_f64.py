"""Which side is off?  Crop scene, teacher-forced joint steps: the HIP path's and the float32 oracle's vertex gradients against the
oracle's differentiable part run in FLOAT64 on the SAME fragment selection (the C rasteriser's float32 face ids / K-buffers of the
float32 run are injected, so that the three sides differentiate the same fragments; the nearest-neighbour pairs and the
intersection count are not differentiated through).  python scripts/diag_crop_grad_f64.py [n_steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from followmyhold_amd import engine as E, synthetic
from oracle import clib, step_ref as S, ref_ops as R
clib.set_threads(32); torch.set_num_threads(32)
H = W = 512
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=H, W=W, seed=0, crop="hoi")
sct = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
sc64 = {k: (v.double() if isinstance(v, torch.Tensor) and v.dtype == torch.float32 else v) for k, v in sct.items()}
st = S.JointStepper(sct, S.make_params(), denoise_i=19, grid_res=64)
gb = E.GuidanceBatch([sc])
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
edges = R.unique_edges(sct["obj_faces"])
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
real_select = R.rasterize_select


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


for k in range(n_steps):
    p_k = {kk: v.detach().clone() for kk, v in st.p.items()}
    gb.set_params(0, **{kk: v.numpy() for kk, v in p_k.items()})
    total, terms, aux, grads = st.step(update=True)
    gb.step(cfg); torch.cuda.synchronize()
    gh = gb.grad_obj_verts(0).cpu().numpy().astype(np.float64)
    gr = grads["obj_verts"].numpy().astype(np.float64)
    # float64 differentiable part on the float32 run's selections (hand render first, then hand + object: the order phase_c_loss asks)
    sels = [aux["hand"]["render"]["sel"], aux["render"]["sel"]]
    R.rasterize_select = lambda *a, **kw: sels.pop(0)
    try:
        p64 = S.leafify({kk: v.double() for kk, v in p_k.items()}, S.PARAM_KEYS)
        ov64 = sc64["obj_verts"].detach().clone().requires_grad_(True)
        t64, terms64, aux64 = S.phase_c_loss(sc64, p64, ov64, edges, 19, 20, grid_res=64)
        t64.backward(retain_graph=True)
    finally:
        R.rasterize_select = real_select
    g64 = ov64.grad.numpy()
    same_knn = bool((aux64["knn_idx"] == aux["knn_idx"]).all())
    dv = np.linalg.norm(gh - gr, axis=1)
    top = np.argsort(-dv)[:6]
    print(f"step {k}: total f32 {float(total):.7f} f64 {float(t64):.7f} hip {gb.loss_dict(0)['total']:.7f} | |g| {np.linalg.norm(g64):.4e} | "
          f"hip-vs-ref {rel(gh, gr):.2e}  hip-vs-f64 {rel(gh, g64):.2e}  ref32-vs-f64 {rel(gr, g64):.2e}  same knn pairs {same_knn}", flush=True)
    for v in top:
        dh, dr = np.linalg.norm(gh[v] - g64[v]), np.linalg.norm(gr[v] - g64[v])
        print(f"    vertex {v:6d}: |hip-ref| {dv[v]:.3e} (|g_v| {np.linalg.norm(g64[v]):.3e})  |hip-f64| {dh:.3e}  |ref32-f64| {dr:.3e}  -> "
              f"{'ORACLE fp32 is off' if dr > 3 * dh else ('HIP is off' if dh > 3 * dr else 'both')}", flush=True)
    # without the top outliers
    keep = np.ones(len(dv), bool); keep[top[:4]] = False
    print(f"    rest (all but the 4 worst): hip-vs-f64 {np.linalg.norm((gh - g64)[keep]) / np.linalg.norm(g64):.2e}  ref32-vs-f64 {np.linalg.norm((gr - g64)[keep]) / np.linalg.norm(g64):.2e}", flush=True)

    if rel(gh, gr) > 1e-4:
        # which loss term carries the float32 oracle's error?  Per-term vertex gradients of the oracle in float32 against float64
        # (torch autograd both) on the vertices where the float32 sides disagree
        p32 = S.leafify(p_k, S.PARAM_KEYS)
        ov32 = sct["obj_verts"].detach().clone().requires_grad_(True)
        sels = [aux["hand"]["render"]["sel"], aux["render"]["sel"]]
        R.rasterize_select = lambda *a, **kw: sels.pop(0)
        try:
            t32, terms32, _ = S.phase_c_loss(sct, p32, ov32, edges, 19, 20, grid_res=64)
        finally:
            R.rasterize_select = real_select
        wts = dict(normal_hoi=10.0, disp_hoi=10.0, sil_hoi=10.0, contact=10.0, edge=1.0, verts_obj=1e-3)
        for name, wgt in wts.items():
            a = torch.autograd.grad(wgt * terms32[name], ov32, retain_graph=True, allow_unused=True)[0]
            b = torch.autograd.grad(wgt * terms64[name], ov64, retain_graph=True, allow_unused=True)[0]
            if a is None or b is None:
                continue
            a, b = a.numpy().astype(np.float64), b.numpy()
            d = np.linalg.norm(a - b, axis=1)
            print(f"    term {name:10s}: |g64| {np.linalg.norm(b):.3e}  ref32-vs-f64 {rel(a, b):.2e}; on the outliers " +
                  ", ".join(f"{v}: {d[v]:.2e}" for v in top[:4]), flush=True)
