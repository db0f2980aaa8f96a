"""Developer diagnostic: stage-by-stage HIP vs oracle comparison on one small scene (run on the GPU box)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import make_scene
from oracle import ref_ops as R, step_ref as S
from followmyhold_amd import engine as E

H = int(os.environ.get("FOHO_H", 64)); kind = os.environ.get("FOHO_OBJ", "ico2"); res = int(os.environ.get("FOHO_RES", 16))
sc = make_scene(kind, H, H, seed=0)
p = S.make_params(scale_hand=torch.tensor([1.02]), trans_hand=torch.tensor([0.004, -0.003, 0.002]),
                  rot_hand=torch.tensor([0.999, 0.02, -0.01, 0.03]), scale_obj=torch.tensor([0.97]),
                  trans_obj=torch.tensor([-0.002, 0.003, 0.001]), rot_obj=torch.tensor([0.998, -0.03, 0.02, 0.01]))
t0 = time.time()
st = S.JointStepper(sc, p, denoise_i=19, grid_res=res)
total, terms, aux, grads = st.step(update=True)
print("oracle step s", time.time() - t0)
npsc = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
gb = E.GuidanceBatch([npsc], grid_res=res)
gb.set_params(0, **{k: v.numpy() for k, v in p.items()})
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
gb.step(cfg); torch.cuda.synchronize()
print("flags", gb.flags.cpu().tolist())
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
world_ref = torch.cat([aux["hand"]["verts"], aux["obj_verts_t"]], 0).detach().numpy()
world = gb.region("world", torch.float32, (-1, 3)).cpu().numpy()
print("world equal", np.array_equal(world, world_ref), np.abs(world - world_ref).max())
cam = R.Camera(sc["fov"], H, H)
ndc_ref = R.world_to_ndc(torch.from_numpy(world_ref), cam).numpy()
ndc = gb.region("ndc", torch.float32, (-1, 3)).cpu().numpy()
print("ndc equal", np.array_equal(ndc, ndc_ref), np.abs(ndc - ndc_ref).max())
vn_ref = torch.cat([R.vertex_normals(aux["hand"]["verts"].detach(), sc["hand_faces"]), R.vertex_normals(aux["obj_verts_t"].detach(), sc["obj_faces"])], 0).numpy()
vn = gb.region("vn", torch.float32, (-1, 3)).cpu().numpy()
print("vn equal", np.array_equal(vn, vn_ref), np.abs(vn - vn_ref).max())
Vh = sc["hand_verts"].shape[0]
idx = gb.region("knn_idx", torch.int32)[:Vh].cpu().numpy()
print("knn idx equal", np.array_equal(idx, aux["knn_idx"].numpy()))
P = H * H
p2f = gb.region("p2f", torch.int32, (2, P)).cpu().numpy(); zb = gb.region("zbuf", torch.float32, (2, P)).cpu().numpy()
sd = gb.region("sdist", torch.float32, (2, P)).cpu().numpy(); prod = gb.region("prod", torch.float32, (2, P)).cpu().numpy()
for r, ren in enumerate([aux["hand"]["render"], aux["render"]]):
    sel = ren["sel"]; ref = sel["pix_to_face"].reshape(-1)
    print("render", r, "hits", (ref >= 0).sum(), "p2f mism", (p2f[r] != ref).sum(), "z equal", np.array_equal(zb[r], sel["zbuf"].reshape(-1)),
          "sd equal", np.array_equal(sd[r], sel["dists"].reshape(-1)), "max|dz|", np.abs(zb[r] - sel["zbuf"].reshape(-1)).max())
sil_ref = aux["render"]["sil"].detach().numpy().reshape(-1)
print("sil max diff", np.abs((1 - prod[1]) - sil_ref).max(), "nonbinary px", ((sil_ref > 0) & (sil_ref < 1)).sum())
print("frac counts", gb.region("frac_count", torch.int32).cpu().tolist())
l = gb.loss_dict(0)
t = {k: float(v) for k, v in terms.items()}
print("n_int", l["n_intersect"], aux["n_int"], "w_int", l["w_int"], aux["w_int"])
for a, b in [("contact", "contact"), ("kps", "kps"), ("trans_hand", "trans_hand"), ("trans_obj", "trans_obj"), ("verts_obj", "verts_obj"), ("edge", "edge"),
             ("normal0", "normal_hand"), ("disp0", "disp_hand"), ("normal1", "normal_hoi"), ("disp1", "disp_hoi"), ("sil1", "sil_hoi"), ("intersection", "intersection")]:
    print(f"  {a:12s} hip {l[a]:.8g} ref {t[b]:.8g} rel {abs(l[a]-t[b])/max(abs(t[b]),1e-12):.2e}")
print("total", l["total"], float(total))
g = gb.grad_params[0].cpu().numpy()
gref = np.concatenate([grads[k].numpy().reshape(-1) for k in E.PARAM_NAMES])
for k, sl in E.PARAM_SLICES.items():
    print(f"  grad {k:10s} rel {rel(g[sl], gref[sl]):.2e} hip {g[sl]} ref {gref[sl]}")
gv = gb.grad_obj_verts(0).cpu().numpy()
print("grad obj_verts rel", rel(gv, grads["obj_verts"].numpy()), np.abs(gv).max(), np.abs(grads["obj_verts"].numpy()).max())
pa = gb.get_params(0)
for k in E.PARAM_NAMES:
    print("  after", k, pa[k].numpy(), st.p[k].detach().numpy())
bc = gb.region("bin_count", torch.int32).cpu().numpy().reshape(2, -1)
print("bin max", bc.max(1), "nonempty", (bc > 0).sum(1), "sum", bc.sum(1))
for _ in range(3): prof = gb.step_profiled(cfg)
print({k: round(v * 1e3, 1) for k, v in prof.items()}, "us; total", round(sum(prof.values()) * 1e3, 1))
