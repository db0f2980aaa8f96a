"""The pytorch3d-shaped facade, driven exactly the way the reference's loop drives pytorch3d
(run.py:84-116 builds the renderers; pipelines.py:272-289, 1323-1349, 1529-1553 use them), vs the oracle."""
import numpy as np
import pytest
import torch

from helpers import make_scene
from oracle import ref_ops as R
from oracle import step_ref as S

gpu = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def render_normal_and_disparity(renderer, mesh):
    """The call sequence of pipelines.py:272-289 (two rasterisations, global min-max normalisation)."""
    norms = renderer(mesh)
    depth = renderer.rasterizer(mesh).zbuf.squeeze(-1)
    mask = norms[..., 3] > 0.0
    n = norms[..., :3]
    nn = (n - n.min()) / (n.max() - n.min() + 1e-6)
    nn = torch.where(mask[..., None], nn, torch.zeros_like(nn))
    depth = torch.where(depth < 0, torch.full_like(depth, 10.0), depth)
    disp = 1 / (depth + 1e-6)
    disp = (disp - disp.min()) / (disp.max() - disp.min() + 1e-6)
    return nn, disp


def build_renderers(p3d, fov, H, W):
    dev = "cuda"
    Rm = torch.tensor([[-1.0, 0, 0], [0, 1.0, 0], [0, 0, -1.0]], device=dev).unsqueeze(0)
    cams = p3d.FoVPerspectiveCameras(device=dev, R=Rm, T=torch.zeros(1, 3, device=dev), znear=0.01, zfar=100.0, fov=fov)
    blend = p3d.BlendParams(sigma=1e-8, gamma=1e-8)
    blur = float(np.float32(np.log(1.0 / 1e-4 - 1.0) * np.float32(1e-8)))
    ren = p3d.MeshRenderer(p3d.MeshRasterizer(cams, p3d.RasterizationSettings((H, W), blur, 1, bin_size=-1)),
                           p3d.PhongNormalShader(cameras=cams, blend_params=blend))
    sil = p3d.MeshRenderer(p3d.MeshRasterizer(cams, p3d.RasterizationSettings((H, W), blur, 100, bin_size=None)),
                           p3d.SoftSilhouetteShader(blend_params=blend))
    return cams, ren, sil


@gpu
def test_facade_render_matches_oracle_and_backpropagates():
    from followmyhold_amd import facade as p3d
    H = W = 64
    sc = make_scene("ico2", H, W, seed=2)
    cams, ren, sil = build_renderers(p3d, sc["fov"], H, W)
    p = S.make_params(scale_hand=torch.tensor([1.03]), rot_hand=torch.tensor([0.99, 0.03, -0.02, 0.01]))
    # ---- oracle: hand + object joined scene (pipelines.py:1544-1547)
    hv = S.hand_transform(sc, p)
    ov = S.obj_transform(sc, p, sc["obj_verts"])
    vref = torch.cat([hv, ov], 0).detach().requires_grad_(True)
    faces = torch.cat([sc["hand_faces"], sc["obj_faces"] + hv.shape[0]], 0)
    cam = R.Camera(sc["fov"], H, W)
    r = S.render_all(vref, faces, cam, R.blur_radius_from_sigma(), True)
    tgt_n, tgt_d = sc["moge_normal"], sc["moge_disp"]
    hoi = sc["hand_mask"] | sc["obj_mask"]
    loss_ref = 10 * R.normal_alignment_loss(r["normal"], tgt_n, hoi) + 10 * (r["disp"] - tgt_d).abs().mean()
    loss_ref.backward()
    # ---- facade, called like the reference calls pytorch3d
    vd = torch.cat([hv, ov], 0).detach().cuda().requires_grad_(True)
    hand = p3d.Meshes([vd[:778]], [sc["hand_faces"].cuda()])
    obj = p3d.Meshes([vd[778:]], [sc["obj_faces"].cuda()])
    hoi_mesh = p3d.join_meshes_as_scene([hand, obj])
    nn, dd = render_normal_and_disparity(ren, hoi_mesh)
    alpha = sil(hoi_mesh)[..., 3]
    assert np.abs(nn[0].detach().cpu().numpy() - r["normal"].detach().numpy()).max() < 2e-6
    assert np.abs(dd[0].detach().cpu().numpy() - r["disp"].detach().numpy()).max() < 2e-6
    assert np.abs(alpha[0].detach().cpu().numpy() - r["sil"].detach().numpy()).max() < 1e-6
    frag = ren.rasterizer(hoi_mesh)
    assert frag.pix_to_face.shape == (1, H, W, 1) and frag.pix_to_face.dtype == torch.int64
    assert np.array_equal(frag.pix_to_face[0, ..., 0].cpu().numpy(), r["sel"]["pix_to_face"])
    valid = torch.nn.functional.normalize(nn, dim=-1), torch.nn.functional.normalize(tgt_n.cuda()[None], dim=-1)
    l_n = (1 - (valid[0] * valid[1]).sum(-1))[hoi.cuda()[None]].mean()
    loss = 10 * l_n + 10 * (dd - tgt_d.cuda()[None]).abs().mean()
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * abs(float(loss_ref))
    loss.backward()
    assert rel_err(vd.grad.cpu().numpy(), vref.grad.numpy()) < 2e-4


@gpu
def test_facade_knn_edge_loss_screen_points_and_sdf():
    from followmyhold_amd import facade as p3d
    sc = make_scene("ico2", 32, 32, seed=3)
    cams, _, _ = build_renderers(p3d, sc["fov"], 32, 32)
    p = S.make_params()
    hv, ov = S.hand_transform(sc, p).detach(), S.obj_transform(sc, p, sc["obj_verts"]).detach()
    a, b = hv.clone().requires_grad_(True), ov.clone().requires_grad_(True)
    d_ref, _ = R.knn1(a, b)
    torch.clamp(d_ref - 0.01, min=0).mean().backward()
    ad, bd = hv.cuda().requires_grad_(True), ov.cuda().requires_grad_(True)
    d, idx, _ = p3d.knn_points(ad[None], bd[None], K=1)
    assert d.shape == (1, 778, 1) and np.allclose(d[0, :, 0].detach().cpu().numpy(), d_ref.detach().numpy(), rtol=1e-6)
    torch.clamp(d.squeeze() - 0.01, min=0).mean().backward()
    assert rel_err(ad.grad.cpu().numpy(), a.grad.numpy()) < 1e-5 and rel_err(bd.grad.cpu().numpy(), b.grad.numpy()) < 1e-5
    m = p3d.Meshes([ov.cuda()], [sc["obj_faces"].cuda()])
    assert float(p3d.mesh_edge_loss(m)) == pytest.approx(float(R.mesh_edge_loss(ov, R.unique_edges(sc["obj_faces"]))), rel=1e-5)
    kp = R.mano_vert_to_3dkps(hv, sc["J_regressor"])
    s_ref = R.ndc_to_screen(R.world_to_ndc(kp, R.Camera(sc["fov"], 32, 32)), 32, 32)
    s = cams.transform_points_screen(kp.cuda()[None], image_size=(32, 32))[0, :, :2]
    assert np.abs(s.cpu().numpy() - s_ref.numpy()).max() < 1e-3
    sdf1, sdf2 = p3d.get_sdf_of_meshes(p3d.Meshes([hv.cuda()], [sc["hand_faces"].cuda()]), m, "cuda", 12)
    grid = R.joint_grid(hv, ov, 12)
    assert np.array_equal(sdf1.cpu().numpy(), R.mesh_sdf(hv, sc["hand_faces"], grid))
    assert np.array_equal(sdf2.cpu().numpy(), R.mesh_sdf(ov, sc["obj_faces"], grid))
    n_int = int(((sdf1 < 0) & (sdf2 < 0)).sum())
    assert n_int == R.intersection_count(hv, sc["hand_faces"], ov, sc["obj_faces"], 12)


@gpu
def test_alignment_stage_end_to_end(tmp_path):
    """foho.alignment.h2m.run on files: recovers a known similarity between a mesh and its transformed copy."""
    from foho.alignment import h2m
    from followmyhold_amd import meshio, synthetic
    v, f = synthetic.make_object("20k")
    v = v.astype(np.float64) * 4
    M = np.eye(4)
    M[:3, :3] = 1.25 * synthetic.axis_angle_matrix([0.04, -0.03, 0.05])
    M[:3, 3] = [0.3, -0.1, 0.2]
    (tmp_path / "hy").mkdir()
    (tmp_path / "moge" / "7_cropped_hoi").mkdir(parents=True)
    meshio.save_ply(str(tmp_path / "hy" / "7_hoi_mesh.ply"), v, f)
    meshio.save_ply(str(tmp_path / "moge" / "7_cropped_hoi" / "pointcloud.ply"), v @ M[:3, :3].T + M[:3, 3])
    h2m.run(str(tmp_path / "hy"), str(tmp_path / "moge"), str(tmp_path / "rt"))
    T = np.load(str(tmp_path / "rt" / "7_hoi_mesh.npy"))
    assert T.shape == (4, 4) and T.dtype == np.float64
    moved = v @ T[:3, :3].T + T[:3, 3]
    err = np.linalg.norm(moved - (v @ M[:3, :3].T + M[:3, 3]), axis=1).mean()
    assert err < 0.05 * 0.4, err      # point-to-point ICP on sampled points: within 5 % of the 0.4 m object diameter


@gpu
def test_align_meshes_on_surface(tmp_path):
    """align_meshes_impl(..., on_surface=True) (ICP:106-107, CLI flag -os): mesh-to-mesh alignment against the target
    triangles; at least as tight as the sampled-point variant of the same call."""
    from foho.alignment import mesh_align as MA
    from followmyhold_amd import meshio, synthetic
    v, f = synthetic.make_object("20k")
    v = v.astype(np.float64) * 4
    M = np.eye(4)
    M[:3, :3] = 1.2 * synthetic.axis_angle_matrix([0.05, 0.02, -0.04])
    M[:3, 3] = [0.2, 0.1, -0.15]
    a, b = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    meshio.save_ply(a, v, f)
    meshio.save_ply(b, v @ M[:3, :3].T + M[:3, 3], f)
    want = v @ M[:3, :3].T + M[:3, 3]
    errs = {}
    for os_ in (False, True):
        T = MA.align_meshes_impl(a, b, None, None, False, 0.2, False, False, os_, 30, 1000, 5000, 40, 2000, 10000, 0.7, 3.0, False)
        errs[os_] = np.linalg.norm(v @ T[:3, :3].T + T[:3, 3] - want, axis=1).mean()
    assert errs[True] < 0.05 * 0.4 and errs[True] < 1.05 * errs[False], errs    # a near-sphere slides tangentially: 5 % of 0.4 m
    with pytest.raises(ValueError):
        MA.icp(MA.Mesh(v, f), MA.Mesh(v), 2, on_surface=True)


@gpu
def test_facade_silhouette_is_differentiable():
    """`sil_renderer(mesh)[..., 3]` carries gradients (RUN:106-116; the BCE silhouette terms PL:1341, 1423, 1569): the facade's
    alpha = 1 - prod_k(1 - sigmoid(-d_k / sigma)) back-propagates through every fragment of the pixels with fractional
    coverage (foho_raster_sil_bwd), sigma taken from the shader's blend_params.  A wider sigma than the path's 1e-8 puts
    hundreds of fragments in the blur band: d BCE / d verts against the oracle's autograd."""
    from followmyhold_amd import facade as p3d
    H = W = 96
    sigma = 2e-5
    sc = make_scene("ico2", H, W, seed=4)
    p = S.make_params(scale_hand=torch.tensor([1.02]), rot_obj=torch.tensor([0.99, 0.02, 0.03, -0.01]))
    hv, ov = S.hand_transform(sc, p), S.obj_transform(sc, p, sc["obj_verts"])
    faces = torch.cat([sc["hand_faces"], sc["obj_faces"] + hv.shape[0]], 0)
    cam = R.Camera(sc["fov"], H, W)
    blur = float(np.float32(np.log(1.0 / 1e-4 - 1.0) * np.float32(sigma)))
    target = (sc["hand_mask"] | sc["obj_mask"]).float()
    # ---- oracle
    vref = torch.cat([hv, ov], 0).detach().requires_grad_(True)
    sel = R.rasterize_select(R.world_to_ndc(vref, cam), faces, H, W, blur)
    sil_ref = R.render_silhouette(vref, faces, cam, sel, sigma=sigma)
    frac = ((sil_ref.detach() > 0) & (sil_ref.detach() < 1)).sum()
    assert int(frac) >= 50, int(frac)
    loss_ref = torch.nn.functional.binary_cross_entropy(sil_ref, target)
    loss_ref.backward()
    # ---- facade
    dev = "cuda"
    Rm = torch.tensor([[-1.0, 0, 0], [0, 1.0, 0], [0, 0, -1.0]], device=dev).unsqueeze(0)
    cams = p3d.FoVPerspectiveCameras(device=dev, R=Rm, T=torch.zeros(1, 3, device=dev), znear=0.01, zfar=100.0, fov=sc["fov"])
    sil = p3d.MeshRenderer(p3d.MeshRasterizer(cams, p3d.RasterizationSettings((H, W), blur, 100, bin_size=None)),
                           p3d.SoftSilhouetteShader(blend_params=p3d.BlendParams(sigma=sigma, gamma=1e-8)))
    vd = vref.detach().cuda().requires_grad_(True)
    mesh = p3d.join_meshes_as_scene([p3d.Meshes([vd[:778]], [sc["hand_faces"].cuda()]), p3d.Meshes([vd[778:]], [sc["obj_faces"].cuda()])])
    alpha = sil(mesh)[..., 3]
    assert alpha.requires_grad
    assert np.abs(alpha[0].detach().cpu().numpy() - sil_ref.detach().numpy()).max() < 2e-5
    loss = torch.nn.functional.binary_cross_entropy(alpha[0], target.cuda())
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * abs(float(loss_ref))
    loss.backward()
    assert float(vref.grad.norm()) > 0
    assert rel_err(vd.grad.cpu().numpy(), vref.grad.numpy()) < 1e-3
    # the rasteriser alone takes the sigma its blur radius was derived from
    fr = sil.rasterizer(mesh)
    assert np.abs((1 - fr.sil_prod[0]).detach().cpu().numpy() - sil_ref.detach().numpy()).max() < 2e-5
