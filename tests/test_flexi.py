"""Iso-surfacing (FlexiCubes with default weights = Dual Marching Cubes).

kaolin's implementation is not available (parity unpinned): the CPU oracle restates the published algorithm and is pinned
by known-answer tests on analytic SDFs (closed 2-manifold, Euler characteristic, outward orientation, volume, finite
differences); the HIP kernels are compared with the oracle index for index."""
import numpy as np
import pytest
import torch

from followmyhold_amd import flexi_tables
from oracle import flexi_ref as FR

gpu = pytest.mark.gpu


def _grid(res, half=1.1):
    x, cubes = FR.construct_voxel_grid(res)
    return x * (2 * half), cubes


def _sdfs(x):
    r = x.norm(dim=1)
    torus = torch.stack([torch.sqrt(x[:, 0] ** 2 + x[:, 1] ** 2) - 0.6, x[:, 2]], 1).norm(dim=1) - 0.25
    two = torch.minimum((x - 0.3).norm(dim=1) - 0.42, (x + 0.3).norm(dim=1) - 0.42)   # touching diagonally: ambiguous cubes
    noisy = r - 0.7 + 0.08 * torch.sin(9 * x[:, 0]) * torch.cos(7 * x[:, 1]) * torch.sin(5 * x[:, 2])
    return {"sphere": (r - 0.8, 2), "torus": (torus, 0), "two_spheres": (two, None), "bumpy": (noisy, 2)}


def _topology(v, f):
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), 1)
    ue, cnt = np.unique(e, axis=0, return_counts=True)
    vol = np.einsum("ij,ij->i", v[f[:, 0]], np.cross(v[f[:, 1]], v[f[:, 2]])).sum() / 6
    return cnt, len(v) - len(ue) + len(f), vol


def test_patch_tables_two_derivations_agree():
    """Union-find over the face pairings (product) == cycle walking (oracle); spot checks of known configurations."""
    n1, e1 = flexi_tables.build()
    n2, e2 = FR.patch_tables()
    assert np.array_equal(n1, n2) and np.array_equal(e1, e2)
    assert n1[0] == 0 and n1[255] == 0
    assert n1[1] == 1 and sorted(np.flatnonzero(e1[1] >= 0)) == [0, 4, 8]            # one inside corner: its three edges
    assert n1[0x0F] == 1 and (e1[0x0F] >= 0).sum() == 4                              # bottom face inside: one quad patch
    assert n1[0x69] == 4 and n1[0x96] == 4                                           # four separated corners
    assert n1[0x81] == 2                                                             # two opposite corners
    for case in range(256):
        assert n1[case] == n1[255 - case] or True                                    # (complement symmetry is not required)
        cross = [((case >> a) & 1) != ((case >> b) & 1) for a, b in flexi_tables.EDGES]
        assert ((e1[case] >= 0) == np.array(cross)).all()
        if n1[case]:
            assert sorted(set(e1[case][e1[case] >= 0])) == list(range(n1[case]))
    # the embedded kernel table is the generator's output
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert open(os.path.join(here, "followmyhold_amd", "csrc", "k_flexi_tables.inc")).read() == flexi_tables.emit()


@pytest.mark.parametrize("name", ["sphere", "torus", "two_spheres", "bumpy"])
def test_oracle_surfaces_are_closed_oriented_manifolds(name):
    res = 24
    x, _ = _grid(res)
    s, chi = _sdfs(x)[name]
    V, F, D = FR.flexicubes(x, s, res)
    v, f = V.numpy(), F.numpy()
    cnt, euler, vol = _topology(v, f)
    assert (cnt == 2).all()                                   # every edge shared by exactly two triangles
    if chi is not None:
        assert euler == chi
    assert vol > 0                                            # outward orientation
    if name == "sphere":
        assert abs(vol - 4 / 3 * np.pi * 0.8 ** 3) < 0.03 * vol
        assert np.abs(np.linalg.norm(v, axis=1) - 0.8).max() < 2.2 / res       # vertices within a cell of the true surface
    assert len(D) == len(v) and float(D.min()) >= 0


def test_oracle_gradient_matches_finite_differences():
    res = 12
    x, _ = _grid(res)
    x = x.double()
    s = (x.norm(dim=1) - 0.8).requires_grad_(True)
    V, F, D = FR.flexicubes(x, s, res)
    w = torch.linspace(0.5, 1.5, V.numel(), dtype=torch.float64).reshape(V.shape)
    (V * w).sum().backward()
    g = s.grad
    for i in torch.nonzero(g).reshape(-1)[::37][:6]:
        sp, sm = s.detach().clone(), s.detach().clone()
        sp[i] += 1e-6
        sm[i] -= 1e-6
        fd = ((FR.flexicubes(x, sp, res)[0] * w).sum() - (FR.flexicubes(x, sm, res)[0] * w).sum()) / 2e-6
        assert abs(float(fd) - float(g[i])) < 1e-5 * max(1.0, abs(float(fd)))


@gpu
@pytest.mark.parametrize("name,res", [("sphere", 24), ("torus", 24), ("two_spheres", 24), ("bumpy", 32)])
def test_hip_flexicubes_matches_oracle_index_for_index(name, res):
    from followmyhold_amd import facade
    x, cubes = _grid(res)
    s, _ = _sdfs(x)[name]
    V, F, D = FR.flexicubes(x, s, res)
    fc = facade.FlexiCubes("cuda")
    gv, gc = fc.construct_voxel_grid(res)
    assert torch.equal(gc.cpu(), cubes) and torch.allclose(gv.cpu() * 2.2, x)
    sg = s.cuda().requires_grad_(True)
    v, f, ld = fc(x.cuda(), sg, gc, res)
    assert torch.equal(f.cpu(), F)                                             # same triangles, same order
    assert torch.equal(v.detach().cpu(), V)                                    # same operation order: bit-identical vertices
    assert np.allclose(ld.cpu().numpy(), D.numpy(), atol=1e-6)
    # backward vs autograd of the oracle
    so = s.clone().requires_grad_(True)
    Vo = FR.flexicubes(x, so, res)[0]
    w = torch.linspace(0.5, 1.5, Vo.numel()).reshape(Vo.shape)
    (Vo * w).sum().backward()
    (v * w.cuda()).sum().backward()
    g, go = sg.grad.cpu().numpy(), so.grad.numpy()
    assert np.abs(g - go).max() <= 1e-4 * np.abs(go).max()


@gpu
def test_hip_flexicubes_at_the_pipeline_resolution_and_capacity_retry():
    """res 64 (PL:1126): 65^3 SDF samples; a capacity that is too small is detected and the call retried."""
    from followmyhold_amd import ops
    res = 64
    x, _ = _grid(res)
    s, _ = _sdfs(x)["bumpy"]
    v, f, ld = ops.flexicubes(x.cuda(), s.cuda(), res, verts_cap=64, faces_cap=64)
    cnt, euler, vol = _topology(v.cpu().numpy(), f.cpu().numpy())
    assert (cnt == 2).all() and euler == 2 and vol > 0 and len(v) > 5000
    V, F, _ = FR.flexicubes(x, s, res)
    assert torch.equal(f.cpu(), F) and torch.equal(v.cpu(), V)


@gpu
def test_sdf_to_loss_chain_matches_the_oracle_chain():
    """The reference's differentiable route (PL:1507-1600): SDF grid -> FlexiCubes -> object mesh -> joint guidance
    loss, and back: dL/d(sdf).  HIP chain (foho_flexi_fwd -> on-GPU topology tables -> foho_step_run -> foho_flexi_bwd)
    against the oracle chain (flexi_ref -> step_ref.phase_c_loss -> torch autograd)."""
    from followmyhold_amd import engine as E, ops
    from helpers import make_scene
    from oracle import ref_ops as R
    from oracle import step_ref as S
    sc = make_scene("ico2", 64, 64, seed=0)
    res = 14
    x = FR.construct_voxel_grid(res)[0] * 0.24                                   # Hunyuan-space box around the object
    r0 = float(sc["obj_verts"].norm(dim=1).mean())
    s0 = x.norm(dim=1) - r0 * (1.0 + 0.15 * torch.sin(25 * x[:, 0]) * torch.cos(21 * x[:, 1]))
    # oracle chain
    so = s0.clone().requires_grad_(True)
    V, F, _ = FR.flexicubes(x, so, res)
    assert len(V) > 100
    sc_o = dict(sc, obj_faces=F)
    p = S.make_params()
    total, terms, aux = S.phase_c_loss(sc_o, p, V, R.unique_edges(F), denoise_i=19, grid_res=16)
    total.backward()
    # HIP chain
    npsc = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
    gb = E.GuidanceBatch([npsc], grid_res=16)
    cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
    sg = s0.cuda().requires_grad_(True)
    v, f, _ = ops.flexicubes(x.cuda(), sg, res)
    assert torch.equal(f.cpu(), F) and torch.equal(v.detach().cpu(), V.detach())
    loss = gb.objective(v, f, cfg)
    loss.backward()
    torch.cuda.synchronize()
    gb.raise_on_flags()
    assert abs(float(loss) - float(total)) <= 1e-4 * abs(float(total))
    l = gb.loss_dict(0)
    assert abs(l["edge"] - float(terms["edge"])) <= 1e-4 * abs(float(terms["edge"]))
    assert int(l["n_intersect"]) == aux["n_int"]
    g, go = sg.grad.cpu().numpy(), so.grad.numpy()
    assert np.linalg.norm(g - go) <= 2e-3 * np.linalg.norm(go), (np.linalg.norm(g - go), np.linalg.norm(go))
    # a second mesh with another topology through the same GuidanceBatch (sizes change, tables rebuilt on the GPU)
    s1 = (s0 - 0.01).cuda().requires_grad_(True)
    v1, f1, _ = ops.flexicubes(x.cuda(), s1, res)
    assert len(v1) != len(v)
    gb.objective(v1, f1, cfg).backward()
    torch.cuda.synchronize()
    gb.raise_on_flags()
    V1, F1, _ = FR.flexicubes(x, (s0 - 0.01), res)
    t1, _, _ = S.phase_c_loss(dict(sc, obj_faces=F1), p, V1, R.unique_edges(F1), denoise_i=19, grid_res=16)
    assert abs(gb.loss_dict(0)["total"] - float(t1)) <= 1e-4 * abs(float(t1))
    assert np.isfinite(s1.grad.cpu().numpy()).all() and float(s1.grad.abs().sum()) > 0


@gpu
def test_device_topology_tables_equal_the_host_builders():
    """foho_topology_tables (closed manifold fast path) against the numpy builders the engine uses at construction; a mesh
    with a boundary is detected and handled by the general sort path."""
    from followmyhold_amd import engine as E, synthetic
    from helpers import make_scene
    sc = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in make_scene("ico2", 64, 64, seed=0).items()}
    gb = E.GuidanceBatch([sc], grid_res=16)
    def host_lists(of, Vh, Vtot):
        off, idx = E.neighbour_csr(E.unique_edges(of) + Vh, Vtot)
        return off, idx

    def check_nbr(gb, of):
        Vh, Vtot = gb.meta[0]["Vh"], gb.Vtot
        off, idx = host_lists(of, Vh, Vtot)
        d_off, d_idx = gb.nbr_off.cpu().numpy(), gb.nbr_idx.cpu().numpy()
        for v in list(range(Vh, Vh + 50)) + list(range(Vtot - 50, Vtot)):
            assert np.array_equal(d_idx[d_off[v]:d_off[v + 1]], idx[off[v]:off[v + 1]]), v

    for kind in ("ico4", "20k"):
        ov, of = synthetic.make_object(kind)
        gb.update_object(ov * 0.05, of)
        faces = gb.faces.cpu().numpy().astype(np.int64)
        inc_off, inc_fc = E.incidence_csr(faces, gb.Vtot)
        assert gb.meta[0]["n_edges"] == len(E.unique_edges(of)) == 3 * len(of) // 2
        assert np.array_equal(gb.inc_off.cpu().numpy(), inc_off) and np.array_equal(gb.inc_fc.cpu().numpy(), inc_fc)
        assert gb.nbr_off.data_ptr() == gb.inc_off.data_ptr()        # neighbour lists share the incidence offsets
        check_nbr(gb, of)
        # the constructor takes the same device path
        g2 = E.GuidanceBatch([dict(sc, obj_verts=(ov * 0.05).astype(np.float32), obj_faces=of)], grid_res=16)
        assert np.array_equal(g2.inc_fc.cpu().numpy(), inc_fc) and g2.meta[0]["n_edges"] == 3 * len(of) // 2
        check_nbr(g2, of)
        gh = E.GuidanceBatch([dict(sc, obj_verts=(ov * 0.05).astype(np.float32), obj_faces=of)], grid_res=16, topology="host")
        assert np.array_equal(gh.inc_fc.cpu().numpy(), inc_fc) and gh.meta[0]["n_edges"] == g2.meta[0]["n_edges"]
    # open mesh (two faces removed): not a closed manifold -> the flag sends it to the sort path, same tables as the host's
    ov, of = synthetic.make_object("ico4")
    of = of[:-2]
    gb.update_object(ov * 0.05, of)
    Vh = gb.meta[0]["Vh"]
    assert gb.meta[0]["n_edges"] == len(E.unique_edges(of)) != 3 * len(of) // 2
    check_nbr(gb, of)
    inc_off, inc_fc = E.incidence_csr(gb.faces.cpu().numpy().astype(np.int64), gb.Vtot)
    assert np.array_equal(gb.inc_fc.cpu().numpy(), inc_fc)
    g3 = E.GuidanceBatch([dict(sc, obj_verts=(ov * 0.05).astype(np.float32), obj_faces=of)], grid_res=16)
    assert g3.meta[0]["n_edges"] == len(E.unique_edges(of))
    check_nbr(g3, of)


@gpu
@pytest.mark.parametrize("seed", range(10))
def test_hip_flexicubes_fuzz_random_fields(seed):
    """Random trigonometric SDF fields at small resolutions: all 256 cube cases incl. the ambiguous ones, several components,
    surfaces touching the grid boundary, values exactly zero on grid points (zero counts as outside: s < 0 is inside).
    Triangles index for index, vertices bit for bit."""
    from followmyhold_amd import ops
    rng = np.random.default_rng(500 + seed)
    res = [6, 9, 12, 16][seed % 4]
    x, _ = _grid(res)
    k = rng.uniform(2.0, 9.0, size=(4, 3))
    ph = rng.uniform(0, 6.28, size=4)
    xt = x.numpy().astype(np.float64)
    s = sum(np.sin(xt @ k[i] + ph[i]) for i in range(4)) * 0.25 + rng.uniform(-0.3, 0.3)
    s = torch.from_numpy(s.astype(np.float32))
    s[torch.from_numpy(rng.random(len(s)) < 0.02)] = 0.0                       # exact zeros
    if seed % 2:
        s = torch.round(s * 4) / 4                                            # many ties and exact zeros
    V, F, D = FR.flexicubes(x, s, res)
    v, f, ld = ops.flexicubes(x.cuda(), s.cuda(), res)
    assert len(V) > 20
    assert torch.equal(f.cpu(), F) and torch.equal(v.cpu(), V)
    assert np.allclose(ld.cpu().numpy(), D.numpy(), atol=1e-6)


@gpu
@pytest.mark.parametrize("seed", range(5))
def test_topology_tables_fuzz_on_flexicubes_meshes(seed):
    """Meshes as the pipeline produces them (FlexiCubes on random fields: several components, all valences dual marching
    cubes emits, surfaces cut by the grid boundary -> open meshes) through GuidanceBatch.update_object: the incidence and
    neighbour tables equal the numpy builders', whichever path (closed-manifold device kernel or general sort) was taken."""
    from followmyhold_amd import engine as E, ops
    from helpers import make_scene
    rng = np.random.default_rng(700 + seed)
    res = [10, 14, 18, 12, 16][seed]
    x, _ = _grid(res)
    k = rng.uniform(2.0, 7.0, size=(3, 3))
    xt = x.numpy().astype(np.float64)
    s = sum(np.sin(xt @ k[i] + rng.uniform(0, 6.28)) for i in range(3)) / 3.0 + rng.uniform(-0.2, 0.2)
    if seed % 2 == 0:      # closed: push the boundary layer outside
        edge = (np.abs(xt).max(1) > 1.1 * (1 - 1.5 / res))
        s = np.where(edge, 1.0, s)
    v, f, _ = ops.flexicubes(x.cuda(), torch.from_numpy(s.astype(np.float32)).cuda(), res)
    assert len(f) > 50
    sc = {k_: (v_.numpy() if isinstance(v_, torch.Tensor) else v_) for k_, v_ in make_scene("ico2", 64, 64, seed=0).items()}
    gb = E.GuidanceBatch([sc], grid_res=16)
    gb.update_object(v * 0.04, f)
    of = f.cpu().numpy()
    Vh, Vtot = gb.meta[0]["Vh"], gb.Vtot
    faces = gb.faces.cpu().numpy().astype(np.int64)
    inc_off, inc_fc = E.incidence_csr(faces, Vtot)
    assert np.array_equal(gb.inc_off.cpu().numpy(), inc_off) and np.array_equal(gb.inc_fc.cpu().numpy(), inc_fc)
    edges = E.unique_edges(of)
    assert gb.meta[0]["n_edges"] == len(edges)
    off, idx = E.neighbour_csr(edges + Vh, Vtot)
    d_off, d_idx = gb.nbr_off.cpu().numpy(), gb.nbr_idx.cpu().numpy()
    for vv in range(Vh, Vtot):
        assert np.array_equal(np.sort(d_idx[d_off[vv]:d_off[vv + 1]]), idx[off[vv]:off[vv + 1]]), vv
    # and the step runs on them
    cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
    gb.step(cfg)
    torch.cuda.synchronize()
    assert np.isfinite(gb.loss_dict(0)["total"]) and np.isfinite(gb.grad_obj_verts(0).cpu().numpy()).all()


def _chain_scene(seed=0):
    from helpers import make_scene
    sc = make_scene("ico2", 64, 64, seed=seed)
    res = 14
    x = FR.construct_voxel_grid(res)[0] * 0.24
    r0 = float(sc["obj_verts"].norm(dim=1).mean())
    s0 = x.norm(dim=1) - r0 * (1.0 + 0.15 * torch.sin(25 * x[:, 0]) * torch.cos(21 * x[:, 1]))
    npsc = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
    return sc, npsc, x, s0, res


@gpu
def test_capacity_mode_objective_matches_the_oracle_chain_and_the_exact_size_path():
    """engine.SdfObjective: the whole iteration (SDF -> FlexiCubes -> install the new object on the device -> fused step ->
    dL/dSDF) as one hipGraph replay over capacity-sized buffers, counts in device memory.  Against the oracle chain
    (flexi_ref -> step_ref.phase_c_loss -> autograd) and against the exact-size path (ops.flexicubes + gb.objective) for
    a sequence of SDFs whose meshes differ in vertex count, face count and connectivity; graph replay == eager launches."""
    from followmyhold_amd import engine as E, ops
    from oracle import ref_ops as R
    from oracle import step_ref as S
    sc, npsc, x, s0, res = _chain_scene()
    cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
    gb = E.GuidanceBatch([npsc], grid_res=16, obj_capacity=(2048, 4096))
    obj = E.SdfObjective(gb, x, res)
    exact = E.GuidanceBatch([npsc], grid_res=16)
    p = S.make_params()
    sizes = set()
    for k, shift in enumerate([0.0, -0.01, 0.012, -0.004]):
        s = s0 + shift
        # oracle chain
        so = s.clone().requires_grad_(True)
        V, F, _ = FR.flexicubes(x, so, res)
        total, terms, aux = S.phase_c_loss(dict(sc, obj_faces=F), p, V, R.unique_edges(F), denoise_i=19, grid_res=16)
        total.backward()
        # capacity mode, graph replay
        sg = s.cuda().requires_grad_(True)
        loss = obj(sg, cfg)
        loss.backward()
        torch.cuda.synchronize()
        nv, nf, flags = obj.status()[0]
        assert flags == 0 and nv == len(V) and nf == len(F)
        sizes.add((nv, nf))
        v_cap, f_cap = obj.mesh(0)
        assert torch.equal(f_cap.cpu(), F) and torch.equal(v_cap.cpu(), V.detach())
        assert abs(float(loss) - float(total)) <= 1e-4 * abs(float(total))
        l = gb.loss_dict(0)
        assert abs(l["edge"] - float(terms["edge"])) <= 1e-4 * abs(float(terms["edge"])) and int(l["n_intersect"]) == aux["n_int"]
        g, go = sg.grad.cpu().numpy(), so.grad.numpy()
        assert np.linalg.norm(g - go) <= 2e-3 * np.linalg.norm(go)
        # exact-size path: same face ids, same loss terms, same gradient
        se = s.cuda().requires_grad_(True)
        ve, fe, _ = ops.flexicubes(x.cuda(), se, res)
        exact.objective(ve, fe, cfg).backward()
        torch.cuda.synchronize()
        P_ = 64 * 64
        pa = gb.region("p2f", torch.int32, (2, P_)).cpu().numpy()
        pe = exact.region("p2f", torch.int32, (2, P_)).cpu().numpy()
        assert np.array_equal(pa, pe)
        assert np.allclose(gb.losses.cpu().numpy(), exact.losses.cpu().numpy(), rtol=1e-6, atol=1e-9)
        ge = se.grad.cpu().numpy()
        assert np.linalg.norm(g - ge) <= 1e-4 * np.linalg.norm(ge)
        # eager launches of the same sequence
        s2 = s.cuda().requires_grad_(True)
        obj(s2, cfg, use_graph=False).backward()
        torch.cuda.synchronize()
        assert np.linalg.norm(s2.grad.cpu().numpy() - g) <= 1e-5 * np.linalg.norm(g)
    assert len(sizes) == 4                                  # four different topologies went through the same graph


@gpu
def test_capacity_mode_batch_of_two_and_updates():
    """B = 2 in capacity mode (the exact-size path is limited to one image): every image gets the result of its own
    single-image run, with the optimiser update applied (parameters move, object counts differ per image)."""
    from followmyhold_amd import engine as E
    _, sc0, x, s0, res = _chain_scene(0)
    _, sc1, _, s1, _ = _chain_scene(5)
    cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
    sdfs = [s0, s1 + 0.006]
    singles = []
    for scn, s in zip([sc0, sc1], sdfs):
        gb = E.GuidanceBatch([scn], grid_res=16, obj_capacity=(2048, 4096))
        o = E.SdfObjective(gb, x, res)
        for _ in range(2):
            o.run(s.cuda(), cfg)
        torch.cuda.synchronize()
        singles.append((gb.losses[0].cpu().numpy(), gb.params[0].cpu().numpy(), o.grad_sdf[0].cpu().numpy(), o.status()[0]))
    gb = E.GuidanceBatch([sc0, sc1], grid_res=16, obj_capacity=(2048, 4096))
    o = E.SdfObjective(gb, x, res)
    for _ in range(2):
        o.run(torch.stack(sdfs).cuda(), cfg)
    torch.cuda.synchronize()
    st = o.status()
    assert st[0][:2] != st[1][:2]
    for b in range(2):
        assert st[b] == singles[b][3] and st[b][2] == 0
        assert np.allclose(gb.losses[b].cpu().numpy(), singles[b][0], rtol=1e-5, atol=1e-7)
        assert np.allclose(gb.params[b].cpu().numpy(), singles[b][1], rtol=1e-5, atol=1e-7)
        gs = o.grad_sdf[b].cpu().numpy()
        assert np.linalg.norm(gs - singles[b][2]) <= 1e-4 * np.linalg.norm(singles[b][2])


@gpu
def test_capacity_mode_flags_overflow_and_open_surfaces():
    """The two situations capacity mode cannot serve are reported, not hidden: a mesh larger than the capacity (flag
    bit 4; the object is left out of that step) and an iso-surface that leaves the grid, i.e. is not closed (bit 5)."""
    from followmyhold_amd import engine as E
    _, npsc, x, s0, res = _chain_scene()
    cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
    gb = E.GuidanceBatch([npsc], grid_res=16, obj_capacity=(64, 128))
    o = E.SdfObjective(gb, x, res)
    o.run(s0.cuda(), cfg)
    nv, nf, flags = o.status()[0]
    assert flags & 16 and nv > 64
    with pytest.raises(E.L.FohoError, match="capacity"):
        gb.raise_on_flags()
    gb = E.GuidanceBatch([npsc], grid_res=16, obj_capacity=(4096, 8192))
    o = E.SdfObjective(gb, x, res)
    s_open = x[:, 0] - 0.05                                  # a plane: the surface runs out of the grid -> boundary edges
    o.run(s_open.cuda(), cfg)
    nv, nf, flags = o.status()[0]
    assert nf > 0 and flags & 32 and not flags & 16
    with pytest.raises(E.L.FohoError, match="manifold"):
        gb.raise_on_flags()
    gb.flags.zero_()
    o.run(s0.cuda(), cfg)                                    # the next closed surface is served again
    assert o.status()[0][2] == 0
