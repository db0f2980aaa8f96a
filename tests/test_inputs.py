"""On-disk formats at the edges of the path: .glb / .ply / .npy / .png / fov.json in the
reference's naming (src/foho/guidance/run.py:210-222), and the file-based mesh-level guidance driver."""
import json
import os
import struct

import numpy as np
import pytest
import torch

from followmyhold_amd import inputs, meshio, synthetic
from foho.guidance import run as G
from helpers import oracle_render_fn

gpu = pytest.mark.gpu


def _dirs(root):
    names = ["cropped_obj_img_dir", "mask_dir", "moge_out_dir", "hunyuan_hoi_mesh_dir", "hamer_out_dir", "h2m_rt_dir",
             "aligned_mano_dir", "guidance_out_dir"]
    return {n: os.path.join(str(root), n) for n in names}


def _gt_mesh(sc):
    """The "MoGe image mesh" of a synthetic scene: ground-truth hand + object in the MoGe world."""
    T = sc["T_h2m"].astype(np.float64)
    ov = sc["obj_verts"].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    v = np.concatenate([sc["gt_hand_verts"], ov.astype(np.float32)], 0)
    f = np.concatenate([sc["hand_faces"], sc["obj_faces"] + len(sc["gt_hand_verts"])], 0)
    return v, f


def test_glb_roundtrip_and_node_transforms(tmp_path):
    v, f = synthetic.icosphere(1, 0.3)
    p = str(tmp_path / "m.glb")
    inputs.save_glb(p, v, f)
    v2, f2 = inputs.load_glb(p)
    assert v2.dtype == np.float32 and f2.dtype == np.int64
    assert np.array_equal(v2, v) and np.array_equal(f2, f)
    # same geometry behind a node with translation + scale, 16-bit indices, interleaved (strided) positions
    inter = np.zeros((len(v), 6), "<f4")
    inter[:, :3] = v
    idx = f.astype("<u2").reshape(-1)
    binary = inter.tobytes() + idx.tobytes()
    binary += b"\0" * ((-len(binary)) % 4)
    gltf = {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0]}],
            "nodes": [{"children": [1], "translation": [1.0, 2.0, 3.0]}, {"mesh": 0, "scale": [2.0, 2.0, 2.0]}],
            "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}],
            "buffers": [{"byteLength": len(binary)}],
            "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": inter.nbytes, "byteStride": 24},
                            {"buffer": 0, "byteOffset": inter.nbytes, "byteLength": idx.nbytes}],
            "accessors": [{"bufferView": 0, "componentType": 5126, "count": len(v), "type": "VEC3"},
                          {"bufferView": 1, "componentType": 5123, "count": len(idx), "type": "SCALAR"}]}
    js = json.dumps(gltf).encode()
    js += b" " * ((-len(js)) % 4)
    p2 = str(tmp_path / "n.glb")
    with open(p2, "wb") as fh:
        fh.write(struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + 8 + len(binary)))
        fh.write(struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(binary), 0x004E4942) + binary)
    v3, f3 = inputs.load_glb(p2)
    assert np.allclose(v3, 2.0 * v + np.array([1.0, 2.0, 3.0]), atol=1e-6) and np.array_equal(f3, f)
    with pytest.raises(ValueError):
        open(p2, "wb").write(b"not a glb file....")
        inputs.load_glb(p2)


def test_scene_files_roundtrip_cpu(tmp_path):
    """Write one image's inputs in the reference's formats / names and read them back through derive_paths."""
    sc = synthetic.build_scene(oracle_render_fn, obj_kind="ico2", H=48, W=48, seed=3)
    d = _dirs(tmp_path)
    mv, mf = _gt_mesh(sc)
    inputs.save_scene_files(sc, mv, mf, {k: v for k, v in d.items() if k != "guidance_out_dir"}, "0007")
    name = "0007_cropped_hoi_1.png"
    assert sorted(os.listdir(d["cropped_obj_img_dir"])) == [name]
    p = G.derive_paths(name, **d)
    assert p["index"] == "0007" and p["is_right"] == "1"
    for k in ["cropped_hand_mask_path", "cropped_obj_mask_path", "moge_mesh_path", "moge_fov_path", "T_h2m_path",
              "aligned_mano_mesh_path", "hunyuan_hoi_mesh_path", "hamer_for_guid_path"]:
        assert os.path.exists(p[k]), k
    back = inputs.load_scene_from_files(p, sc["J_regressor"], oracle_render_fn)
    assert back["H"] == 48 and back["W"] == 48 and back["fov"] == sc["fov"]
    assert np.array_equal(back["hand_mask"], sc["hand_mask"]) and np.array_equal(back["obj_mask"], sc["obj_mask"])
    assert np.array_equal(back["kps_2d"], sc["kps_2d"]) and np.array_equal(back["obj_verts"], sc["obj_verts"])
    assert np.array_equal(back["obj_faces"], sc["obj_faces"]) and np.array_equal(back["hand_faces"], sc["hand_faces"])
    assert np.allclose(back["T_h2m"], sc["T_h2m"]) and np.allclose(back["hand_verts"], sc["hand_verts"], atol=2e-6)
    # targets: the MoGe mesh is the ground-truth scene (up to float32 rounding of the object's vertices), so rendering it
    # reproduces the scene's own target maps except for the odd pixel on a silhouette
    bad = (np.abs(back["moge_disp"] - sc["moge_disp"]) > 1e-4) | (np.abs(back["moge_normal"] - sc["moge_normal"]).max(-1) > 1e-3)
    assert bad.mean() < 0.01


@gpu
def test_file_based_mesh_level_guidance(tmp_path, monkeypatch):
    """`foho.guidance.run.run` end to end on files, without the diffusion model: inputs read from the reference's
    formats, targets rendered on the GPU, phases A/B/C, {idx}_obj.ply / {idx}_hand.ply written."""
    from followmyhold_amd import engine as E
    sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="ico4", H=128, W=128, seed=5)
    d = _dirs(tmp_path)
    mv, mf = _gt_mesh(sc)
    inputs.save_scene_files(sc, mv, mf, {k: v for k, v in d.items() if k != "guidance_out_dir"}, "12")
    jr = str(tmp_path / "J.npy")
    np.save(jr, sc["J_regressor"])
    monkeypatch.setenv("FOHO_J_REGRESSOR", jr)
    monkeypatch.setenv("FOHO_MESH_LEVEL_GUIDANCE", "1")
    # the scene read back from the files drives the same first iterations as the in-memory scene
    p = G.derive_paths("12_cropped_hoi_1.png", **d)
    back = inputs.load_scene_from_files(p, sc["J_regressor"], E.hip_render_fn("cuda"))
    bad = (np.abs(back["moge_disp"] - sc["moge_disp"]) > 1e-4) | (np.abs(back["moge_normal"] - sc["moge_normal"]).max(-1) > 1e-3)
    assert bad.mean() < 0.005
    cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
    ga, gb = E.GuidanceBatch([sc], grid_res=32), E.GuidanceBatch([back], grid_res=32)
    for _ in range(2):
        ga.step(cfg)
        gb.step(cfg)
    torch.cuda.synchronize()
    assert abs(ga.loss_dict(0)["total"] - gb.loss_dict(0)["total"]) <= 2e-2 * abs(ga.loss_dict(0)["total"])
    assert np.allclose(ga.params.cpu().numpy(), gb.params.cpu().numpy(), atol=2e-2)
    # the driver (short schedule so that the test stays fast)
    from foho import configs
    short = configs.OptimizationConfig()
    short.optimization_steps_hand, short.optimization_steps_scale, short.optimization_steps_joint = 6, 4, 3
    monkeypatch.setattr(G, "OptimizationConfig", lambda: short)
    G.run(project_root=str(tmp_path), task_list_file=None, **d)
    out_obj, out_hand = os.path.join(d["guidance_out_dir"], "12_obj.ply"), os.path.join(d["guidance_out_dir"], "12_hand.ply")
    assert os.path.exists(out_obj) and os.path.exists(out_hand)
    ov, of = meshio.load_ply(out_obj)
    hv, hf = meshio.load_ply(out_hand)
    assert ov.shape == sc["obj_verts"].shape and np.array_equal(of, sc["obj_faces"])
    assert hv.shape == sc["hand_verts"].shape and np.array_equal(hf, sc["hand_faces"])
    assert np.isfinite(ov).all() and np.isfinite(hv).all()
    assert np.abs(hv - sc["gt_hand_verts"]).max() < 0.25                  # same neighbourhood as the ground truth (metres)
    # second call: outputs exist -> skipped (RUN:224-226)
    t0 = os.path.getmtime(out_obj)
    G.run(project_root=str(tmp_path), task_list_file=None, **d)
    assert os.path.getmtime(out_obj) == t0


def _image_mesh(n, fov=60.0):
    """MoGe-style image mesh (utils3d.image_mesh, src/foho/geometry/moge.py:137-158): one vertex per pixel of an n x n
    depth map, two faces per pixel quad."""
    ys, xs = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    t = np.tan(np.radians(fov) / 2)
    z = 0.5 + 0.1 * np.sin(xs / n * 6.0) * np.cos(ys / n * 5.0)
    x = (xs + 0.5 - n / 2) / (n / 2) * t * z * 0.9
    y = -(ys + 0.5 - n / 2) / (n / 2) * t * z * 0.9
    v = np.stack([x, y, -z], -1).reshape(-1, 3).astype(np.float32)
    i = (ys[:-1, :-1] * n + xs[:-1, :-1]).reshape(-1)
    f = np.concatenate([np.stack([i, i + n, i + 1], 1), np.stack([i + 1, i + n, i + n + 1], 1)], 0).astype(np.int64)
    return v, f


@gpu
def test_image_mesh_target_render_matches_oracle():
    """the once-per-image render of the MoGe image mesh (large F, screen-filling) into the
    target maps -- face indices bit-exact, maps equal to the oracle's."""
    from followmyhold_amd import engine as E
    v, f = _image_mesh(96)
    n_ref, d_ref, p_ref = oracle_render_fn(v, f, 128, 128, 60.0)
    n_hip, d_hip, p_hip = E.hip_render_fn("cuda")(v, f, 128, 128, 60.0)
    assert (p_ref >= 0).mean() > 0.7
    assert np.array_equal(p_hip, p_ref)
    assert np.abs(d_hip - d_ref).max() < 1e-5 and np.abs(n_hip - n_ref).max() < 1e-4


@gpu
def test_device_side_target_renderer_equals_the_per_image_render():
    """engine.TargetRenderer (the target maps rendered inside an image's job, straight into the batch's tgt_normal / tgt_disp)
    against the oracle and engine.hip_render_fn: image meshes of different sizes through ONE renderer, one after the other and
    back again, masked with the hand-object mask."""
    from followmyhold_amd import engine as E
    H = W = 128
    rd = E.TargetRenderer(H, W, H * W, 2 * (H - 1) * (W - 1))
    rng = np.random.default_rng(5)
    mask = torch.from_numpy((rng.random((H, W)) < 0.6).astype(np.uint8)).cuda()
    out_n, out_d = torch.zeros(H, W, 3, device="cuda"), torch.zeros(H, W, device="cuda")
    render = E.hip_render_fn("cuda")
    for k, (n, fov) in enumerate(((96, 60.0), (40, 47.0), (128, 60.0), (96, 60.0))):
        v, f = _image_mesh(n)
        rd.render_into(("t", k % 2), v, f.astype(np.int32), fov, mask, out_n, out_d)
        fl = rd.flags.cpu().numpy()
        assert fl[0] == 0 and fl[1] == 0
        n_ref, d_ref, p_ref = render(v, f, H, W, fov)
        m = mask.cpu().numpy().astype(np.float32)
        assert np.array_equal(out_n.cpu().numpy(), n_ref * m[..., None]) and np.array_equal(out_d.cpu().numpy(), d_ref * m)
        if k == 0:
            n_or, d_or, _ = oracle_render_fn(v, f, H, W, fov)
            assert np.abs(out_d.cpu().numpy() - d_or * m).max() < 1e-5 and np.abs(out_n.cpu().numpy() - n_or * m[..., None]).max() < 1e-4


@gpu
def test_guidance_driver_under_torchrun_two_ranks(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m foho.guidance.run ...` on a two-image tree (the product
    driver's N>1 path; RUN:178-185, 208-259): each rank binds a GPU (both share the one GPU of this box, so the
    metrics all-reduce goes through gloo; on a node with one rank per GPU it is RCCL), writes its image's meshes, and
    rank 0 prints the all-reduced totals.  Full reference schedule: 200 + 100 + 9 x 50 iterations per image."""
    import subprocess
    import sys
    from followmyhold_amd import engine as E
    d = _dirs(tmp_path)
    jr = None
    for k, idx in enumerate(["0004", "0009"]):
        sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="ico3", H=96, W=96, seed=20 + k)
        mv, mf = _gt_mesh(sc)
        inputs.save_scene_files(sc, mv, mf, {k2: v for k2, v in d.items() if k2 != "guidance_out_dir"}, idx)
        jr = sc["J_regressor"]
    jr_path = str(tmp_path / "J.npy")
    np.save(jr_path, jr)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FOHO_J_REGRESSOR=jr_path, FOHO_MESH_LEVEL_GUIDANCE="1", FOHO_DIST_BACKEND="gloo",
               PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", "-m", "foho.guidance.run", "--project_root", str(tmp_path)]
    for k, v in d.items():
        cmd += [f"--{k}", v]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-4000:]
    for idx in ["0004", "0009"]:
        for kind in ["obj", "hand"]:
            v, f = meshio.load_ply(os.path.join(d["guidance_out_dir"], f"{idx}_{kind}.ply"))
            assert np.isfinite(v).all() and len(f) > 0
    line = [l for l in r.stdout.splitlines() if "Batch metrics:" in l]
    assert len(line) == 1, r.stdout[-4000:]
    tot = json.loads(line[0].split("Batch metrics:", 1)[1])
    assert tot["world_size"] == 2 and tot["n_images"] == 2 and tot["n_steps"] == 2 * 750
    assert np.isfinite(tot["sum_total_loss"]) and tot["sum_total_loss"] > 0 and tot["n_nan"] == 0


def _tame(cfg, a, b, c):
    """A short schedule with learning rates 1/500 of the reference's: at its own settings (quaternion learning rate 0.5
    with eps 1e-4, CFG:21-26) the reference's optimisation is chaotic -- the 1e-7 noise of atomic float sums decides single
    Adam steps -- and two runs of the SAME code on the same inputs part ways within a few iterations.
    (scripts/dev/dev_determinism.py: at the reference's rates a fresh runner and the exact-size driver end 4 + 2 + 9 x 2
    iterations 4e-2 m apart, at a fiftieth of them 1.6e-4 m).  Tests that compare two executions of a schedule (batched
    against one by one, slot re-use against a fresh slot) run it where the trajectory is a function of the inputs."""
    cfg.optimization_steps_hand, cfg.optimization_steps_scale, cfg.optimization_steps_joint = a, b, c
    for name in ("phase1_hand_lrs", "phase2_hand_lrs", "obj_2half_lrs", "obj_lrs"):
        setattr(cfg, name, {k: v / 500.0 for k, v in getattr(cfg, name).items()})
    return cfg


def _write_tree(tmp_path, n, size=64, kinds=("ico2", "ico3")):
    """n scene folders with objects of different sizes; returns (dirs, scenes by index, J regressor path)."""
    from followmyhold_amd import engine as E
    d = _dirs(tmp_path)
    scenes = {}
    for k in range(n):
        idx = f"{3 + 4 * k:04d}"
        sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind=kinds[k % len(kinds)], H=size, W=size, seed=40 + k)
        mv, mf = _gt_mesh(sc)
        inputs.save_scene_files(sc, mv, mf, {k2: v for k2, v in d.items() if k2 != "guidance_out_dir"}, idx)
        scenes[idx] = sc
    jr = str(tmp_path / "J.npy")
    np.save(jr, next(iter(scenes.values()))["J_regressor"])
    return d, scenes, jr


@gpu
def test_batched_driver_equals_one_image_at_a_time(tmp_path, monkeypatch, capsys):
    """SURVEY.md 8(e) "within a GPU, batch the rank's images": `foho.guidance.run.run` takes nine scene folders (two object
    sizes) through MeshGuidanceRunner four at a time -- two full image sets and a padded one, the second and third loaded
    INTO the slots of the first -- and every image's meshes equal what the same driver produces one image at a time and
    what the exact-size single-image driver (`inputs.run_mesh_guidance`) produces.  One empty mask and one missing file
    exercise the per-image skip / error isolation of RUN:224-236, 257-259 inside an image set."""
    from foho import configs
    from followmyhold_amd import engine as E
    d, scenes, jr = _write_tree(tmp_path, 9)
    monkeypatch.setenv("FOHO_J_REGRESSOR", jr)
    monkeypatch.setenv("FOHO_MESH_LEVEL_GUIDANCE", "1")
    short = _tame(configs.OptimizationConfig(), 2, 2, 1)
    monkeypatch.setattr(G, "OptimizationConfig", lambda: short)
    idxs = sorted(scenes)
    # two more list entries that must not disturb the others: an empty hand mask (skipped) and a missing key-point file (error)
    for extra, seed in (("0100", 70), ("0101", 71)):
        sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="ico2", H=64, W=64, seed=seed)
        mv, mf = _gt_mesh(sc)
        inputs.save_scene_files(sc, mv, mf, {k2: v for k2, v in d.items() if k2 != "guidance_out_dir"}, extra)
    inputs.save_mask(os.path.join(d["mask_dir"], "0100_cropped_hand_mask.png"), np.zeros((64, 64), bool))
    os.remove(os.path.join(d["hamer_out_dir"], "0101_kps_for_guidance.npy"))

    def drive(in_flight, out):
        monkeypatch.setenv("FOHO_IMAGES_IN_FLIGHT", str(in_flight))
        dd = dict(d, guidance_out_dir=str(tmp_path / out))
        tot = G.run(project_root=str(tmp_path), task_list_file=None, **dd)
        txt = capsys.readouterr().out
        res = {}
        for idx in idxs:
            res[idx] = (meshio.load_ply(os.path.join(dd["guidance_out_dir"], f"{idx}_obj.ply")),
                        meshio.load_ply(os.path.join(dd["guidance_out_dir"], f"{idx}_hand.ply")))
        return tot, txt, res

    tot4, txt4, r4 = drive(4, "out4")
    tot1, txt1, r1 = drive(1, "out1")
    for tot, txt in ((tot4, txt4), (tot1, txt1)):
        assert tot["n_images"] == 9 and tot["n_steps"] == 9 * (2 + 2 + 9) and tot["n_failed"] == 1 and tot["n_nan"] == 0
        assert "Skipping 0100 due to empty mask" in txt and "Error in processing 0101_cropped_hoi_1.png" in txt
        assert txt.count("Reconstructed object") == 9
        assert not os.path.exists(os.path.join(str(tmp_path), "out4", "0100_obj.ply"))
    # (the loss itself is only compared loosely: one pixel crossing the BCE clamp moves an image's total by 84 w_sil / P = 0.2)
    assert abs(tot4["sum_total_loss"] - tot1["sum_total_loss"]) <= 5e-2 * abs(tot1["sum_total_loss"])
    for idx in idxs:
        sc = scenes[idx]
        for (v4, f4), (v1, f1) in zip(r4[idx], r1[idx]):
            assert np.array_equal(f4, f1) and v4.shape == v1.shape
            assert np.abs(v4 - v1).max() < 5e-5, idx
        assert np.array_equal(r4[idx][0][1], sc["obj_faces"]) and np.array_equal(r4[idx][1][1], sc["hand_faces"])
    # ... and against the exact-size driver (host / device topology builders, its own graphs) for three of the images
    for idx in (idxs[0], idxs[4], idxs[8]):
        p = G.derive_paths(f"{idx}_cropped_hoi_1.png", **d)
        back = inputs.load_scene_from_files(p, scenes[idx]["J_regressor"], E.hip_render_fn("cuda"))
        gb = inputs.run_mesh_guidance([back], short)
        (ov, _), (hv, _) = inputs.export_meshes(gb, 0, str(tmp_path / "e_obj.ply"), str(tmp_path / "e_hand.ply"))
        assert np.abs(ov - r4[idx][0][0]).max() < 5e-5 and np.abs(hv - r4[idx][1][0]).max() < 5e-5, idx
    # different images really are different (the comparison above is not vacuous)
    assert np.abs(r4[idxs[0]][1][0] - r4[idxs[2]][1][0]).max() > 1e-2
    # the outputs carry the LAST optimiser update (one lagging update at these rates is ~1e-5 m: not resolved here, see
    # test_exported_meshes_follow_the_final_parameters)


@gpu
def test_runner_reuses_slots_and_graphs_across_image_sets():
    """MeshGuidanceRunner: later jobs run on the slots and hipGraphs of the first ones (no new slot, no new capture), also
    when their objects have other vertex / face counts; a list longer than `in_flight` brings in the stream's second slot
    (jobs alternate between the two) and nothing more; a larger object than the capacity rebuilds one slot; an open object
    mesh is handed back for the exact-size driver."""
    from followmyhold_amd import engine as E
    short = _tame(E.OptimizationConfig(), 4, 2, 2)
    rf = E.hip_render_fn("cuda")
    mk = lambda kind, seed: synthetic.build_scene(rf, obj_kind=kind, H=64, W=64, seed=seed)
    runner = inputs.MeshGuidanceRunner(short, in_flight=2, grid_res=16)
    a = runner.run([mk("ico3", 1), mk("ico2", 2)])
    assert runner.stats["slots_built"] == 1 and runner.stats["captures"] == 4    # phase A, phase B, phase C without / with the intersection gate
    b = runner.run([mk("ico2", 3), mk("ico3", 4), mk("ico2", 5)])       # two jobs, the second one padded
    assert runner.stats["slots_built"] == 1 and runner.stats["captures"] == 4 and runner.stats["jobs"] == 3
    # a long list (length unknown to the runner): the stream's second slot comes in, jobs alternate between the two
    more = [mk("ico2", 3), mk("ico3", 4), mk("ico2", 5), mk("ico2", 8), mk("ico3", 9)]
    b2 = [r for _, r in runner.run_stream(((i, s) for i, s in enumerate(more)))]
    assert runner.stats["slots_built"] == 2 and runner.stats["captures"] == 8 and runner.stats["jobs"] == 6
    for r, r2 in zip(b, b2):     # whichever slot an image lands in, and whatever ran there before
        assert np.abs(r["hand"][0] - r2["hand"][0]).max() < 5e-5 and np.abs(r["obj"][0] - r2["obj"][0]).max() < 5e-5
    assert all(r["ok"] and np.isfinite(r["hand"][0]).all() and np.isfinite(r["obj"][0]).all() for r in a + b)
    assert [len(r["obj"][0]) for r in b] == [162, 642, 162]
    # a repeat of the first set on the re-used slots gives the first set's answer
    a2 = runner.run([mk("ico3", 1), mk("ico2", 2)])
    for r, r2 in zip(a, a2):
        assert np.abs(r["hand"][0] - r2["hand"][0]).max() < 5e-5 and np.abs(r["obj"][0] - r2["obj"][0]).max() < 5e-5
    big = runner.run([mk("ico4", 6)])
    assert big[0]["ok"] and runner.stats["slots_built"] == 3 and len(big[0]["obj"][0]) == 2562
    sc = mk("ico2", 7)
    sc["obj_faces"] = sc["obj_faces"][:-2]         # a hole: not a closed manifold
    opened = runner.run([sc])
    assert not opened[0]["ok"] and opened[0]["reason"] == "fallback"


@gpu
def test_runner_takes_mixed_image_sizes_and_image_meshes_in_one_list():
    """One list with two image sizes, scenes that bring their target maps and scenes that bring the image mesh instead (rendered
    on the device inside the job): jobs are formed per shape, slots and renderers are rebuilt where a shape does not fit, and
    every image's result equals its own single-image run."""
    from followmyhold_amd import engine as E
    short = _tame(E.OptimizationConfig(), 4, 2, 2)
    rf = E.hip_render_fn("cuda")

    def mk(kind, seed, size, deferred):
        sc = synthetic.build_scene(rf, obj_kind=kind, H=size, W=size, seed=seed)
        if deferred:      # the "MoGe mesh" of the synthetic scene: ground-truth hand + object; its render gives the target maps
            v, f = _gt_mesh(sc)
            n, d, _ = rf(v, f, size, size, sc["fov"])
            hoi = (sc["hand_mask"] | sc["obj_mask"]).astype(np.float32)
            sc["moge_normal"], sc["moge_disp"] = (n * hoi[..., None]).astype(np.float32), (d * hoi).astype(np.float32)
            dsc = {k: v_ for k, v_ in sc.items() if k not in ("moge_normal", "moge_disp")}
            dsc["moge_mesh"] = (v.astype(np.float32), f.astype(np.int32))
            return sc, dsc
        return sc, sc

    # jobs of two images: (maps, mesh) at 64 x 64, (mesh, maps) at 80 x 80 -- mixed within a job --, then a padded one
    cases = [mk("ico2", 21, 64, False), mk("ico2", 23, 64, True), mk("ico3", 22, 80, True), mk("ico2", 24, 80, False), mk("ico3", 25, 64, False)]
    runner = inputs.MeshGuidanceRunner(short, in_flight=2, grid_res=16)
    got = [r for _, r in runner.run_stream(((i, c[1]) for i, c in enumerate(cases)))]
    assert len(got) == len(cases) and all(r["ok"] for r in got)
    for (full, _), r in zip(cases, got):
        ref = inputs.MeshGuidanceRunner(short, in_flight=1, grid_res=16).run([full])[0]
        assert np.abs(ref["hand"][0] - r["hand"][0]).max() < 5e-5 and np.abs(ref["obj"][0] - r["obj"][0]).max() < 5e-5


@gpu
def test_exported_meshes_follow_the_final_parameters():
    """The meshes a job returns are the input meshes under the FINAL parameters (the reference builds debug_mano /
    debug_transformed_obj_mesh after the last optimiser step, PL:1614-1618, 1653-1657), not the vertices the last iteration
    started from: both drivers against a float64 similarity transform of the inputs with the parameters they report."""
    from followmyhold_amd import engine as E
    from oracle import ref_ops as R
    cfg = E.OptimizationConfig()            # the reference's learning rates: one update moves the hand by centimetres
    cfg.optimization_steps_hand, cfg.optimization_steps_scale, cfg.optimization_steps_joint = 3, 2, 1
    sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="ico2", H=64, W=64, seed=11)

    def expect(params):
        p = torch.from_numpy(np.asarray(params, np.float64))
        hv = torch.from_numpy(sc["hand_verts"].astype(np.float64))
        ov = torch.from_numpy(sc["obj_verts"].astype(np.float64))
        hand = R.transform_around_center_w_scale(hv, R.quaternion_to_matrix(p[4:8]), p[1:4], p[0:1])
        moge = R.transform_hunyuan2moge(ov, torch.from_numpy(sc["T_h2m"].astype(np.float64)))
        obj = R.transform_around_center_w_scale(moge, R.quaternion_to_matrix(p[12:16]), p[9:12], p[8:9])
        return hand.numpy(), obj.numpy()

    res = inputs.MeshGuidanceRunner(cfg, in_flight=1, grid_res=16).run([sc])[0]
    hand, obj = expect(res["params"])
    assert np.abs(res["hand"][0] - hand).max() < 1e-5 and np.abs(res["obj"][0] - obj).max() < 1e-5
    assert np.abs(res["hand"][0] - sc["hand_verts"]).max() > 1e-3               # the hand did move
    gb = inputs.run_mesh_guidance([sc], cfg)
    hand, obj = expect(gb.params[0].cpu().numpy())
    m = gb.meta[0]
    world = gb.region("world", torch.float32, (-1, 3)).cpu().numpy()
    assert np.abs(world[:m["Vh"]] - hand).max() < 1e-5 and np.abs(world[m["Vh"]:m["Vh"] + m["Vo"]] - obj).max() < 1e-5
