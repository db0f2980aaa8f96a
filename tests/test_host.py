"""Host-side logic on CPU: CSR builders, phase recipes, mesh IO, image sharding and the gloo metrics all-reduce."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from followmyhold_amd import engine as E
from followmyhold_amd import meshio, sharding, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_incidence_csr_reproduces_index_add_order():
    v, f = synthetic.icosphere(1)
    off, fc = E.incidence_csr(f, len(v))
    assert off[-1] == 3 * len(f)
    for vi in range(len(v)):
        ent = fc[off[vi]:off[vi + 1]]
        keys = [((int(e) & 3), int(e) >> 2) for e in ent]
        assert keys == sorted(keys)                        # corner-major, then face id
        for c, fa in keys:
            assert f[fa, c] == vi
    # normals accumulated in CSR order == three index_add passes
    vt = torch.from_numpy(v)
    fn = torch.cross(vt[f[:, 2]] - vt[f[:, 1]], vt[f[:, 0]] - vt[f[:, 1]], dim=1)
    ref = torch.zeros_like(vt).index_add(0, torch.from_numpy(f[:, 0]), fn).index_add(0, torch.from_numpy(f[:, 1]), fn) \
        .index_add(0, torch.from_numpy(f[:, 2]), fn)
    acc = torch.zeros_like(vt)
    for vi in range(len(v)):
        for e in fc[off[vi]:off[vi + 1]]:
            acc[vi] = acc[vi] + fn[int(e) >> 2]
    assert torch.equal(acc, ref)


def test_neighbour_csr_and_unique_edges():
    v, f = synthetic.icosphere(2)
    e = E.unique_edges(f)
    assert len(e) == 3 * len(f) // 2 and len(v) - len(e) + len(f) == 2   # closed genus-0 surface
    off, idx = E.neighbour_csr(e, len(v))
    assert off[-1] == 2 * len(e)
    deg = np.diff(off)
    assert deg.min() >= 5 and deg.max() <= 6
    for a, b in e[:50]:
        assert b in idx[off[a]:off[a + 1]] and a in idx[off[b]:off[b + 1]]


def test_synthetic_meshes_have_the_reference_sizes():
    hv, hf = synthetic.hand_template()
    assert hv.shape == (778, 3) and hf.shape == (1552, 3)                 # MANO 1538 + 14 wrist-closing faces
    assert len(E.unique_edges(hf)) == 2328                                # watertight: E = 3F/2
    assert synthetic.icosphere(4)[1].shape == (5120, 3)
    ov, of = synthetic.make_object("20k")
    assert ov.shape == (10242, 3) and of.shape == (20480, 3)
    ov, of = synthetic.make_object("40k")
    assert of.shape[0] == 40320 and ov.shape[0] == 20160 and len(E.unique_edges(of)) == 3 * 40320 // 2
    m = synthetic.mano_like_model()
    assert m["posedirs"].shape == (135, 2334) and m["shapedirs"].shape == (778, 3, 10)
    assert m["J_regressor"].shape == (16, 778) and np.allclose(m["lbs_weights"].sum(1), 1, atol=1e-5)


def test_phase_recipes_match_the_reference_weights():
    """Effective weights inside the total loss (pipelines.py:1343-1349, 1433-1440, 1499-1504 + 1578-1588)."""
    a, n = E.phase_cfg("A")
    assert n == 1 and a.render[0].face_set == E.L.FACES_HAND
    assert (a.render[0].w_normal, a.render[0].w_disp, a.render[0].w_sil) == (1.0, 10.0, 1.0)
    assert a.w_kps == pytest.approx(1e-2) and a.w_trans_hand == pytest.approx(1e-2) and a.weight_decay == 0.0
    assert list(a.lr)[:8] == pytest.approx([1e-2] * 4 + [0.5] * 4) and list(a.lr)[8:] == [0.0] * 8
    b, n = E.phase_cfg("B")
    assert n == 1 and b.render[0].face_set == E.L.FACES_OBJ and b.render[0].w_sil == 100.0
    assert b.w_edge == 1.0 and b.w_verts_obj == pytest.approx(1e-3) and b.w_trans_obj == pytest.approx(1e-2)
    assert list(b.lr)[:8] == [0.0] * 8 and b.weight_decay == pytest.approx(0.01)
    c, n = E.phase_cfg("C", denoise_i=16)
    assert n == 2 and c.int_gate_step_ok == 0 and E.phase_cfg("C", denoise_i=17)[0].int_gate_step_ok == 1
    assert c.render[0].w_normal == pytest.approx(1e-2) and c.render[0].w_sil == 0.0 and c.render[1].w_sil == 10.0
    assert c.render[1].disp_mask == E.L.MASK_NONE and c.render[0].disp_mask == E.L.MASK_HAND   # PL:1568 vs PL:1497
    assert c.w_kps == pytest.approx(1e-7) and c.w_trans_hand == pytest.approx(1e-5) and c.w_contact == 10.0
    assert list(c.lr) == pytest.approx([1e-4] * 4 + [1e-2] * 4 + [5e-2] + [1e-2] * 7)
    assert c.eps == pytest.approx(1e-4) and c.blur_radius == pytest.approx(np.log(1 / 1e-4 - 1) * 1e-8, rel=1e-6)
    with pytest.raises(ValueError):
        E.phase_cfg("D")


def test_meshio_roundtrip(tmp_path):
    v, f = synthetic.icosphere(1, 0.3)
    for binary in (True, False):
        p = str(tmp_path / f"m{int(binary)}.ply")
        meshio.save_ply(p, v, f, binary=binary)
        v2, f2 = meshio.load_ply(p)
        assert np.allclose(v2, v, atol=1e-6) and np.array_equal(f2, f)
    # polygonal binary faces: uniform quads (one structured read) and mixed triangles / quads (row-by-row fallback),
    # with a scalar property after the list, big endian
    import struct
    for rows in ([[0, 1, 2, 3], [3, 2, 1, 0]], [[0, 1, 2], [0, 1, 2, 3], [4, 3, 2, 1, 0]]):
        p = str(tmp_path / "poly.ply")
        with open(p, "wb") as fh:
            fh.write((f"ply\nformat binary_big_endian 1.0\nelement vertex 5\nproperty float x\nproperty float y\n"
                      f"property float z\nelement face {len(rows)}\nproperty list uchar int vertex_indices\n"
                      f"property uchar flags\nend_header\n").encode())
            fh.write(np.arange(15, dtype=">f4").tobytes())
            for r in rows:
                fh.write(struct.pack(f">B{len(r)}iB", len(r), *r, 7))
        v2, f2 = meshio.load_ply(p)
        want = [[r[0], r[k], r[k + 1]] for r in rows for k in range(1, len(r) - 1)]
        assert np.array_equal(v2, np.arange(15, dtype=np.float32).reshape(5, 3)) and np.array_equal(f2, want)
    p = str(tmp_path / "pc.ply")
    meshio.save_ply(p, v)
    v2, f2 = meshio.load_ply(p)
    assert np.array_equal(v2, v) and f2.shape == (0, 3)
    p = str(tmp_path / "m.obj")
    meshio.save_obj(p, v, f)
    v2, f2 = meshio.load_mesh(p)
    assert np.allclose(v2, v, atol=1e-6) and np.array_equal(f2, f)
    from followmyhold_amd import inputs
    p = str(tmp_path / "m.glb")
    inputs.save_glb(p, v, f)
    v2, f2 = meshio.load_mesh(p)
    assert np.array_equal(v2, v) and np.array_equal(f2, f)
    with pytest.raises(ValueError):
        meshio.load_mesh(str(tmp_path / "x.stl"))


def test_image_sharding_is_a_partition():
    items = [f"{i}_cropped_hoi_1.png" for i in range(64)]
    parts = [sharding.shard_images(items, r, 8) for r in range(8)]
    assert all(len(p) == 8 for p in parts)
    assert sorted(sum(parts, [])) == sorted(items)
    assert sharding.shard_images(items, 3, 8)[0] == items[3]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from followmyhold_amd import sharding
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank = dist.get_rank()
items = [f"{i}_x_1.png" for i in range(5)]
mine = sharding.shard_images(items, rank, dist.get_world_size())
class FakeBatch:   # what GuidanceBatch exposes to local_metrics
    B = len(mine)
    losses = torch.arange(len(mine) * 24, dtype=torch.float32).reshape(len(mine), 24) + 100 * rank
    flags = torch.tensor([rank] * len(mine), dtype=torch.int32)
vec = sharding.local_metrics(FakeBatch, n_steps=3, wall_ms=10.0 * (rank + 1))
out = sharding.all_reduce_metrics(vec.clone(), dist)
if rank == 0:
    print("RESULT", out.tolist(), flush=True)
dist.barrier(); dist.destroy_process_group()
"""


def test_metrics_all_reduce_world_size_2_gloo(tmp_path):
    """N>1 path on CPU: two processes, gloo, 127.0.0.1 rendezvous; sums must equal the serial computation."""
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT")][0]
    got = eval(line[len("RESULT "):])
    # serial expectation
    exp = np.zeros(len(sharding.METRIC_NAMES))
    for r in range(2):
        n = len(list(range(5))[r::2])
        losses = np.arange(n * 24, dtype=np.float64).reshape(n, 24) + 100 * r
        exp[0] += n
        exp[1] += 3 * n
        for j, col in enumerate([0, 1, 2, 3, 7, 8, 9, 11, 12, 13]):
            exp[2 + j] += losses[:, col].sum()
        exp[12] += 10.0 * (r + 1)
        exp[13] += n * (r & 1)
        exp[14] += n * (1 if (r & 6) else 0)
    assert np.allclose(got, exp)


DRIVER_WORKER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
root = sys.argv[2]
import numpy as np
from PIL import Image
from foho.guidance import run as G
from followmyhold_amd import sharding

seen = []
def fake_run_hunyuan_w_guid(**kw):      # stands in for the GPU work of one image; same keyword contract as RUN:65-79
    idx = int(os.path.basename(kw["save_path_obj"]).split("_")[0])
    assert "device" not in kw                # the reference's keyword set (RUN:237-250); the rank's GPU is the current device
    seen.append(idx)
    v = np.zeros(len(sharding.METRIC_NAMES)); v[0] = 1; v[1] = 750; v[2] = float(idx)
    G._tally(v)
    open(kw["save_path_obj"], "w").write("x"); open(kw["save_path_hand"], "w").write("x")
    return object(), object()
G.run_hunyuan_w_guid = fake_run_hunyuan_w_guid
d = {n: os.path.join(root, n) for n in ["cropped_obj_img_dir", "mask_dir", "moge_out_dir", "hunyuan_hoi_mesh_dir",
                                        "hamer_out_dir", "h2m_rt_dir", "aligned_mano_dir", "guidance_out_dir"]}
out = G.run(project_root=root, task_list_file=None, **d)
print("SEEN", int(os.environ["RANK"]), sorted(seen), flush=True)
if int(os.environ["RANK"]) == 0:
    print("TOTALS", json.dumps(out), flush=True)
else:
    assert out is None
"""


def test_guidance_driver_world_size_2_gloo(tmp_path):
    """foho.guidance.run.run() as two ranks (gloo, 127.0.0.1): every rank processes its round-robin share of the image
    list (the torch.distributed counterpart of the SLURM array, RUN:178-185), the metrics vector is all-reduced once
    at the end of the batch and rank 0 reports the totals."""
    from PIL import Image
    root = tmp_path / "tree"
    dirs = {n: root / n for n in ["cropped_obj_img_dir", "mask_dir", "moge_out_dir", "hunyuan_hoi_mesh_dir",
                                  "hamer_out_dir", "h2m_rt_dir", "aligned_mano_dir"]}
    for p in dirs.values():
        p.mkdir(parents=True)
    idxs = [3, 10, 11, 25, 40]
    for i in idxs:
        Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(dirs["cropped_obj_img_dir"] / f"{i:04d}_cropped_hoi_1.png")
        m = np.full((8, 8), 255, np.uint8)
        if i == 25:
            m[:] = 0                                             # empty hand mask: skipped (RUN:232-236)
        Image.fromarray(m).save(dirs["mask_dir"] / f"{i:04d}_cropped_hand_mask.png")
        Image.fromarray(np.full((8, 8), 255, np.uint8)).save(dirs["mask_dir"] / f"{i:04d}_cropped_obj_mask.png")
        (dirs["moge_out_dir"] / f"{i:04d}_cropped_hoi").mkdir()
        (dirs["moge_out_dir"] / f"{i:04d}_cropped_hoi" / "fov.json").write_text('{"fov_x": 60.0}')
    script = tmp_path / "w.py"
    script.write_text(DRIVER_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FOHO_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(root)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    seen = {}
    for o in outs:
        for l in o.splitlines():
            if l.startswith("SEEN"):
                seen[int(l.split()[1])] = eval(l.split(None, 2)[2])
    assert seen == {0: [3, 11, 40], 1: [10]}                    # sorted list [3,10,11,25,40] dealt round-robin; 25 skipped
    assert "Skipping 0025 due to empty mask" in outs[1]
    assert sum("Batch metrics:" in o for o in outs) == 1        # rank 0 only
    import json
    tot = json.loads([l for l in outs[0].splitlines() if l.startswith("TOTALS")][0][len("TOTALS "):])
    assert tot["world_size"] == 2 and tot["n_images"] == 4 and tot["n_steps"] == 3000
    assert tot["sum_total_loss"] == 3 + 11 + 40 + 10 and tot["sum_wall_ms"] > 0
    for i in [3, 10, 11, 40]:
        assert (root / "guidance_out_dir" / f"{i:04d}_obj.ply").exists()


def test_bench_traffic_tables_read_the_committed_profiles():
    """bench.py attaches HBM traffic to its record from the committed rocprofv3 PMC summaries: the guide's 2 x FETCH_SIZE +
    WRITE_SIZE and, beside it, the bytes counted by request size class.  Both tables must parse, cover the six kernels of the
    step for the benchmark workload, agree with each other within 25 % (they are separate profiler runs) and stay silent for a
    workload nobody profiled."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("foho_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for wl in (("20k", 512, 1), ("20k", 512, 8)):
        pmc, src = bench.pmc_table(wl)
        ea, ea_src = bench.ea_table(wl)
        assert src and ea_src and os.path.exists(os.path.join(root, src)) and os.path.exists(os.path.join(root, ea_src))
        for k in bench.STEP_KERNELS:
            assert pmc[k] > 0 and ea[k] > 0
        tot_pmc, tot_ea = sum(pmc[k] for k in bench.STEP_KERNELS), sum(ea[k] for k in bench.STEP_KERNELS)
        assert abs(tot_pmc - tot_ea) <= 0.25 * tot_pmc, (tot_pmc, tot_ea)
    assert bench.pmc_table(("40k", 512, 1)) == ({}, None) and bench.ea_table(("ico4", 64, 1)) == ({}, None)


def test_hy3dgen_layout_check_script_says_so_when_hy3dgen_is_absent():
    """scripts/check_hy3dgen_layout.py -- the real-module readiness check a maintainer runs where Hunyuan3D-2 is installed -- exits 0
    with a plain message on a machine without hy3dgen (this one) instead of failing."""
    import importlib.util
    import subprocess
    import sys
    if importlib.util.find_spec("hy3dgen") is not None:
        pytest.skip("hy3dgen is installed here: run the script itself")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_hy3dgen_layout.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "hy3dgen not installed" in r.stdout, r.stdout[-2000:]
