"""Drop-in proof against the reference checkout (build container only: /root/reference does not travel to the GPU box).

Two ways the hot-path modules meet the reference's `foho` package:
  * path overlay -- this repository before the checkout's src/ on sys.path: `pkgutil.extend_path` keeps every module
    this repository does not provide (foho.main, foho.configs.pipeline, foho.utils.runner, ...) resolving to the
    checkout, while foho.guidance.run / foho.alignment.* resolve here;
  * scripts/install_overlay.py -- for the UNCHANGED orchestrator, which puts its own src/ first on every stage's
    PYTHONPATH (src/foho/main.py:19-23): the stage modules are placed inside the checkout.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "src", "foho", "main.py")),
                                reason="reference checkout not mounted")

CHECK_OVERLAY = r"""
import os, shlex, sys
repo, ref = sys.argv[1], sys.argv[2]
import foho, foho.guidance.run as G, foho.alignment.h2m as H2M, foho.alignment.mano as MANO, foho.alignment.mesh_align as MA
for m in (foho, G, H2M, MANO, MA):
    assert os.path.abspath(m.__file__).startswith(repo + os.sep), m.__file__
import foho.main as MAIN, foho.utils.runner as RUNNER, foho.configs as C, foho.configs.pipeline as CP
for m in (MAIN, RUNNER, CP):
    assert os.path.abspath(m.__file__).startswith(ref + os.sep), m.__file__
from foho.configs import PipelineConfig, load_config, OptimizationConfig, third_party_root   # main.py:11, run.py:28-29
assert load_config.__module__ == "foho.configs.pipeline" and callable(RUNNER.run_in_conda)
import foho.hand, foho.preprocess, foho.geometry                     # the reference's other sub-packages stay visible
assert os.path.abspath(foho.hand.__file__).startswith(ref + os.sep)
# the orchestrator's own command lines (main.py:229-278) parse with this repository's stage parsers
names = ["project_root", "cropped_obj_img_dir", "mask_dir", "moge_out_dir", "hunyuan_hoi_mesh_dir", "hamer_out_dir",
         "h2m_rt_dir", "aligned_mano_dir", "guidance_out_dir"]
cmd = MAIN._cmd("foho.guidance.run", {n: f"/data/my dir/{n}" for n in names})
parts = shlex.split(cmd)
assert parts[:3] == ["python3", "-m", "foho.guidance.run"]
a = G.build_parser().parse_args(parts[3:])
assert all(getattr(a, n) == f"/data/my dir/{n}" for n in names) and a.task_list_file is None
for mod, M in (("foho.alignment.h2m", H2M), ("foho.alignment.mano", MANO)):
    parts = shlex.split(MAIN._cmd(mod, {n: "/x/" + n for n in M.FLAGS}))
    a = H2M.build_parser(*M.FLAGS).parse_args(parts[3:])
    assert all(getattr(a, n) == "/x/" + n for n in M.FLAGS)
print("OVERLAY OK")
"""


def _run(code, args, env, cwd=None):
    r = subprocess.run([sys.executable, "-c", code] + args, env=env, cwd=cwd, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    return r.stdout


def _clean_env(pythonpath):
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PYTHONPATH"] = pythonpath
    return env


def test_path_overlay_keeps_the_reference_packages_visible():
    out = _run(CHECK_OVERLAY, [ROOT, REF], _clean_env(ROOT + os.pathsep + os.path.join(REF, "src")), cwd="/tmp")
    assert "OVERLAY OK" in out


CHECK_INSTALLED = r"""
import os, sys
co, repo = sys.argv[1], sys.argv[2]
import foho.guidance.run as G, foho.alignment.mesh_align as MA, foho.main as MAIN, followmyhold_amd
assert os.path.abspath(G.__file__).startswith(co + os.sep) and hasattr(G, "build_parser") and hasattr(G, "_dist_setup")
assert os.path.abspath(MAIN.__file__).startswith(co + os.sep)
assert os.path.realpath(followmyhold_amd.__file__).startswith(repo + os.sep)
from foho.configs import OptimizationConfig
assert OptimizationConfig.__module__ == "foho.configs.guid_config"      # the checkout's own config drives the kernels
assert MAIN._foho_src() == os.path.join(co, "src")
print("INSTALLED OK")
"""


def test_install_overlay_into_a_checkout(tmp_path):
    """The unchanged orchestrator's stage environment (PYTHONPATH = <checkout>/src only) picks up the HIP-backed stage
    modules after scripts/install_overlay.py; --uninstall restores the checkout byte for byte."""
    co = tmp_path / "FollowMyHold"
    shutil.copytree(os.path.join(REF, "src", "foho"), co / "src" / "foho", ignore=shutil.ignore_patterns("__pycache__"))
    inst = [sys.executable, os.path.join(ROOT, "scripts", "install_overlay.py"), "--foho-root", str(co)]
    subprocess.run(inst, check=True, stdout=subprocess.PIPE)
    env = _clean_env(str(co / "src"))                                    # main.py:19-23
    assert "INSTALLED OK" in _run(CHECK_INSTALLED, [str(co), ROOT], env, cwd=str(tmp_path))
    # the stage command exactly as main.py:259-278 spells it, on an empty (but existing) set of directories
    dirs = {}
    for n in ["cropped_obj_img_dir", "mask_dir", "moge_out_dir", "hunyuan_hoi_mesh_dir", "hamer_out_dir", "h2m_rt_dir",
              "aligned_mano_dir", "guidance_out_dir"]:
        dirs[n] = tmp_path / "data" / n
        dirs[n].mkdir(parents=True, exist_ok=True)
    cmd = ["python3", "-m", "foho.guidance.run", "--project_root", str(co)]
    for n, p in dirs.items():
        cmd += [f"--{n}", str(p)]
    r = subprocess.run(cmd, env=env, cwd=str(co), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "Finished processing all images" in r.stdout and "Batch metrics:" in r.stdout
    r = subprocess.run(["python3", "-m", "foho.alignment.h2m", "--hunyuan_mesh_dir", str(dirs["hunyuan_hoi_mesh_dir"]),
                        "--moge_out_dir", str(dirs["moge_out_dir"]), "--h2m_rt_dir", str(dirs["h2m_rt_dir"])],
                       env=env, cwd=str(co), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "No Hunyuan HOI meshes found" in r.stdout, r.stdout[-2000:]
    subprocess.run(inst + ["--uninstall"], check=True, stdout=subprocess.PIPE)
    for rel in ["guidance/run.py", "alignment/h2m.py", "alignment/mano.py", "alignment/mesh_align.py"]:
        a = open(os.path.join(REF, "src", "foho", rel), "rb").read()
        assert open(co / "src" / "foho" / rel, "rb").read() == a
    assert not os.path.lexists(co / "src" / "followmyhold_amd")
    assert not [f for f in os.listdir(co / "src" / "foho" / "guidance") if f.endswith(".reference")]
