#!/usr/bin/env python
"""Install / remove the MI355X hot path inside a FollowMyHold checkout, so that the UNCHANGED orchestrator
(`python -m foho.main --config ...`) runs its alignment and guidance stages on libfoho_hip.so.

Why an installer: src/foho/main.py:19-23 puts the checkout's own `src/` FIRST on the PYTHONPATH of every stage it
spawns (`python3 -m foho.alignment.h2m`, `... foho.alignment.mano`, `... foho.guidance.run`, main.py:229-278), so a
package that merely sits elsewhere on the path is never picked.  The four stage modules are therefore placed where the
orchestrator looks, and the kernels' package is linked next to them:

    <checkout>/src/foho/guidance/run.py          <- foho/guidance/run.py          (original kept as run.py.reference)
    <checkout>/src/foho/alignment/h2m.py         <- foho/alignment/h2m.py
    <checkout>/src/foho/alignment/mano.py        <- foho/alignment/mano.py
    <checkout>/src/foho/alignment/mesh_align.py  <- foho/alignment/mesh_align.py
    <checkout>/src/followmyhold_amd              -> <this repo>/followmyhold_amd  (symlink; --copy copies instead)

Everything else of the checkout (foho.main, foho.configs, foho.hand, foho.preprocess, foho.geometry, foho.utils, the
third_party trees) is left alone; the installed modules import `OptimizationConfig` / `third_party_root` from the
checkout's own foho.configs.

    python scripts/install_overlay.py --foho-root /path/to/FollowMyHold            # install
    python scripts/install_overlay.py --foho-root /path/to/FollowMyHold --uninstall
"""
import argparse
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODULES = ["guidance/run.py", "alignment/h2m.py", "alignment/mano.py", "alignment/mesh_align.py"]
BACKUP = ".reference"


def install(foho_root: str, copy: bool = False) -> None:
    src = os.path.join(foho_root, "src")
    pkg = os.path.join(src, "foho")
    if not os.path.isfile(os.path.join(pkg, "main.py")):
        raise SystemExit(f"{foho_root} does not look like a FollowMyHold checkout (no src/foho/main.py)")
    so = os.path.join(REPO, "followmyhold_amd", "libfoho_hip.so")
    if not os.path.exists(so):
        raise SystemExit("libfoho_hip.so is not built: run `python __graft_entry__.py` in this repository first")
    for rel in MODULES:
        dst = os.path.join(pkg, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        if os.path.exists(dst) and not os.path.exists(dst + BACKUP):
            shutil.move(dst, dst + BACKUP)
        shutil.copyfile(os.path.join(REPO, "foho", rel), dst)
        print("installed", dst)
    link = os.path.join(src, "followmyhold_amd")
    if os.path.islink(link) or os.path.exists(link):
        shutil.rmtree(link) if os.path.isdir(link) and not os.path.islink(link) else os.remove(link)
    if copy:
        shutil.copytree(os.path.join(REPO, "followmyhold_amd"), link, ignore=shutil.ignore_patterns("__pycache__", "csrc"))
    else:
        os.symlink(os.path.join(REPO, "followmyhold_amd"), link)
    print("linked" if not copy else "copied", link)


def uninstall(foho_root: str) -> None:
    src = os.path.join(foho_root, "src")
    pkg = os.path.join(src, "foho")
    for rel in MODULES:
        dst = os.path.join(pkg, rel)
        if os.path.exists(dst + BACKUP):
            shutil.move(dst + BACKUP, dst)
            print("restored", dst)
    link = os.path.join(src, "followmyhold_amd")
    if os.path.islink(link):
        os.remove(link)
    elif os.path.isdir(link):
        shutil.rmtree(link)


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--foho-root", required=True, help="root of the FollowMyHold checkout (holds src/foho/main.py)")
    ap.add_argument("--uninstall", action="store_true")
    ap.add_argument("--copy", action="store_true", help="copy followmyhold_amd instead of linking it")
    a = ap.parse_args(argv)
    (uninstall if a.uninstall else lambda r: install(r, a.copy))(os.path.abspath(a.foho_root))


if __name__ == "__main__":
    sys.exit(main())
