"""Drop-in module paths of the reference's `foho` package for the guidance/alignment hot path.

`python3 -m foho.guidance.run`, `python3 -m foho.alignment.h2m` and `python3 -m foho.alignment.mano` keep the
reference's flags (src/foho/guidance/run.py:264-289, src/foho/alignment/h2m.py:57-68, mano.py:46-57) so the
orchestrator src/foho/main.py:229-278 drives them unchanged; the arithmetic runs in libfoho_hip.so.

Overlay-safe: this package only holds the hot-path modules.  `pkgutil.extend_path` adds every other `foho/`
directory found further down `sys.path` (a FollowMyHold checkout's `src/foho`), so `foho.main`, `foho.hand`,
`foho.preprocess`, `foho.geometry`, `foho.utils`, `foho.configs.pipeline` ... keep resolving to the reference's files
while `foho.guidance.run` / `foho.alignment.*` resolve here.  For the unchanged orchestrator -- which puts its own
`src/` first on every stage's PYTHONPATH (src/foho/main.py:19-23) -- scripts/install_overlay.py places the same
modules into the checkout instead.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
