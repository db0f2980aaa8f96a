"""Drop-in module paths of the reference's `foho` package for the guidance/alignment hot path.

`python3 -m foho.guidance.run`, `python3 -m foho.alignment.h2m` and `python3 -m foho.alignment.mano` keep the
reference's flags (src/foho/guidance/run.py:264-289, src/foho/alignment/h2m.py:57-68, mano.py:46-57) so the
orchestrator src/foho/main.py:229-278 drives them unchanged; the arithmetic runs in libfoho_hip.so.
"""
