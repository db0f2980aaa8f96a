"""Workload runner for rocprofv3: joint guidance steps of a synthetic scene, nothing else.
    python scripts/run_steps.py [--crop hoi] [--images 1] [--streams 1] [--steps 500] [--obj 20k] [--eager]
--crop hoi = the reference's crop regime (synthetic.hoi_crop); default = the 60-degree benchmark scene.  Prints steps/s."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--crop", default=None)
ap.add_argument("--fov", type=float, default=60.0, help="field of view of the synthetic camera when --crop is not given (22: the round-3 close-up)")
ap.add_argument("--images", type=int, default=1)
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--obj", default="20k")
ap.add_argument("--eager", action="store_true", help="plain launches instead of 50-iteration hipGraph replays (kernel names in every trace)")
ap.add_argument("--listed-cap", type=int, default=0)
a = ap.parse_args()
rf = E.hip_render_fn("cuda")
scenes = [synthetic.build_scene(rf, obj_kind=a.obj, H=512, W=512, seed=100 + j, crop=a.crop, fov=a.fov) for j in range(a.images)]
group = E.GuidanceGroup(scenes, a.streams, device="cuda")
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
cfg.listed_cap = a.listed_cap
if not a.eager:
    group.capture(cfg, steps_per_graph=50)
ident = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0] * 2, dtype=torch.float32, device="cuda")
best = 0.0
for rep in range(3):
    done = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while done < a.steps:
        group.restart(ident)
        group.run(cfg, 50)
        done += 50
    torch.cuda.synchronize()
    best = max(best, a.images * done / (time.perf_counter() - t0))
for g in group.batches:
    g.raise_on_flags()
gb = group.batches[0]
p2f = gb.region("p2f", torch.int32, (2, gb.B, 512 * 512))[1, 0]
print(f"crop={a.crop} images={a.images} streams={a.streams}: {best / 1e3:.2f} k steps/s; fov {scenes[0]['fov']:.1f}; hit pixels {int((p2f >= 0).sum())}", flush=True)
