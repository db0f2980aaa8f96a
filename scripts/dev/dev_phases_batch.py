"""Per-phase cost of an image-iteration with images in flight (4 streams x n images, 50-iteration hipGraphs): where the 750
iterations of a job (200 A + 100 B + 450 C) spend the GPU when 16 / 32 images run together.  Usage: dev_phases_batch.py [n_img ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic

counts = [int(a) for a in sys.argv[1:]] or [16]
rf = E.hip_render_fn("cuda")
ident = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0] * 2, dtype=torch.float32, device="cuda")
for n_img in counts:
    scenes = [synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, seed=100 + j) for j in range(n_img)]
    tot = 0.0
    for phase, iters in (("A", 200), ("B", 100), ("C", 450)):
        group = E.GuidanceGroup(scenes, 4, device="cuda")
        cfg, nr = E.phase_cfg(phase, denoise_i=19, do_update=True)
        for gb in group.batches:
            gb.set_n_renders(nr)
        group.capture(cfg, steps_per_graph=50)
        ts = []
        for rep in range(6):
            group.restart(ident)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            group.run(cfg, 100)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 100)
        t = min(ts[1:])
        tot += t * iters
        print(f"{n_img} images, phase {phase}: {t*1e6:7.1f} us per batch step = {t*1e6/n_img:5.2f} us per image-iteration "
              f"({n_img/t/1e3:6.1f} k image-steps/s); x{iters} = {t*iters*1e3/n_img:5.2f} ms per image", flush=True)
        del group
    print(f"{n_img} images: schedule total {tot*1e3/n_img:.2f} ms per image ({n_img/tot:.0f} images/s)", flush=True)
