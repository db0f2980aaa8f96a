"""Experiment: N independent one-image guidance loops on N streams (one hipGraph each) vs one batched loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic
render = E.hip_render_fn("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scenes = [synthetic.build_scene(render, obj_kind="20k", H=512, W=512, seed=i) for i in range(N)]
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
def bench_streams(n, per=1):
    gbs = [E.GuidanceBatch(scenes[i * per:(i + 1) * per]) for i in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    graphs = []
    for gb, st in zip(gbs, streams):
        with torch.cuda.stream(st):
            graphs.append(gb.capture(cfg))
    torch.cuda.synchronize()
    def run(k):
        for _ in range(k):
            for g, st in zip(graphs, streams):
                with torch.cuda.stream(st):
                    g.replay()
    run(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return n * per * 100 / dt
for n, per in [(4, 2), (8, 1), (2, 8), (4, 4), (8, 2), (4, 8), (8, 4), (2, 16), (1, 32)]:
    if n * per <= N:
        print("streams %d x images/stream %d : %.0f steps/s" % (n, per, bench_streams(n, per)), flush=True)
