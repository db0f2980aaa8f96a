"""Host cost of hipGraph capture and replay against iterations per graph (run on the GPU box): one 4-image slot, 750 iterations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic
rf = E.hip_render_fn("cuda")
scs = [synthetic.build_scene(rf, obj_kind="20k", H=512, W=512, seed=s) for s in range(4)]
gb = E.GuidanceBatch(scs, n_renders=2, obj_capacity=(12288, 24576)); gb.load_scenes(scs)
cfg, _ = E.phase_cfg("C", E.OptimizationConfig(), denoise_i=15, do_update=True)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for spg in (10, 25, 50, 10):
        t0 = time.perf_counter(); g = gb.capture(cfg, steps_per_graph=spg); tc = time.perf_counter() - t0
        g.replay(); st.synchronize()
        for rep in range(2):
            t0 = time.perf_counter()
            for _ in range(750 // spg): g.replay()
            th = time.perf_counter() - t0
            st.synchronize(); tt = time.perf_counter() - t0
            print(f"spg {spg}: capture {tc*1e3:.1f} ms; 750 iterations: host enqueue {th*1e3:.1f} ms ({th*1e6/(750//spg):.0f} us per replay), done after {tt*1e3:.1f} ms", flush=True)
