"""In-kernel time stamps of the bench scene at iteration 25 (needs `make -C followmyhold_amd/csrc STAMPS=1`)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from followmyhold_amd import _lib as L
L.SO_PATH = os.path.join(ROOT, "followmyhold_amd", "libfoho_hip_stamps.so")
from followmyhold_amd import engine as E, synthetic
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0, crop=os.environ.get("CROP"))
gb = E.GuidanceBatch([sc]); cfgu, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
for _ in range(25): gb.step(cfgu)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 1024)()
graph = gb.capture(cfgu)
for rep in range(4):
    gb.lib.foho_debug_clear()
    if rep < 2:
        gb.step(cfgu)
    else:
        graph.replay()
    torch.cuda.synchronize()
    gb.lib.foho_debug_stamps(out)
    a = np.array(out[:], dtype=np.int64)
    d = lambda i, j: (a[j] - a[i]) / 100.0
    print("loss  blk0: slots %.2f loop %.2f blocksum %.2f publish+ticket %.2f | last blk: since blk0 start %.2f, finalize %.2f" % (
        d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(0, 5), d(5, 6)))
    print("pixbwd mid tile: loads+or %.2f body %.2f barrier %.2f flush %.2f" % (d(20, 21), d(21, 22), d(22, 23), d(23, 24)))
    print("vbwd  mid blk: body %.2f blocksum %.2f prefetch+fence+ticket %.2f | last blk: since mid start %.2f rows %.2f blocksum %.2f rest %.2f" % (
        d(10, 11), d(11, 12), d(12, 13), d(10, 14), d(14, 15), d(15, 16), d(16, 17)))
    print("vbwd  mid blk body: offsets %.2f pair records + gathers %.2f pair arithmetic %.2f barrier %.2f per-vertex sums %.2f per-vertex rest %.2f" % (
        d(10, 18), d(18, 19), d(19, 45), d(45, 46), d(46, 47), d(47, 11)))
    print("vbwd  hand blk 0 per-vertex: projection %.2f contact %.2f keypoints %.2f" % (d(40, 41), d(41, 42), d(42, 43)))
    print("xform blk 0 deferred prologue: prefetch + pending %.2f sums ready %.2f gradients / centre path || loss assembly %.2f barrier %.2f Adam + stores %.2f rest %.2f | total %.2f" % (
        d(95, 90), d(90, 91), d(91, 92), d(92, 93), d(93, 94), d(94, 96), d(95, 96)))
    print("gaps (%s): loss end -> pixbwd mid tile start %.2f | pixbwd mid tile end -> vbwd mid start %.2f | loss blk0 start -> vbwd last end %.2f" % (
        "eager" if rep < 2 else "graph", d(6, 20), d(24, 10), d(0, 17)))
    print("resolve centre tile (render 1): flags %.2f key %.2f face verts %.2f eval %.2f (sil planes) %.2f colour gather %.2f stores+shade %.2f barrier %.2f reduce+atomics %.2f | total %.2f" % (
        d(70, 71), d(71, 72), d(72, 73), d(73, 74), d(74, 75), d(75, 76), d(76, 77), d(77, 78), d(78, 79), d(70, 79)))
    for nm, o in (("hand", 50), ("obj", 60)):
        print("raster %s blk setup: ids %.2f ndc %.2f math+stores issued %.2f scan %.2f" % (nm, d(o, o + 6), d(o + 6, o + 7), d(o + 7, o + 9), d(o + 9, o + 1)))
        print("raster %s blk setup detail: stores issued %.2f cull+flags %.2f pix range %.2f tile marks %.2f" % (nm, d(o + 7, o + 30), d(o + 30, o + 31), d(o + 31, o + 32), d(o + 32, o + 9)))
        print("raster %s blk: setup %.2f barrier %.2f enumerate %.2f barrier %.2f evaluate %.2f (T=%d candidates)" % (
            nm, d(o, o + 1), d(o + 1, o + 2), d(o + 2, o + 3), d(o + 3, o + 4), d(o + 4, o + 5), a[o + 8]))
