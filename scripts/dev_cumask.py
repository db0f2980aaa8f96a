"""Experiment: run the one-image step on a CU-masked stream (hipExtStreamCreateWithCUMask) -- all CUs, one XCD
(every 8th bit), one XCD (first 32 bits), two XCDs -- and report the step time."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import engine as E, synthetic
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << (b - 32 * w) for b in bits if 32 * w <= b < 32 * (w + 1)) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
masks = {"all 256": list(range(256)), "every 8th bit (32 CUs)": list(range(0, 256, 8)), "bits 0-31 (32 CUs)": list(range(32)),
         "every 4th bit (64 CUs)": list(range(0, 256, 4)), "bits 0-127 (128 CUs)": list(range(128))}
for name, bits in masks.items():
    st = masked_stream(bits)
    gb = E.GuidanceBatch([sc])
    with torch.cuda.stream(st):
        for _ in range(20): gb.step(cfg)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): gb.step(cfg)
        st.synchronize()
        dt = (time.perf_counter() - t0) / 200
    print("%-28s eager step %.1f us" % (name, dt * 1e6), flush=True)
