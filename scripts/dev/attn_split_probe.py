"""What would splitting the KEYS of the ShapeVAE self-attention forward (192 workgroups at 16 heads x 3072 queries: under-filled) buy?  The
main loop of a k-way split is k_geo_attn on (k x 3072 queries) x (3072 / k keys): timed here through foho_geo_attention.
python scripts/dev/attn_split_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda", 0)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
H, W = 16, 1024
for k in (1, 2, 3, 4):
    M, Lk = 3072 * k, 3072 // k
    Q = torch.randn(M, W, device=dev).half() * 0.2
    KV = torch.randn(Lk, 2 * W, device=dev).half()
    Vt = torch.empty(W * Lk, device=dev, dtype=torch.float16)
    O = torch.empty(M, W, device=dev, dtype=torch.float16)
    for _ in range(3):
        lib.foho_geo_attention(P(Q), P(KV), P(Vt), P(O), M, Lk, H, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.foho_geo_attention(P(Q), P(KV), P(Vt), P(O), M, Lk, H, st)
    e1.record(); torch.cuda.synchronize()
    print(f"{k}-way: {M} queries x {Lk} keys: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (pack_vt included)", flush=True)
