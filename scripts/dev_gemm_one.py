"""Ten launches of the geometry decoder's GEMM at one shape (for rocprofv3 --pmc): python scripts/dev_gemm_one.py M N K [flag]"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import _lib as L
lib = L.lib()
M, N, K = (int(x) for x in sys.argv[1:4])
flag = int(sys.argv[4]) if len(sys.argv) > 4 else 0
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).half(); b = torch.randn(N, device=dev)
C = torch.empty(M, N, dtype=torch.float16, device=dev)
for _ in range(10):
    lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, flag, ctypes.c_float(1.0), st)
torch.cuda.synchronize()
