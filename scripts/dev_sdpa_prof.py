"""Ten forward + backward passes of followmyhold_amd.sdpa.attention at (1, 16, 3072, 64) -- for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from followmyhold_amd import sdpa
dev = torch.device("cuda", 0)
q, k, v = (torch.randn(1, 16, 3072, 64, device=dev, dtype=torch.float16, requires_grad=True) for _ in range(3))
go = torch.randn(1, 16, 3072, 64, device=dev, dtype=torch.float16)
for _ in range(10):
    for t in (q, k, v): t.grad = None
    sdpa.attention(q, k, v).backward(go)
torch.cuda.synchronize()
