"""Which torch streams run concurrently?  HIP maps streams onto a few hardware queues; two streams on one queue serialise.
Pairwise test with a spin kernel, then the 8-image / 4-stream step on stream quadruples (run on the GPU box)."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda")
N = 12
streams = [torch.cuda.Stream(dev) for _ in range(N)]
print("stream handles", [hex(s.cuda_stream) for s in streams])
def spin(ss, cycles=4_000_000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in ss:
        with torch.cuda.stream(s): torch.cuda._sleep(cycles)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
spin(streams[:1]); one = min(spin(streams[:1]) for _ in range(3))
print("one spin: %.2f ms" % one)
rows = []
for i in range(N):
    rows.append(" ".join("%.1f" % (min(spin([streams[i], streams[j]]) for _ in range(2)) / one) if j != i else " - " for j in range(N)))
    print("stream %2d:" % i, rows[-1], flush=True)
print("all %d at once: %.2f x one" % (N, spin(streams) / one))
for k in (2, 3, 4, 5, 6, 8):
    print("first %d at once: %.2f x one" % (k, min(spin(streams[:k]) for _ in range(2)) / one))
