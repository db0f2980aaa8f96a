"""Per-workgroup timeline of the phased GEMM (development build libfoho_hip_p8tl.so: make VARIANT=p8tl EXTRA=-DP8_TIMELINE): shader-clock
stamps {entry, first K tile landed, loop end, wave groups re-joined, exit} of every workgroup + the CU it ran on -> where a tile's
time outside its K loop goes, and the gap between two workgroups on one CU."""
import ctypes, math, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lib = ctypes.CDLL(os.path.join(ROOT, "followmyhold_amd", "libfoho_hip_p8tl.so"))
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, gelu) in ((49152, 4096, 1024, 0), (49152, 4096, 1024, 1), (49152, 1024, 4096, 0), (49152, 1024, 1024, 0)):
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).half(); b = torch.randn(N, device=dev)
    C = torch.empty(M, N, dtype=torch.float16, device=dev)
    for _ in range(3):
        lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, gelu, ctypes.c_float(1.0), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.foho_geo_gemm(P(A), P(W), P(b), None, P(C), M, N, K, gelu, ctypes.c_float(1.0), st); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (4096 * 8))()
    lib.foho_geo_p8_timeline(buf)
    ntiles = min(4096, (M // 256) * (N // 256))
    rows = [[buf[i * 8 + j] for j in range(8)] for i in range(ntiles)]
    rows = [r for r in rows if r[4]]
    seg = [[r[j + 1] - r[j] for j in range(4)] for r in rows]
    mean = [sum(s[j] for s in seg) / len(seg) for j in range(4)]
    tot = sum(mean)
    # per CU: sort by entry, gap = next entry - previous exit
    cu = collections.defaultdict(list)
    for r in rows:
        if r[4]: cu[(r[6] & 0xf, r[5] >> 8 & 0xff)].append((r[0], r[4]))      # (XCC_ID, se / sh / cu bits of HW_ID)
    gaps = []
    for k, v in cu.items():
        v.sort()
        gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    span = max(r[4] for r in rows) - min(r[0] for r in rows)
    gaps.sort()
    print(f"M={M} N={N} K={K} gelu={gelu}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {ntiles} tiles on {len(cu)} CUs; stamp span {span} cycles")
    print(f"  per workgroup (cycles): prologue {mean[0]:.0f} | K loop {mean[1]:.0f} ({mean[1] / (K // 64):.0f} per K tile) | re-join {mean[2]:.0f} | epilogue {mean[3]:.0f} | total {tot:.0f}")
    a = sum(r[7] >> 32 for r in rows) / len(rows); b_ = sum(r[7] & 0xffffffff for r in rows) / len(rows)
    print(f"  prologue: entry -> 14 DMA pieces issued {a:.0f} | wait for the first K tile {b_:.0f} | write vectors, barrier, first fragment reads {mean[0] - a - b_:.0f}")
    if gaps:
        print(f"  gap between workgroups on one CU: median {gaps[len(gaps) // 2]}, mean {sum(gaps) / len(gaps):.0f}, p90 {gaps[int(len(gaps) * 0.9)]} cycles ({len(gaps)} gaps)")
