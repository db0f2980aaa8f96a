"""Gaps on the GPU timeline of one per-image job: run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python
scripts/dev/dev_job_trace.py run`, then `python scripts/dev/dev_job_trace.py report DIR`."""
import os, sys, csv, glob
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import torch, time
    from followmyhold_amd import engine as E, synthetic, inputs
    sc = synthetic.build_scene(E.hip_render_fn("cuda"), obj_kind="20k", H=512, W=512, seed=0)
    inputs.run_mesh_guidance([sc]); torch.cuda.synchronize()
    time.sleep(0.5)                                  # a hole in the trace marks the start of the measured job
    t0 = time.perf_counter(); inputs.run_mesh_guidance([sc]); torch.cuda.synchronize()
    print("job %.1f ms" % ((time.perf_counter() - t0) * 1e3))
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f))))
    # the measured job starts after the largest idle hole
    holes = [(rows[i + 1][0] - rows[i][1], i + 1) for i in range(len(rows) - 1)]
    start = max(holes)[1]
    job = rows[start:]
    span = (job[-1][1] - job[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in job) / 1e6
    print("kernels %d span %.2f ms busy %.2f ms idle %.2f ms" % (len(job), span, busy, span - busy))
    gaps = sorted(((job[i + 1][0] - job[i][1]) / 1e3, job[i][2], job[i + 1][2]) for i in range(len(job) - 1))[::-1]
    short = lambda n: n.replace("void at::native::vectorized_elementwise_kernel<4, at::native::", "").replace("void at::native::vectorized_elementwise_kernel<16, at::native::", "")[:28]
    idx = sorted(range(len(job) - 1), key=lambda i: job[i][1] - job[i + 1][0])[:8]
    for i in sorted(idx):
        print("gap %.0f us at +%.2f ms: ... %s || %s ..." % ((job[i + 1][0] - job[i][1]) / 1e3, (job[i][1] - job[0][0]) / 1e6,
              " ".join(short(job[k][2]) for k in range(max(0, i - 5), i + 1)), " ".join(short(job[k][2]) for k in range(i + 1, min(len(job), i + 7)))))
    print("largest gaps (us, after kernel, before kernel):")
    for g in gaps[:25]: print("  %.1f  %s -> %s" % g)
    import collections
    hist = collections.Counter()
    for g, a, b in gaps: hist[(a, b)] += g
    print("gap time by kernel pair (ms):")
    for k, v in hist.most_common(12): print("  %.2f  %s -> %s" % (v / 1e3, k[0], k[1]))
    byk = collections.Counter()
    for s, e, n in job: byk[n] += e - s
    print("busy by kernel (ms):", {k: round(v / 1e6, 2) for k, v in byk.most_common(10)})
