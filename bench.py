#!/usr/bin/env python
"""bench.py -- guidance-steps/sec of the HIP hot path on MI355X (BASELINE.json metric).

A "step" is one iteration of the reference's joint loop (pipelines.py:1480-1601) for one image: transforms,
hand render, hand+object render with silhouette, keypoints, nearest-neighbour contact, edge loss, 65^3
intersection count, all losses, backward and the AdamW update.  Workload at every N: configs[1] of
BASELINE.json -- one synthetic 512x512 frame per GPU, 778-vertex hand + 10 242-vertex / 20 480-face object
(image-sharded: rank r owns its own frame, no data-path collective; one RCCL all-reduce of the metrics
vector at the end of the batch).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel (hipEvent-timed on the launch
stream), `cpu_baseline` the CPU oracle ("port" of the reference path) on a bounded sample of the same
workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW")


def algorithmic_bytes(H, W, Vh, Vo, Fh, Fo):
    """B_step = 206*H*W + 96*(Vh+Vo) + 48*(Fh+Fo) + 12*Fo bytes per guidance step."""
    return 206 * H * W + 96 * (Vh + Vo) + 48 * (Fh + Fo) + 12 * Fo


# Share of B_step that each kernel of the step touches algorithmically (DESIGN.md "Kernels"); per pixel P,
# per vertex V, per face F, for the two live renders of phase C.
def kernel_bytes(name, H, W, Vh, Vo, Fh, Fo):
    P, V, F = H * W, Vh + Vo, Fh + Fo
    table = {
        "k_xform": 36 * V,
        # vertex normals, KNN, keypoints, edge loss; scatter rasteriser reads the faces' NDC vertices (hand faces feed
        # both renders); inside test's face pass (ids + vertices)
        "k_stage2": 48 * V + 60 * F + 12 * Fo + 36 * (Fh + F) + 24 * F,
        "k_resolve": 2 * 16 * P + 12 * V,                           # write the 16 B/px G-buffer of 2 renders, read normals
        "k_loss": (2 * 16 + 17) * P,                                # read the G-buffers of 2 renders + the targets (12+4+1 B/px) once
        "k_pix_bwd": 2 * (16 + 17) * P + 48 * (Fh + F),             # read G-buffer + targets, accumulate 48 B/face x 2 renders
        "k_vert_bwd": 108 * V + 12 * F,
        "k_final": 0,
    }
    return table.get(name)


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_rocprofv3_pmc_fetch_write_b1.csv; FETCH_SIZE / WRITE_SIZE are in KiB and FETCH_SIZE reads half of a
    wide coalesced stream on gfx950 -- MI355X_MICROARCH.md "HBM" -- hence 2 x FETCH + WRITE)."""
    path = os.path.join(ROOT, "profiles", "r01_rocprofv3_pmc_fetch_write_b1.csv")
    if not os.path.exists(path):
        return None
    vals = {}
    for line in open(path).read().splitlines()[1:]:
        k, cn, _, mean = line.split(",")
        if k == kernel:
            vals[cn] = float(mean)
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--images-per-gpu", type=int, default=1)
    ap.add_argument("--streams", type=int, default=0, help="independent image groups, one HIP stream + hipGraph each "
                    "(0 = auto: min(4, images per GPU))")
    ap.add_argument("--obj", default="20k", choices=["ico4", "20k", "40k"])
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--joint-graph", action="store_true", help="one hipGraph with a branch per stream instead of one graph "
                    "per stream (measured slower on ROCm 7.2)")
    ap.add_argument("--steps-per-graph", type=int, default=0, help="cap on the iterations captured per hipGraph (0 = auto)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=30)
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # FOHO_BENCH_BACKEND=gloo lets the N>1 code path be exercised with several ranks on ONE GPU (development aid);
    # the driver's multi-GPU runs use RCCL ("nccl") with one rank per GPU.
    backend = os.environ.get("FOHO_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from followmyhold_amd import engine as E
    from followmyhold_amd import synthetic
    from followmyhold_amd import sharding

    H = W = args.size
    ipg = args.images_per_gpu
    render_fn = E.hip_render_fn(dev)
    # image-sharded: global image index = rank * ipg + j (seed per image)
    scenes = [synthetic.build_scene(render_fn, obj_kind=args.obj, H=H, W=W, seed=rank * ipg + j) for j in range(ipg)]
    # measured (8 images: 1 / 2 / 4 / 8 streams = 38 / 48 / 58 / 40 k steps/s): up to four independent groups overlap well
    n_streams = args.streams if args.streams > 0 else min(4, ipg)
    group = E.GuidanceGroup(scenes, n_streams, device=dev)
    gb = group.batches[0]
    cfg, _ = E.phase_cfg("C", denoise_i=19, do_update=True)
    # iterations per hipGraph: the reference's inner loop is 50 iterations per denoising step, and nothing in the step
    # needs the host, so a slice of that loop is ONE graph replay (no host work between iterations)
    spg = 1
    if not args.no_graph and not args.joint_graph:
        for cand in (50, 25, 10, 5):
            if args.steps % cand == 0:
                spg = cand
                break
        spg = min(spg, args.steps_per_graph) if args.steps_per_graph > 0 else spg
    if not args.no_graph:
        group.capture(cfg, joint=args.joint_graph, steps_per_graph=spg)
    ident = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0] * 2, dtype=torch.float32, device=dev)  # PL:1207-1215

    def new_denoise_step():
        """Every CFG.joint_iters = 50 iterations the reference starts a new denoising step with a fresh AdamW
        (PL:1478).  The pose parameters are put back to the scene's start point as well, so that the measured
        workload stays the configs[1] scene instead of whatever the synthetic optimisation drifts to (with the
        reference's learning rates the synthetic object shrinks away after ~150 iterations, which would make the
        rasteriser's job easier than the benchmark claims)."""
        group.restart(ident)

    def run_steps(n):
        done = 0
        while done < n:
            new_denoise_step()                  # every 50 iterations
            k = min(50, n - done)
            group.run(cfg, k)
            done += k

    run_steps(50)            # setup: let clocks / caches settle before the counted warm-up
    torch.cuda.synchronize(dev)

    run_steps(args.warmup)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    # end-of-batch metrics all-reduce (the only collective of the path)
    metrics = sum(sharding.local_metrics(g, n_steps=args.steps, wall_ms=dt * 1e3 if i == 0 else 0.0)
                  for i, g in enumerate(group.batches))
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        metrics = sharding.all_reduce_metrics(metrics, dist)
    dt = float(tmax.item())
    for g in group.batches:
        g.raise_on_flags()

    m0 = gb.meta[0]
    value = world * ipg * args.steps / dt
    out = {
        "metric": "guidance-steps/sec (512x512, 778+20k verts)", "value": value, "unit": "guidance-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[1]: single {H}x{W} synthetic frame per GPU, {m0['Vh']}-vert hand + "
                               f"{m0['Vo']}-vert/{m0['Fo']}-face object, joint guidance step (phase C)",
                   "images_per_gpu": ipg, "global_images": world * ipg, "parallelism": f"image-sharded x{world}",
                   "hip_graph": not args.no_graph, "steps_per_graph": spg, "streams": len(group.batches),
                   "restart_every": 50},
        "final_loss_mean": float(metrics[2] / max(metrics[0], 1.0)), "nan_images": int(metrics[-2]),
    }

    if rank == 0:
        # ---- roofline of the dominant kernel: hipEvents around every launch, averaged over the timed step count
        acc = {}
        nprof = 20
        run_steps(25)            # profile in the middle of a 50-iteration window
        torch.cuda.synchronize(dev)
        cfg_frozen, _ = E.phase_cfg("C", denoise_i=19, do_update=False)
        for _ in range(nprof):
            for k, v in gb.step_profiled(cfg_frozen).items():
                acc[k] = acc.get(k, 0.0) + v / nprof
        dom = max(acc, key=lambda k: acc[k])
        kb = kernel_bytes(dom, H, W, m0["Vh"], m0["Vo"], m0["Fh"], m0["Fo"])
        bstep = algorithmic_bytes(H, W, m0["Vh"], m0["Vo"], m0["Fh"], m0["Fo"])
        if kb is None:
            kb = bstep
        ipb = gb.B               # images in the profiled batch (the first stream's)
        achieved = kb * ipb / (acc[dom] * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(dom) if ipg == 1 else None,
                           "traffic_source": "profiles/r01_rocprofv3_pmc_fetch_write_b1.csv (2*FETCH_SIZE+WRITE_SIZE, KiB)",
                           "kernel_ms": acc[dom],
                           "algorithmic_bytes_per_launch": kb * ipb}
        out["kernel_ms"] = {k: round(v, 5) for k, v in acc.items()}
        out["kernels"] = {}
        for k, v in acc.items():           # every launch of the step against the HBM roofline (algorithmic bytes / duration)
            kbk = kernel_bytes(k, H, W, m0["Vh"], m0["Vo"], m0["Fh"], m0["Fo"]) or 0
            gbs = kbk * ipb / (v * 1e-3) / 1e9
            out["kernels"][k] = {"ms": round(v, 5), "algorithmic_MB": round(kbk * ipb / 1e6, 3), "GBs": round(gbs, 1),
                                 "frac": round(gbs / HBM_PEAK_GBS, 4)}
        out["step_hbm_GBs"] = bstep * value / world / 1e9  # whole-step algorithmic bytes x steps/s per GPU
        out["step_roofline_frac"] = out["step_hbm_GBs"] / HBM_PEAK_GBS

        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scenes[0], args.cpu_steps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(scene, n_steps):
    """The CPU oracle (a port of the reference's PyTorch-CPU path; the reference itself cannot run without
    pytorch3d/kaolin) timed on this box's host cores on the SAME scene: 1 warm-up + n_steps joint steps."""
    import torch
    from oracle import clib
    from oracle import step_ref as S
    sc = {k: (torch.from_numpy(v) if hasattr(v, "dtype") and not isinstance(v, torch.Tensor) else v) for k, v in scene.items()}
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))  # the oracle stops scaling (and oversubscribes) beyond a few dozen threads
    torch.set_num_threads(cores)
    clib.set_threads(cores)
    st = S.JointStepper(sc, S.make_params(), denoise_i=19)
    t0 = time.perf_counter()
    st.step()
    warm = time.perf_counter() - t0
    n_steps = max(1, min(n_steps, int(25.0 / max(warm, 1e-3))))  # bound the sample to ~25 s of CPU work
    t0 = time.perf_counter()
    for _ in range(n_steps):
        st.step()
    dt = time.perf_counter() - t0
    return {"value": n_steps / dt, "unit": "guidance-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_steps} joint steps (after 1 warm-up) of the same 512x512 / 20k-face scene, oracle/step_ref.py "
                      f"with OpenMP C rasteriser + torch CPU autograd"}


if __name__ == "__main__":
    main()
