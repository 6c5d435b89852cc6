#!/usr/bin/env python3
"""bench.py -- edge-traces/sec of the MI355X GP edge tracer (BASELINE.json metric).

One "step" = every rank traces its batch of independent 500x500 edges to completion
(device-resident GP loop: fit -> factor -> normals -> sample GEMM -> scoring -> KDE -> pixel
selection, then the final hyper-parameter fit), inputs (gradient image, inits) resident in HBM.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

README_KW = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}, noise_y=1, N_samples=1000,
                 score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
STAGES = ["fit_predict_cov", "factor", "normals", "sample_gemm", "score_topk", "curve_kde"]
# gpet_profile_stage ids of the single kernels of one iteration (include/gpet_hip.h)
# structured loop path (prior eigenbasis; what gpet_trace_iterate runs for on-grid batches): 120-123;
# generic path (per-stage API, off-grid observations, GPET_NO_STRUCT=1): 100-113
KERNEL_IDS_STRUCT = {120: "k_fit", 121: "k_struct_H", 122: "k_jacobi_lds", 123: "k_struct_rows"}
KERNEL_IDS_GENERIC = {100: "k_fit", 101: "k_predict", 102: "k_cov_mfma", 110: "k_pchol_reg", 111: "k_gram",
                      112: "k_jacobi_lds", 113: "k_factor_rows"}
KERNEL_IDS_COMMON = {130: "k_sample_gemm_mfma_r", 140: "k_score", 141: "k_topk", 150: "k_kde_prep",
                     151: "k_kde_fused"}  # (152 k_kde_normalise: stage API only; the loop normalises inside the pixel kernels)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X FP64 vector/matrix peak (spec)


def synth_image(N, seed):
    """Sinusoidal step edge + gaps + Gaussian noise (recipe of gpet_utils.py:163-253)."""
    from gaussian_process_edge_trace_amd import gpet_utils
    return gpet_utils.construct_test_img((N, N), 200 if N == 500 else int(0.4 * N), 4, 0.05, 'sinusoidal', 0.3,
                                         gaps=True, seed=seed)


def cpu_baseline(N, img_seed, n_traces, per_curve=True):
    """The CPU oracle (NumPy/SciPy port of the reference algorithm: LAPACK-SVD sampling, per-curve
    Simpson scoring, scipy L-BFGS-B final fit) timed on this host.  Checker code used as the timed
    baseline only.  Timed with 1 BLAS thread and with 16 (the reference is BLAS-thread sensitive,
    BASELINE.md section 2); the faster setting is reported with its thread count."""
    from oracle import gpet_oracle as orc
    from threadpoolctl import threadpool_limits
    img, edge = orc.synth_sinusoid_image(N, img_seed)
    grad = orc.comp_grad_img(img, orc.kernel_builder((11, 5)))
    init = edge[[0, -1], :][:, [1, 0]]
    best = None
    for threads in (1, min(16, usable_cpus())):
        with threadpool_limits(limits=threads):
            t0 = time.time()
            iters = []
            for k in range(n_traces):
                _, _, info = orc.trace(init, grad, per_curve=per_curve, seed=1 + k, **README_KW)
                iters.append(info["n_iter"])
            dt = time.time() - t0
        log("cpu baseline: %d BLAS thread(s): %.2f s per trace" % (threads, dt / n_traces))
        if best is None or dt < best[0]:
            best = (dt, threads, iters)
    dt, threads, iters = best
    return dict(value=n_traces / dt, unit="edge-traces/s", cores=int(threads), kind="port",
                sample="%d full trace(s) of the 500x500 README edge (RBF sf=75 l=20, S=1000, dx=5, pixel_thresh=5), "
                       "%s iterations, oracle/gpet_oracle.py: per-curve scoring loop + LAPACK SVD sampling + "
                       "scipy L-BFGS-B x13; best of 1 and 16 BLAS threads" % (n_traces, iters),
                seconds=dt)


def log(msg):
    print("[bench %.1fs] %s" % (time.time() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.time()


def usable_cpus():
    """CPUs this job may actually use: the smallest of the machine's count, the affinity mask and the cgroup quota
    (a container with `cpu.max = 1600000 100000` on a 256-thread host has 16)."""
    n = os.cpu_count() or 2
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(2, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--edges", type=int, default=1024,
                    help="independent edges per GPU and step (BASELINE config 4 is a batch of independent 500x500 edges; "
                         "measured on one MI355X: 256 edges 3.6 k traces/s, 512 4.0 k, 1024 4.2 k, 2048 4.2 k -- the loop's "
                         "kernels are latency-bound below ~4 workgroups per CU)")
    ap.add_argument("--size", type=int, default=500)
    ap.add_argument("--fit-workers", type=int, default=int(os.environ.get("GPET_FIT_WORKERS", "0")),
                    help="0: final fits on the GPU (batched LML kernel); >1: host worker processes instead")
    ap.add_argument("--lbfgs-workers", type=int, default=max(1, min(12, usable_cpus() - 2)),
                    help="worker processes advancing scipy's L-BFGS-B routine in lock step (final fits)")
    ap.add_argument("--pipeline-depth", type=int, default=3,
                    help="how many steps' converged fits may be in flight behind the device loops (batch objects = depth+1)")
    ap.add_argument("--concurrent-steps", action="store_true",
                    help="experiment: one driver thread per batch object (device loop + fits of a step back to back), "
                         "so that device loops of different batch objects overlap too (measured: slower, DESIGN.md)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="trace the steps strictly one after the other (default: the converged fits of step k overlap "
                         "the device loop of step k+1 on a second batch object / HIP stream)")
    ap.add_argument("--cpu-traces", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("warning: WORLD_SIZE=%d but --gpus=%d; using WORLD_SIZE" % (world, args.gpus), file=sys.stderr)
    # worker processes for the host-side final fits: forked BEFORE anything touches the GPU
    from gaussian_process_edge_trace_amd.gpet import make_fit_pool
    from gaussian_process_edge_trace_amd._lbfgsb_lockstep import LockstepFarm
    pool = make_fit_pool(args.fit_workers) if args.fit_workers > 1 else None
    # L-BFGS-B state machines of the final fits advance in worker processes (the objective runs on
    # the GPU); also created before HIP is initialised
    n_lbfgs = max(2, min(args.lbfgs_workers, (usable_cpus() - 2) // max(1, world))) if args.lbfgs_workers > 1 else 0
    pipeline = (not args.no_pipeline) and pool is None
    depth = max(1, args.pipeline_depth) if pipeline else 0
    n_farms = depth + 1 if pipeline else 1
    # one farm, one job slot per batch object: a fit running alone (the last of a run) gets every worker
    big_farm = LockstepFarm(n_lbfgs, slots=n_farms, pmax=max(16384, 13 * args.edges)) if (n_lbfgs > 1 and pool is None) else None
    farms = [big_farm.slot(i) if big_farm is not None else None for i in range(n_farms)]
    farm = farms[0]
    import torch
    dist = None
    # GPET_BENCH_BACKEND=gloo + GPET_BENCH_SHARE_GPU=1: rehearsal of the N>1 path on a one-GPU box
    # (ranks share device 0, collectives on CPU tensors); the real run uses nccl (= RCCL), one rank per GPU
    backend = os.environ.get("GPET_BENCH_BACKEND", "nccl")
    share_gpu = os.environ.get("GPET_BENCH_SHARE_GPU", "0") == "1"
    dev_index = 0 if share_gpu else local_rank
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    import gaussian_process_edge_trace_amd as pkg
    L = pkg._lib
    ctx = L.Context(dev_index)

    # ---- inputs: one shared gradient image, produced on rank 0's GPU, broadcast over RCCL/xGMI
    N = args.size
    img, truth = synth_image(N, 3)
    init = truth[[0, -1], :][:, [1, 0]]
    t_b = 0.0
    if world > 1:
        g = torch.empty((N, N), dtype=torch.float32, device=coll_dev)
        if rank == 0:
            g.copy_(torch.from_numpy(pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)))
        torch.cuda.synchronize()
        tb0 = time.time()
        dist.broadcast(g, src=0)
        torch.cuda.synchronize()
        t_b = time.time() - tb0
        grad = g.cpu().numpy()
    else:
        grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)

    E = args.edges
    seeds = [1 + rank * E + e for e in range(E)]  # independent edges: distinct RNG streams
    tracer = pkg.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx,
                                       fit_pool=pool, fit_farm=farm)
    tracers = [tracer]
    for d_ in range(depth):  # more batch objects, each with its own context (= HIP stream) and worker farm
        ctx_d = L.Context(dev_index)
        tracers.append(pkg.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx_d,
                                                 fit_farm=farms[d_ + 1]))

    def barrier():
        for tr_ in tracers:
            tr_._ctx.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    from concurrent.futures import ThreadPoolExecutor
    executor = ThreadPoolExecutor(max_workers=max(1, depth))
    drivers = [ThreadPoolExecutor(max_workers=1) for _ in tracers] if args.concurrent_steps else []

    fit_walls = []  # wall time of every step's converged fits (they run concurrently with later device loops)

    def run_steps(n_steps, timed):
        """n_steps passes of the hot path.  Pipelined: while the converged fits of step k run (host-driven
        lock-step L-BFGS-B + LML kernels on stream A) the device loop of step k+1 runs on stream B."""
        loop_s = fit_s = 0.0
        iters_, traces_, pending = [], None, []
        fit_walls.clear()
        if pipeline and args.concurrent_steps:
            def one_step(tr__):
                t_a = time.time()
                tr__.reset()
                it__ = tr__.run_loop()
                t_m = time.time()
                out_ = tr__.finish(it__)
                return it__, out_, t_m - t_a, time.time() - t_m
            futs = [drivers[k % len(tracers)].submit(one_step, tracers[k % len(tracers)]) for k in range(n_steps)]
            for f_ in futs:
                iters_, traces_, lw, fw = f_.result()
                loop_s += lw
                fit_walls.append(fw)
            return loop_s, 0.0, iters_, traces_
        for k in range(n_steps):
            tr_ = tracers[k % len(tracers)]
            # a batch object is reused only after its previous fits (depth+1 steps ago) were collected
            while len(pending) > depth:
                traces_ = pending.pop(0).result()
            t_a = time.time()
            tr_.reset()
            iters_ = tr_.run_loop()
            t_b2 = time.time()
            loop_s += t_b2 - t_a
            if pipeline:
                def timed_finish(tr__=tr_, it__=iters_):
                    t_f = time.time()
                    out_ = tr__.finish(it__)
                    fit_walls.append(time.time() - t_f)
                    return out_
                pending.append(executor.submit(timed_finish))
            else:
                traces_ = tr_.finish(iters_)
                fit_s += time.time() - t_b2
        t_c = time.time()
        while pending:
            traces_ = pending.pop(0).result()
        if pipeline:
            fit_s += time.time() - t_c
        if timed:
            log("%d step(s): device loops %.3fs, fits %s" % (n_steps, loop_s, ("%.3fs" % fit_s) if not pipeline else "overlapped (tail %.3fs)" % fit_s))
        return loop_s, fit_s, iters_, traces_

    log("rank %d: batch of %d edges ready%s" % (rank, E, " (pipelined, %d batch objects)" % len(tracers) if pipeline else ""))
    run_steps(max(args.warmup, len(tracers) if args.warmup else 0), False)
    barrier()
    t0 = time.time()
    loop_s, fit_s, iters, traces = run_steps(args.steps, True)
    barrier()
    elapsed = time.time() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    total_traces = E * world * args.steps
    value = total_traces / elapsed

    if rank != 0:
        if pool is not None:
            pool.terminate()
        if big_farm is not None:
            big_farm.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- quality of this rank's traces vs ground truth (sanity band, not the metric)
    mse = float(np.mean([pkg.gpet_utils.trace_MSE(tr, truth) for tr in traces]))

    # ---- one step alone (nothing else on the GPU): device time of the LML kernel launches of its converged fits,
    #      hipEvents around every launch on the fit stream (gpet_lml_stats)
    tracer.reset()
    it_alone = tracer.run_loop()
    tracer._batch.lml_stats(reset=True)
    tracer.finish(it_alone)
    lml = tracer._batch.lml_stats()
    n_fit = float(np.mean([len(o) for o in tracer._batch.read_obs_all()])) + len(init)  # training points of a converged fit
    iters_per_trace = float(np.mean(it_alone))

    # ---- per-stage and per-kernel device time at a mid-trace state (batch of E edges, 7 iterations in),
    #      measured live with hipEvents on the library's stream (gpet_profile_stage)
    tracer.reset()
    tracer._batch.iterate(seeds, 7)
    sc_mid = tracer._batch.scalars(0)
    n_mid, rank_mid, sweeps_mid = sc_mid.n, sc_mid.rank, max(1, int(sc_mid.lml))
    ring = 16  # the normals stage fills a ring of 16 upcoming iterations per launch: report per iteration
    def per_iter(d):
        d["normals"] = d["normals"] / ring
        return d
    stage_ms = per_iter({name: tracer._batch.profile_stage(i, 20) for i, name in enumerate(STAGES)})
    info0 = tracer._batch.info()
    structured = bool(info0.get("structured", 0))
    kernel_ids = dict(KERNEL_IDS_STRUCT if structured else KERNEL_IDS_GENERIC)
    kernel_ids.update(KERNEL_IDS_COMMON)
    kernel_ms = {name: tracer._batch.profile_stage(kid, 20) for kid, name in kernel_ids.items()}
    # single edge (BASELINE config 2): latency view
    one = pkg.GP_Edge_Tracing_Batch([init], grad, [1], **README_KW, _ctx=ctx)
    one(); one.reset()
    ts = time.time(); one(); single_s = time.time() - ts
    one.reset(); one._batch.iterate([1], 7)
    one_ms = per_iter({name: one._batch.profile_stage(i, 20) for i, name in enumerate(STAGES)})

    # ---- roofline of the dominant kernel: algorithmic bytes / flops per edge per launch
    S, Lg, M_ = README_KW["N_samples"], N, N
    nk = int(README_KW["keep_ratio"] * README_KW["N_samples"])
    r, n_ = rank_mid, n_mid
    alg = {
        "k_fit": dict(flops=n_ ** 3 / 3.0 + 2.0 * n_ * n_, bytes=8.0 * (n_ * n_ + 4 * n_)),
        "k_predict": dict(flops=1.0 * n_ * n_ * Lg + 4.0 * n_ * Lg, bytes=8.0 * (n_ * Lg + n_ * n_ / 2 + 2 * Lg)),
        "k_cov_mfma": dict(flops=1.0 * Lg * Lg * n_ + 30.0 * Lg * Lg / 2, bytes=8.0 * (Lg * Lg + n_ * Lg)),
        "k_pchol_reg": dict(flops=2.0 * Lg * r * r / 2 * 2, bytes=8.0 * (2 * r * Lg)),
        "k_gram": dict(flops=1.0 * r * r * Lg, bytes=8.0 * (r * Lg + r * r)),
        "k_factor_rows": dict(flops=2.0 * r * r * Lg, bytes=8.0 * (2 * r * Lg + r * r)),
        "k_struct_H": dict(flops=1.0 * n_ * n_ * r + 2.0 * r * r * n_ + 2.0 * r * Lg, bytes=8.0 * (n_ * n_ / 2 + r * r + r * Lg)),
        "k_jacobi_lds": dict(flops=6.0 * sweeps_mid * r ** 3, bytes=8.0 * (3 * r * r)),
        "k_struct_rows": dict(flops=2.0 * r * r * Lg, bytes=8.0 * (2 * r * Lg + r * r)),
        "k_sample_gemm_mfma_r": dict(flops=2.0 * S * Lg * r, bytes=8.0 * (S * r + r * Lg + S * Lg)),
        "k_score": dict(flops=60.0 * S * Lg, bytes=8.0 * S * Lg + 4.0 * M_ * N),
        "k_topk": dict(flops=2.0 * S * S, bytes=16.0 * S),
        "k_kde_prep": dict(flops=2.0 * nk * Lg, bytes=8.0 * nk * Lg),
        "k_kde_fused": dict(flops=36.0 * M_ * N + 10.0 * nk * Lg, bytes=8.0 * nk * Lg + 4.0 * M_ * N),
    }
    # the normals kernel fills a ring of `ring` iterations per launch on a side stream: per-iteration share
    zc = info0["z_cols"]
    kernel_ms["k_mt_normals"] = stage_ms["normals"] * ring
    alg["k_mt_normals"] = dict(flops=40.0 * S * zc * ring, bytes=8.0 * S * zc * ring)
    per_iter = {k: (v / ring if k == "k_mt_normals" else v) for k, v in kernel_ms.items()}
    # dominant kernel of a step = largest device time per step of E traces, kernels timed alone: a loop kernel runs
    # once per iteration of the trace, the LML kernel of the converged fits ~80 times per step (one launch per
    # lock-step round of the optimiser).  The LML kernel evaluates `evaluations` problems of n_fit points each:
    # n^3 flops (sweep of the bordered matrix, n^3 / 2 FMAs) and 3 n + 4 doubles of HBM traffic per evaluation.
    per_step = {k: v * iters_per_trace for k, v in per_iter.items()}
    per_step["k_lml"] = lml["kernel_ms"]
    kernel_ms["k_lml"] = lml["kernel_ms"] / max(1, lml["launches"])
    per_iter["k_lml"] = lml["kernel_ms"] / iters_per_trace
    ev_per_launch = lml["evaluations"] / max(1, lml["launches"])
    alg["k_lml"] = dict(flops=n_fit ** 3 * ev_per_launch / E, bytes=8.0 * (3 * n_fit + 4) * ev_per_launch / E)
    dom = max(per_step, key=per_step.get)
    d_ms = kernel_ms[dom]
    a_bytes = alg[dom]["bytes"] * E
    a_flops = alg[dom]["flops"] * E
    gbs = a_bytes / (d_ms * 1e-3) / 1e9
    tfl = a_flops / (d_ms * 1e-3) / 1e12
    ridge = FP64_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)  # flop per byte
    use_flops = (a_flops / a_bytes) > ridge
    traffic = None
    tp = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tp):
        prof = json.load(open(tp))
        if prof.get("edges") == E and prof.get("image") == [N, N] and dom in prof.get("kernels", {}):
            traffic = prof["kernels"][dom]["hbm_bytes_per_launch"]
    roofline = dict(kernel=dom, bound="mfma" if use_flops else "hbm",
                    achieved=tfl if use_flops else gbs, peak=FP64_PEAK_TFLOPS if use_flops else HBM_PEAK_GBS,
                    unit="TFLOP/s" if use_flops else "GB/s",
                    frac=(tfl / FP64_PEAK_TFLOPS) if use_flops else (gbs / HBM_PEAK_GBS), traffic=traffic,
                    launch_ms=d_ms, edges_per_launch=E, algorithmic_bytes=a_bytes, algorithmic_flops=a_flops,
                    launches_per_step=(lml["launches"] if dom == "k_lml" else iters_per_trace),
                    device_ms_per_step={k: v for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
                    note=("f64 vector/matrix peak (equal on MI355X); this kernel is an LDS-resident eigen-solver: its practical "
                          "bound is LDS bandwidth and barrier latency, see DESIGN.md section 6" if dom == "k_jacobi_lds" else
                          ("objective of the converged fits: %d evaluations of ~%.0f-point problems in %d launches per step; "
                           "f64 vector peak; latency-bound (one barrier per pivot), DESIGN.md section 6"
                           % (lml["evaluations"], n_fit, lml["launches"]) if dom == "k_lml" else None)),
                    state=dict(n_train=n_mid, factor_rank=rank_mid, jacobi_sweeps=sweeps_mid,
                               loop_path="structured" if structured else "generic", iterations_per_trace=iters_per_trace,
                               n_train_final_fit=n_fit),
                    all_kernels={k: dict(ms=v, ms_per_iteration=per_iter[k],
                                         GBps=alg[k]["bytes"] * E / (v * 1e-3) / 1e9,
                                         TFLOPps=alg[k]["flops"] * E / (v * 1e-3) / 1e12) for k, v in kernel_ms.items()})

    log("stage profile done; dominant stage %s %.3f ms" % (dom, d_ms))
    cpu = None
    if not args.no_cpu_baseline and world == 1:  # (reported at N=1 only)
        cpu = cpu_baseline(N, 3, args.cpu_traces)

    out = {
        "metric": "edge-traces/sec on 500x500 synthetic images; GP-iter ms (Cholesky+sample)",
        "value": value, "unit": "edge-traces/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE config 2 edge (500x500 sinusoidal image, RBF sigma_f=75 l=20, N_samples=1000, "
                               "delta_x=5, pixel_thresh=5) x %d independent edges per GPU and step (config 4's batch of independent edges, sized to fill the GPU), "
                               "shared gradient image%s" % (E, ", RCCL broadcast" if world > 1 else ""),
                   "edges_per_gpu": E, "image": [N, N], "iterations_per_trace": iters[:4],
                   "final_fit": ("scipy L-BFGS-B routine x13 starts in lock step (%d worker processes), objective = batched LML kernel on the GPU" % n_lbfgs
                                 if args.fit_workers <= 1 else "host objective, %d worker processes" % args.fit_workers)},
        "gp_iter_ms": {"batch_of_%d" % E: sum(stage_ms[k] for k in STAGES[:4]),
                       "single_edge": sum(one_ms[k] for k in STAGES[:4])},
        "stage_ms_batch": stage_ms, "stage_ms_single_edge": one_ms,
        "time_split_s": {"device_loop": loop_s, "final_fit_not_overlapped": fit_s, "elapsed": elapsed, "pipelined": pipeline, "pipeline_depth": depth,
                         "fit_wall_mean": (sum(fit_walls) / len(fit_walls)) if fit_walls else None},
        "single_edge": {"traces_per_s": 1.0 / single_s, "ms_per_trace": 1e3 * single_s},
        "trace_mse_vs_truth": mse, "bcast_grad_ms": 1e3 * t_b,
        "roofline": roofline, "cpu_baseline": cpu,
    }
    if cpu:
        out["speedup_vs_cpu_port"] = value / cpu["value"]
    print(json.dumps(out))
    if pool is not None:
        pool.terminate()
    if big_farm is not None:
        big_farm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
