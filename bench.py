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

# The bench keeps eight batch objects in flight, each with three HIP streams (loop, RNG look-ahead, converged fits): 24 streams on
# the runtime's default of FOUR hardware queues per process alias heavily (streams of one queue run in order).  Eight queues:
# +2-3 % on the headline, +9 % on batches of 256 (profiles/r05_hw_queues.txt; 12 and 16 hurt the single-edge latency).  The HIP
# runtime reads this variable when it initialises, so it is set before anything that loads it; a value the caller has set wins.
if os.environ.get("GPET_BENCH_HW_QUEUES") == "runtime-default":
    # (the child run behind secondary.hw_queues_runtime_default: what the HIP runtime does when nobody sets the variable --
    #  the package itself sets 8 when it loads, gaussian_process_edge_trace_amd/_lib.py, unless the environment has a value)
    os.environ["GPU_MAX_HW_QUEUES"] = "4"
else:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

README_KW = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}, noise_y=1, N_samples=1000,
                 score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
STAGES = ["fit_predict_cov", "factor", "normals", "sample_gemm", "score_topk", "curve_kde"]
# gpet_profile_stage ids of the single kernels of one iteration (include/gpet_hip.h)
# structured loop path (prior eigenbasis; what gpet_trace_iterate runs for on-grid batches): 120-123;
# generic path (per-stage API, off-grid observations, option struct_path = 0): 100-113
KERNEL_IDS_STRUCT = {120: "k_fit", 121: "k_struct_H", 122: "k_jacobi_ahead", 123: "k_struct_rows"}
KERNEL_IDS_GENERIC = {100: "k_fit", 101: "k_predict", 102: "k_cov_mfma", 110: "k_pchol_reg", 111: "k_gram",
                      112: "k_jacobi_ahead", 113: "k_factor_rows"}
KERNEL_IDS_COMMON = {130: "k_sample_gemm_mfma_r", 140: "k_score", 141: "k_topk", 150: "k_kde_prep",
                     151: "k_kde_fused", 160: "k_pix_columns"}  # (152 k_kde_normalise: stage API only; the loop normalises inside the pixel kernels)
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
    BASELINE.md section 2); the faster setting is reported with its thread count.  A trace includes the
    constructor (gradient-image KDE), as a reference user pays it per edge; its share is timed separately."""
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
            tc = time.time()
            orc.kde_of_gradient(orc.normalise(grad, (0, 1), np.float64))
            ctor = time.time() - tc
        log("cpu baseline: %d BLAS thread(s): %.2f s per trace (constructor %.2f s of it)" % (threads, dt / n_traces, ctor))
        if best is None or dt < best[0]:
            best = (dt, threads, iters, ctor)
    dt, threads, iters, ctor = best
    return dict(value=n_traces / dt, unit="edge-traces/s", cores=int(threads), kind="port",
                sample="%d full trace(s) of the 500x500 README edge (RBF sf=75 l=20, S=1000, dx=5, pixel_thresh=5), "
                       "%s iterations, oracle/gpet_oracle.py: per-curve scoring loop + LAPACK SVD sampling + "
                       "scipy L-BFGS-B x13, constructor (gradient KDE) included; best of 1 and 16 BLAS threads" % (n_traces, iters),
                seconds=dt, host_cores=usable_cpus(), ctor_s_per_trace=ctor,
                value_ctor_excluded=n_traces / max(1e-9, dt - n_traces * ctor))


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


def timed_steps(tracers, n_steps, depth, executor, fit_walls, loop_walls=None):
    """n_steps passes of the hot path over the batch objects in `tracers`.  depth > 0: len(tracers) WHOLE traces in flight
    -- one host thread per batch object runs reset -> device loop -> converged fits for its share of the steps (step k on
    object k mod W), every object on its own HIP stream.  The GPU then always has the loop of one batch, the fits of
    another (device-resident L-BFGS-B, ~80 rounds of small launches) and the thinly populated last iterations of a third
    to choose from: 7.2-7.4 k edge-traces/s with three objects against 6.8-6.9 k when only one device loop ran at a time
    beside the previous steps' fits (tools/time_concurrent_loops.py).  depth = 0: one object, loop then fits, nothing
    overlapped.  Returns (loop wall time per worker, fit wall time per worker -- sums over its steps, averaged over the
    workers --, iterations and traces of the last step)."""
    if loop_walls is None:
        loop_walls = []
    if depth <= 0 or len(tracers) == 1:
        loop_s = fit_s = 0.0
        iters_, traces_ = [], None
        tr_ = tracers[0]
        for k in range(n_steps):
            t_a = time.time()
            tr_.reset()
            iters_ = tr_.run_loop()
            t_b = time.time()
            traces_ = tr_.finish(iters_)
            t_c = time.time()
            loop_walls.append(t_b - t_a)
            fit_walls.append(t_c - t_b)
            loop_s += t_b - t_a
            fit_s += t_c - t_b
        return loop_s, fit_s, iters_, traces_
    W = min(len(tracers), n_steps)
    lw, fw = [[] for _ in range(W)], [[] for _ in range(W)]

    def worker(w):
        tr_, its, out = tracers[w], [], None
        for k in range(w, n_steps, W):  # (a batch object is only ever used by its own thread)
            t_a = time.time()
            tr_.reset()
            its = tr_.run_loop()
            t_b = time.time()
            out = tr_.finish(its)
            lw[w].append(t_b - t_a)
            fw[w].append(time.time() - t_b)
        return its, out

    res = list(executor.map(worker, range(W)))
    for w in range(W):
        loop_walls.extend(lw[w])
        fit_walls.extend(fw[w])
    iters_, traces_ = res[(n_steps - 1) % W]
    return sum(map(sum, lw)) / W, sum(map(sum, fw)) / W, iters_, traces_


def job_ms(objs, pool, reps=4):
    """Wall time (ms, median of `reps` after one warm-up) of ONE job: every batch object in `objs` traces its edges once --
    reset -> device loop -> converged fits -- all of them in flight together, one host thread and one HIP stream each."""
    def one(o):
        o.reset()
        return o.finish(o.run_loop())
    ts = []
    for _ in range(reps + 1):
        for o in objs:
            o._ctx.sync()
        ta = time.time()
        list(pool.map(one, objs))
        for o in objs:
            o._ctx.sync()
        ts.append(1e3 * (time.time() - ta))
    return float(np.median(ts[1:])) if reps > 0 else ts[0]


def config4_literal(pkg, L, dev_index, make_tracer, seeds, n_total=256, n_gpus=8):
    """BASELINE config 4 AS STATED -- 256 independent 500x500 edges over 8 GPUs -- the way it would run: the whole job on one
    GPU as k batch objects in flight (k = 1, 4, 8: a job of 256 edges is small enough for the tail of one loop to leave the
    GPU idle, several smaller loops fill it), and the per-GPU share of 256 / 8 = 32 edges as 1 x 32, 2 x 16 and 4 x 8
    objects in flight.  predicted_speedup_8_gpus = best t(256) / best t(32): what eight GPUs can reach on that job (no
    collective on the path; the broadcast of the 1 MB image is bcast_grad_ms at N > 1)."""
    from concurrent.futures import ThreadPoolExecutor
    share = n_total // n_gpus
    pool = ThreadPoolExecutor(max_workers=8)
    ctxs = [L.Context(dev_index) for _ in range(8)]
    out = {"edges": n_total, "gpus": n_gpus, "share_edges": share, "whole_job_ms": {}, "share_ms": {}}

    def split(n, k):
        objs, per = [], n // k
        for j in range(k):
            objs.append(make_tracer(per, ctxs[j], sds=seeds[j * per:(j + 1) * per]))
        return objs
    for k in (1, 4, 8):
        objs = split(n_total, k)
        out["whole_job_ms"]["%d x %d" % (k, n_total // k)] = job_ms(objs, pool)
        for o in objs:
            o._batch.close()
    for k in (1, 2, 4):
        objs = split(share, k)
        out["share_ms"]["%d x %d" % (k, share // k)] = job_ms(objs, pool)
        for o in objs:
            o._batch.close()
    pool.shutdown()
    out["best_whole_job_ms"] = min(out["whole_job_ms"].values())
    out["best_share_ms"] = min(out["share_ms"].values())
    out["predicted_speedup_8_gpus"] = out["best_whole_job_ms"] / out["best_share_ms"]
    out["note"] = ("one job = every object traces its edges once (device loop + converged fits), all objects in flight together, "
                   "median of 4; constructor excluded; the same seeds however the edges are split")
    return out


def readme_literal(pkg, L, dev_index, ctx, init, grad, seeds, with_cpu=True):
    """The README's own call (README.md:75-76) predates the pixel_thresh parameter: its positional arguments bind seed = 1 to
    pixel_thresh (clamped to 2), return_std = True to seed (= 1) and fix_endpoints to return_std (quirk Q6) -- 29 iterations
    instead of 14.  One edge (latency), a batch of 1 024 edges (three objects in flight) and one trace of the CPU port."""
    from concurrent.futures import ThreadPoolExecutor
    kw = dict(README_KW)
    kw["pixel_thresh"] = 2
    one = pkg.GP_Edge_Tracing_Batch([init], grad, [1], **kw, _ctx=ctx)
    one()
    one.reset()
    runs = []
    for _ in range(5):
        ts = time.time()
        one()
        runs.append(time.time() - ts)
        one.reset()
    iters_one = list(one.timings["iters"])
    one._batch.close()
    E = 1024
    objs = [pkg.GP_Edge_Tracing_Batch([init] * E, grad, seeds[:E] if k == 0 else [sd + 7 * k for sd in seeds[:E]], **kw,
                                      _ctx=(ctx if k == 0 else L.Context(dev_index))) for k in range(3)]
    pool = ThreadPoolExecutor(max_workers=3)
    t_job = job_ms(objs, pool, reps=3)
    its = sorted(set(objs[0]._iters()))
    for o in objs:
        o._batch.close()
    pool.shutdown()
    out = dict(config="README.md:75-76 as written: pixel_thresh = 2 (Q6), seed = 1; otherwise the headline's edge",
               single_edge_ms_per_trace=1e3 * float(np.median(runs)), single_edge_iterations=iters_one,
               batch_traces_per_s=3 * E / (1e-3 * t_job), batch_ms_per_1024=t_job / 3, batch_iterations=[its[0], its[-1]],
               note="batch: three objects of 1 024 edges in flight, each one whole trace (loop + converged fits), median of 3 jobs")
    if with_cpu:
        from oracle import gpet_oracle as orc
        from threadpoolctl import threadpool_limits
        img, edge = orc.synth_sinusoid_image(grad.shape[0], 3)
        g_o = orc.comp_grad_img(img, orc.kernel_builder((11, 5)))
        i_o = edge[[0, -1], :][:, [1, 0]]
        with threadpool_limits(limits=1):
            t0 = time.time()
            _, _, info = orc.trace(i_o, g_o, per_curve=True, seed=1, **kw)
            dt = time.time() - t0
        out["cpu_port"] = dict(s_per_trace=dt, traces_per_s=1.0 / dt, iterations=info["n_iter"], cores=1,
                               note="oracle/gpet_oracle.py, one BLAS thread, one trace, constructor included")
        out["batch_speedup_vs_cpu_port"] = out["batch_traces_per_s"] * dt
    return out


def secondary_config3(pkg, ctx, n_edges=1):
    """BASELINE config 3's shape: 2048x2048 image, 1498 observations (+2 inits = 1500 training points), N_samples=4000:
    ms of one GP iteration (fit + predict + covariance, factor, sample GEMM) and of the scoring, per stage, timed with
    hipEvents on the library's stream (gpet_timer_*).  n_edges > 1: the same edge n_edges times in one batch (different
    observation noise and seeds) -- a single edge of this size is a latency chain of ~100 launches, a batch is what shows
    the blocked f64-MFMA path's rate."""
    N = 2048
    img, truth = synth_image(N, 0)
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 300, 'length_scale': 80}, noise_y=1, N_samples=4000,
              score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    obs_all = []
    for e in range(n_edges):
        rng = np.random.default_rng(e)
        cols = np.sort(rng.choice(np.arange(1, N - 1), size=1498, replace=False))
        obs_all.append(np.stack([cols, np.clip(truth[cols, 0] + rng.integers(-2, 3, size=cols.size), 0, N - 1)], axis=1).astype(np.int64))
    bt = pkg.GP_Edge_Tracing_Batch([init] * n_edges, grad, [1 + 97 * e for e in range(n_edges)], obs=obs_all, _ctx=ctx, **kw)
    b = bt._batch

    def timed(fn, reps=3):
        """device time of a per-stage entry point: hipEvents around `reps` calls on the context's stream"""
        fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        return ctx.timer_stop_ms() / reps
    ms = dict(fit_predict_cov=timed(lambda: b.fit_predict(True)), factor=timed(b.factor))
    ms["normals"] = timed(lambda: b.normals([7 + e for e in range(n_edges)]))  # RandomState(seed).standard_normal((4000, 2048)) per edge
    ms["sample_gemm"] = timed(b.sample)
    ms["score_topk"] = timed(b.score)
    s_ = b.scalars()
    n, Lg = int(s_.n), N
    gflop = n_edges * (n ** 3 / 3.0 + n * n * Lg + Lg * Lg * n) * 1e-9  # Cholesky + V = L^-1 K*^T + K** - V^T V
    out = dict(config="2048x2048, n=1500 training points, N_samples=4000, RBF sigma_f=300 l=80; per-stage entry points "
                      "(1498 observations exceed algo_thresh, so the loop itself would not iterate: SURVEY 8d C3); %d edge(s) per launch" % n_edges,
               edges=n_edges, gp_iter_ms=ms["fit_predict_cov"] + ms["factor"] + ms["normals"] + ms["sample_gemm"],
               gp_iter_ms_note="fit + predict + covariance, factor, the 8.2 M normals per edge of the iteration, sample GEMM",
               scoring_ms=ms["score_topk"], stage_ms=ms, fit_predict_cov_tflops=gflop / ms["fit_predict_cov"],
               fit_predict_cov_frac_of_f64_peak=gflop / ms["fit_predict_cov"] / FP64_PEAK_TFLOPS,
               sample_gemm_tflops=n_edges * 2.0 * 4000 * Lg * int(s_.rank) * 1e-9 / ms["sample_gemm"],
               n_train=n, factor_rank=int(s_.rank), timing="hipEvents on the library's stream, 3 repetitions")
    b.close()
    return out


def secondary_config5(pkg, ctx, n_chains=8, frames_per_chain=8):
    """BASELINE config 5's shape on ONE GPU: a 1024x1024 image sequence, Matern-5/2 (sigma_f ~154, l ~41), frames
    chained by the warm start, `n_chains` chains traced as batches of `n_chains` edges (one frame per chain and step)."""
    N = 1024
    T = n_chains * frames_per_chain
    frames = []
    for t in range(T):
        img, truth = pkg.gpet_utils.construct_test_img((N, N), int(0.4 * N * (1.0 + 0.01 * (t % frames_per_chain))), 4, 0.05,
                                                       'sinusoidal', 0.3, gaps=True, seed=100 + t)
        frames.append(pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx))
    init = truth[[0, -1], :][:, [1, 0]]
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 154, 'length_scale': 41}, noise_y=1, N_samples=1000,
              score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    st = pkg.SequenceTracer(frames, init, n_chains=n_chains, warm_every=16, seed=3, _ctx=ctx, **kw)
    t0 = time.time()
    st()
    dt = time.time() - t0
    # the factor alone at a mid-trace state: 8 edges, warm-start observations, not finished
    warm = truth[16:-16:16][:, [1, 0]].astype(np.int64)
    fb = pkg.GP_Edge_Tracing_Batch([init] * n_chains, frames[:n_chains], [3] * n_chains, obs=[warm] * n_chains, _ctx=ctx, **kw)
    fb._batch.fit_predict(True)
    fb._batch.factor()
    fac = fb._batch.profile_stage(1, 2)
    sweeps = int(fb._batch.scalars().lml)
    fb._batch.close()
    one1 = pkg.GP_Edge_Tracing_Batch([init], frames[:1], [3], obs=[warm], _ctx=ctx, **kw)
    one1._batch.fit_predict(True)
    one1._batch.factor()
    fac1 = one1._batch.profile_stage(1, 2)
    one1._batch.close()

    # ... and the factor as every iteration but a trace's first runs it: three iterations into a trace, the covariance of the
    # current observation set, the previous iteration's rows in the ring (warm start, option oj_warm)
    def warm_factor(nb):
        bt_ = pkg.GP_Edge_Tracing_Batch([init] * nb, frames[:nb], [3] * nb, _ctx=ctx, **kw)
        bt_._batch.iterate([3] * nb, 3)
        bt_._batch.profile_stage(0, 1)
        ms_ = bt_._batch.profile_stage(1, 2)
        sw_ = int(bt_._batch.scalars().lml)
        bt_._batch.close()
        return ms_, sw_
    fac_w, sweeps_w = warm_factor(n_chains)
    fac1_w, _ = warm_factor(1)
    one = pkg.SequenceTracer(frames[:2], init, n_chains=1, warm_every=16, seed=3, _ctx=ctx, **kw)
    t1 = time.time()
    one()
    dt1 = time.time() - t1
    # the literal per-GPU shares of config 5 on 8 GPUs, each timed alone on this one: (a) 8 chains x 8 frames = ONE chain per
    # GPU (a chain is serial: frame t warm-starts frame t + 1); (b) 16 chains x 4 frames = TWO chains per GPU (more cold starts,
    # but the any-rank factor of two edges costs little more than that of one)
    sh_a = pkg.SequenceTracer(frames[:frames_per_chain], init, n_chains=1, warm_every=16, seed=3, _ctx=ctx, **kw)
    t2 = time.time()
    sh_a()
    dt_a = time.time() - t2
    half = max(1, frames_per_chain // 2)
    sh_b = pkg.SequenceTracer(frames[:2 * half], init, n_chains=2, warm_every=16, seed=3, _ctx=ctx, **kw)
    t3 = time.time()
    sh_b()
    dt_b = time.time() - t3
    return dict(config="1024x1024, Matern-5/2 sigma_f=154 l=41, N_samples=1000, delta_x=8; %d frames = %d chains x %d "
                       "(cold first frame, warm-started later ones), one GPU, constructor of the batch included" % (T, n_chains, frames_per_chain),
                frames_per_s=T / dt, seconds_total=dt, iterations_per_frame=st.iterations,
                single_chain_s_per_frame=dt1 / 2, single_chain_iterations=one.iterations,
                one_chain_per_gpu_s=dt_a, one_chain_per_gpu_iterations=sh_a.iterations,
                predicted_speedup_8_gpus_one_chain_each=dt / dt_a,
                two_chains_per_gpu_s=dt_b, two_chains_per_gpu_iterations=sh_b.iterations,
                predicted_speedup_8_gpus_two_half_chains_each=dt / dt_b,
                per_gpu_share_note="config 5 on 8 GPUs as stated is one chain of %d frames per GPU; the prediction is this GPU's time for all "
                                   "%d frames over its time for that share alone (no collective on the path); with 16 chains of %d frames a GPU "
                                   "holds two edges per step" % (frames_per_chain, T, half),
                factor_ms_batch_of_chains=fac, factor_ms_single_edge=fac1, factor_jacobi_sweeps=sweeps,
                factor_ms_batch_of_chains_warm=fac_w, factor_ms_single_edge_warm=fac1_w, factor_jacobi_sweeps_warm=sweeps_w,
                factor_note="cold: the first iteration of a trace (pivoted Cholesky rows); warm: every later iteration (rows from the "
                            "previous iteration's factor, k_ojw_*)", chains=n_chains)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)  # (a multiple of the objects in flight: no half-empty last round)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--edges", type=int, default=1024,
                    help="independent edges per GPU and step (BASELINE config 4 is a batch of independent 500x500 edges).  With "
                         "several steps in flight (--pipeline-depth) the throughput no longer depends on the batch size: 7.3-7.4 k "
                         "traces/s at 1024 x 6, 2048 x 3 or 4096 x 3 objects; one step at a time it was 6.6 k at 1024, 6.9 k at 2048, "
                         "7.1 k at 4096.  The 256-edge figure of config 4 is reported next to it")
    ap.add_argument("--size", type=int, default=500)
    ap.add_argument("--pipeline-depth", type=int, default=7,
                    help="whole traces in flight = depth + 1 batch objects, each driven by its own host thread on its own HIP stream (0 = one object, nothing overlapped)")
    ap.add_argument("--cpu-traces", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (ctor included, 256 edges, "
                    "distinct images, configs 3 and 5)")
    ap.add_argument("--headline-only", action="store_true", help="print {value, ms_per_step, hip_hw_queues} after the timed steps and stop "
                    "(the child runs of secondary.hw_queues_runtime_default)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("warning: WORLD_SIZE=%d but --gpus=%d; using WORLD_SIZE" % (world, args.gpus), file=sys.stderr)
    depth = max(0, args.pipeline_depth)
    pipeline = depth > 0
    import torch
    dist = None
    # GPET_BENCH_BACKEND=gloo + GPET_BENCH_SHARE_GPU=1: rehearsal of the N>1 path on a one-GPU box
    # (ranks share device 0, collectives on CPU tensors); the real run uses nccl (= RCCL), one rank per GPU
    backend = os.environ.get("GPET_BENCH_BACKEND", "nccl")
    share_gpu = os.environ.get("GPET_BENCH_SHARE_GPU", "0") == "1"
    dev_index = 0 if share_gpu else local_rank
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    # GPET_BENCH_FORCE_DIST=1: take the N>1 code path (process group, broadcast, device-pointer hand-over, barriers)
    # with a single rank too -- the only way to run the nccl path on a one-GPU box
    use_dist = world > 1 or os.environ.get("GPET_BENCH_FORCE_DIST", "0") == "1"
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"),
                              RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    import gaussian_process_edge_trace_amd as pkg
    L = pkg._lib
    ctx = L.Context(dev_index)

    # ---- inputs: one shared gradient image, produced on rank 0's GPU, broadcast over RCCL/xGMI and consumed where the
    #      collective put it (device pointer -> gpet_batch_create2 / GPET_GRAD_ON_DEVICE)
    N = args.size
    img, truth = synth_image(N, 3)
    init = truth[[0, -1], :][:, [1, 0]]
    t_b = 0.0
    grad_kw = {}
    if use_dist:
        g = torch.empty((N, N), dtype=torch.float32, device=coll_dev)
        if rank == 0:
            g.copy_(torch.from_numpy(pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)))
        torch.cuda.synchronize()
        tb0 = time.time()
        dist.broadcast(g, src=0)
        torch.cuda.synchronize()
        t_b = time.time() - tb0
        if coll_dev == "cuda":
            grad, grad_kw = None, dict(grad_device_ptrs=[g.data_ptr()], grad_shape=(N, N))
        else:
            grad = g.numpy()
    else:
        grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)

    E = args.edges
    # independent edges: iteration k of an edge draws from RandomState(seed + k + 1) (gpet.py:839), so consecutive seeds
    # would hand edge e + 1's iteration-k normals to edge e at iteration k + 1; seeds 997 apart (a trace takes < 64
    # iterations) give every (edge, iteration) of every batch object in flight on every rank its own stream
    SEED_STRIDE = 997
    n_objs = max(0, args.pipeline_depth) + 1

    def seeds_of(obj):
        return [1 + SEED_STRIDE * ((rank * n_objs + obj) * E + e) for e in range(E)]
    seeds = seeds_of(0)

    def make_tracer(n_edges, ctx_, images=None, inits=None, sds=None):
        return pkg.GP_Edge_Tracing_Batch(inits if inits is not None else [init] * n_edges,
                                         images if images is not None else grad, sds if sds is not None else seeds[:n_edges],
                                         **README_KW, _ctx=ctx_, **(grad_kw if images is None else {}))

    tracer = make_tracer(E, ctx)
    tracers = [tracer]
    for d_ in range(depth):  # more batch objects, each with its own context (= HIP stream) and its own seeds
        tracers.append(make_tracer(E, L.Context(dev_index), sds=seeds_of(d_ + 1)))
    if grad is None:
        grad = g.cpu().numpy()  # (the secondary figures below build batches from host arrays)

    def barrier():
        for tr_ in tracers:
            tr_._ctx.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    from concurrent.futures import ThreadPoolExecutor
    executor = ThreadPoolExecutor(max_workers=depth + 1)
    fit_walls, loop_walls = [], []  # wall time of every step's converged fits / device loop (other steps' run beside them)

    def run_steps(n_steps, timed):
        fit_walls.clear()
        loop_walls.clear()
        loop_s, fit_s, iters_, traces_ = timed_steps(tracers, n_steps, depth, executor, fit_walls, loop_walls)
        if timed:
            log("%d step(s), %d in flight: per host thread %.3fs in device loops, %.3fs in converged fits" % (n_steps, len(tracers), loop_s, fit_s))
        return loop_s, fit_s, iters_, traces_

    log("rank %d: batch of %d edges ready%s" % (rank, E, " (pipelined, %d batch objects)" % len(tracers) if pipeline else ""))
    n_in_flight = len(tracers)

    def release_pipeline_objects():
        """The latency figures that follow (configs 3 / 4-literal / 5, one edge, the stage profiles) are of ONE object on the GPU: the
        other pipeline objects are freed first -- their idle streams would share the process's few hardware queues with the object
        under test, and whether its RNG look-ahead stream then lands on the queue of its own loop is luck (one edge 9.5 or 11.9 ms,
        one Matern chain 0.85 or 1.2 s: profiles/r05_hw_queues.txt)."""
        for tr_ in tracers[1:]:
            tr_._batch.close()
            tr_._ctx.close()
        del tracers[1:]

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
    if args.headline_only:
        if rank == 0:
            print(json.dumps({"value": value, "ms_per_step": 1e3 * elapsed / args.steps, "steps": args.steps, "warmup": args.warmup,
                              "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "objects_in_flight": len(tracers)}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- N > 1: BASELINE config 4 as stated (256 edges over the ranks) measured for real: every rank traces its 256 / world
    #      edges -- as 1, 2 or 4 objects in flight, the best of them -- between two barriers; the job is the slowest rank's time
    c4_measured = None
    if dist is not None and world > 1 and 256 % world == 0:
        from concurrent.futures import ThreadPoolExecutor as _TPE
        share = 256 // world
        pool4 = _TPE(max_workers=4)
        best = None
        for k in (1, 2, 4):
            if share % k or share // k < 1:
                continue
            per = share // k
            objs = [make_tracer(per, tracers[j % len(tracers)]._ctx, sds=[1 + SEED_STRIDE * (rank * share + j * per + e) for e in range(per)]) for j in range(k)]
            job_ms(objs, pool4, reps=1)  # (warm-up)
            barrier()
            ta = time.time()
            job_ms(objs, pool4, reps=0)
            barrier()
            tj = torch.tensor([time.time() - ta], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tj, op=dist.ReduceOp.MAX)
            for o in objs:
                o._batch.close()
            if best is None or float(tj.item()) < best[0]:
                best = (float(tj.item()), k)
        pool4.shutdown()
        if best is not None:
            c4_measured = dict(edges=256, ranks=world, share_edges=share, job_ms=1e3 * best[0], objects_in_flight_per_rank=best[1],
                               traces_per_s=256 / best[0],
                               note="BASELINE config 4 as stated, strong scaling: max over ranks between two barriers, best of 1 / 2 / 4 "
                                    "objects in flight per rank; compare with secondary.config4_literal.best_whole_job_ms of the N = 1 line")
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- quality of this rank's traces vs ground truth (sanity band, not the metric)
    mse = float(np.mean([pkg.gpet_utils.trace_MSE(tr, truth) for tr in traces]))

    # ---- secondary figures (N = 1 only): what the headline leaves out, each a short run of its own.  Every one runs inside
    #      its own try: a failure (an allocation, a mode that does not apply) becomes {"error": ...} in its slot and can never
    #      cost the contract line
    secondary = None
    n_sec = 2 * len(tracers)  # steps of a secondary run: two per batch object in flight

    def sync_all(objs):
        for o in objs:
            o._ctx.sync()

    def run_secondary(name, fn):
        import traceback
        try:
            secondary[name] = fn()
        except Exception as ex:  # noqa: BLE001 -- anything: the headline must survive
            secondary[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            log("secondary %s FAILED: %s\n%s" % (name, ex, traceback.format_exc()))
        return secondary[name]

    def sec_ctor_included():
        # (a) constructor INCLUDED: a fresh batch object per step (arena, image upload + re-normalisation, gradient KDE,
        #     prior eigenbasis of every edge), then loop + fits, nothing overlapped
        ctx2 = L.Context(dev_index)
        t_c = []
        for _ in range(3):
            t1 = time.time()
            fresh = make_tracer(E, ctx2)
            t2 = time.time()
            fresh()
            ctx2.sync()
            t_c.append((time.time() - t1, t2 - t1))
            fresh._batch.close()
            del fresh
        best = min(t_c)
        return dict(traces_per_s=E / best[0], s_per_step=best[0], ctor_s=best[1], edges=E,
                    note="fresh GP_Edge_Tracing_Batch per step, loop + converged fits not overlapped with anything")

    def sec_no_pipeline():
        # (b) the same without pipelining (ctor excluded): the loop and the fits of ONE batch object back to back
        l1, f1, _, _ = timed_steps([tracer], 3, 0, executor, [])
        return dict(traces_per_s=3 * E / (l1 + f1), loop_s_per_step=l1 / 3, fit_s_per_step=f1 / 3, edges=E)

    def timed_mode(objs, n_steps):
        """warm-up (one step per object), then n_steps pipelined steps between syncs: (seconds, traces of the last step)"""
        sync_all(objs)
        timed_steps(objs, len(objs), depth, executor, [])
        sync_all(objs)
        t1 = time.time()
        _, _, it_, tr_ = timed_steps(objs, n_steps, depth, executor, [])
        sync_all(objs)
        return time.time() - t1, it_, tr_

    def sec_f32_samples():
        # (b2) the opt-in f32 storage of the samples (BASELINE config 2: "fp64 Cholesky + fp32 posterior samples"): the same
        #      pipelined steps with gpet_batch_set_sample_dtype(1); never the headline (parity is stated on f64 samples)
        try:
            for tr_ in tracers:
                tr_._ctx.sync()
                tr_._batch.set_sample_dtype("f32")
            dt_f32, _, tr_f32 = timed_mode(tracers, n_sec)
        finally:
            for tr_ in tracers:
                tr_._ctx.sync()
                tr_._batch.set_sample_dtype("f64")
        return dict(traces_per_s=n_sec * E / dt_f32, ms_per_step=1e3 * dt_f32 / n_sec, edges=E,
                    trace_mse_vs_truth=float(np.mean([pkg.gpet_utils.trace_MSE(t_, truth) for t_ in tr_f32])),
                    note="sample GEMM stores f32, scorer/KDE widen; all arithmetic f64; opt-in, "
                         "tests/test_gpu_trace.py::test_full_trace_f32_samples_vs_oracle")

    def sec_philox():
        # (b3) the opt-in counter-based generator (Philox4x32-10 + Box-Muller instead of numpy's RandomState stream):
        #      other numbers than the reference draws, so never the headline
        try:
            for tr_ in tracers:
                tr_._ctx.sync()
                tr_._batch.set_rng("philox")
            dt_px, _, tr_px = timed_mode(tracers, n_sec)
            px_normals_ms = tracer._batch.profile_stage(2, 5)
        finally:
            for tr_ in tracers:
                tr_._ctx.sync()
                tr_._batch.set_rng("mt19937")
        return dict(traces_per_s=n_sec * E / dt_px, ms_per_step=1e3 * dt_px / n_sec, edges=E, normals_ms_per_ring=px_normals_ms,
                    trace_mse_vs_truth=float(np.mean([pkg.gpet_utils.trace_MSE(t_, truth) for t_ in tr_px])),
                    note="gpet_batch_set_rng(1): not the reference's random numbers; "
                         "tests/test_gpu_stages.py::test_full_trace_philox_mode_vs_oracle")

    def alone_ms(obj, reps=5):
        ts_ = []
        for _ in range(reps + 1):
            obj._ctx.sync()
            ta = time.time()
            obj.reset()
            obj.finish(obj.run_loop())
            obj._ctx.sync()
            ts_.append(1e3 * (time.time() - ta))
        return float(np.median(ts_[1:]))

    def close_all(objs):
        for o in objs:
            try:
                o._batch.close()
            except Exception:  # noqa: BLE001
                pass

    def sec_edges_256_and_32():
        # (c) BASELINE config 4's batch size: 256 edges per step (pipelined like the headline)
        small, e32s = [], []
        try:
            small = [make_tracer(256, tr_._ctx) for tr_ in tracers]
            dt2, it2, _ = timed_mode(small, 2 * n_sec)
            out256 = dict(traces_per_s=2 * n_sec * 256 / dt2, ms_per_step=1e3 * dt2 / (2 * n_sec), iterations=sorted(set(it2)),
                          in_flight=len(small))
            # (c2) config 4 AS STATED is 256 edges over 8 GPUs = 32 edges per GPU: one job, nothing to pipeline with.  One batch
            #      object alone on the GPU at 256 and at 32 edges (median of five steps, loop + converged fits back to back);
            #      t(256) / t(32) is the strong-scaling factor 8 GPUs can reach on that job (no collective on the path; the
            #      broadcast of the 1 MB image is reported as bcast_grad_ms at N > 1)
            t256 = alone_ms(small[0])
            e32s = [make_tracer(32, tracers[0]._ctx)]
            t32 = alone_ms(e32s[0])
            e32s += [make_tracer(32, tr_._ctx, sds=seeds_of(k_ + 1)[:32]) for k_, tr_ in enumerate(tracers[1:])]
            dt32, _, _ = timed_mode(e32s, 4 * n_sec)
            secondary["edges_32"] = dict(ms_per_step_alone=t32, traces_per_s_alone=32 / (1e-3 * t32),
                                         traces_per_s_in_flight=4 * n_sec * 32 / dt32, in_flight=len(e32s),
                                         edges_256_ms_per_step_alone=t256, predicted_strong_scaling_8_gpus=t256 / t32,
                                         note="BASELINE config 4 as stated: 256 edges / 8 GPUs = 32 per GPU; one batch object alone "
                                              "(loop + converged fits) beside the idle pipeline objects, median of 5; prediction = "
                                              "t(256 alone) / t(32 alone); config4_literal repeats it with the pipeline objects freed")
            return out256
        finally:
            close_all(e32s)
            close_all(small)

    def distinct_inputs(n_img, base_seed):
        imgs, inits_d = [], []
        for e in range(n_img):
            im_e, tr_e = synth_image(N, base_seed + e)
            imgs.append(pkg.gpet_utils.comp_grad_img(im_e, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx))
            inits_d.append(tr_e[[0, -1], :][:, [1, 0]])
        return imgs, inits_d

    def sec_distinct_images():
        # (d) distinct images and inits: 256 edges, every edge its own noise realisation of the image (own gradient image,
        #     own gradient KDE) and its own end points -- nothing shared through L2
        dist_tr = []
        try:
            imgs, inits_d = distinct_inputs(256, 1000)
            dist_tr = [make_tracer(256, tr_._ctx, images=imgs, inits=inits_d, sds=seeds[:256]) for tr_ in tracers[:2]]
            timed_steps(dist_tr, 2, min(depth, 1), executor, [])
            t1 = time.time()
            timed_steps(dist_tr, 6, min(depth, 1), executor, [])
            sync_all(dist_tr)
            dt3 = time.time() - t1
            return dict(traces_per_s=6 * 256 / dt3, ms_per_step=1e3 * dt3 / 6)
        finally:
            close_all(dist_tr)

    def sec_fresh_images_pipelined():
        # (e) a NEW image and new end points for every step, pipelined like the headline: batch objects of 256 edges, one image
        #     each (256 distinct gradient images per step), every step starts with set_frame (gpet_batch_set_images: upload,
        #     re-normalisation, gradient KDE of all 256 images; next_frame=False: unrelated images, nothing carried over).  The
        #     end points stay (a batch object's geometry is fixed at construction): what a server tracing one set of edge
        #     geometries through fresh images pays per step
        objs = []
        try:
            pools = [distinct_inputs(256, 3000 + 256 * k) for k in range(2)]
            objs = [make_tracer(256, tr_._ctx, images=pools[0][0], inits=pools[0][1], sds=seeds[:256]) for tr_ in tracers]
            W = len(objs)

            def worker(w):
                tr_, t_set = objs[w], 0.0
                for k in range(4):
                    ta = time.time()
                    tr_.set_frame(pools[(k + 1) % 2][0], next_frame=False)
                    t_set += time.time() - ta
                    tr_.finish(tr_.run_loop())
                return t_set
            list(executor.map(lambda w: (objs[w].reset(), objs[w].finish(objs[w].run_loop())), range(W)))
            sync_all(objs)
            t1 = time.time()
            t_sets = list(executor.map(worker, range(W)))
            sync_all(objs)
            dt = time.time() - t1
            return dict(traces_per_s=4 * W * 256 / dt, ms_per_step=1e3 * dt / (4 * W), in_flight=W, edges=256, steps=4 * W,
                        set_frame_ms_per_step=1e3 * float(np.mean(t_sets)) / 4,
                        note="every step: set_frame with 256 new gradient images (host -> device, re-normalisation, gradient KDE), then "
                             "loop + converged fits; %d objects in flight; compare edges_256_distinct_images (images resident) and "
                             "edges_256 (one shared image)" % W)
        finally:
            close_all(objs)

    def sec_default_kernel():
        # (f) the reference's DEFAULT kernel: GP_Edge_Tracing(init, grad) with kernel_options=(1, 3, 3) (gpet.py:25,139-151) is
        #     Matern-5/2 with sigma_f = M // 6, length_scale = edge_length // 2 -> a full-rank posterior covariance, the any-rank
        #     factor (csrc/gpet_eig.hip).  Same image, S, delta_x, pixel_thresh as the headline.  One edge and a batch of 256.
        kw = dict(README_KW)
        kw["kernel_options"] = (1, 3, 3)
        objs = []
        try:
            one_ = pkg.GP_Edge_Tracing_Batch([init], grad, [1], **kw, _ctx=ctx)
            objs.append(one_)
            one_()
            runs = []
            out_one = None
            for _ in range(3):
                one_.reset()
                ts = time.time()
                out_one = one_()
                runs.append(time.time() - ts)
            it_one = list(one_.timings["iters"])
            mse_one = float(pkg.gpet_utils.trace_MSE(out_one[0], truth))
            nb = 64
            bt_ = pkg.GP_Edge_Tracing_Batch([init] * nb, grad, seeds[:nb], **kw, _ctx=ctx)
            objs.append(bt_)
            bt_()
            tb = []
            for _ in range(2):
                bt_.reset()
                ts = time.time()
                out_b = bt_()
                tb.append(time.time() - ts)
            out = dict(config="kernel_options=(1, 3, 3): Matern nu=2.5, sigma_f=%d, length_scale=%d (gpet.py:139-151); otherwise the headline's edge"
                              % (N // 6, N // 2),
                       single_edge_ms_per_trace=1e3 * float(np.median(runs)), single_edge_iterations=it_one,
                       batch_edges=nb, batch_ms=1e3 * min(tb), batch_traces_per_s=nb / min(tb),
                       batch_trace_mse_vs_truth=float(np.mean([pkg.gpet_utils.trace_MSE(t_, truth) for t_ in out_b])),
                       single_trace_mse_vs_truth=mse_one)
            if not args.no_cpu_baseline:
                from oracle import gpet_oracle as orc
                from threadpoolctl import threadpool_limits
                img_o, edge_o = orc.synth_sinusoid_image(N, 3)
                g_o = orc.comp_grad_img(img_o, orc.kernel_builder((11, 5)))
                with threadpool_limits(limits=1):
                    t0_ = time.time()
                    et_o, _, info = orc.trace(edge_o[[0, -1], :][:, [1, 0]], g_o, per_curve=True, seed=1, **kw)
                    dt_ = time.time() - t0_
                out["cpu_port"] = dict(s_per_trace=dt_, iterations=info["n_iter"], cores=1,
                                       trace_mse_vs_truth=float(pkg.gpet_utils.trace_MSE(et_o, edge_o)),
                                       note="oracle/gpet_oracle.py, one BLAS thread, one trace (LAPACK's signs), constructor included; this "
                                            "kernel's length scale (N // 2 = 250 px) cannot follow the image's 125-px waves: the reference's "
                                            "default is not tuned for this image, the figure is about time")
                out["single_edge_speedup_vs_cpu_port"] = dt_ / float(np.median(runs))
            return out
        finally:
            close_all(objs)

    def sec_hw_queues_runtime_default():
        # (g) the headline's steps in a CHILD process with the HIP runtime's own default of four hardware queues (this process
        #     and the package run with eight: GPU_MAX_HW_QUEUES is read once, when the runtime initialises)
        import subprocess
        env = dict(os.environ, GPET_BENCH_HW_QUEUES="runtime-default")
        env.pop("GPU_MAX_HW_QUEUES", None)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(max(8, args.steps)), "--warmup", str(max(8, args.warmup)),
               "--edges", str(E), "--size", str(N), "--pipeline-depth", str(args.pipeline_depth), "--headline-only"]
        r_ = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r_.stdout.splitlines() if ln.startswith("{")]
        if r_.returncode != 0 or not lines:
            raise RuntimeError("child bench failed (rc %d): %s" % (r_.returncode, r_.stderr[-400:]))
        out = json.loads(lines[-1])
        out["note"] = "the same steps with GPU_MAX_HW_QUEUES = 4 (the HIP runtime's default) in a child process; the headline runs with 8"
        out["headline_over_this"] = value / out["value"]
        return out

    if world == 1 and not args.no_secondary:
        secondary = {}
        run_secondary("ctor_included", sec_ctor_included)
        run_secondary("no_pipeline", sec_no_pipeline)
        run_secondary("f32_samples", sec_f32_samples)
        run_secondary("philox_rng", sec_philox)
        run_secondary("edges_256", sec_edges_256_and_32)
        run_secondary("edges_256_distinct_images", sec_distinct_images)
        run_secondary("fresh_images_pipelined", sec_fresh_images_pipelined)
        log("secondary: " + ", ".join("%s %s" % (k, ("%.0f traces/s" % v["traces_per_s"]) if isinstance(v, dict) and "traces_per_s" in v else "(see entry)")
                                      for k, v in secondary.items()))
        release_pipeline_objects()
        run_secondary("hw_queues_runtime_default", sec_hw_queues_runtime_default)
        run_secondary("default_kernel_500", sec_default_kernel)
        run_secondary("config3", lambda: secondary_config3(pkg, ctx))
        run_secondary("config3_batch", lambda: secondary_config3(pkg, ctx, n_edges=8))
        run_secondary("config5", lambda: secondary_config5(pkg, ctx))
        run_secondary("config4_literal", lambda: config4_literal(pkg, L, dev_index, make_tracer, seeds))
        run_secondary("readme_literal", lambda: readme_literal(pkg, L, dev_index, ctx, init, grad, seeds, with_cpu=not args.no_cpu_baseline))
        for k in ("hw_queues_runtime_default", "default_kernel_500", "config3", "config3_batch", "config5", "config4_literal", "readme_literal"):
            v = secondary.get(k) or {}
            keys = [q for q in ("value", "single_edge_ms_per_trace", "batch_traces_per_s", "gp_iter_ms", "scoring_ms", "frames_per_s",
                                "whole_job_ms", "share_ms", "predicted_speedup_8_gpus", "error") if q in v]
            log("secondary %s: %s" % (k, ", ".join("%s=%s" % (q, ("%.3f" % v[q]) if isinstance(v[q], float) else v[q]) for q in keys)))

    # ---- one step alone (nothing else on the GPU): device time of the LML kernel launches of its converged fits,
    #      hipEvents around every launch on the fit stream (gpet_lml_stats)
    release_pipeline_objects()
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
    # the normals stage fills the batch's whole ring of upcoming iterations per launch (gpet_batch_info: 9 slots above 64
    # edges, 16 up to 64 edges): report per iteration
    ring = int(tracer._batch.info().get("z_ring", 16))
    def per_iter(d, ring_):
        d["normals"] = d["normals"] / ring_
        return d
    stage_ms = per_iter({name: tracer._batch.profile_stage(i, 20) for i, name in enumerate(STAGES)}, ring)
    info0 = tracer._batch.info()
    structured = bool(info0.get("structured", 0))
    kernel_ids = dict(KERNEL_IDS_STRUCT if structured else KERNEL_IDS_GENERIC)
    kernel_ids.update(KERNEL_IDS_COMMON)
    kernel_ms = {name: tracer._batch.profile_stage(kid, 20) for kid, name in kernel_ids.items()}
    # single edge (BASELINE config 2): latency view
    one = pkg.GP_Edge_Tracing_Batch([init], grad, [1], **README_KW, _ctx=ctx)
    one(); one.reset()
    single_runs = []
    for _ in range(5):  # (median of five: one trace is 12 ms of host-visible latency, a single sample carries its jitter)
        ts = time.time(); one(); single_runs.append(time.time() - ts)
        one.reset()
    single_s = float(np.median(single_runs))
    one.reset(); one._batch.iterate([1], 7)
    one_ms = per_iter({name: one._batch.profile_stage(i, 20) for i, name in enumerate(STAGES)}, int(one._batch.info().get("z_ring", 16)))

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
        "k_jacobi_ahead": dict(flops=6.0 * sweeps_mid * r ** 3, bytes=8.0 * (3 * r * r)),
        "k_struct_rows": dict(flops=2.0 * r * r * Lg, bytes=8.0 * (2 * r * Lg + r * r)),
        "k_sample_gemm_mfma_r": dict(flops=2.0 * S * Lg * r, bytes=8.0 * (S * r + r * Lg + S * Lg)),
        "k_score": dict(flops=60.0 * S * Lg, bytes=8.0 * S * Lg + 4.0 * M_ * N),
        "k_topk": dict(flops=2.0 * S * S, bytes=16.0 * S),
        "k_kde_prep": dict(flops=2.0 * nk * Lg, bytes=8.0 * nk * Lg),
        "k_kde_fused": dict(flops=36.0 * M_ * N + 10.0 * nk * Lg, bytes=8.0 * nk * Lg + 4.0 * M_ * N),
        # column scan of the pixel selection: the raw density of the band rows (4 B, ~1/3 of the rows at this state) and,
        # where it exceeds the threshold, the gradient KDE (4 B); ~30 flops per scanned pixel
        "k_pix_columns": dict(flops=30.0 * M_ * N / 3.0, bytes=2 * 4.0 * M_ * N / 3.0),
    }
    # the normals kernel fills a ring of `ring` iterations per launch on a side stream: per-iteration share
    # (a structured batch stores only the r0 leading normals of a row, rounded up to 4: its factors have no more rows)
    zc = ((info0["r0"] + 3) & ~3) if structured and info0.get("r0", 0) > 0 else info0["z_cols"]
    kernel_ms["k_mt_normals"] = stage_ms["normals"] * ring
    # algorithmic work of the generator: numpy's stream must be WALKED (every polar attempt decided) although only zc of Lg
    # columns are stored: S Lg / 2 accepted pairs at an acceptance of pi / 4, four MT19937 words per attempt, ~15 32-bit vector
    # operations per word (twist 7, tempering of the two words the accept test needs 5, the test itself 3); bytes = what it stores
    mt_words = 4.0 * (S * Lg / 2.0) / (np.pi / 4.0)
    alg["k_mt_normals"] = dict(flops=15.0 * mt_words * ring, bytes=8.0 * S * zc * ring)
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
    import glob
    tps = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))  # (the latest round's PMC passes)
    tp = tps[-1] if tps else ""
    if tp and os.path.exists(tp):
        prof = json.load(open(tp))
        if prof.get("edges") == E and prof.get("image") == [N, N] and dom in prof.get("kernels", {}):
            traffic = prof["kernels"][dom]["hbm_bytes_per_launch"]
    # the generator is neither HBM- nor MFMA-shaped: integer work on the vector ALUs (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz)
    VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12
    is_valu = dom == "k_mt_normals"
    if is_valu:
        use_flops = False  # (the contract's two roofs: of them HBM -- the bytes it must store -- is the one that applies)
    roofline = dict(kernel=dom, bound="mfma" if use_flops else "hbm",
                    achieved=tfl if use_flops else gbs, peak=FP64_PEAK_TFLOPS if use_flops else HBM_PEAK_GBS,
                    unit="TFLOP/s" if use_flops else "GB/s",
                    frac=(tfl / FP64_PEAK_TFLOPS) if use_flops else (gbs / HBM_PEAK_GBS), traffic=traffic,
                    valu=(dict(achieved=tfl, peak=VALU_PEAK_TOPS, unit="Top/s (32-bit vector operations)", frac=tfl / VALU_PEAK_TOPS,
                               note="the roof that binds this kernel: integer work on the vector ALUs, ~15 operations per MT19937 "
                                    "word of the stream it has to walk") if is_valu else None),
                    launch_ms=d_ms, edges_per_launch=E, algorithmic_bytes=a_bytes, algorithmic_flops=a_flops,
                    launches_per_step=(lml["launches"] if dom == "k_lml" else iters_per_trace),
                    device_ms_per_step={k: v for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
                    note=("f64 vector/matrix peak (equal on MI355X); this kernel is an LDS-resident eigen-solver: its practical "
                          "bound is LDS store bandwidth and barrier latency, see DESIGN.md section 6" if dom == "k_jacobi_ahead" else
                          ("bound by the samples it stores (8 S Lg bytes per edge = 4.1 GB per launch, rows padded to 128 bytes): a kernel "
                           "that ONLY stores them in the same row-tile shape takes 1.13-1.19 ms = 3.4-3.6 TB/s "
                           "(profiles/r03_gemm_store_counters.txt, tools/ubench/hbm_write.hip) -- the one store-only ceiling, also quoted in "
                           "DESIGN.md section 6; the matrix instructions alone 1.12 ms; this kernel overlaps the two to 1.5 ms (1.8 in round "
                           "3: line-aligned rows, 16-byte stores of column pairs, no wait for the stores anywhere in the loop)"
                           if dom == "k_sample_gemm_mfma_r" else
                          ("numpy's RandomState(seed).standard_normal stream, bit for bit: MT19937 + polar method, every attempt of the "
                           "S x Lg stream decided, zc of Lg columns stored; vector-ALU bound (integer; the `valu` entry), not HBM or MFMA: "
                           "DESIGN.md section 6b; the sample GEMM (next in device time) is at %.2f of the f64 MFMA peak"
                           % (alg["k_sample_gemm_mfma_r"]["flops"] * E / (kernel_ms["k_sample_gemm_mfma_r"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS)
                           if dom == "k_mt_normals" else
                          ("objective of the converged fits: %d evaluations of ~%.0f-point problems in %d launches per step; "
                           "f64 peak (vector = matrix on MI355X); k_lml16: block-4 sweep on v_mfma_f64_16x16x4, two waves per problem -- MFMA "
                           "cycles and VALU issue share the vector unit's FMA lanes, DESIGN.md section 6b"
                           % (lml["evaluations"], n_fit, lml["launches"]) if dom == "k_lml" else None)))),
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
                   "edges_per_gpu": E, "steps_in_flight_per_gpu": n_in_flight, "image": [N, N],
                   "seeds": "1 + 997 * ((rank * objects + object) * edges + edge): every (edge, iteration) its own RandomState stream", "iterations_per_trace": iters[:4],
                   "final_fit": "device-resident: standardisation, 13 starts, L-BFGS-B state machines and the batched LML objective "
                                "all on the GPU (gpet_final_fit_all); no host workers"},
        "host": {"cpus_usable": usable_cpus(), "cpus_machine": os.cpu_count(), "lbfgs_workers": 0,
                 "host_threads": depth + 1, "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "note": "the host only enqueues launches and waits: one driver thread per batch object "
                 "in flight (its device loop, then its converged fits)"},
        "secondary": secondary,
        "gp_iter_ms": {"batch_of_%d" % E: sum(stage_ms[k] for k in STAGES[:4]),
                       "single_edge": sum(one_ms[k] for k in STAGES[:4])},
        "stage_ms_batch": stage_ms, "stage_ms_single_edge": one_ms,
        "time_split_s": {"elapsed": elapsed, "steps_in_flight": n_in_flight, "pipelined": pipeline, "pipeline_depth": depth,
                         "loop_wall_per_thread": loop_s, "fit_wall_per_thread": fit_s,
                         "loop_wall_mean": (sum(loop_walls) / len(loop_walls)) if loop_walls else None,
                         "fit_wall_mean": (sum(fit_walls) / len(fit_walls)) if fit_walls else None,
                         "note": "wall times of a step's device loop and converged fits while the other objects' steps run beside them"},
        "single_edge": {"traces_per_s": 1.0 / single_s, "ms_per_trace": 1e3 * single_s,
                        "runs_ms": [round(1e3 * v, 3) for v in single_runs], "note": "median of five traces of one edge"},
        "trace_mse_vs_truth": mse, "bcast_grad_ms": 1e3 * t_b, "config4_literal_measured": c4_measured,
        "roofline": roofline, "cpu_baseline": cpu,
    }
    if cpu:
        out["speedup_vs_cpu_port"] = value / cpu["value"]
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
