#!/usr/bin/env python3
"""T3 as a distribution on INDEPENDENT RNG seeds: quality of whole traces on the README configuration
over 240 seeds spaced 997 apart (iteration k of seed s draws from RandomState(s + k + 1), gpet.py:839: seeds closer than
a trace's iteration count share normal streams), image seeds {1, 3}:

  * ``ref_quality``     -- round 3's rows, kept: the UNMODIFIED reference on 24 consecutive + 48 spaced seeds, run with
                           OpenBLAS's default thread count (8 here);
  * ``ref_quality_t1``  -- the unmodified reference (/root/reference through ref_harness.py) on the first 60 spaced seeds
                           with ONE BLAS thread;
  * ``oracle_quality``  -- the oracle (oracle/gpet_oracle.py) on all 240 seeds: convention 0 = LAPACK's own eigenvector
                           signs with one BLAS thread (= ``ref_quality_t1`` row by row: asserted here), 1 = the library's
                           "harmonic" sign convention (what the device reproduces bit for bit), 2 = LAPACK's signs with
                           EIGHT BLAS threads on the first 60 seeds (= ``ref_quality``'s spaced rows: counted here);
                           3 / 4 = the harmonic convention again with TWO / FOUR BLAS threads on all 240 seeds: the
                           oracle perturbed by nothing but its own rounding (another partition of every dot product,
                           another draw of LAPACK's noise in the ~430 numerically-zero singular directions).  A seed
                           whose rows 1, 3 and 4 are identical is DECIDED: the GPU test demands the device's trace to
                           be that row, no exception; on the few others the trace hangs on a near-tie that rounding
                           noise settles, and the device must give one of the three answers.
LAPACK's singular-vector signs are implementation-defined -- and they depend on the THREAD COUNT of the BLAS underneath:
the same reference code on the same seed traces a different edge with 1, 4 and 8 threads (image 1, seed 1000: MSE 907,
522, 6 490).  A sign convention is therefore not something the reference has; its traces are one draw per environment.

Runs only in the build container (it imports the reference); the .npz (a few KB) is committed.
    python tests/golden/make_quality_fixture.py [workers] [--redo-harmonic]
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings("ignore")

README = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}, noise_y=1, N_samples=1000, score_thresh=1,
              delta_x=5, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
SPACED = [1000 + 997 * k for k in range(240)]
REF_SEEDS = list(range(1, 25)) + SPACED[:60]
IMG_SEEDS = (1, 3)


def _metrics(et, edge):
    from oracle import gpet_oracle as orc  # (the three formulas of gpet_utils.py:256-313, restated there)
    n = et.shape[0]
    mse = float(np.round((1 / n) * np.sum((et[:, 0] - edge[:, 0]) ** 2), 4))
    rows = np.arange(n)[:, None]
    pb = (rows >= et[:, 0].astype(int)[None, :]).astype(float)
    tb = (rows >= edge[:, 0].astype(int)[None, :]).astype(float)
    jacc = np.sum(pb * tb) / np.sum(np.clip(pb + tb, 0, 1))
    dice = float(np.round(2 * jacc / (jacc + 1), 4))
    ta = np.sum(n - edge[:, 0]) / n ** 2
    pa = np.sum(n - et[:, 0]) / n ** 2
    return mse, dice, float(np.round(np.abs((ta - pa) / ta), 5))


def _oracle_one(args):
    img_seed, seed, conv = args
    threads = {0: 1, 1: 1, 2: 8, 3: 2, 4: 4}[conv]
    from oracle import gpet_oracle as orc
    img, edge = orc.synth_sinusoid_image(500, img_seed)
    grad = orc.comp_grad_img(img, orc.kernel_builder((11, 5)))
    init = edge[[0, -1], :][:, [1, 0]]
    et, _, info = orc.trace(init, grad, seed=seed, sign_convention="harmonic" if conv in (1, 3, 4) else None, blas_threads=threads,
                            **README)
    return (img_seed, seed, conv, info["n_iter"]) + _metrics(et, edge)


def _reference_one(args):
    img_seed, seed = args
    import ref_harness
    ref_harness.load_reference()
    from gp_edge_tracing import gpet, gpet_utils
    from threadpoolctl import threadpool_limits
    from oracle import gpet_oracle as orc
    img, edge = orc.synth_sinusoid_image(500, img_seed)
    grad = gpet_utils.comp_grad_img(img, gpet_utils.kernel_builder(size=(11, 5), unit=False))
    init = edge[[0, -1], :][:, [1, 0]]
    with threadpool_limits(limits=1):
        tr = gpet.GP_Edge_Tracing(init, grad, seed=seed, return_std=False, **README)
        et, (all_samples, all_obs, iter_curves) = tr(return_lines=True)
    return (img_seed, seed, len(all_obs) - 2, float(gpet_utils.trace_MSE(et, edge)), float(gpet_utils.trace_dicecoef(et, edge)),
            float(gpet_utils.trace_relarea(et, edge)))


def main():
    import multiprocessing as mp
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    workers = int(args[0]) if args else 6
    path = os.path.join(HERE, "quality_rbf500.npz")
    old = dict(np.load(path)) if os.path.exists(path) else {}
    ref_old = old.get("ref_quality", np.zeros((0, 6)))
    have_ref = old.get("ref_quality_t1", np.zeros((0, 6)))
    have_orc = old.get("oracle_quality", np.zeros((0, 7)))
    if "--redo-harmonic" in sys.argv:
        have_orc = have_orc[have_orc[:, 2] != 1]

    def save(ref, orc_rows):
        np.savez_compressed(path, ref_quality=ref_old, columns=np.array(["img_seed", "seed", "n_iter", "mse", "dice", "relarea"]),
                            ref_quality_t1=ref, oracle_quality=orc_rows,
                            oracle_columns=np.array(["img_seed", "seed", "convention (0 LAPACK 1 thread, 1 harmonic 1 thread, 2 LAPACK 8 threads, 3 harmonic 2 threads, 4 harmonic 4 threads)",
                                                     "n_iter", "mse", "dice", "relarea"]),
                            kde_standin=1)

    done_ref = {(int(r[0]), int(r[1])) for r in have_ref}
    done_orc = {(int(r[0]), int(r[1]), int(r[2])) for r in have_orc}
    jobs_ref = [(a, b) for a in IMG_SEEDS for b in SPACED[:60] if (a, b) not in done_ref]
    jobs_orc = [(a, b, c) for c in (1, 0) for a in IMG_SEEDS for b in SPACED if (a, b, c) not in done_orc]
    # (two / four BLAS threads per run: fewer workers at a time)
    jobs_orc_mt = [(a, b, c) for c in (3, 4) for a in IMG_SEEDS for b in SPACED if (a, b, c) not in done_orc]
    # (eight BLAS threads per run: one run at a time, on the 60-seed subset -- five such workers oversubscribe this machine)
    jobs_orc8 = [(a, b, 2) for a in IMG_SEEDS for b in SPACED[:60] if (a, b, 2) not in done_orc]
    print("reference runs to do: %d, oracle runs: %d (+ %d with 8 BLAS threads), %d workers"
          % (len(jobs_ref), len(jobs_orc), len(jobs_orc8), workers), flush=True)
    ref, orc_rows = have_ref, have_orc
    with mp.get_context("fork").Pool(workers) as pool:
        for i0 in range(0, len(jobs_orc), 120):  # (saved as it goes)
            rows = pool.map(_oracle_one, jobs_orc[i0:i0 + 120], chunksize=4)
            orc_rows = np.asarray(sorted([tuple(r) for r in orc_rows] + rows), dtype=np.float64).reshape(-1, 7)
            save(ref, orc_rows)
            print("oracle runs done: %d" % min(len(jobs_orc), i0 + 120), flush=True)
        for i0 in range(0, len(jobs_ref), 30):
            rows = pool.map(_reference_one, jobs_ref[i0:i0 + 30], chunksize=1)
            ref = np.asarray(sorted([tuple(r) for r in ref] + rows), dtype=np.float64).reshape(-1, 6)
            save(ref, orc_rows)
            print("reference runs done: %d" % min(len(jobs_ref), i0 + 30), flush=True)
    with mp.get_context("fork").Pool(max(1, workers // 3)) as pool:
        for i0 in range(0, len(jobs_orc_mt), 60):
            rows = pool.map(_oracle_one, jobs_orc_mt[i0:i0 + 60], chunksize=2)
            orc_rows = np.asarray(sorted([tuple(r) for r in orc_rows] + rows), dtype=np.float64).reshape(-1, 7)
            save(ref, orc_rows)
            print("oracle runs with 2 / 4 BLAS threads done: %d of %d" % (min(len(jobs_orc_mt), i0 + 60), len(jobs_orc_mt)), flush=True)
    for i0 in range(0, len(jobs_orc8), 20):
        rows = [_oracle_one(j) for j in jobs_orc8[i0:i0 + 20]]
        orc_rows = np.asarray(sorted([tuple(r) for r in orc_rows] + rows), dtype=np.float64).reshape(-1, 7)
        save(ref, orc_rows)
        print("oracle runs with 8 BLAS threads done: %d" % min(len(jobs_orc8), i0 + 20), flush=True)
    # the oracle under LAPACK's signs with one BLAS thread IS the reference with one BLAS thread, seed by seed
    lut0 = {(int(r[0]), int(r[1])): r[3:] for r in orc_rows if int(r[2]) == 0}
    lut2 = {(int(r[0]), int(r[1])): r[3:] for r in orc_rows if int(r[2]) == 2}
    n_cmp = 0
    for r in ref:
        k = (int(r[0]), int(r[1]))
        if k in lut0:
            assert np.array_equal(lut0[k], r[2:]), (k, lut0[k], r[2:])
            n_cmp += 1
    print("oracle (LAPACK signs, 1 BLAS thread) == unmodified reference (1 BLAS thread) on %d (image, seed) pairs: "
          "n_iter, MSE, DICE, rel. area identical" % n_cmp)
    n8 = sum(1 for r in ref_old if (int(r[0]), int(r[1])) in lut2)
    e8 = sum(1 for r in ref_old if (int(r[0]), int(r[1])) in lut2 and np.array_equal(lut2[(int(r[0]), int(r[1]))], r[2:]))
    print("oracle (LAPACK signs, 8 BLAS threads) == round 3's reference rows (default threads) on %d of %d pairs" % (e8, n8))
    both = [k for k in lut0 if k in lut2]
    t1_vs_t8 = sum(1 for k in both if np.array_equal(lut0[k], lut2[k]))
    print("LAPACK signs, 1 thread vs 8 threads: the same trace quality on %d of %d (image, seed) pairs" % (t1_vs_t8, len(both)))
    for a in IMG_SEEDS:
        rows = {c: {int(r[1]): tuple(r[3:]) for r in orc_rows if int(r[0]) == a and int(r[2]) == c} for c in (1, 3, 4)}
        if rows[3] and rows[4]:
            und = [sd for sd in rows[1] if not (rows[1][sd] == rows[3].get(sd) == rows[4].get(sd))]
            print("image seed %d: harmonic rows with 1 / 2 / 4 BLAS threads identical on %d of %d seeds; undecided: %s" % (a, len(rows[1]) - len(und), len(rows[1]), und))
    for a in IMG_SEEDS:
        for c in (0, 2, 1):
            r = orc_rows[(orc_rows[:, 0] == a) & (orc_rows[:, 2] == c)]
            if not len(r):
                continue
            good = r[:, 4] < 2000.0
            print("image seed %d, %s, %d seeds: good-branch fraction %.3f (s.e. %.3f), MSE quartiles %s, iterations %d..%d"
                  % (a, ["LAPACK signs 1 thread", "harmonic signs", "LAPACK signs 8 threads"][c], len(r), good.mean(),
                     np.sqrt(good.mean() * (1 - good.mean()) / len(r)), np.percentile(r[:, 4], [25, 50, 75]).round(0), r[:, 3].min(), r[:, 3].max()))


if __name__ == "__main__":
    main()
