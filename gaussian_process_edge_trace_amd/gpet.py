"""Drop-in ``GP_Edge_Tracing`` whose arithmetic runs in libgpet_hip.so on an MI355X.

Mirrors the constructor / ``__call__`` surface of the reference
(``gp_edge_tracing/gpet.py:22-35, 768-773``): same argument names, order, defaults, silent
clamping (gpet.py:95-119) and return conventions (gpet.py:902-908).  The per-iteration
stages keep the reference's method names -- ``fit_predict_GP`` (gpet.py:182),
``get_best_curves`` (:414), ``cost_funct`` (:371), ``get_best_pixels`` (:622) -- and are thin
calls into the C ABI (include/gpet_hip.h).  There is no CPU fallback: without the HIP
library / a GPU the constructor raises.
"""
from __future__ import annotations

import time as t

import numpy as np

from . import _lib
from . import gpet_utils


def resolve_params(init, grad_shape, kernel_options=(1, 3, 3), noise_y=1, obs=np.array([], dtype=np.int8),
                   N_samples=500, score_thresh=1, delta_x=20, keep_ratio=0.1, pixel_thresh=5, seed=42,
                   return_std=False, fix_endpoints=True):
    """Host-side parameter clamping and derived sizes of the reference constructor
    (gpet.py:95-151), including its quirks: x_st/x_en come from the UNSORTED ``init``
    (gpet.py:96), ``N_keep`` uses the raw ``keep_ratio*N_samples`` (gpet.py:118), the 3-tuple
    presets index with ``[opt-1]`` (gpet.py:146,150), the dict needs key 'kernel' (gpet.py:133)."""
    init = np.asarray(init)
    p = dict(noise_y=noise_y, seed=seed, return_std=return_std, fix_endpoints=fix_endpoints)
    p["init"] = init[np.argsort(init[:, 0])].astype(int)
    p["x_st"], p["x_en"] = int(init[0, 0]), int(init[-1, 0])
    p["N_samples"] = int(N_samples) if N_samples > 100 else 1000
    p["obs"] = np.asarray(obs).reshape(-1, 2).astype(np.int64)
    p["keep_ratio"] = float(keep_ratio) if 0 < keep_ratio <= 1 else 0.1
    p["pixel_thresh"] = int(pixel_thresh) if pixel_thresh >= 2 else 2
    p["score_thresh"] = float(score_thresh) if 0 < score_thresh <= 1 else 1
    p["delta_x"] = int(delta_x) if delta_x > 3 else 2
    p["half_delta"] = p["delta_x"] // 2
    p["N_inits"] = p["init"].shape[0]
    p["M"], p["N"] = grad_shape
    p["x_grid"] = p["x_st"] + np.arange(p["x_en"] - p["x_st"] + 1).astype(int)
    p["edge_length"] = p["x_grid"].shape[0]
    p["N_subints"] = int(p["edge_length"] // p["delta_x"])
    p["N_keep"] = int(keep_ratio * N_samples)
    p["algo_thresh"] = p["N_subints"] - (p["pixel_thresh"] - 1)
    if type(kernel_options) == dict:
        p["sigma_f"] = kernel_options["sigma_f"]
        p["sigma_l"] = kernel_options["length_scale"]
        p["kernel_type"] = kernel_options["kernel"]
        p["kernel_nu"] = kernel_options["nu"] if kernel_options["kernel"] == "Matern" else 2.5
    else:
        rbf_matern, sigmaf_opt, sigmal_opt = kernel_options
        p["kernel_type"] = ["RBF", "Matern"][int(rbf_matern > 0)]
        p["kernel_nu"] = [2.5, 1.5][int(rbf_matern > 1)]
        sf = [10, 8, 6, 4, 2, 1][sigmaf_opt - 1] if (sigmaf_opt >= 0) and (sigmaf_opt <= 5) else 1
        p["sigma_f"] = p["M"] // sf
        sl = [1, 4 / 3, 2, 4, 10][sigmal_opt - 1] if (sigmal_opt >= 0) and (sigmal_opt <= 4) else 10
        p["sigma_l"] = p["edge_length"] // sl
    return p


def auto_factor_cap(p):
    """Rows kept by the eigen-factor sampler.  The RBF correlation matrix on a unit grid has
    eigenvalues ~ exp(-(pi k l / N)^2 / 2): they fall below 1e-14 of the largest at
    k ~ 2.6 N / l, and the posterior covariance cannot have higher rank.  <= 96 rows take the
    LDS-resident Jacobi path; Matern spectra decay polynomially, so keep every row."""
    Lg = p["edge_length"]
    if p["kernel_type"] != "RBF":
        return Lg
    est = int(2.6 * Lg / float(p["sigma_l"])) + 12
    return 0 if est <= 96 else min(Lg, int(est * 1.2))


def to_abi_params(p, obs_cap=None, factor_cap=0, z_cols=0):
    """gpet_params (include/gpet_hip.h) from the resolved constructor state."""
    if not factor_cap:
        factor_cap = auto_factor_cap(p)
    q = _lib.GpetParams()
    q.kernel_type = _lib.KERNEL_MATERN if p["kernel_type"] == "Matern" else _lib.KERNEL_RBF
    q.nu = float(p["kernel_nu"])
    if q.kernel_type == _lib.KERNEL_MATERN and np.isinf(q.nu):  # sklearn's Matern(nu=inf) IS the RBF kernel
        q.kernel_type, q.nu = _lib.KERNEL_RBF, 2.5
    q.sigma_f = float(p["sigma_f"])
    q.length_scale = float(p["sigma_l"])
    q.noise_y = float(p["noise_y"])
    q.n_samples = int(p["N_samples"])
    q.n_keep = int(p["N_keep"])
    q.delta_x = int(p["delta_x"])
    q.pixel_thresh = int(p["pixel_thresh"])
    q.score_thresh = float(p["score_thresh"])
    q.fix_endpoints = 1 if p["fix_endpoints"] else 0
    q.x_st, q.x_en = int(p["x_st"]), int(p["x_en"])
    q.n_init = int(p["N_inits"])
    n_bins = (p["edge_length"] - 1) // p["delta_x"] + 3
    q.obs_cap = int(obs_cap if obs_cap is not None else max(n_bins, p["obs"].shape[0] + 1))
    q.factor_cap = int(factor_cap)
    q.z_cols = int(z_cols)
    q.jitter = 1e-6  # gpet.py:155
    return q


class GP_Edge_Tracing(object):
    """Traces one edge with Gaussian-process regression on the GPU (gpet.py:17-35).

    The seam methods of the reference class are device calls on ONE resident state (observation set, samples, scores,
    KDE, score threshold).  Unlike the reference's pure functions, ``fit_predict_GP(obs)``, ``compute_new_obs(...,
    pre_fobs)`` and ``get_best_pixels(..., pre_fobs=...)`` leave the observation set they were given on the device, and
    ``get_best_curves(y_samples)`` the samples: a later argument-less call continues from there.  ``__call__`` starts
    from ``self.obs`` again (it resets the device state first), so a trace is not affected by earlier seam calls.

    Keywords beyond the reference's signature (keyword-only): ``device``, ``stream``, ``factor_cap``, ``z_cols``,
    ``sample_dtype`` ("f32": samples stored in single precision) and ``rng`` ("philox": counter-based generator) --
    the last two are opt-in modes outside the reference-parity statements."""

    def __init__(self, init, grad_img, kernel_options=(1, 3, 3), noise_y=1, obs=np.array([], dtype=np.int8),
                 N_samples=500, score_thresh=1, delta_x=20, keep_ratio=0.1, pixel_thresh=5, seed=42,
                 return_std=False, fix_endpoints=True, *, device=0, stream=None, factor_cap=0, z_cols=0,
                 sample_dtype=None, rng=None, _ctx=None):
        p = resolve_params(init, np.asarray(grad_img).shape, kernel_options, noise_y, obs, N_samples, score_thresh,
                           delta_x, keep_ratio, pixel_thresh, seed, return_std, fix_endpoints)
        self._p = p
        for k in ("init", "x_st", "x_en", "noise_y", "N_samples", "obs", "seed", "keep_ratio", "pixel_thresh",
                  "score_thresh", "delta_x", "half_delta", "return_std", "fix_endpoints", "N_inits", "M", "N",
                  "x_grid", "edge_length", "N_subints", "N_keep", "algo_thresh", "sigma_f", "sigma_l",
                  "kernel_type", "kernel_nu"):
            setattr(self, k, p[k])
        self.kde_thresh = 1e-3
        self.X = np.repeat(self.x_grid.reshape(-1, 1), self.N_samples, axis=-1)
        self._ctx = _ctx if _ctx is not None else _lib.Context(device, stream)
        # the library re-normalises the gradient image in float32 (gpet.py:97)
        g32 = np.asarray(grad_img).astype(np.float32)
        self._abi = to_abi_params(p, factor_cap=factor_cap, z_cols=z_cols)
        self._batch = _lib.Batch(self._ctx, [g32], [self._abi], [p["init"]])
        if sample_dtype is not None:  # (keyword beyond the reference's signature: "f32" stores the samples in single precision)
            self._batch.set_sample_dtype(sample_dtype)
        if rng is not None:  # ("philox": the counter-based generator instead of numpy's RandomState stream)
            self._batch.set_rng(rng)
        self.grad_img = self._batch.read(_lib.BUF_GRAD).astype(np.float64)
        self._n_iter = 0

    # ---- gpet.py:182-268 ---------------------------------------------------------------
    def fit_predict_GP(self, obs, converged=False, seed=0):
        """Not-converged branch: returns ``N_samples`` posterior curves, shape (N, N_samples),
        exactly like the reference (a transposed view of the row-per-sample device buffer)."""
        b = self._batch
        if converged:
            # gpet.py:232-248,262-266: hyper-parameters optimised (1 + 12 L-BFGS-B starts), returns (mean in pixels,
            # std in standardised units -- the reference does not rescale it)
            obs = np.asarray(obs).reshape(-1, 2).astype(np.int64)
            fits, _ = device_final_fits(b, [dict(self._p, seed=int(seed))], [obs], [0])
            y_mean_optim, y_std, self._theta = fits[0]
            return y_mean_optim, y_std
        b.set_obs(0, obs)
        b.fit_predict(want_cov=True)
        b.factor()
        b.normals([seed])
        b.sample()
        return b.read(_lib.BUF_SAMPLES).T

    # ---- gpet.py:371-451 ---------------------------------------------------------------
    def get_best_curves(self, y_samples=None):
        """Scores the samples currently on the device (``y_samples`` given => they are
        uploaded first) and returns (best_curves, best_costs, (optimal_curve, optimal_cost))."""
        b = self._batch
        if y_samples is not None:
            b.write(_lib.BUF_SAMPLES, np.ascontiguousarray(np.asarray(y_samples, dtype=np.float64).T))
        b.score()
        idx = b.read(_lib.BUF_BEST_IDX)
        costs = b.read(_lib.BUF_BEST_COSTS)
        Y = b.read(_lib.BUF_SAMPLES)
        curves = np.stack((np.repeat(self.x_grid.reshape(-1, 1), idx.shape[0], axis=-1).astype(np.float64),
                           Y[idx].T), axis=-1)
        return curves, costs, (curves[:, 0, :], costs[0])

    def cost_funct(self, edge):
        """Cost of one curve given as (N, 2) xy (gpet.py:371-410).  The device scores whole sample sets, so the curve
        takes the place of sample 0 for one scoring pass; the samples and the scores of the set are restored after it."""
        edge = np.asarray(edge, dtype=np.float64)
        edge = edge[edge[:, 0].argsort(), :]
        b = self._batch
        have = b.have_scores
        Y = b.read(_lib.BUF_SAMPLES)
        saved = (Y[0].copy(), b.read(_lib.BUF_COSTS), b.read(_lib.BUF_BEST_IDX), b.read(_lib.BUF_BEST_COSTS))
        Y[0] = edge[:, 1]
        b.write(_lib.BUF_SAMPLES, Y)
        b.score()
        cost = float(b.read(_lib.BUF_COSTS)[0])
        Y[0] = saved[0]
        b.write(_lib.BUF_SAMPLES, Y)
        if have:
            b.write(_lib.BUF_COSTS, saved[1])
            b.write(_lib.BUF_BEST_IDX, saved[2])
            b.write(_lib.BUF_BEST_COSTS, saved[3])
        return cost

    # ---- gpet.py:455-662 ---------------------------------------------------------------
    def _upload_best(self, best_curves, costs):
        """A caller's own best curves (N, n_keep, 2) xy + costs (n_keep,) become samples 0..n_keep-1 of the device's
        sample set, selected in that order."""
        bc = np.asarray(best_curves, dtype=np.float64)
        costs = np.asarray(costs, dtype=np.float64).reshape(-1)
        b = self._batch
        inf = b.info()
        if bc.ndim != 3 or bc.shape[0] != inf["Lg"] or bc.shape[1] != inf["n_keep"] or costs.shape[0] != inf["n_keep"]:
            raise ValueError("best_curves must be (edge_length, N_keep, 2) = (%d, %d, 2) with N_keep costs"
                             % (inf["Lg"], inf["n_keep"]))
        if not np.array_equal(bc[:, 0, 0], self.x_grid.astype(np.float64)):
            raise ValueError("best_curves must be sampled on the tracer's x-grid")
        Y = b.read(_lib.BUF_SAMPLES)
        Y[:inf["n_keep"]] = bc[:, :, 1].T
        b.write(_lib.BUF_SAMPLES, Y)
        b.write(_lib.BUF_BEST_IDX, np.arange(inf["n_keep"], dtype=np.int32))
        b.write(_lib.BUF_BEST_COSTS, costs)

    def kernel_density_estimate(self, best_curves=None, costs=None):
        """(M, N) float64 density image, min-max normalised in float32 (gpet.py:455-529): of the gradient image when
        called without curves (what the constructor stores as ``grad_kde``, gpet.py:127), else the cost-weighted KDE of
        ``best_curves`` (N, N_keep, 2) with ``costs`` (N_keep,)."""
        b = self._batch
        if best_curves is None:
            return b.read(_lib.BUF_GRAD_KDE).astype(np.float64)
        self._upload_best(best_curves, costs)
        b.curve_kde()
        return b.read(_lib.BUF_KDE).astype(np.float64)

    def compute_new_obs(self, pixel_idx, kde_arr, pre_fobs):
        """Pixel scoring, threshold decay, binning and per-bin argmax (gpet.py:532-618) on ``kde_arr``.
        ``pre_fobs`` is in yx order like in the reference; the new observation set comes back as xy int64.
        ``pixel_idx`` is implied by ``kde_arr`` (gpet.py:651-657); a different candidate list is rejected."""
        kde_arr = np.asarray(kde_arr)
        cand = np.argwhere(kde_arr > self.kde_thresh)
        if self.fix_endpoints:
            cand = cand[(cand[:, 1] > self.x_st) & (cand[:, 1] < self.x_en)]
        if pixel_idx is not None and not np.array_equal(np.asarray(pixel_idx), cand):
            raise ValueError("pixel_idx must be the candidates get_best_pixels derives from kde_arr (gpet.py:651-657)")
        b = self._batch
        thresh = b.scalars().score_thresh
        pre = np.asarray(pre_fobs).reshape(-1, 2)[:, [1, 0]].astype(np.int64)
        b.set_obs(0, pre)
        sc = b.scalars()
        sc.score_thresh = thresh
        b.write_scalars(sc)
        b.write(_lib.BUF_KDE, kde_arr.astype(np.float32))
        b.select_pixels_only()
        self.score_thresh = b.scalars().score_thresh
        return b.read(_lib.BUF_OBS)

    def get_best_pixels(self, best_curves=None, costs=None, pre_fobs=None):
        """KDE of the best curves + pixel scoring / binning / non-max suppression on the device (gpet.py:622-662).
        Arguments given => they are what is used (``pre_fobs`` in yx order, as the reference passes it); arguments
        omitted => the curves scored last and the observation set already on the device.  Returns the new
        observation set (xy int64)."""
        b = self._batch
        if pre_fobs is not None:
            thresh = b.scalars().score_thresh
            b.set_obs(0, np.asarray(pre_fobs).reshape(-1, 2)[:, [1, 0]].astype(np.int64))
            sc = b.scalars()
            sc.score_thresh = thresh
            b.write_scalars(sc)
        if best_curves is not None:
            self._upload_best(best_curves, costs)
        b.select_pixels()
        self.score_thresh = b.scalars().score_thresh
        return b.read(_lib.BUF_OBS)

    def __call__(self, print_final_diagnostics=False, show_init_post=False, show_post_iter=False, verbose=False,
                 return_lines=False, max_iter=1000):
        """Runs the trace (gpet.py:768-908).  The while-loop of the reference (gpet.py:829-870)
        lives on the device; the host only polls the per-edge ``done`` flag."""
        all_samples, all_obs = [], [self.obs]
        iter_optimal_curves, iter_optimal_costs = [], []
        if show_init_post or show_post_iter or print_final_diagnostics:
            import warnings
            warnings.warn("plotting flags are accepted for API compatibility but ignored by the GPU tracer")
        alg_st = t.time()
        b = self._batch
        b.set_obs(0, self.obs)
        n_iter = 0
        while not b.scalars().done:
            if n_iter >= max_iter:
                raise _lib.GpetError(_lib.ERR_ITER_CAP, f"trace did not converge in {max_iter} iterations")
            step = min(1 if (return_lines or verbose) else 4, max_iter - n_iter)
            st = t.time()
            if verbose:
                print('Fitting Gaussian process and computing next set of observations...')
            b.iterate([self.seed], step)
            s = b.scalars()
            n_iter = s.iter
            if return_lines:
                all_samples.append(b.read(_lib.BUF_SAMPLES).T)
                idx = b.read(_lib.BUF_BEST_IDX)
                iter_optimal_curves.append(np.stack((self.x_grid.astype(np.float64), all_samples[-1][:, idx[0]]), -1))
                iter_optimal_costs.append(b.read(_lib.BUF_BEST_COSTS)[0])
                all_obs.append(b.read(_lib.BUF_OBS))
            if verbose:
                print(f'Number of observations: {s.n_obs}')
                print(f'Iteration {n_iter + 1} - Time Elapsed: {round(t.time() - st, 4)}\n\n')
        self._n_iter = n_iter
        pre_fobs = b.read(_lib.BUF_OBS)
        self.score_thresh = b.scalars().score_thresh
        # final hyper-parameter-optimised fit (gpet.py:874-876), seed = seed + N_iter; on the device
        fits, _ = device_final_fits(b, [dict(self._p, seed=self.seed)], None, [n_iter])
        y_mean_optim, y_std, self._theta = fits[0]
        cred_interval = (y_mean_optim - 1.96 * y_std, y_mean_optim + 1.96 * y_std)
        all_samples.append(y_mean_optim)
        all_obs.append(pre_fobs)
        optim_mean_curve = np.concatenate([self.x_grid[:, np.newaxis], y_mean_optim[:, np.newaxis]], axis=1)
        edge_trace = np.rint(optim_mean_curve[:, [1, 0]]).astype(int)
        iter_optimal_curves.append(edge_trace[:, [1, 0]])
        if verbose:
            print(f'Time elapsed before algorithm converged: {round(t.time() - alg_st, 3)}')
        if self.return_std:
            return edge_trace, cred_interval
        if not return_lines:
            return edge_trace
        return edge_trace, (all_samples, all_obs, iter_optimal_curves)


def device_final_fits(batch, ps, obs_list, iters):
    """Converged fits (gpet.py:874) of every edge of a batch at once, entirely on the device (gpet_final_fit_all):
    training sets from the observation sets on the device (``obs_list`` given => those are set first), standardised
    like the reference does, theta0 + 12 restarts from ``RandomState(seed + iters[e])``, L-BFGS-B for all 13 B
    problems in lock step with one batched objective launch per round, best restart, posterior at the optimum.
    Returns ([(mean in pixels, std in standardised units, theta)] per edge, number of optimiser rounds)."""
    if obs_list is not None:
        for e, o in enumerate(obs_list):
            batch.set_obs(e, np.asarray(o).reshape(-1, 2).astype(np.int64))
    mean, std, theta, fmin, rounds = batch.final_fit_all([p["seed"] + iters[e] for e, p in enumerate(ps)])
    out = [(mean[e, :len(p["x_grid"])].copy(), std[e, :len(p["x_grid"])].copy(), theta[e].copy()) for e, p in enumerate(ps)]
    return out, rounds


class GP_Edge_Tracing_Batch(object):
    """B independent edges traced together on one GPU (BASELINE config 4's per-GPU share).

    Not in the reference (which traces one edge per object); it is the batched form of the same
    algorithm: every kernel takes the edge index from blockIdx, finished edges are skipped, and
    edge e's result equals what ``GP_Edge_Tracing`` returns for the same arguments.

    ``inits``: list of (n_init, 2) xy arrays; ``grad_imgs``: one (M, N) image shared by all edges
    or a list with one image per edge; ``seeds``: per-edge RNG seed (gpet.py:33,839).
    Remaining keyword arguments are the reference constructor's (gpet.py:22-35), common to all edges.
    """

    def __init__(self, inits, grad_imgs, seeds, kernel_options=(1, 3, 3), noise_y=1, N_samples=500, score_thresh=1,
                 delta_x=20, keep_ratio=0.1, pixel_thresh=5, return_std=False, fix_endpoints=True, *, obs=None,
                 device=0, stream=None, factor_cap=0, z_cols=0, _ctx=None, grad_device_ptrs=None, grad_shape=None,
                 sample_dtype=None, rng=None):
        """``obs``: optional list of per-edge warm-start observation sets (xy), the reference's ``obs`` constructor
        argument (gpet.py:57-61,100,820).  ``grad_device_ptrs`` + ``grad_shape``: the gradient image(s) already live
        on this GPU (e.g. a torch tensor an RCCL broadcast filled): integer device addresses of f32 (M, N) arrays,
        consumed in place instead of ``grad_imgs``."""
        on_dev = grad_device_ptrs is not None
        if on_dev:
            ptrs = list(grad_device_ptrs) if isinstance(grad_device_ptrs, (list, tuple)) else [grad_device_ptrs]
            share = len(ptrs) == 1
            shapes = [tuple(grad_shape)] * len(inits)
        else:
            share = not isinstance(grad_imgs, (list, tuple))
            imgs = [grad_imgs] if share else list(grad_imgs)
            shapes = [np.asarray(imgs[0 if share else e]).shape for e in range(len(inits))]
        B = len(inits)
        assert len(seeds) == B and (share or (len(ptrs) if on_dev else len(imgs)) == B)
        obs = [np.array([])] * B if obs is None else list(obs)
        inits = list(inits)  # (an ndarray of shape (B, n, 2) makes a fresh view per access: materialise the items once)
        # (edges with the same init points, observations and image shape resolve to the same parameters but for the seed, which
        #  nothing derived depends on: resolved once per distinct triple -- a batch of 1 024 equal edges spent 8 ms here.  The key
        #  is the CONTENT: object identities can be reused by temporaries)
        def content(a):
            a = np.asarray(a)
            return (a.shape, a.dtype.str, a.tobytes())
        self._ps, abi, memo, by_id = [], [], {}, {}
        for e in range(B):
            ik = (id(inits[e]), id(obs[e]))  # (the items are alive in `inits` / `obs`, so these identities are stable here)
            ck = by_id.get(ik)
            if ck is None:
                ck = by_id[ik] = (content(inits[e]), tuple(shapes[e]), content(obs[e]))
            key = ck
            hit = memo.get(key)
            if hit is None:
                pe = resolve_params(inits[e], shapes[e], kernel_options, noise_y, obs[e], N_samples, score_thresh,
                                    delta_x, keep_ratio, pixel_thresh, int(seeds[e]), return_std, fix_endpoints)
                hit = memo[key] = (pe, to_abi_params(pe, factor_cap=factor_cap, z_cols=z_cols))
            else:
                pe = dict(hit[0], seed=int(seeds[e]))
                pe["init"] = hit[0]["init"].copy()
                pe["obs"] = np.array(hit[0]["obs"], copy=True)
            self._ps.append(pe)
            abi.append(hit[1])
        self._ctx = _ctx if _ctx is not None else _lib.Context(device, stream)
        if on_dev:
            self._batch = _lib.Batch(self._ctx, None, abi, [p["init"] for p in self._ps], share_image=share,
                                     device_ptrs=ptrs, shape=grad_shape)
        else:
            g32 = [np.ascontiguousarray(g, dtype=np.float32) for g in imgs]
            self._batch = _lib.Batch(self._ctx, g32, abi, [p["init"] for p in self._ps], share_image=share)
        if sample_dtype is not None:
            self._batch.set_sample_dtype(sample_dtype)
        if rng is not None:
            self._batch.set_rng(rng)
        self._set_obs()
        self.B = B
        self.return_std = return_std
        self.seeds = [int(s) for s in seeds]
        self.timings = {}

    def _set_obs(self):
        for e, p in enumerate(self._ps):
            if p["obs"].shape[0]:
                self._batch.set_obs(e, p["obs"])

    def reset(self):
        """Back to the state right after construction (the warm-start observations included): nothing of the trace the
        object ran before is used by the next one."""
        self._batch.reset()
        self._set_obs()

    def set_frame(self, grad_imgs=None, obs=None, seeds=None, grad_device_ptrs=None, next_frame=True):
        """The next frame of an image sequence for the same edges (gpet.py:57-61: the previous trace warm-starts the
        next through ``obs``): new gradient image(s) -- host arrays, or device addresses with ``grad_device_ptrs`` --
        new warm-start observations and, optionally, new seeds.  Geometry, kernel and every other parameter stay, so
        what depends only on them (arena, streams, the prior eigenbasis of the structured loop path) is reused.
        ``next_frame`` (default): the images continue the sequences just traced, so the any-rank (Matern) factor of the
        new trace's first iteration may start from the last trace's rows -- an iterative solve, the same rows to its
        tolerance.  ``next_frame=False``: unrelated images; the trace is what a fresh object would compute, bit for bit."""
        if grad_device_ptrs is not None:
            self._batch.set_images(device_ptrs=list(grad_device_ptrs) if isinstance(grad_device_ptrs, (list, tuple))
                                   else [grad_device_ptrs], next_frame=next_frame)
        else:
            imgs = list(grad_imgs) if isinstance(grad_imgs, (list, tuple)) else [grad_imgs]
            self._batch.set_images([np.ascontiguousarray(g, dtype=np.float32) for g in imgs], next_frame=next_frame)  # (no copy of f32 input)
        obs = [np.array([])] * self.B if obs is None else list(obs)
        for e, p in enumerate(self._ps):
            p["obs"] = np.asarray(obs[e]).reshape(-1, 2).astype(np.int64)
            if seeds is not None:
                p["seed"] = int(seeds[e])
        if seeds is not None:
            self.seeds = [int(v) for v in seeds]
        self._set_obs()

    def run_loop(self, max_iter=1000, chunk=64):
        """The device-resident while-loops of all edges (gpet.py:829-870); returns iterations per edge.

        One library call advances the loop by up to `chunk` iterations and returns as soon as every edge has finished
        (gpet_trace_iterate enqueues the iterations in shrinking groups and checks the `done` flags in between)."""
        b = self._batch
        n_active = self.B
        done_iters = 0
        while n_active > 0:
            n_active = b.iterate(self.seeds, chunk)
            done_iters += chunk
            if done_iters >= max_iter and n_active > 0:
                raise _lib.GpetError(_lib.ERR_ITER_CAP, f"{n_active} edges did not converge in {max_iter} iterations")
        return self._iters()

    def _iters(self):
        return [s.iter for s in self._batch.all_scalars()]

    def final_fits(self, iters):
        fits, self._fit_rounds = device_final_fits(self._batch, self._ps, None, iters)
        return fits

    def finish(self, iters):
        """Converged fits + rounding of the means to pixel indices (gpet.py:874-886) for every edge."""
        fits = self.final_fits(iters)
        out = []
        for p, (mean, std, theta) in zip(self._ps, fits):
            curve = np.concatenate([p["x_grid"][:, None], mean[:, None]], axis=1)
            et = np.rint(curve[:, [1, 0]]).astype(int)
            out.append((et, (mean - 1.96 * std, mean + 1.96 * std)) if self.return_std else et)
        return out

    def __call__(self, max_iter=1000):
        t0 = t.time()
        iters = self.run_loop(max_iter)
        t1 = t.time()
        out = self.finish(iters)
        t2 = t.time()
        self.timings = dict(loop_s=t1 - t0, final_fit_s=t2 - t1, iters=iters)
        return out
