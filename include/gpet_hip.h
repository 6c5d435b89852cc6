/*
 * gpet_hip.h -- C ABI of libgpet_hip.so: the MI355X (gfx950) implementation of the
 * GP posterior-update + posterior-sampling + curve-scoring hot path of
 * gp_edge_tracing.gpet.GP_Edge_Tracing (reference: jaburke166/gaussian_process_edge_trace).
 *
 * The reference is pure Python and has no FFI; each entry point below names the reference
 * interface (file:line under /root/reference) whose arithmetic it replaces.  The Python host
 * (gaussian_process_edge_trace_amd/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain C, POD only, no exceptions; every call returns a gpet_status (0 = ok);
 *     gpet_last_error(ctx) gives the message of the last failure on that context.
 *   - M rows (y), N columns (x) image; Lg = edge_length = x_en - x_st + 1 grid points;
 *     n = training points (inits + observations); S = posterior samples.
 *   - a "batch" holds B independent edges that are processed together (blockIdx.y = edge).
 *     A single edge is a batch of 1.
 *   - host pointers unless a parameter says "device".  All work is enqueued on the
 *     context's HIP stream; calls that return data to the host synchronise that stream.
 *   - handles are not thread-safe individually; distinct contexts may be used from
 *     distinct host threads.
 */
#ifndef GPET_HIP_H
#define GPET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPET_ABI_VERSION 1

typedef enum gpet_status {
  GPET_OK = 0,
  GPET_ERR_BAD_ARG = 1,
  GPET_ERR_HIP = 2,          /* a HIP runtime call failed (message has the HIP error) */
  GPET_ERR_NOT_PD = 3,       /* Cholesky met a non-positive pivot (sklearn_gpr.py:306-314 LinAlgError) */
  GPET_ERR_ITER_CAP = 4,     /* trace did not converge within the iteration cap */
  GPET_ERR_RANK_CAP = 5,     /* posterior covariance rank exceeded the factor capacity */
  GPET_ERR_UNSUPPORTED = 6,  /* e.g. a Matern nu that is not a positive finite number */
  GPET_ERR_NO_DEVICE = 7,
  GPET_ERR_STATE = 8         /* call sequence error (stage input not produced yet) */
} gpet_status;

typedef enum gpet_kernel_type { GPET_KERNEL_RBF = 0, GPET_KERNEL_MATERN = 1 } gpet_kernel_type;

/* Clamped constructor arguments of GP_Edge_Tracing.__init__ (gpet.py:95-158), resolved by the host. */
typedef struct gpet_params {
  int32_t kernel_type;   /* gpet_kernel_type                  gpet.py:133,142 */
  double nu;             /* Matern smoothness: 0.5 / 1.5 / 2.5 in closed form, any other nu > 0 through the
                          * Bessel form (quadrature; lag table for the loop)      gpet.py:134,143 */
  double sigma_f;        /* amplitude                         gpet.py:131,147 */
  double length_scale;   /*                                   gpet.py:132,151 */
  double noise_y;        /*                                   gpet.py:98 */
  int32_t n_samples;     /* S (already clamped)               gpet.py:99 */
  int32_t n_keep;        /* int(keep_ratio*N_samples) raw     gpet.py:118 */
  int32_t delta_x;       /* clamped                           gpet.py:105 */
  int32_t pixel_thresh;  /* clamped                           gpet.py:103 */
  double score_thresh;   /* clamped initial value             gpet.py:104 */
  int32_t fix_endpoints; /*                                   gpet.py:108,161 */
  int32_t x_st, x_en;    /* from the UNSORTED init            gpet.py:96 */
  int32_t n_init;        /* rows of init                      gpet.py:112 */
  int32_t obs_cap;       /* capacity for observations (>= any obs set passed later) */
  int32_t factor_cap;    /* max rank kept by the eigen-factor sampler (0 = library default) */
  int32_t z_cols;        /* normal columns stored per sample (0 = default: factor_cap; Lg = full stream) */
  double jitter;         /* GPR alpha, 1e-6                    gpet.py:155 */
} gpet_params;

/* Named per-edge device buffers readable/writable through gpet_batch_read/_write (tests, injection). */
typedef enum gpet_buf {
  GPET_BUF_X_TRAIN = 0,   /* f64 [n]        sorted training x             gpet.py:212-223 */
  GPET_BUF_Y_TRAIN = 1,   /* f64 [n]        y/y_s - mean                  sklearn_gpr.py:227 */
  GPET_BUF_CHOL = 2,      /* f64 [n_cap*n_cap] row-major, lower = L       sklearn_gpr.py:307 */
  GPET_BUF_ALPHA = 3,     /* f64 [n]                                       sklearn_gpr.py:316 */
  GPET_BUF_MEAN = 4,      /* f64 [Lg]       posterior mean (scaled units)  sklearn_gpr.py:382-385 */
  GPET_BUF_STD = 5,       /* f64 [Lg]                                      sklearn_gpr.py:414-436 */
  GPET_BUF_COV = 6,       /* f64 [Lg*Lg]                                   sklearn_gpr.py:398-403 */
  GPET_BUF_FACTOR = 7,    /* f64 [rows*Lg]  rows of sqrt(s)*v              numpy mvn via sklearn_gpr.py:464 */
  GPET_BUF_EIGVALS = 8,   /* f64 [factor rows]                             */
  GPET_BUF_NORMALS = 9,   /* f64 [S*z_cols] standard normals, row = sample sklearn_gpr.py:464 */
  GPET_BUF_SAMPLES = 10,  /* f64 [S*Lg]     row = sample, pixel units      gpet.py:260-261 */
  GPET_BUF_COSTS = 11,    /* f64 [S]                                       gpet.py:438-440 */
  GPET_BUF_BEST_IDX = 12, /* i32 [n_keep]                                  gpet.py:443 */
  GPET_BUF_BEST_COSTS = 13, /* f64 [n_keep]                                gpet.py:445 */
  GPET_BUF_SCALARS = 14,  /* gpet_scalars                                  */
  GPET_BUF_OBS = 15,      /* i64 [n_obs*2] xy                              gpet.py:857 */
  GPET_BUF_KDE = 16,      /* f32 [M*N]      normalised curve KDE           gpet.py:648
                           *                (as left by gpet_select_pixels; gpet_trace_iterate keeps the raw density of
                           *                 the band rows only and normalises inside its pixel kernels) */
  GPET_BUF_GRAD_KDE = 17, /* f32 [M*N]      normalised gradient KDE        gpet.py:127 */
  GPET_BUF_GRAD = 18,     /* f32 [M*N]      normalised gradient image      gpet.py:97 */
  GPET_BUF_NOISE_W = 19,  /* f64 [n]        per-point noise weights        gpet.py:209-213 */
  GPET_BUF_FIN_TRAIN = 20,  /* f64 [3][n_cap] converged fit: standardised x | y | noise weights, n = n_init + n_obs used
                             *                (gpet.py:235-238; sklearn_gpr.py:229-234), as gpet_final_fit_all built them */
  GPET_BUF_FIN_PAR = 21,  /* f64 [12]       optimum and transforms: constant, length_scale, noise_level, X_m, X_s, y_m,
                           *                y_s, m2, s2, 0, 0, 0 */
  GPET_BUF_FIN_STARTS = 22  /* f64 [13][3]  theta0 + the 12 restart points of the last gpet_final_fit_all
                             *                (gpet.py:244-245; sklearn_gpr.py:283-288) */
} gpet_buf;

/* Per-edge scalar state kept on the device (GPET_BUF_SCALARS). */
typedef struct gpet_scalars {
  double y_s;          /* std(y)+1                              gpet.py:228 */
  double amp;          /* sigma_f^2 / y_s^2                     gpet.py:230 */
  double y_mean;       /* _y_train_mean                         sklearn_gpr.py:222 */
  double y_std;        /* _y_train_std                          sklearn_gpr.py:223 */
  double score_thresh; /* persists across iterations            gpet.py:595 */
  double lml;          /* diagnostics (Jacobi sweeps of the last factorisation) */
  int32_t n;           /* training points of the last fit */
  int32_t n_obs;       /* current observation count */
  int32_t rank;        /* rows of the factor */
  int32_t status;      /* gpet_status raised on the device */
  int32_t iter;        /* iterations done                       gpet.py:865 */
  int32_t done;        /* n_obs >= algo_thresh                  gpet.py:829 */
  int32_t n_removed;   /* curve points outside the image        gpet.py:498-500 */
  int32_t force;       /* set by the per-stage entry points: run the stage even if `done` */
} gpet_scalars;

typedef struct gpet_ctx gpet_ctx;
typedef struct gpet_batch gpet_batch;

/* ---- context ------------------------------------------------------------------------ */
int gpet_abi_version(void);

/* Process-wide tuning switches (no reference counterpart): ONE table (csrc/gpet_options.hip; INTEGRATION.md section 5
 * lists every name with its default, range and meaning).  An option's initial value is the environment variable
 * GPET_<NAME IN CAPITALS> if set, else the table's default; results never depend on an option except where its
 * description says so (cross-check solvers, opt-in modes).
 * gpet_set_option returns the previous value -- "chosen automatically" (-1) is reported as the option's largest value
 * + 1 -- or -1 for an unknown name; values outside an option's range are clamped.  gpet_get_option stores the current
 * value (-1 = automatic) and returns 0, or GPET_ERR_BAD_ARG for an unknown name.  gpet_option_count / gpet_option_info
 * enumerate the table (any out pointer may be NULL; strings are static). */
int gpet_set_option(const char* name, int value);
int gpet_get_option(const char* name, int* value);
/* A batch keeps its OWN copy of the table, taken when it is created: what a batch does is fixed by the options in force at
 * gpet_batch_create and by these two calls, never by a gpet_set_option another thread makes while it runs (every batch entry
 * point reads the batch's copy).  Same return conventions as gpet_set_option / gpet_get_option. */
int gpet_batch_set_option(gpet_batch* b, const char* name, int value);
int gpet_batch_get_option(const gpet_batch* b, const char* name, int* value);
int gpet_option_count(void);
int gpet_option_info(int index, const char** name, int* value, int* def, int* lo, int* hi, const char** doc);
/* stream: a hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream), or NULL
 * to let the library create its own. */
int gpet_ctx_create(int device, void* stream, gpet_ctx** out);
void gpet_ctx_destroy(gpet_ctx* ctx);
const char* gpet_last_error(const gpet_ctx* ctx);
int gpet_sync(gpet_ctx* ctx);
void* gpet_ctx_stream(gpet_ctx* ctx);
/* hipEvent timing on the context's stream (bench.py roofline leg). */
int gpet_timer_start(gpet_ctx* ctx);
int gpet_timer_stop_ms(gpet_ctx* ctx, float* ms);

/* ---- a1: gpet_utils.comp_grad_img + normalise (gpet_utils.py:65-119) ------------------ */
/* img f64 [M*N], kern f64 [kh*kw] (host); out f32 [M*N] (host).  True convolution with
 * clamp-to-edge padding, negatives -> 0, float32 min-max normalisation. */
int gpet_grad_image(gpet_ctx* ctx, const double* img, int M, int N, const double* kern, int kh, int kw,
                    float* out);
/* gpet_utils.normalise(img, (0,1)) for an f32 image (gpet.py:97): out f32 [count] (host). */
int gpet_normalise_f32(gpet_ctx* ctx, const float* img, size_t count, float* out);

/* ---- batch of edges (GP_Edge_Tracing.__init__, gpet.py:95-178) ------------------------- */
/* grad: B pointers (host memory) to f32 [M*N] gradient images as the user passes them; the
 * library re-normalises them (gpet.py:97).  If share_image != 0 only grad[0] is used for all
 * edges.  params: B structs.  init_xy: B pointers to i64 [n_init*2], already sorted by x. */
int gpet_batch_create(gpet_ctx* ctx, int B, int M, int N, const float* const* grad, int share_image,
                      const gpet_params* params, const int64_t* const* init_xy, gpet_batch** out);
/* The same with flags.  GPET_GRAD_ON_DEVICE: grad[] are DEVICE pointers on the context's device (e.g. the data_ptr()
 * of the torch tensor an RCCL broadcast over xGMI has just filled): the images are consumed in place, with
 * no copy through host memory.  init_xy and params stay host pointers. */
#define GPET_GRAD_ON_DEVICE 1u
int gpet_batch_create2(gpet_ctx* ctx, int B, int M, int N, const float* const* grad, int share_image,
                       const gpet_params* params, const int64_t* const* init_xy, unsigned int flags, gpet_batch** out);
void gpet_batch_destroy(gpet_batch* b);
int gpet_batch_size(const gpet_batch* b);
/* out[0..count): Lg, S, n_keep, n_cap, factor_cap, z_cols, factor_rows_cap, n_bins, obs_cap, algo_thresh,
 * structured (1: the loop uses the prior-eigenbasis path), r0 (rank of the grid's correlation matrix), z_ring (slots
 * of pre-generated normals per edge), arena size of the batch in MiB */
int gpet_batch_info(const gpet_batch* b, int e, int32_t* out, int count);

/* Back to the state right after gpet_batch_create: no observations, initial score threshold,
 * iteration counter 0 (a GP_Edge_Tracing instance is single-use in the reference; benches re-run). */
int gpet_batch_reset(gpet_batch* b);

/* New gradient image(s) for an existing batch of the same geometry and parameters -- the next frame of an image
 * sequence (gpet.py:57-61: a trace warm-starts the next one through `obs`): grad as in gpet_batch_create2 (one pointer
 * if the batch shares its image, else B), re-normalised, gradient KDE recomputed (gpet.py:97,127), then
 * gpet_batch_reset.  The per-edge work that depends only on the geometry and the kernel (the prior eigenbasis of the
 * structured loop path) is kept. */
int gpet_batch_set_images(gpet_batch* b, const float* const* grad, unsigned int flags);
/* flags: GPET_GRAD_ON_DEVICE as above, and GPET_IMAGES_NEXT_FRAME: the new images are the NEXT FRAMES of the sequences
 * the edges have just been traced through (gpet.py:57-61), so the any-rank factor of the new trace's first iteration may
 * start from the last trace's factor rows (an iterative solve: the same rows to its tolerance, 4e-9 relative).  Without the
 * flag -- and after gpet_batch_reset / gpet_batch_set_obs always -- nothing of an earlier trace is used: a trace depends
 * on (image, seed, observations) only, like the reference's single-use object. */
#define GPET_IMAGES_NEXT_FRAME 2u

/* set / get the observation set (xy int64) of edge e (gpet.py:100,820,857). */
int gpet_batch_set_obs(gpet_batch* b, int e, const int64_t* obs_xy, int n_obs);
int gpet_batch_read(gpet_batch* b, int e, int which, void* dst, size_t bytes);
/* Writable: FACTOR (rows = factor rows; marks the factor as injected so gpet_gp_factor leaves
 * it alone), NORMALS, SAMPLES, COSTS, BEST_IDX, BEST_COSTS (a caller's own choice of best curves for
 * gpet_curve_kde / gpet_select_pixels, gpet.py:622-648), MEAN, COV, KDE, GRAD_KDE, SCALARS. */
int gpet_batch_write(gpet_batch* b, int e, int which, const void* src, size_t bytes, int rows);
int gpet_batch_clear_injected_factor(gpet_batch* b, int e);

/* ---- a2-a5: fit_predict_GP not-converged, deterministic part -------------------------- */
/* gpet.py:209-231 + sklearn_gpr.py:221-227,304-320 (fit) + :381-436 (predict mean/std/cov).
 * want_cov: also materialise the Lg x Lg covariance. */
int gpet_gp_fit_predict(gpet_batch* b, int want_cov);

/* ---- a6: sample_y (sklearn_gpr.py:440-473) --------------------------------------------- */
/* Factor the posterior covariance into rows sqrt(s_k) v_k (k by descending s_k), the object
 * numpy's legacy multivariate_normal builds from LAPACK SVD.  Eigenvector signs follow the
 * library convention: sum_j row[j] / (j + 1) >= 0 (LAPACK's are implementation-defined). */
int gpet_gp_factor(gpet_batch* b);
/* Fill the normals with RandomState(seed[e]).standard_normal((S, Lg)) (first z_cols columns
 * of every row are stored).  seeds: B values. */
int gpet_gp_normals(gpet_batch* b, const uint32_t* seeds);
/* samples[s, :] = y_s * (Z[s, :rows] @ factor + mean)            (gpet.py:260-261) */
int gpet_gp_sample(gpet_batch* b);

/* ---- a7: get_best_curves / cost_funct (gpet.py:371-451) -------------------------------- */
int gpet_score_curves(gpet_batch* b);

/* ---- f1: get_best_pixels (gpet.py:455-662) --------------------------------------------- */
/* kernel_density_estimate(best_curves, costs) alone (gpet.py:455-529): the normalised KDE of the curves
 * GPET_BUF_BEST_IDX / _BEST_COSTS select among GPET_BUF_SAMPLES goes to GPET_BUF_KDE. */
int gpet_curve_kde(gpet_batch* b);
int gpet_select_pixels(gpet_batch* b);
/* Pixel scoring / threshold decay / per-bin argmax only (gpet.py:532-618), on whatever curve KDE
 * is currently in GPET_BUF_KDE (tests inject the reference's). */
int gpet_select_pixels_only(gpet_batch* b);

/* ---- a8: the outer loop (gpet.py:829-870) ---------------------------------------------- */
/* Runs up to max_iters iterations of fit->factor->normals->sample->score->pixels for every
 * edge that is not done; seed of iteration k (0-based) of edge e is base_seed[e] + k + 1
 * (gpet.py:839).  The iterations are enqueued in groups (8, 4, then 2 once edges start to finish); after
 * each group the `done` flags are read, the call returns early when no edge is left, and the next group is
 * launched over the edges still running only.  Returns the number of edges still not done in *n_active. */
int gpet_trace_iterate(gpet_batch* b, const uint32_t* base_seeds, int max_iters, int* n_active);

/* ---- f2: converged fit (gpet.py:232-248; sklearn_gpr.py:254-295, 475-585) ----------------- */
/* Upload edge e's standardised training set (x, y standardised as gpet.py:235-238 and
 * sklearn_gpr.py:229-234 do; w = per-point noise weights), n <= the batch's training-set capacity (register-tile objective
 * kernels up to 250 points; above, the blocked HBM path: Cholesky, L^-1 and K^-1 tiles on the matrix cores). */
int gpet_final_set_training(gpet_batch* b, int e, const double* xs, const double* ys, const double* w, int n);
/* The same for every edge of the batch in one call: xs/ys/w are [B*stride], n [B]. */
int gpet_final_set_training_all(gpet_batch* b, const double* xs, const double* ys, const double* w, const int32_t* n,
                                int stride);
/* Scalar state of every edge in one copy: dst [B]. */
int gpet_batch_read_scalars_all(gpet_batch* b, gpet_scalars* dst);
/* Observation sets of every edge in one call: dst i64 [B*stride_obs*2] xy, counts [B]. */
int gpet_batch_read_obs_all(gpet_batch* b, int64_t* dst, int32_t* counts, int stride_obs);
/* Posterior of the converged fit at the optimum (gpet.py:262-266) for every edge: par [B*12] =
 * constant, length_scale, noise_level (values, not logs), X_m, X_s, y_m, y_s, m2, s2, 0, 0, 0;
 * mean_out (pixels) and std_out (standardised units, as the reference returns it) are [B*stride]. */
int gpet_final_predict_all(gpet_batch* b, const double* par, double* mean_out, double* std_out, int stride);
/* GaussianProcessRegressor.predict(return_cov=True) (sklearn_gpr.py:398-403) for the fit gpet_final_predict_all has
 * just evaluated: the Lg x Lg posterior covariance on the standardised grid goes to GPET_BUF_COV of every edge. */
int gpet_final_cov(gpet_batch* b);
/* Objective of the reference's L-BFGS-B runs for P problems at once: problem i evaluates
 * -log_marginal_likelihood and its gradient wrt theta_i = log(constant, length_scale, noise_level)
 * on edge edge_of[i]'s training set.  theta [P*3], f_out [P], g_out [P*3] (host).  A non-PD
 * kernel matrix gives f = +inf, g = 0 (sklearn_gpr.py:521-522). */
int gpet_lml_batch(gpet_batch* b, int P, const int32_t* edge_of, const double* theta, double* f_out, double* g_out);

/* The whole converged fit of every edge of the batch on the device (gpet.py:874-876 -> :232-248, 262-266;
 * sklearn_gpr.py:254-295): training sets from the current observation sets, standardised as the reference does;
 * theta0 + 12 restarts from RandomState(seeds[e]).uniform (the caller passes seed + N_iter, gpet.py:874); L-BFGS-B
 * (scipy.optimize.minimize's algorithm and defaults) for all 13 B problems in lock step, one batched objective launch
 * per round; best restart; posterior at the optimum.  mean_out (pixels) and std_out (standardised units, as the
 * reference returns it) are [B*stride]; theta_out (optional) [B*4] = log(constant, length_scale, noise_level) and the
 * minimum of -log marginal likelihood; rounds_out (optional) = objective rounds.  Any number of training points the
 * batch was created for (more than 250: the blocked objective, ~100 launches per round -- a rare, slower path). */
int gpet_final_fit_all(gpet_batch* b, const uint32_t* seeds, double* mean_out, double* std_out, double* theta_out,
                       int stride, int32_t* rounds_out);

/* Random numbers of this batch: 0 = MT19937 + polar method, the stream of numpy's RandomState(seed).standard_normal that
 * sklearn_gpr.py:460-464 draws from (default; every parity statement is made on it); 1 = Philox4x32-10 + Box-Muller, a
 * counter-based generator (normal (s, j) of an iteration is a pure function of seed, s, j): NOT the reference's numbers, an
 * opt-in mode with its own oracle (oracle.philox_standard_normal); the per-iteration seed rule (gpet.py:839) is the same. */
int gpet_batch_set_rng(gpet_batch* b, int mode);

/* Storage type of the posterior samples of this batch: 0 = f64 (default: the reference's sample_y, sklearn_gpr.py:440-473,
 * every parity statement is made on it), 1 = f32 -- BASELINE config 2's "fp32 posterior samples": the sample GEMM rounds
 * each sample to f32 when it stores it, scorer / KDE / pixel kernels widen it again, all arithmetic stays f64; GPET_BUF_SAMPLES
 * is f64 on the interface either way.  Results equal the reference with `y_samples.astype(float32)` inserted after
 * sample_y (oracle mode sample_dtype="f32").  Call between traces, not while a loop is enqueued. */
int gpet_batch_set_sample_dtype(gpet_batch* b, int f32);

/* The optimiser alone, for a caller's own training sets (GaussianProcessRegressor.fit with optimizer="fmin_l_bfgs_b",
 * sklearn_gpr.py:254-295, 587-607): L-BFGS-B from n_starts start points per edge (starts [B][n_starts][3], theta = log
 * (constant, length_scale, noise_level)) inside bounds [3][2] = (lo, hi) per component, on the training sets of
 * gpet_final_set_training(_all); theta_out [B*4] = the best start's optimum (first minimum, np.argmin) and its objective
 * value (-log marginal likelihood); it is also left in GPET_BUF_FIN_PAR[0..2] as exp(theta). */
int gpet_final_optimize(gpet_batch* b, int n_starts, const double* starts, const double* bounds, double* theta_out,
                        int32_t* rounds_out);

/* Device time of the LML kernel launches of this batch since the last reset (hipEvents around each launch), the
 * number of objective evaluations and of launches; any of the outputs may be NULL.  (bench.py's roofline leg.) */
int gpet_lml_stats(gpet_batch* b, int reset, double* kernel_ms, int64_t* evaluations, int32_t* launches);

/* ---- collectives: one process per GPU ---------------------------------------------------------------
 * The reference has no multi-GPU path; independent edges (and whole chains of an image sequence) shard over ranks with no
 * collective on a trace's data path: ONE broadcast of the shared gradient image(s) and ONE gather of the finished traces.
 * These are the RCCL call sites for a host that is not Python (the Python package does the same through torch.distributed,
 * sharding.py).  RCCL is bound at run time (dlopen); a world of 1 needs none.
 *   rank 0: gpet_comm_unique_id(id)  -> ship the GPET_COMM_ID_BYTES bytes to every rank by any means (file, socket, MPI)
 *   all:    gpet_comm_create(ctx, id, world, rank, &comm)        (ncclCommInitRank on the context's device)
 *           gpet_comm_block(comm, n_edges, &lo, &hi)             this rank's contiguous block of edges [lo, hi)
 *           gpet_bcast_grad(comm, d_grad, count, root)           in place on DEVICE memory, on the context's stream; hand
 *                                                                d_grad to gpet_batch_create2(..., GPET_GRAD_ON_DEVICE)
 *           gpet_gather_traces(comm, local, n_edges, len, all)   host int64 [n_local][len][2] -> [n_edges][len][2] on every
 *                                                                rank, in global edge order (blocks as gpet_comm_block)
 *           gpet_allgather_i64(comm, local, counts, all)         the same for blocks of counts[r] int64 per rank (sequences) */
typedef struct gpet_comm gpet_comm;
#define GPET_COMM_ID_BYTES 128
int gpet_comm_unique_id(void* id128);
int gpet_comm_create(gpet_ctx* ctx, const void* id128, int world, int rank, gpet_comm** out);
void gpet_comm_destroy(gpet_comm* comm);
int gpet_comm_rank(const gpet_comm* comm);
int gpet_comm_world(const gpet_comm* comm);
int gpet_comm_block(const gpet_comm* comm, int64_t n_units, int64_t* lo, int64_t* hi);
int gpet_bcast_grad(gpet_comm* comm, float* d_grad, size_t count, int root);
/* device memory on the context's device for the broadcast buffer (hosts that do not call the HIP runtime themselves);
 * gpet_dev_copy: host -> device (to_host = 0) or device -> host (1), on the context's stream, complete on return */
int gpet_dev_alloc(gpet_ctx* ctx, size_t bytes, void** out);
int gpet_dev_free(gpet_ctx* ctx, void* ptr);
int gpet_dev_copy(gpet_ctx* ctx, void* dst, const void* src, size_t bytes, int to_host);
int gpet_allgather_i64(gpet_comm* comm, const int64_t* h_local, const int64_t* counts, int64_t* h_all);
int gpet_gather_traces(gpet_comm* comm, const int64_t* h_local, int64_t n_edges, int64_t edge_len, int64_t* h_all);

/* ---- measurement -------------------------------------------------------------------------- */
/* Enqueue one stage `reps` times between two hipEvents on the context's stream and return the
 * mean milliseconds per repetition.  stage: 0 fit+predict+cov, 1 factor, 2 normals, 3 sample
 * GEMM, 4 scoring+top-k, 5 curve KDE; single kernels: 100 fit, 101 predict, 102 covariance, 110 pivoted
 * Cholesky, 111 Gram, 112 Jacobi, 113 factor rows, 130 sample GEMM, 140 scoring, 141 top-k, 150 KDE prep,
 * 151 fused KDE (5, 150, 151: the loop form -- raw density, band rows only), 152 KDE normalise (stage-API form),
 * 160 column scan of the pixel selection (loop form);
 * structured loop path: 120 fit, 121 U/H/mean, 122 Jacobi, 123 factor rows + sign pass.
 * (bench.py's roofline leg; leaves the loop state as-is.) */
int gpet_profile_stage(gpet_batch* b, int stage, int reps, float* ms_per_rep);

#ifdef __cplusplus
}
#endif
#endif /* GPET_HIP_H */
