// C ABI, part 2: batches of edges -- arena layout and creation, destruction, images, observation sets, reset, reads and writes.
#include "gpet_api_internal.h"

namespace {
int nu_to_code(double nu) {
  if (nu == 0.5) return 0;
  if (nu == 1.5) return 1;
  if (nu == 2.5) return 2;
  if (nu > 0.0 && nu <= 1e6) return 3;  // any other smoothness: Bessel form by quadrature (gpet.py:134)
  return -1;
}

// lays out one edge's buffers; with base == nullptr only measures
void carve_edge(Carver& cv, EdgeDev& E, bool own_image, int batch_lg) {
  const size_t Lg = E.Lg, nc = E.n_cap, rc = E.r_cap, S = E.S;
  const size_t px = (size_t)E.M * E.N, gpx = (size_t)(E.M + 2) * (E.N + 2);
  E.init_xy = cv.take<long long>(2 * (size_t)E.n_init);
  E.obs_xy = cv.take<long long>(2 * (size_t)E.obs_cap);
  E.obs_new = cv.take<long long>(2 * (size_t)E.obs_cap);
  E.xt = cv.take<double>(nc);
  E.yt = cv.take<double>(nc);
  E.wt = cv.take<double>(nc);
  E.alpha = cv.take<double>(nc);
  E.chol_inv = cv.take<double>(nc > 128 ? (nc / 64 + 1) * 4096 : 1);
  E.solve_z = cv.take<double>(nc > 128 ? nc : 1);
  E.solve_flag = cv.take<int>(nc > 128 ? 2 * (nc / 64 + 1) : 2);
  E.K = cv.take<double>(nc * nc);
  E.V = cv.take<double>(nc * Lg);
  E.mean = cv.take<double>(Lg);
  E.std = cv.take<double>(Lg);
  E.cov = cv.take<double>(Lg * Lg);
  E.G = cv.take<double>(rc * Lg);
  E.perm = cv.take<int>(rc);
  E.C = cv.take<double>(rc * rc);
  E.W = cv.take<double>(rc * rc);
  E.Wq = cv.take<double>(2 * rc * rc);
  E.Cw = cv.take<double>(rc * rc);
  E.wq_tag = cv.take<int>(2);
  E.theta = cv.take<double>(rc);
  E.order = cv.take<int>(rc);
  E.Q0 = cv.take<double>(rc * Lg);
  E.lam0 = cv.take<double>(rc);
  E.beta = cv.take<double>(rc);
  E.h0 = cv.take<double>(rc);
  E.rho_tab = cv.take<double>((size_t)E.N);
  E.eig = cv.take<EigState>(1);
  // (transposed copy of G: the any-rank factor, and the multi-workgroup pivoted Cholesky -- which launch_factor picks per BATCH
  //  from the widest edge, pchol_multi_applies, and which then writes Gt of EVERY edge of the batch: the condition is the batch's)
  E.Gt = cv.take<double>((rc > 96 || batch_lg > 1024) ? Lg * rc : 1);
  E.Ap = cv.take<double>(rc > 96 ? 2 * Lg * rc : 1);
  E.ap_tag = cv.take<int>(3);
  E.pcx_d = cv.take<double>(Lg);
  E.pcx_cand = cv.take<double>(16 * ((size_t)E.N / 32 + 2));  // (indexed with the widest edge of the batch)
  E.jlog = cv.take<double>(E.jlog_cap > 0 ? (size_t)E.jlog_cap * 2 * rc * (rc / 2 + 1) : 2);
  E.A = cv.take<double>((size_t)E.a_rows_cap * Lg + 64);  // (+ 64: the sample GEMM loads whole 64-column tiles of the last row)
  E.Z = cv.take<double>((size_t)E.z_ring * S * (size_t)E.z_cols);
  E.Yp = (int)((Lg + 15) & ~(size_t)15);
  // (+ the spare region of the sample GEMM's idle lanes: they write 16 bytes at (Sround + 4 g) Yp + 2 lane doubles, g < 4, lane < 64 --
  //  up to 128 doubles into row Sround + 12 whatever the pitch is, so the slack is sized in elements, not in rows)
  E.Y = cv.take<double>((((S + 127) & ~(size_t)127) + 12) * (size_t)E.Yp + 128 + (size_t)E.Yp);
  E.costs = cv.take<double>(S);
  E.cost_part = cv.take<double>(S * 2 * (Lg / 30 + 2));  // (15 Simpson pairs = 30 columns per tile of the scorer)
  E.best_costs = cv.take<double>((size_t)E.n_keep + 1);
  E.best_idx = cv.take<int>((size_t)E.n_keep + 1);
  E.bins = cv.take<double>(gpx);
  E.tmpk = cv.take<double>(gpx);
  E.kde = cv.take<float>(px);
  E.kde_band = cv.take<int>(2 * ((size_t)E.N / 16 + 2));
  E.colsum = cv.take<double>((size_t)E.N);
  E.kde_wsum = cv.take<double>(2);
  E.colbest = cv.take<double>((size_t)E.N);
  E.colbest_y = cv.take<int>((size_t)E.N);
  E.mm = cv.take<unsigned int>(4);
  E.binbest = cv.take<unsigned long long>((size_t)E.n_bins);
  E.binarg = cv.take<long long>((size_t)E.n_bins);
  E.fin_x = cv.take<double>(nc);
  E.fin_y = cv.take<double>(nc);
  E.fin_w = cv.take<double>(nc);
  E.fin_par = cv.take<double>(12);
  if (own_image) {
    E.grad = cv.take<float>(px);
    E.grad_kde = cv.take<float>(px);
  }
}
}  // namespace

extern "C" {

namespace {
// frees a half-built batch (arena, streams, events, tables, staging buffer) on every early return
struct BatchGuard {
  gpet_batch* b = nullptr;
  ~BatchGuard() {
    if (b) gpet_batch_destroy(b);
  }
};
}  // namespace

// gradient image(s) as the user passes them -> re-normalised f32 on the device (gpet.py:97).
// GPET_GRAD_ON_DEVICE: grad[] are device pointers (e.g. the tensor an RCCL broadcast has just filled): consumed in
// place, no trip through host memory.
static int upload_images(gpet_batch* b, const float* const* grad, unsigned int flags) {
  gpet_ctx* c = b->ctx;
  const size_t px = (size_t)b->bd.M * b->bd.N;
  const int n_img = b->share_image ? 1 : b->B;
  const bool on_dev = (flags & GPET_GRAD_ON_DEVICE) != 0;
  for (int g = 0; g < n_img; ++g)
    if (!grad[g]) return fail(c, GPET_ERR_BAD_ARG, "gradient image %d is a null pointer", g);
  // every image its own (min, max) slot, all of them reset by ONE copy; the images then follow each other on the stream --
  // staging copy -> min / max -> normalise -- with a single wait at the end (round 5 waited after every image: 0.8 ms per
  // image of a 256-image set_frame, most of it the waits).  A device image (GPET_GRAD_ON_DEVICE) is read where it lies.
  b->h_mm0.assign((size_t)2 * n_img, 0u);
  for (int g = 0; g < n_img; ++g) b->h_mm0[2 * (size_t)g] = 0xFFFFFFFFu;
  HIPCHK(c, hipMemcpyAsync(b->d_minmax, b->h_mm0.data(), sizeof(unsigned int) * 2 * n_img, hipMemcpyHostToDevice, c->stream));
  for (int g = 0; g < n_img; ++g) {
    unsigned int* mm = b->d_minmax + 2 * (size_t)g;
    const float* src = grad[g];
    if (!on_dev) {
      HIPCHK(c, hipMemcpyAsync(b->d_raw, grad[g], px * sizeof(float), hipMemcpyHostToDevice, c->stream));
      src = b->d_raw;
    }
    HIPCHK(c, launch_minmax(c->stream, src, px, mm));
    HIPCHK(c, launch_normalise(c->stream, src, px, mm, (float*)b->h_edges[g].grad));
  }
  HIPCHK(c, gpet_wait(c->stream));  // (pageable sources and h_mm0 must stay valid until the copies have run)
  return GPET_OK;
}

int gpet_batch_create(gpet_ctx* c, int B, int M, int N, const float* const* grad, int share_image,
                      const gpet_params* params, const int64_t* const* init_xy, gpet_batch** out) {
  return gpet_batch_create2(c, B, M, N, grad, share_image, params, init_xy, 0u, out);
}

int gpet_batch_create2(gpet_ctx* c, int B, int M, int N, const float* const* grad, int share_image,
                       const gpet_params* params, const int64_t* const* init_xy, unsigned int flags, gpet_batch** out) {
  if (!c || !out || B <= 0 || M < 2 || N < 2 || !grad || !params || !init_xy)
    return fail(c, GPET_ERR_BAD_ARG, "gpet_batch_create: bad argument");
  *out = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  gpet_batch* b = new (std::nothrow) gpet_batch();
  if (!b) return fail(c, GPET_ERR_HIP, "out of host memory");
  BatchGuard guard;
  guard.b = b;
  option_snapshot(&b->opts);  // the process-wide table as it is NOW is this batch's for its lifetime
  GPET_BATCH_SCOPE(b);
  b->ctx = c;
  b->B = B;
  b->share_image = share_image ? 1 : 0;
  b->h_edges.resize(B);
  b->params.assign(params, params + B);
  BatchDims bd{};
  bd.M = M;
  bd.N = N;
  bd.lg_even = 1;
  bool any_big = false, any_gen_nu = false;
  for (int e = 0; e < B; ++e) {
    const int Lg_e = params[e].x_en - params[e].x_st + 1;
    const int cap = params[e].factor_cap > 0 ? params[e].factor_cap : 96;
    if ((cap < Lg_e ? cap : Lg_e) > 96) any_big = true;
  }
  for (int e = 0; e < B; ++e) {
    const gpet_params& p = params[e];
    EdgeDev& E = b->h_edges[e];
    memset(&E, 0, sizeof E);
    const int Lg = p.x_en - p.x_st + 1;
    if (p.x_st < 0 || p.x_en >= N || Lg < 4 || p.n_init < 1 || p.n_samples < 1 || p.n_keep < 0 ||
        p.n_keep > p.n_samples || p.delta_x < 1 || p.length_scale <= 0)
      return fail(c, GPET_ERR_BAD_ARG, "gpet_batch_create: edge %d has inconsistent parameters", e);
    if (p.kernel_type == GPET_KERNEL_MATERN && nu_to_code(p.nu) < 0)
      return fail(c, GPET_ERR_UNSUPPORTED, "Matern nu=%g is not a positive finite smoothness", p.nu);
    E.M = M;
    E.N = N;
    E.x_st = p.x_st;
    E.x_en = p.x_en;
    E.Lg = Lg;
    E.S = p.n_samples;
    E.n_keep = p.n_keep;
    E.n_init = p.n_init;
    // bins of np.round((x - x_st)/delta_x) over every image column (gpet.py:605-606)
    E.bin_lo = (int)rint((double)(0 - p.x_st) / (double)p.delta_x);
    E.n_bins = (int)rint((double)(N - 1 - p.x_st) / (double)p.delta_x) - E.bin_lo + 2;
    E.obs_cap = p.obs_cap > E.n_bins ? p.obs_cap : E.n_bins;
    E.n_cap = E.n_init + E.obs_cap;
    E.r_cap = p.factor_cap > 0 ? p.factor_cap : 96;  // <= 96: the LDS-resident Jacobi path
    if (E.r_cap > Lg) E.r_cap = Lg;
    // A capacity above 96 selects the whole-GPU Jacobi on the full covariance; that path is
    // chosen per batch, so then every edge keeps all Lg directions.
    if (any_big) E.r_cap = Lg;
    E.z_cols = p.z_cols > 0 ? p.z_cols : E.r_cap;
    if (E.z_cols > Lg) E.z_cols = Lg;
    if (any_big) E.z_cols = Lg;
    if (E.z_cols < E.r_cap) E.r_cap = E.z_cols;
    E.a_rows_cap = (E.z_cols >= Lg) ? Lg : E.r_cap;
    // ring of pre-generated normals: look-ahead + 2 slots.  Small batches are latency-bound in the generator and draw
    // 8 iterations ahead (gpet_trace_iterate); a batch that fills the GPU draws 1 ahead (up to 3 by option): 4 slots
    // instead of 16 -- at 1024 edges of the bench shape 2.4 GB instead of 9.4 GB of an 18 GB arena.  Full-stream mode
    // (z_cols == Lg: full-rank covariances, tests) holds whole 8 MB streams per slot: 2.
    // slots of the normals ring: 16 for small batches (eight iterations ahead on the side stream), 9 above 64 edges (the
    // eight iterations of a group are generated by one launch), 2 when a row holds the whole grid (config 3)
    E.z_ring = (E.z_cols >= Lg && Lg > 128) ? 2 : (B <= 64 ? 16 : 9);
    // small batches are bound by the chain of Jacobi rounds: their rotations are logged and the eigenvectors formed by a
    // second kernel (k_jacobi_wpass); 40 sweeps x (m - 1) rounds x m / 2 pairs x 16 bytes = 2.9 MB per edge at rank 96
    const int jlog_max_b = option("jlog_max_b");  // (32: the rotation-log form pays while the chain of rounds is the time)
    E.jlog_cap = (B <= jlog_max_b && E.r_cap <= 96) ? 40 : 0;
    E.kernel_type = p.kernel_type;
    E.nu_code = p.kernel_type == GPET_KERNEL_MATERN ? nu_to_code(p.nu) : 2;
    E.nu_gen = p.nu;
    E.inv_gamma_nu = (E.nu_code == 3) ? 1.0 / tgamma(p.nu) : 1.0;
    E.tab_ok = 1;
    if (E.nu_code == 3) any_gen_nu = true;
    E.fix_endpoints = p.fix_endpoints;
    E.delta_x = p.delta_x;
    E.pixel_thresh = p.pixel_thresh;
    E.algo_thresh = Lg / p.delta_x - (p.pixel_thresh - 1);  // gpet.py:117-119
    E.sigma_f = p.sigma_f;
    E.length_scale = p.length_scale;
    E.noise_y = p.noise_y;
    E.jitter = p.jitter;
    if (Lg > bd.Lg) bd.Lg = Lg;
    if (Lg & 1) bd.lg_even = 0;
    if (E.S > bd.S) bd.S = E.S;
    if (E.n_keep > bd.n_keep) bd.n_keep = E.n_keep;
    if (E.z_cols > bd.z_cols) bd.z_cols = E.z_cols;
    if (E.r_cap > bd.r_cap) bd.r_cap = E.r_cap;
    if (E.n_cap > bd.n_cap) bd.n_cap = E.n_cap;
    if (E.n_bins > bd.n_bins) bd.n_bins = E.n_bins;
    if (E.obs_cap > bd.obs_cap) bd.obs_cap = E.obs_cap;
    if (E.a_rows_cap > bd.a_rows_cap) bd.a_rows_cap = E.a_rows_cap;
    if (bd.z_ring == 0 || E.z_ring < bd.z_ring) bd.z_ring = E.z_ring;
    bd.jlog = E.jlog_cap > 0 ? 1 : 0;
  }
  bd.rng4 = normals4_applies(b->h_edges.data(), B) ? 1 : 0;
  b->bd = bd;
  // measure, allocate, carve
  const size_t px = (size_t)M * N;
  Carver meas;
  std::vector<EdgeDev> tmp = b->h_edges;
  float* shared_grad = nullptr;
  float* shared_kde = nullptr;
  if (b->share_image) {
    shared_grad = meas.take<float>(px);
    shared_kde = meas.take<float>(px);
  }
  (void)meas.take<gpet_scalars>((size_t)B);
  (void)meas.take<double>((size_t)B * 2 * bd.Lg);
  (void)meas.take<double>((size_t)B * 12);
  (void)meas.take<long long>((size_t)B * 2 * bd.obs_cap);
  int n_init_max = 1;
  for (int e = 0; e < B; ++e) n_init_max = b->h_edges[e].n_init > n_init_max ? b->h_edges[e].n_init : n_init_max;
  (void)meas.take<long long>((size_t)B * 2 * (size_t)n_init_max);
  for (int e = 0; e < B; ++e) carve_edge(meas, tmp[e], !b->share_image, bd.Lg);
  (void)shared_grad;
  (void)shared_kde;
  b->arena_bytes = meas.off + 256;
  hipError_t he = hipMalloc(&b->arena, b->arena_bytes);
  if (he != hipSuccess) {
    b->arena = nullptr;
    return fail(c, GPET_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", b->arena_bytes, hipGetErrorString(he));
  }
  HIPCHK(c, hipMemsetAsync(b->arena, 0, b->arena_bytes, c->stream));
  Carver cv;
  cv.base = b->arena;
  if (b->share_image) {
    shared_grad = cv.take<float>(px);
    shared_kde = cv.take<float>(px);
  }
  b->d_scalars = cv.take<gpet_scalars>((size_t)B);
  b->d_fin_out = cv.take<double>((size_t)B * 2 * bd.Lg);
  b->d_fin_par = cv.take<double>((size_t)B * 12);
  b->d_obs = cv.take<long long>((size_t)B * 2 * bd.obs_cap);
  b->d_init = cv.take<long long>((size_t)B * 2 * (size_t)n_init_max);
  b->h_scalars.resize(B);
  for (int e = 0; e < B; ++e) {
    EdgeDev& E = b->h_edges[e];
    E.sc = b->d_scalars + e;
    E.fin_out = b->d_fin_out + (size_t)e * 2 * bd.Lg;
    carve_edge(cv, E, !b->share_image, bd.Lg);
    E.fin_par = b->d_fin_par + (size_t)e * 12;                 // (batch-contiguous; the per-edge carve is unused)
    E.obs_xy = b->d_obs + (size_t)e * 2 * bd.obs_cap;
    E.init_xy = b->d_init + (size_t)e * 2 * (size_t)n_init_max;  // (batch-contiguous; the per-edge carve is unused)
    if (b->share_image) {
      E.grad = shared_grad;
      E.grad_kde = shared_kde;
    }
  }
  HIPCHK(c, hipMalloc(&b->d_edges, sizeof(EdgeDev) * B));
  HIPCHK(c, hipMalloc(&b->d_seeds, sizeof(unsigned int) * B));
  HIPCHK(c, hipMalloc(&b->d_minmax, sizeof(unsigned int) * 2 * (size_t)B));
  {
    // (the stream the normals run ahead of the loop on: default priority -- lowest / highest were measured, +-0)
    // Option side_own_queue: the HIP runtime maps a process's streams onto a pool of GPU_MAX_HW_QUEUES hardware queues, and a
    // batch whose look-ahead stream lands on the queue of its own loop runs the two IN ORDER -- the loop then stands still for
    // every 3 ms generator launch (a 32-edge loop 11.8 or 15.9 ms by luck, in a process that has created other contexts
    // before).  A stream created with a CU mask gets a hardware queue of its own from the runtime: with all CUs enabled it is
    // an ordinary stream that shares its queue with nobody.
    bool made = false;
    if (option("side_own_queue")) {  // (off by default: measured, see the option's description)
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) {
        const unsigned int words = (unsigned int)((prop.multiProcessorCount + 31) / 32);
        std::vector<uint32_t> mask(words, 0xFFFFFFFFu);
        if (prop.multiProcessorCount % 32) mask[words - 1] = (1u << (prop.multiProcessorCount % 32)) - 1u;
        made = hipExtStreamCreateWithCUMask(&b->side, words, mask.data()) == hipSuccess;
        if (!made) (void)hipGetLastError();
      }
    }
    if (!made) HIPCHK(c, hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
  }
  {
    int pr_least = 0, pr_greatest = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
    // the stream the converged fits' objective runs on has the highest priority (its launches are small and many)
    const int prio = pr_greatest;
    HIPCHK(c, hipStreamCreateWithPriority(&b->fit, hipStreamNonBlocking, prio));
  }
  for (int i = 0; i < 16; ++i) HIPCHK(c, hipEventCreateWithFlags(&b->ev_norm[i], hipEventDisableTiming));
  for (int i = 0; i < 16; ++i) HIPCHK(c, hipEventCreateWithFlags(&b->ev_gemm[i], hipEventDisableTiming));
  for (int i = 0; i < 16; ++i) HIPCHK(c, hipEventCreateWithFlags(&b->ev_pix[i], hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&b->ev_main, hipEventDisableTiming));
  // upload: gradient image(s) re-normalised on the device (gpet.py:97), inits, initial scalars
  HIPCHK(c, hipMalloc(&b->d_raw, px * sizeof(float)));
  {
    int rcu = upload_images(b, grad, flags);
    if (rcu) return rcu;
  }
  // (one copy each for the init points and the initial scalars of all edges: 3 x B small copies were most of the constructor)
  std::vector<long long> h_init((size_t)B * 2 * (size_t)n_init_max, 0);
  auto pristine_scalars = [&]() {
    for (int e = 0; e < B; ++e) {
      gpet_scalars& s0 = b->h_scalars[e];
      memset(&s0, 0, sizeof s0);
      s0.score_thresh = params[e].score_thresh;
      s0.done = (0 >= b->h_edges[e].algo_thresh) ? 1 : 0;  // gpet.py:829 with no observations yet
    }
    return hipMemcpyAsync(b->d_scalars, b->h_scalars.data(), sizeof(gpet_scalars) * (size_t)B, hipMemcpyHostToDevice, c->stream);
  };
  for (int e = 0; e < B; ++e)
    memcpy(&h_init[(size_t)e * 2 * (size_t)n_init_max], init_xy[e], sizeof(long long) * 2 * (size_t)b->h_edges[e].n_init);
  HIPCHK(c, hipMemcpyAsync(b->d_init, h_init.data(), sizeof(long long) * h_init.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, pristine_scalars());
  HIPCHK(c, hipMemcpyAsync(b->d_edges, b->h_edges.data(), sizeof(EdgeDev) * B, hipMemcpyHostToDevice, c->stream));
  if (any_gen_nu) HIPCHK(c, launch_rho_tab(c->stream, b->d_edges, B, N));
  // gradient KDE of every distinct image (gpet.py:127)
  HIPCHK(c, launch_kde(c->stream, b->d_edges, b->share_image ? 1 : B, b->bd, 1));
  HIPCHK(c, gpet_wait(c->stream));
  // structured loop path: eigenbasis of the grid's correlation matrix, once per edge.  Usable when the
  // LDS Jacobi applies (capacity <= 96) and every init x lies on the grid; option "struct_path" = 0 disables it.
  b->structured = false;
  if (!any_big && option("struct_path")) {
    bool ok = true;
    // without fix_endpoints the pixel selection admits every image column (gpet.py:655-657 only filters when it is
    // set), so the loop can accept observations outside [x_st, x_en] unless the edge spans the whole image: those are
    // not on the grid the prior eigenbasis indexes
    for (int e = 0; e < B && ok; ++e)
      if (!b->h_edges[e].fix_endpoints && !(b->h_edges[e].x_st == 0 && b->h_edges[e].x_en == N - 1)) ok = false;
    for (int e = 0; e < B && ok; ++e)
      for (int i = 0; i < b->h_edges[e].n_init; ++i) {
        const int64_t x = init_xy[e][2 * i];
        if (x < b->h_edges[e].x_st || x > b->h_edges[e].x_en) ok = false;
      }
    if (ok) {
      // Edges of the same grid length, first column, kernel and length scale have the same prior eigenbasis bit for bit
      // (k_rho_fill forms the lags as fl((x_st+i)/l) - fl((x_st+j)/l), which depends on x_st in the last bits unless l is
      // a power of two -- so x_st is part of the match; the amplitude is not: the matrix has unit amplitude).  It is computed
      // for the first edge of every such class only (a batch of 1 024 equal edges: one factorisation instead of 1 024, 7.5 ms
      // of the constructor) and the others read that edge's copy, which then stays in L2 for the whole batch (k_struct_H
      // gathers its rows, k_struct_rows streams it: 288 KB per edge at rank 72, Lg 500).  Option "shared_basis" = 0: every
      // edge computes and keeps its own.
      std::vector<int> rep_of((size_t)B), reps;
      for (int e = 0; e < B; ++e) {
        const EdgeDev& E = b->h_edges[e];
        int found = -1;
        if (option("shared_basis"))
          for (size_t k = reps.size() > 8 ? reps.size() - 8 : 0; k < reps.size() && found < 0; ++k) {  // (batches are homogeneous or nearly so: a short search)
            const EdgeDev& F = b->h_edges[reps[k]];
            if (F.Lg == E.Lg && F.x_st == E.x_st && F.kernel_type == E.kernel_type && F.nu_code == E.nu_code && F.nu_gen == E.nu_gen &&
                F.length_scale == E.length_scale && F.r_cap == E.r_cap)
              found = reps[k];
          }
        if (found < 0) {
          found = e;
          reps.push_back(e);
        }
        rep_of[(size_t)e] = found;
      }
      if ((int)reps.size() == B) {
        HIPCHK(c, launch_struct_basis(c->stream, b->d_edges, B, b->bd));
      } else {
        std::vector<EdgeDev> h_rep(reps.size());
        for (size_t k = 0; k < reps.size(); ++k) h_rep[k] = b->h_edges[reps[k]];
        EdgeDev* d_rep = nullptr;
        HIPCHK(c, hipMalloc(&d_rep, sizeof(EdgeDev) * reps.size()));
        hipError_t e1 = hipMemcpyAsync(d_rep, h_rep.data(), sizeof(EdgeDev) * reps.size(), hipMemcpyHostToDevice, c->stream);
        if (e1 == hipSuccess) e1 = launch_struct_basis(c->stream, d_rep, (int)reps.size(), b->bd);
        if (e1 == hipSuccess) e1 = gpet_wait(c->stream);
        (void)hipFree(d_rep);
        HIPCHK(c, e1);
      }
      int rc2 = fetch_all_scalars(b);
      if (rc2) return rc2;
      int r0_max = 0;
      for (int e = 0; e < B; ++e) {
        EdgeDev& E = b->h_edges[e];
        const EdgeDev& F = b->h_edges[rep_of[(size_t)e]];
        const gpet_scalars& s = b->h_scalars[rep_of[(size_t)e]];
        if (s.status != GPET_OK || s.rank < 1 || s.rank >= E.r_cap) ok = false;  // rank capacity reached
        E.r0 = s.rank;
        if (rep_of[(size_t)e] != e) {
          E.Q0 = F.Q0;
          E.lam0 = F.lam0;
          E.h0 = F.h0;  // (the sign convention's weights in that basis)
        }
        if (s.rank > r0_max) r0_max = s.rank;
      }
      b->bd.r0_max = r0_max;
      // (n_cap <= 128: k_struct_H keeps U in LDS -- it fits with L streamed row by row; larger: U in HBM, blocked)
      if (b->bd.n_cap <= 128 &&
          ((size_t)b->bd.n_cap * (r0_max | 1) + b->bd.n_cap + b->bd.r_cap) * sizeof(double) > (size_t)STRUCT_H_LDS_MAX)
        ok = false;
      // back to the pristine scalar state
      for (int e = 0; e < B; ++e) b->h_edges[e].structured = ok ? 1 : 0;
      HIPCHK(c, pristine_scalars());
      HIPCHK(c, hipMemcpyAsync(b->d_edges, b->h_edges.data(), sizeof(EdgeDev) * B, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, gpet_wait(c->stream));
      b->structured = ok;
    }
  }
  guard.b = nullptr;  // success: the caller owns the batch
  *out = b;
  return GPET_OK;
}

void gpet_batch_destroy(gpet_batch* b) {
  GPET_BATCH_SCOPE(b);
  if (!b) return;
  (void)hipSetDevice(b->ctx->device);
  (void)hipStreamSynchronize(b->ctx->stream);
  if (b->arena) (void)hipFree(b->arena);
  if (b->d_edges) (void)hipFree(b->d_edges);
  if (b->d_edges_act) (void)hipFree(b->d_edges_act);
  if (b->d_seeds_act) (void)hipFree(b->d_seeds_act);
  if (b->d_seeds) (void)hipFree(b->d_seeds);
  if (b->d_minmax) (void)hipFree(b->d_minmax);
  if (b->d_raw) (void)hipFree(b->d_raw);
  if (b->ev_l0) (void)hipEventDestroy(b->ev_l0);
  if (b->ev_l1) (void)hipEventDestroy(b->ev_l1);
  if (b->d_fin_stage) (void)hipFree(b->d_fin_stage);
  if (b->d_fin_n) (void)hipFree(b->d_fin_n);
  if (b->fit) {
    (void)hipStreamSynchronize(b->fit);
    (void)hipStreamDestroy(b->fit);
  }
  if (b->side) {
    (void)hipStreamSynchronize(b->side);
    (void)hipStreamDestroy(b->side);
  }
  for (int i = 0; i < 16; ++i)
    if (b->ev_norm[i]) (void)hipEventDestroy(b->ev_norm[i]);
  for (int i = 0; i < 16; ++i)
    if (b->ev_gemm[i]) (void)hipEventDestroy(b->ev_gemm[i]);
  for (int i = 0; i < 16; ++i)
    if (b->ev_pix[i]) (void)hipEventDestroy(b->ev_pix[i]);
  if (b->ev_main) (void)hipEventDestroy(b->ev_main);
  if (b->d_edge_of) (void)hipFree(b->d_edge_of);
  if (b->d_theta) (void)hipFree(b->d_theta);
  if (b->d_f) (void)hipFree(b->d_f);
  if (b->d_g) (void)hipFree(b->d_g);
  if (b->lb_mem) (void)hipFree(b->lb_mem);
  if (b->mtj_work) (void)hipFree(b->mtj_work);
  if (b->d_mtj_poly) (void)hipFree(b->d_mtj_poly);
  if (b->big_mem) (void)hipFree(b->big_mem);
  for (hipEvent_t ev : b->lb_events) (void)hipEventDestroy(ev);
  delete b;
}

int gpet_batch_size(const gpet_batch* b) { return b ? b->B : 0; }

int gpet_batch_info(const gpet_batch* b, int e, int32_t* out, int count) {
  if (!b || e < 0 || e >= b->B || !out) return GPET_ERR_BAD_ARG;
  const EdgeDev& E = b->h_edges[e];
  const int32_t v[14] = {E.Lg, E.S, E.n_keep, E.n_cap, E.r_cap, E.z_cols, E.a_rows_cap, E.n_bins, E.obs_cap, E.algo_thresh,
                         b->structured ? 1 : 0, E.r0, E.z_ring, (int32_t)(b->arena_bytes >> 20)};
  for (int i = 0; i < count && i < 14; ++i) out[i] = v[i];
  return GPET_OK;
}

}  // extern "C"

// (shared with the other units: C++ linkage)
static int read_scalars(gpet_batch* b, int e, gpet_scalars* s) {
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipMemcpyAsync(s, b->h_edges[e].sc, sizeof *s, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

int fetch_all_scalars(gpet_batch* b) {
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipMemcpyAsync(b->h_scalars.data(), b->d_scalars, sizeof(gpet_scalars) * b->B, hipMemcpyDeviceToHost,
                           c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

int check_device_status(gpet_batch* b) {
  gpet_ctx* c = b->ctx;
  int rc = fetch_all_scalars(b);
  if (rc) return rc;
  for (int e = 0; e < b->B; ++e) {
    const gpet_scalars& s = b->h_scalars[e];
    if (s.status == GPET_ERR_NOT_PD)
      return fail(c, GPET_ERR_NOT_PD, "edge %d: the kernel matrix is not positive definite (n=%d)", e, s.n);
    if (s.status == GPET_ERR_RANK_CAP)
      return fail(c, GPET_ERR_RANK_CAP, "edge %d: posterior covariance rank exceeds factor_cap=%d", e, b->h_edges[e].r_cap);
    if (s.status == GPET_ERR_ITER_CAP)
      return fail(c, GPET_ERR_ITER_CAP, "edge %d: no score threshold yields enough new pixels (the reference would loop forever, gpet.py:591-609)", e);
    if (s.status != GPET_OK) return fail(c, s.status, "edge %d: device status %d", e, s.status);
  }
  return GPET_OK;
}

extern "C" {

// The any-rank factor's rows of the trace that ends here may serve as the FIRST warm start of the next one -- only when
// the caller says the next trace is the next frame of a sequence (gpet_batch_set_images with GPET_IMAGES_NEXT_FRAME: the
// same chain, a similar covariance).  They are in slot (iters_done - 1) & 1 of the ring if its tag says "iteration
// iters_done - 1, full rank, converged".  Every other restart (gpet_batch_reset, gpet_batch_set_obs) clears all tags: a
// trace is then a function of (image, seed, observations) alone, whatever the batch object ran before.
static int clear_factor_rows(gpet_batch* b, int e) {
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipMemsetAsync(b->h_edges[e].ap_tag, 0, 3 * sizeof(int), c->stream));
  return GPET_OK;
}

// all edges at once: one wait for the tags, one for their replacements (iters[e] = iterations the edge's last trace ran)
static int carry_factor_rows_all(gpet_batch* b, const std::vector<int>& iters) {
  gpet_ctx* c = b->ctx;
  const int B = b->B;
  std::vector<int> tags((size_t)3 * B, 0);
  for (int e = 0; e < B; ++e)
    HIPCHK(c, hipMemcpyAsync(&tags[3 * e], b->h_edges[e].ap_tag, 3 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  for (int e = 0; e < B; ++e) {
    const int it = iters[e], slot = (it - 1) & 1;
    const int keep = (b->h_edges[e].r_cap > 96 && it >= 1 && tags[3 * e + slot] == it) ? slot + 1 : 0;
    tags[3 * e] = tags[3 * e + 1] = 0;
    tags[3 * e + 2] = keep;
    HIPCHK(c, hipMemcpyAsync(b->h_edges[e].ap_tag, &tags[3 * e], 3 * sizeof(int), hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(c, gpet_wait(c->stream));  // (tags is a local)
  return GPET_OK;
}

int gpet_batch_set_obs(gpet_batch* b, int e, const int64_t* obs_xy, int n_obs) {
  GPET_BATCH_SCOPE(b);
  if (!b || e < 0 || e >= b->B || n_obs < 0 || (n_obs > 0 && !obs_xy)) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  EdgeDev& E = b->h_edges[e];
  if (n_obs > E.obs_cap) return fail(c, GPET_ERR_BAD_ARG, "n_obs=%d exceeds obs_cap=%d", n_obs, E.obs_cap);
  // the pixel kernels index the density images with the observations (gpet.py:568: kde_arr[pre_fobs[:,0], pre_fobs[:,1]]
  // raises IndexError in the reference for pixels outside the image)
  for (int i = 0; i < n_obs; ++i)
    if (obs_xy[2 * i] < 0 || obs_xy[2 * i] >= E.N || obs_xy[2 * i + 1] < 0 || obs_xy[2 * i + 1] >= E.M)
      return fail(c, GPET_ERR_BAD_ARG, "observation %d = (%lld, %lld) lies outside the %d x %d image", i,
                  (long long)obs_xy[2 * i], (long long)obs_xy[2 * i + 1], E.M, E.N);
  HIPCHK(c, hipSetDevice(c->device));
  gpet_scalars s;
  int rc = read_scalars(b, e, &s);
  if (rc) return rc;
  s.n_obs = n_obs;
  s.done = (n_obs >= E.algo_thresh) ? 1 : 0;
  s.status = GPET_OK;
  const int iters_done = s.iter;
  s.iter = 0;            // a new observation set restarts the edge's loop (gpet.py:820-828)
  HIPCHK(c, hipMemsetAsync(E.wq_tag, 0, 2 * sizeof(int), c->stream));  // (and forgets the last trace's eigenvectors)
  if (iters_done >= 1) {  // (0: gpet_batch_reset / gpet_batch_set_images has been here already and decided what stays)
    int rc3 = clear_factor_rows(b, e);
    if (rc3) return rc3;
  }
  b->iters_issued = 0;   // (all edges of a batch are restarted together)
  if ((int)b->h_nobs_prev.size() != b->B) b->h_nobs_prev.assign(b->B, 0);
  b->h_nobs_prev[e] = n_obs;  // (the loop's group sizes follow the growth of the observation sets from here)
  b->norm_issued = 0;
  if (b->structured)
    for (int i = 0; i < n_obs; ++i)
      if (obs_xy[2 * i] < E.x_st || obs_xy[2 * i] > E.x_en) {  // off-grid training point: generic path from now on
        b->structured = false;
        break;
      }
  if (n_obs > 0)
    HIPCHK(c, hipMemcpyAsync(E.obs_xy, obs_xy, sizeof(long long) * 2 * n_obs, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(E.sc, &s, sizeof s, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

int gpet_batch_read(gpet_batch* b, int e, int which, void* dst, size_t bytes) {
  GPET_BATCH_SCOPE(b);
  if (!b || e < 0 || e >= b->B || !dst) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  const EdgeDev& E = b->h_edges[e];
  gpet_scalars s;
  int rc = read_scalars(b, e, &s);
  if (rc) return rc;
  const void* src = nullptr;
  size_t avail = 0;
  const size_t Lg = E.Lg, n = s.n, px = (size_t)E.M * E.N;
  switch (which) {
    case GPET_BUF_X_TRAIN: src = E.xt; avail = n * 8; break;
    case GPET_BUF_Y_TRAIN: src = E.yt; avail = n * 8; break;
    case GPET_BUF_NOISE_W: src = E.wt; avail = n * 8; break;
    case GPET_BUF_ALPHA: src = E.alpha; avail = n * 8; break;
    case GPET_BUF_MEAN: src = E.mean; avail = Lg * 8; break;
    case GPET_BUF_STD: src = E.std; avail = Lg * 8; break;
    case GPET_BUF_COV: src = E.cov; avail = Lg * Lg * 8; break;
    case GPET_BUF_FACTOR: src = E.A; avail = (size_t)s.rank * Lg * 8; break;
    case GPET_BUF_EIGVALS: src = E.theta; avail = (size_t)s.rank * 8; break;
    case GPET_BUF_NORMALS: src = E.Z + (size_t)(s.iter % E.z_ring) * E.S * E.z_cols; avail = (size_t)E.S * E.z_cols * 8; break;
    case GPET_BUF_SAMPLES: {
      // rows of Yp elements on the device (f32 after gpet_batch_set_sample_dtype); the interface is a dense [S][Lg] f64 matrix
      const size_t cnt = (size_t)E.S * Lg, pitch = (size_t)E.Yp, esz = E.y_f32 ? 4 : 8;
      if (bytes > cnt * 8) bytes = cnt * 8;
      std::vector<char> tmp((size_t)E.S * pitch * esz);
      HIPCHK(c, hipMemcpyAsync(tmp.data(), E.Y, tmp.size(), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, gpet_wait(c->stream));
      double* o = (double*)dst;
      for (size_t i = 0; i < bytes / 8; ++i) {
        const size_t at = (i / Lg) * pitch + i % Lg;
        o[i] = E.y_f32 ? (double)((const float*)tmp.data())[at] : ((const double*)tmp.data())[at];
      }
      return GPET_OK;
    }
    case GPET_BUF_COSTS: src = E.costs; avail = (size_t)E.S * 8; break;
    case GPET_BUF_BEST_IDX: src = E.best_idx; avail = (size_t)E.n_keep * 4; break;
    case GPET_BUF_BEST_COSTS: src = E.best_costs; avail = (size_t)E.n_keep * 8; break;
    case GPET_BUF_OBS: src = E.obs_xy; avail = (size_t)s.n_obs * 16; break;
    case GPET_BUF_KDE: src = E.kde; avail = px * 4; break;
    case GPET_BUF_GRAD_KDE: src = E.grad_kde; avail = px * 4; break;
    case GPET_BUF_GRAD: src = E.grad; avail = px * 4; break;
    case GPET_BUF_SCALARS:
      memcpy(dst, &s, bytes < sizeof s ? bytes : sizeof s);
      return GPET_OK;
    case GPET_BUF_FIN_PAR: src = E.fin_par; avail = 12 * 8; break;
    case GPET_BUF_FIN_STARTS:
      if (!b->lb_starts) return fail(c, GPET_ERR_STATE, "no converged fit has run on this batch yet");
      src = b->lb_starts + (size_t)e * 39;
      avail = 39 * 8;
      break;
    case GPET_BUF_FIN_TRAIN: {
      const size_t nc = E.n_cap;
      if (bytes < 3 * nc * 8) return fail(c, GPET_ERR_BAD_ARG, "FIN_TRAIN read needs %zu bytes", 3 * nc * 8);
      HIPCHK(c, hipMemcpyAsync((char*)dst, E.fin_x, nc * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync((char*)dst + nc * 8, E.fin_y, nc * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync((char*)dst + 2 * nc * 8, E.fin_w, nc * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, gpet_wait(c->stream));
      return GPET_OK;
    }
    case GPET_BUF_CHOL: {
      // compact n x n lower-triangular copy (upper part zeroed)
      if (bytes < n * n * 8) return fail(c, GPET_ERR_BAD_ARG, "CHOL read needs %zu bytes", n * n * 8);
      std::vector<double> full((size_t)E.n_cap * E.n_cap);
      HIPCHK(c, hipMemcpyAsync(full.data(), E.K, full.size() * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, gpet_wait(c->stream));
      double* o = (double*)dst;
      for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) o[i * n + j] = (j <= i) ? full[i * E.n_cap + j] : 0.0;
      return GPET_OK;
    }
    default:
      return fail(c, GPET_ERR_BAD_ARG, "gpet_batch_read: unknown buffer %d", which);
  }
  if (bytes > avail) bytes = avail;
  if (which == GPET_BUF_EIGVALS) {
    std::vector<double> th(s.rank);
    std::vector<int> ord(s.rank);
    if (s.rank > 0) {
      HIPCHK(c, hipMemcpyAsync(th.data(), E.theta, (size_t)s.rank * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(ord.data(), E.order, (size_t)s.rank * 4, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, gpet_wait(c->stream));
    }
    double* o = (double*)dst;
    for (size_t k = 0; k < bytes / 8; ++k) o[k] = th[ord[k]];
    return GPET_OK;
  }
  if (bytes) {
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, gpet_wait(c->stream));
  }
  return GPET_OK;
}

int gpet_batch_write(gpet_batch* b, int e, int which, const void* src, size_t bytes, int rows) {
  GPET_BATCH_SCOPE(b);
  if (!b || e < 0 || e >= b->B || !src) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  EdgeDev& E = b->h_edges[e];
  const size_t Lg = E.Lg, px = (size_t)E.M * E.N;
  void* dst = nullptr;
  size_t cap = 0;
  switch (which) {
    case GPET_BUF_FACTOR: {
      if (rows < 0 || rows > E.a_rows_cap || rows > E.z_cols)
        return fail(c, GPET_ERR_BAD_ARG, "factor rows=%d exceeds capacity (a_rows_cap=%d, z_cols=%d); create the batch with z_cols=Lg", rows, E.a_rows_cap, E.z_cols);
      dst = E.A;
      cap = (size_t)rows * Lg * 8;
      if (bytes != cap) return fail(c, GPET_ERR_BAD_ARG, "factor write: expected %zu bytes", cap);
      gpet_scalars s;
      int rc = read_scalars(b, e, &s);
      if (rc) return rc;
      s.rank = rows;
      HIPCHK(c, hipMemcpyAsync(E.sc, &s, sizeof s, hipMemcpyHostToDevice, c->stream));
      E.factor_injected = 1;
      HIPCHK(c, hipMemcpyAsync(b->d_edges + e, &E, sizeof E, hipMemcpyHostToDevice, c->stream));
      b->have_factor = true;
      break;
    }
    case GPET_BUF_NORMALS: {
      gpet_scalars s;
      int rc = read_scalars(b, e, &s);
      if (rc) return rc;
      dst = E.Z + (size_t)(s.iter % E.z_ring) * E.S * E.z_cols;
      cap = (size_t)E.S * E.z_cols * 8;
      b->have_normals = true;
      break;
    }
    case GPET_BUF_SAMPLES: {
      b->have_samples = true;
      // (dense [S][Lg] f64 in, rows of Yp elements on the device; rounded to f32 here, as the GEMM does when it stores)
      const size_t cnt = (size_t)E.S * Lg, pitch = (size_t)E.Yp, esz = E.y_f32 ? 4 : 8, nel = bytes / 8;
      if (bytes > cnt * 8) return fail(c, GPET_ERR_BAD_ARG, "gpet_batch_write: %zu bytes exceed capacity %zu", bytes, cnt * 8);
      const size_t full = nel / Lg, rest = nel % Lg;
      std::vector<char> tmp((full * pitch + rest) * esz, 0);
      const double* in = (const double*)src;
      for (size_t i = 0; i < nel; ++i) {
        const size_t at = (i / Lg) * pitch + i % Lg;
        if (E.y_f32) ((float*)tmp.data())[at] = (float)in[i];
        else ((double*)tmp.data())[at] = in[i];
      }
      if (!tmp.empty()) HIPCHK(c, hipMemcpyAsync(E.Y, tmp.data(), tmp.size(), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, gpet_wait(c->stream));
      return GPET_OK;
    }
    case GPET_BUF_GRAD_KDE: dst = (void*)E.grad_kde; cap = px * 4; break;
    case GPET_BUF_KDE: dst = E.kde; cap = px * 4; break;
    case GPET_BUF_COSTS: dst = E.costs; cap = (size_t)E.S * 8; break;
    case GPET_BUF_BEST_IDX: dst = E.best_idx; cap = (size_t)E.n_keep * 4; b->have_scores = true; break;
    case GPET_BUF_BEST_COSTS: dst = E.best_costs; cap = (size_t)E.n_keep * 8; break;
    case GPET_BUF_MEAN: dst = E.mean; cap = Lg * 8; break;
    case GPET_BUF_COV: dst = E.cov; cap = Lg * Lg * 8; b->have_fit = true; break;
    case GPET_BUF_SCALARS: dst = E.sc; cap = sizeof(gpet_scalars); break;
    default:
      return fail(c, GPET_ERR_BAD_ARG, "gpet_batch_write: buffer %d is not writable", which);
  }
  if (bytes > cap) return fail(c, GPET_ERR_BAD_ARG, "gpet_batch_write: %zu bytes exceed capacity %zu", bytes, cap);
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

// the batch's own option table (a copy of the process-wide one taken at gpet_batch_create): returns like gpet_set_option
int gpet_batch_set_option(gpet_batch* b, const char* name, int value) {
  if (!b) return -1;
  int prev = 0, idx = -1;
  for (int i = 0; i < option_count(); ++i)
    if (name && strcmp(option_def(i).name, name) == 0) idx = i;
  if (idx < 0 || option_set_in(&b->opts, name, value, &prev) != 0) return -1;
  return prev < 0 ? option_def(idx).hi + 1 : prev;
}
int gpet_batch_get_option(const gpet_batch* b, const char* name, int* value) {
  return (b && option_get_in(&b->opts, name, value) == 0) ? GPET_OK : GPET_ERR_BAD_ARG;
}

int gpet_batch_set_rng(gpet_batch* b, int mode) {
  GPET_BATCH_SCOPE(b);
  if (!b || (mode != 0 && mode != 1)) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, gpet_wait(c->stream));
  if (b->side) HIPCHK(c, gpet_wait(b->side));
  b->rng_mode = mode;
  b->have_normals = false;
  return GPET_OK;
}

int gpet_batch_set_sample_dtype(gpet_batch* b, int f32) {
  GPET_BATCH_SCOPE(b);
  if (!b) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, gpet_wait(c->stream));
  if (b->side) HIPCHK(c, gpet_wait(b->side));
  const int v = f32 ? 1 : 0;
  for (int e = 0; e < b->B; ++e) b->h_edges[e].y_f32 = v;
  b->bd.y_f32 = v;
  HIPCHK(c, hipMemcpyAsync(b->d_edges, b->h_edges.data(), sizeof(EdgeDev) * (size_t)b->B, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  b->have_samples = false;
  return GPET_OK;
}

int gpet_batch_clear_injected_factor(gpet_batch* b, int e) {
  GPET_BATCH_SCOPE(b);
  if (!b || e < 0 || e >= b->B) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  EdgeDev& E = b->h_edges[e];
  E.factor_injected = 0;
  HIPCHK(c, hipMemcpyAsync(b->d_edges + e, &E, sizeof E, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

static int batch_reset(gpet_batch* b, bool next_frame) {
  gpet_ctx* c = b->ctx;
  b->h_nobs_prev.assign(b->B, 0);
  b->iters_issued = 0;
  b->norm_issued = 0;
  HIPCHK(c, hipSetDevice(c->device));
  if (b->bd.r_cap > 96) {  // (any-rank batches keep the last factor rows in a ring)
    if (next_frame && option("oj_warm")) {  // where every edge's last rows are, before the iteration counters go
      int rc = fetch_all_scalars(b);
      if (rc) return rc;
      std::vector<int> iters((size_t)b->B);
      for (int e = 0; e < b->B; ++e) iters[e] = b->h_scalars[e].iter;
      rc = carry_factor_rows_all(b, iters);
      if (rc) return rc;
    } else {
      for (int e = 0; e < b->B; ++e) {
        int rc = clear_factor_rows(b, e);
        if (rc) return rc;
      }
    }
  }
  for (int e = 0; e < b->B; ++e) {
    gpet_scalars& s0 = b->h_scalars[e];
    memset(&s0, 0, sizeof s0);
    s0.score_thresh = b->params[e].score_thresh;
    s0.done = (0 >= b->h_edges[e].algo_thresh) ? 1 : 0;
  }
  HIPCHK(c, hipMemcpyAsync(b->d_scalars, b->h_scalars.data(), sizeof(gpet_scalars) * b->B, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

int gpet_batch_reset(gpet_batch* b) {
  GPET_BATCH_SCOPE(b);
  if (!b) return GPET_ERR_BAD_ARG;
  return batch_reset(b, false);
}

int gpet_batch_set_images(gpet_batch* b, const float* const* grad, unsigned int flags) {
  GPET_BATCH_SCOPE(b);
  if (!b || !grad) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, gpet_wait(c->stream));
  int rc = upload_images(b, grad, flags);
  if (rc) return rc;
  // gradient KDE of every distinct image (gpet.py:127), then the state of a fresh constructor
  HIPCHK(c, launch_kde(c->stream, b->d_edges, b->share_image ? 1 : b->B, b->bd, 1));
  b->have_fit = b->have_factor = b->have_normals = b->have_samples = b->have_scores = false;
  return batch_reset(b, (flags & GPET_IMAGES_NEXT_FRAME) != 0);
}

}  // extern "C"
