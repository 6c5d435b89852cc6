// C ABI, part 4: the converged fit (f2) -- training sets, the batched objective, the device L-BFGS-B, the posterior at the optimum.
#include "gpet_api_internal.h"

static int eval_objective(gpet_batch* b, hipStream_t st, int P, int n_max, const int* d_edge_of, const double* d_theta,
                          double* d_f, double* d_g, const int* d_count = nullptr, int lag_cap = 0);

extern "C" {

int gpet_final_set_training(gpet_batch* b, int e, const double* xs, const double* ys, const double* w, int n) {
  GPET_BATCH_SCOPE(b);
  if (!b || e < 0 || e >= b->B || !xs || !ys || !w || n < 1) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  EdgeDev& E = b->h_edges[e];
  if (n > E.n_cap) return fail(c, GPET_ERR_BAD_ARG, "final fit: n=%d exceeds n_cap=%d", n, E.n_cap);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(E.fin_x, xs, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(E.fin_y, ys, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(E.fin_w, w, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  E.fin_n = n;
  HIPCHK(c, hipMemcpyAsync(b->d_edges + e, &E, sizeof E, hipMemcpyHostToDevice, c->stream));
  double lat[2] = {0.0, 0.0};
  b->fin_lag.resize(b->B, -1);
  b->fin_lag[e] = fin_lattice(xs, n, &lat[0]);
  lat[1] = (double)b->fin_lag[e];
  HIPCHK(c, hipMemcpyAsync(E.fin_par + 9, lat, sizeof lat, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

int gpet_batch_read_scalars_all(gpet_batch* b, gpet_scalars* dst) {
  GPET_BATCH_SCOPE(b);
  if (!b || !dst) return GPET_ERR_BAD_ARG;
  HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
  int rc = fetch_all_scalars(b);
  if (rc) return rc;
  memcpy(dst, b->h_scalars.data(), sizeof(gpet_scalars) * b->B);
  return GPET_OK;
}

int gpet_batch_read_obs_all(gpet_batch* b, int64_t* dst, int32_t* counts, int stride_obs) {
  GPET_BATCH_SCOPE(b);
  if (!b || !dst || !counts || stride_obs < 1) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = fetch_all_scalars(b);
  if (rc) return rc;
  const int cap = b->bd.obs_cap;
  std::vector<long long> host((size_t)b->B * 2 * cap);
  HIPCHK(c, hipMemcpyAsync(host.data(), b->d_obs, host.size() * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  for (int e = 0; e < b->B; ++e) {
    const int n = b->h_scalars[e].n_obs;
    counts[e] = n;
    if (n > stride_obs) return fail(c, GPET_ERR_BAD_ARG, "gpet_batch_read_obs_all: edge %d has %d observations > stride %d", e, n, stride_obs);
    if (n > 0) memcpy(dst + (size_t)e * stride_obs * 2, host.data() + (size_t)e * 2 * cap, sizeof(int64_t) * 2 * n);
  }
  return GPET_OK;
}

int gpet_final_set_training_all(gpet_batch* b, const double* xs, const double* ys, const double* w, const int32_t* n,
                                int stride) {
  GPET_BATCH_SCOPE(b);
  if (!b || !xs || !ys || !w || !n || stride < 1) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  for (int e = 0; e < b->B; ++e) {
    EdgeDev& E = b->h_edges[e];
    if (n[e] < 1 || n[e] > E.n_cap || n[e] > stride) return fail(c, GPET_ERR_BAD_ARG, "final fit: edge %d n=%d out of range", e, n[e]);
    E.fin_n = n[e];
  }
  // three block copies + one scatter kernel (which also sets fin_n on the device) instead of 3 B small copies
  const size_t blk = (size_t)b->B * stride;
  if (3 * blk > b->fin_stage_cap) {
    if (b->d_fin_stage) (void)hipFree(b->d_fin_stage);
    if (b->d_fin_n) (void)hipFree(b->d_fin_n);
    b->d_fin_stage = nullptr;
    b->d_fin_n = nullptr;
    b->fin_stage_cap = 0;
    HIPCHK(c, hipMalloc(&b->d_fin_stage, sizeof(double) * 3 * blk));
    HIPCHK(c, hipMalloc(&b->d_fin_n, sizeof(int) * b->B));
    b->fin_stage_cap = 3 * blk;
  }
  HIPCHK(c, hipMemcpyAsync(b->d_fin_stage, xs, sizeof(double) * blk, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b->d_fin_stage + blk, ys, sizeof(double) * blk, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b->d_fin_stage + 2 * blk, w, sizeof(double) * blk, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b->d_fin_n, n, sizeof(int) * b->B, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, launch_fin_scatter(c->stream, b->d_edges, b->B, b->d_fin_stage, b->d_fin_n, stride));
  std::vector<double> lat((size_t)2 * b->B);
  b->fin_lag.resize(b->B, -1);
  for (int e = 0; e < b->B; ++e) {
    b->fin_lag[e] = fin_lattice(xs + (size_t)e * stride, n[e], &lat[2 * (size_t)e]);
    lat[2 * (size_t)e + 1] = (double)b->fin_lag[e];
  }
  HIPCHK(c, hipMemcpy2DAsync(b->d_fin_par + 9, 12 * sizeof(double), lat.data(), 2 * sizeof(double), 2 * sizeof(double), b->B,
                             hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

int gpet_final_predict_all(gpet_batch* b, const double* par, double* mean_out, double* std_out, int stride) {
  GPET_BATCH_SCOPE(b);
  if (!b || !par || !mean_out || !std_out || stride < b->bd.Lg) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  for (int e = 0; e < b->B; ++e)
    if (b->h_edges[e].fin_n < 1) return fail(c, GPET_ERR_STATE, "gpet_final_predict_all before the training sets are set");
  // (slots 9..11 of every edge stay: the lattice of its training set)
  HIPCHK(c, hipMemcpy2DAsync(b->d_fin_par, 12 * sizeof(double), par, 12 * sizeof(double), 9 * sizeof(double), b->B,
                             hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, launch_final_predict(c->stream, b->d_edges, b->B, b->bd));
  std::vector<double> host((size_t)b->B * 2 * b->bd.Lg);
  HIPCHK(c, hipMemcpyAsync(host.data(), b->d_fin_out, host.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  for (int e = 0; e < b->B; ++e) {
    const int Lg = b->h_edges[e].Lg;
    memcpy(mean_out + (size_t)e * stride, host.data() + (size_t)e * 2 * b->bd.Lg, sizeof(double) * Lg);
    memcpy(std_out + (size_t)e * stride, host.data() + (size_t)e * 2 * b->bd.Lg + b->bd.Lg, sizeof(double) * Lg);
  }
  b->have_fit = false;  // the loop's L/alpha were overwritten by the converged fit
  return check_device_status(b);
}

int gpet_lml_batch(gpet_batch* b, int P, const int32_t* edge_of, const double* theta, double* f_out, double* g_out) {
  GPET_BATCH_SCOPE(b);
  if (!b || P < 1 || !edge_of || !theta || !f_out || !g_out) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  int n_max = 0;
  for (int i = 0; i < P; ++i) {
    if (edge_of[i] < 0 || edge_of[i] >= b->B) return fail(c, GPET_ERR_BAD_ARG, "gpet_lml_batch: bad edge index");
    const int n = b->h_edges[edge_of[i]].fin_n;
    if (n < 1) return fail(c, GPET_ERR_STATE, "gpet_lml_batch before gpet_final_set_training (edge %d)", edge_of[i]);
    if (n > n_max) n_max = n;
  }
  if (P > b->lml_cap) {
    if (b->d_edge_of) (void)hipFree(b->d_edge_of);
    if (b->d_theta) (void)hipFree(b->d_theta);
    if (b->d_f) (void)hipFree(b->d_f);
    if (b->d_g) (void)hipFree(b->d_g);
    b->d_edge_of = nullptr;
    b->d_theta = b->d_f = b->d_g = nullptr;
    b->lml_cap = 0;
    const int cap = P * 2;
    HIPCHK(c, hipMalloc(&b->d_edge_of, sizeof(int) * cap));
    HIPCHK(c, hipMalloc(&b->d_theta, sizeof(double) * 3 * cap));
    HIPCHK(c, hipMalloc(&b->d_f, sizeof(double) * cap));
    HIPCHK(c, hipMalloc(&b->d_g, sizeof(double) * 3 * cap));
    b->lml_cap = cap;
  }
  HIPCHK(c, hipMemcpyAsync(b->d_edge_of, edge_of, sizeof(int) * P, hipMemcpyHostToDevice, b->fit));
  HIPCHK(c, hipMemcpyAsync(b->d_theta, theta, sizeof(double) * 3 * P, hipMemcpyHostToDevice, b->fit));
  if (!b->ev_l0) {
    HIPCHK(c, hipEventCreate(&b->ev_l0));
    HIPCHK(c, hipEventCreate(&b->ev_l1));
  }
  HIPCHK(c, hipEventRecord(b->ev_l0, b->fit));
  {
    // every training set of this call on a lattice the host knows -> tables of (largest lag + 1) entries
    int lag_cap = 1;
    for (int i = 0; i < P && lag_cap > 0; ++i) {
      const int lg = (size_t)edge_of[i] < b->fin_lag.size() ? b->fin_lag[edge_of[i]] : -1;
      lag_cap = lg < 0 ? 0 : (lg + 1 > lag_cap ? lg + 1 : lag_cap);
    }
    int rco = eval_objective(b, b->fit, P, n_max, b->d_edge_of, b->d_theta, b->d_f, b->d_g, nullptr, lag_cap);
    if (rco) return rco;
  }
  HIPCHK(c, hipEventRecord(b->ev_l1, b->fit));
  HIPCHK(c, hipMemcpyAsync(f_out, b->d_f, sizeof(double) * P, hipMemcpyDeviceToHost, b->fit));
  HIPCHK(c, hipMemcpyAsync(g_out, b->d_g, sizeof(double) * 3 * P, hipMemcpyDeviceToHost, b->fit));
  HIPCHK(c, gpet_wait(b->fit));
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, b->ev_l0, b->ev_l1) == hipSuccess) b->lml_ms += (double)ms;
  b->lml_evals += P;
  b->lml_launches += 1;
  return GPET_OK;
}

// -log marginal likelihood + gradient of P problems on stream st: the register-tile kernels up to 250 training points,
// the blocked HBM path (virtual edges, per-problem scratch, evaluated in chunks that fit a 6 GB budget) above.
static int eval_objective(gpet_batch* b, hipStream_t st, int P, int n_max, const int* d_edge_of, const double* d_theta,
                          double* d_f, double* d_g, const int* d_count, int lag_cap) {
  gpet_ctx* c = b->ctx;
  if (n_max <= 250) {
    HIPCHK(c, launch_lml(st, b->d_edges, P, n_max, d_edge_of, d_theta, d_f, d_g, d_count, lag_cap));
    return GPET_OK;
  }
  const int ncap_v = ((n_max + 63) / 64) * 64 + 64;
  if (!b->big_mem || b->big_ncap < ncap_v) {
    if (b->big_mem) {
      HIPCHK(c, gpet_wait(st));
      (void)hipFree(b->big_mem);
      b->big_mem = nullptr;
    }
    const size_t per = lmlbig_scratch_doubles(ncap_v) * sizeof(double);
    int chunk = (int)((6ull << 30) / per);
    if (chunk < 1) chunk = 1;
    if (chunk > 13 * b->B) chunk = 13 * b->B;
    const int nt = ncap_v / 64;
    for (int pass = 0; pass < 2; ++pass) {
      Carver cv;
      cv.base = pass ? b->big_mem : nullptr;
      b->big_vedges = cv.take<EdgeDev>((size_t)chunk);
      b->big_vsc = cv.take<gpet_scalars>((size_t)chunk);
      b->big_part = cv.take<double>((size_t)chunk * nt * nt * 3);
      b->big_scratch = cv.take<double>((size_t)chunk * lmlbig_scratch_doubles(ncap_v));
      if (!pass) {
        hipError_t he = hipMalloc(&b->big_mem, cv.off + 256);
        if (he != hipSuccess) {
          b->big_mem = nullptr;
          return fail(c, GPET_ERR_HIP, "converged fit with %d training points: hipMalloc(%zu bytes) failed: %s", n_max, cv.off + 256, hipGetErrorString(he));
        }
      }
    }
    b->big_chunk = chunk;
    b->big_ncap = ncap_v;
  }
  for (int p0 = 0; p0 < P; p0 += b->big_chunk) {
    const int pc = (P - p0) < b->big_chunk ? (P - p0) : b->big_chunk;
    HIPCHK(c, launch_lml_big(st, b->d_edges, pc, n_max, d_edge_of + p0, d_theta + 3 * (size_t)p0, d_f + p0, d_g + 3 * (size_t)p0,
                             b->big_vedges, b->big_vsc, b->big_scratch, b->big_part, b->big_ncap));
  }
  return GPET_OK;
}

// workspace of the device optimiser for P problems (grown on demand)
static int lb_ensure(gpet_batch* b, int P) {
  gpet_ctx* c = b->ctx;
  const int B = b->B;
  if (b->lb_mem && b->lb_cap_P >= P) return GPET_OK;
  if (b->lb_mem) {
    HIPCHK(c, gpet_wait(b->fit));
    (void)hipFree(b->lb_mem);
    b->lb_mem = nullptr;
  }
  b->lb_scratch_stride = b->bd.n_cap > 256 ? b->bd.n_cap : 256;
  for (int pass = 0; pass < 2; ++pass) {
    Carver cv;
    cv.base = pass ? b->lb_mem : nullptr;
    b->lb_probs = cv.take<char>(lb_prob_bytes() * (size_t)P);
    b->lb_starts = cv.take<double>((size_t)P * 3);
    b->lb_scratch = cv.take<double>((size_t)B * b->lb_scratch_stride);
    b->lb_f = cv.take<double>((size_t)P);
    b->lb_g = cv.take<double>((size_t)P * 3);
    b->lb_theta_out = cv.take<double>((size_t)B * 4);
    for (int h = 0; h < 2; ++h) {
      b->lb_slot_edge[h] = cv.take<int>((size_t)P);
      b->lb_slot_theta[h] = cv.take<double>((size_t)P * 3);
      b->lb_slot_src[h] = cv.take<int>((size_t)P);
    }
    b->lb_count = cv.take<int>(4);
    b->lb_seeds = cv.take<unsigned int>((size_t)B);
    if (!pass) {
      HIPCHK(c, hipMalloc(&b->lb_mem, cv.off + 256));
      // slots beyond the true count of a round are never evaluated (the objective kernels read the count), but the
      // blocked path above 250 points sizes its work by the host's bound: every slot must name a valid edge
      HIPCHK(c, hipMemsetAsync(b->lb_mem, 0, cv.off + 256, b->fit));
    }
  }
  b->lb_cap_P = P;
  return GPET_OK;
}

// The rounds of the device optimiser on stream b->fit: lb_starts holds P = nstart * B start points (edge-major).  The
// number of running problems lives on the device (lb_count[round & 1]); the host reads it only every LB_CHECK rounds and
// sizes the launches by its last known value in between -- workgroups of the objective beyond the true count return at
// once, threads of the advance kernel beyond it too.  A round costs the GPU ~80 us for one edge; a host round trip per
// round would double that.
static int lb_rounds(gpet_batch* b, int P, int n_max, int lag_cap, const LbCfg& cfg, int* rounds_out) {
  gpet_ctx* c = b->ctx;
  hipStream_t st = b->fit;
  HIPCHK(c, launch_lb_init(st, b->lb_probs, P, b->lb_starts, b->lb_slot_edge[0], b->lb_slot_theta[0], b->lb_slot_src[0], cfg));
  // (for problem sets that are resident all at once -- 4 workgroups on each of 256 CUs -- the chain of a problem's ~50
  //  evaluations is what takes the time: 3.2 instead of 4.8 ms for a single edge; bigger sets are throughput-bound and a
  //  workgroup that keeps its registers through the single-threaded state machine costs more than the rounds' launches:
  //  32 instead of 23 ms of objective time per 13 312 problems)
  const bool persistent = opt_fit_persistent() > 0 || (opt_fit_persistent() < 0 && P <= 1024);
  if (persistent && lml16_fit_applies(n_max, lag_cap)) {
    // one launch: a workgroup per problem runs objective and state machine until the problem is done (k_lml16_fit)
    HIPCHK(c, hipMemsetAsync(b->lb_count, 0, 4 * sizeof(int), st));
    if (b->lb_events.size() < 2) {
      hipEvent_t e0, e1;
      HIPCHK(c, hipEventCreate(&e0));
      b->lb_events.push_back(e0);
      HIPCHK(c, hipEventCreate(&e1));
      b->lb_events.push_back(e1);
    }
    HIPCHK(c, hipEventRecord(b->lb_events[0], st));
    HIPCHK(c, launch_lml16_fit(st, b->d_edges, b->lb_probs, P, cfg, lag_cap, 4000, b->lb_count));
    HIPCHK(c, hipEventRecord(b->lb_events[1], st));
    HIPCHK(c, launch_lb_pick(st, b->d_edges, b->B, b->lb_probs, b->lb_theta_out, cfg.nstart));
    int h_cnt[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(h_cnt, b->lb_count, sizeof h_cnt, hipMemcpyDeviceToHost, st));
    HIPCHK(c, gpet_wait(st));
    if (h_cnt[2] > 0)
      return fail(c, GPET_ERR_ITER_CAP, "converged fit: %d problems not finished (4000 evaluations, or a training set the kernel does not serve)", h_cnt[2]);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, b->lb_events[0], b->lb_events[1]) == hipSuccess) b->lml_ms += (double)ms;
    b->lml_evals += h_cnt[0];
    b->lml_launches += 1;
    if (rounds_out) *rounds_out = h_cnt[1];
    return GPET_OK;
  }
  constexpr int LB_CHECK = 4;
  int h_count[2] = {P, 0};
  HIPCHK(c, hipMemcpyAsync(b->lb_count, h_count, sizeof h_count, hipMemcpyHostToDevice, st));
  int n_upper = P, cur = 0, rounds = 0;
  size_t ev_used = 0;
  while (n_upper > 0) {
    if (rounds >= 4000) return fail(c, GPET_ERR_ITER_CAP, "converged fit: %d problems still running after %d rounds", n_upper, rounds);
    int* cnt_cur = b->lb_count + (rounds & 1);
    int* cnt_next = b->lb_count + ((rounds + 1) & 1);
    HIPCHK(c, hipMemsetAsync(cnt_next, 0, sizeof(int), st));
    if (b->lb_events.size() < ev_used + 2) {
      hipEvent_t e0, e1;
      HIPCHK(c, hipEventCreate(&e0));
      b->lb_events.push_back(e0);
      HIPCHK(c, hipEventCreate(&e1));
      b->lb_events.push_back(e1);
    }
    HIPCHK(c, hipEventRecord(b->lb_events[ev_used], st));
    {
      int rco = eval_objective(b, st, n_upper, n_max, b->lb_slot_edge[cur], b->lb_slot_theta[cur], b->lb_f, b->lb_g, cnt_cur, lag_cap);
      if (rco) return rco;
    }
    HIPCHK(c, hipEventRecord(b->lb_events[ev_used + 1], st));
    ev_used += 2;
    HIPCHK(c, launch_lb_advance(st, b->lb_probs, n_upper, cnt_cur, b->lb_slot_src[cur], b->lb_f, b->lb_g, cnt_next,
                                b->lb_slot_edge[1 - cur], b->lb_slot_theta[1 - cur], b->lb_slot_src[1 - cur], cfg));
    b->lml_evals += n_upper;
    b->lml_launches += 1;
    cur ^= 1;
    rounds += 1;
    if (rounds % LB_CHECK == 0) {
      int h_next = 0;
      HIPCHK(c, hipMemcpyAsync(&h_next, cnt_next, sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(c, gpet_wait(st));
      n_upper = h_next;
    }
  }
  for (size_t q = 0; q + 1 < ev_used; q += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, b->lb_events[q], b->lb_events[q + 1]) == hipSuccess) b->lml_ms += (double)ms;
  }
  HIPCHK(c, launch_lb_pick(st, b->d_edges, b->B, b->lb_probs, b->lb_theta_out, cfg.nstart));
  if (rounds_out) *rounds_out = rounds;
  return GPET_OK;
}

int gpet_final_fit_all(gpet_batch* b, const uint32_t* seeds, double* mean_out, double* std_out, double* theta_out,
                       int stride, int32_t* rounds_out) {
  GPET_BATCH_SCOPE(b);
  if (!b || !seeds || !mean_out || !std_out || stride < b->bd.Lg) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = fetch_all_scalars(b);  // (synchronises the loop's stream: the observation sets are final)
  if (rc) return rc;
  const LbCfg cfg = lb_default_cfg();
  const int B = b->B, P = cfg.nstart * B;
  int n_max = 0;
  for (int e = 0; e < B; ++e) {
    const int n = b->h_edges[e].n_init + b->h_scalars[e].n_obs;
    if (n > b->h_edges[e].n_cap) return fail(c, GPET_ERR_BAD_ARG, "converged fit: edge %d n=%d exceeds n_cap", e, n);
    b->h_edges[e].fin_n = n;
    if (n > n_max) n_max = n;
  }
  rc = lb_ensure(b, P);
  if (rc) return rc;
  hipStream_t st = b->fit;
  // the training x are pixel columns of the image: a lattice of fewer than N points (k_fin_prepare leaves the step and
  // the largest lag in fin_par[9..10])
  int lag_cap = b->bd.N > b->bd.Lg ? b->bd.N : b->bd.Lg;
  for (int e = 0; e < B; ++e) {
    const EdgeDev& E = b->h_edges[e];
    if (E.x_st < 0 || E.x_en >= lag_cap) lag_cap = 0;  // (end points outside the image: no bound on the lags)
  }
  b->fin_lag.assign(B, lag_cap > 0 ? lag_cap - 1 : -1);
  HIPCHK(c, hipMemcpyAsync(b->lb_seeds, seeds, sizeof(uint32_t) * B, hipMemcpyHostToDevice, st));
  HIPCHK(c, launch_fin_prepare(st, b->d_edges, B, b->lb_seeds, b->lb_starts, b->lb_scratch, b->lb_scratch_stride, b->bd.n_cap));
  int rounds = 0;
  rc = lb_rounds(b, P, n_max, lag_cap, cfg, &rounds);
  if (rc) return rc;
  HIPCHK(c, launch_final_predict(st, b->d_edges, B, b->bd));
  std::vector<double> host((size_t)B * 2 * b->bd.Lg), th((size_t)B * 4);
  HIPCHK(c, hipMemcpyAsync(host.data(), b->d_fin_out, host.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(th.data(), b->lb_theta_out, th.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(c, gpet_wait(st));
  for (int e = 0; e < B; ++e) {
    const int Lg = b->h_edges[e].Lg;
    memcpy(mean_out + (size_t)e * stride, host.data() + (size_t)e * 2 * b->bd.Lg, sizeof(double) * Lg);
    memcpy(std_out + (size_t)e * stride, host.data() + (size_t)e * 2 * b->bd.Lg + b->bd.Lg, sizeof(double) * Lg);
    if (theta_out) memcpy(theta_out + (size_t)e * 4, th.data() + (size_t)e * 4, sizeof(double) * 4);
  }
  if (rounds_out) *rounds_out = rounds;
  b->have_fit = false;  // the loop's L/alpha were overwritten by the converged fit
  return check_device_status(b);
}

int gpet_final_optimize(gpet_batch* b, int n_starts, const double* starts, const double* bounds, double* theta_out,
                        int32_t* rounds_out) {
  GPET_BATCH_SCOPE(b);
  if (!b || n_starts < 1 || n_starts > 64 || !starts || !bounds || !theta_out) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  const int B = b->B, P = n_starts * B;
  LbCfg cfg;
  cfg.nstart = n_starts;
  for (int k = 0; k < 3; ++k) {
    cfg.lo[k] = bounds[2 * k];
    cfg.hi[k] = bounds[2 * k + 1];
    if (!(cfg.lo[k] <= cfg.hi[k])) return fail(c, GPET_ERR_BAD_ARG, "gpet_final_optimize: empty bound interval %d", k);
  }
  int n_max = 0, lag_cap = 1;
  for (int e = 0; e < B; ++e) {
    const int n = b->h_edges[e].fin_n;
    if (n < 1) return fail(c, GPET_ERR_STATE, "gpet_final_optimize before gpet_final_set_training (edge %d)", e);
    if (n > n_max) n_max = n;
    const int lg = (size_t)e < b->fin_lag.size() ? b->fin_lag[e] : -1;
    lag_cap = (lag_cap == 0 || lg < 0) ? 0 : (lg + 1 > lag_cap ? lg + 1 : lag_cap);
  }
  int rc = lb_ensure(b, P);
  if (rc) return rc;
  HIPCHK(c, gpet_wait(c->stream));  // (the training sets were written on the context's stream)
  HIPCHK(c, hipMemcpyAsync(b->lb_starts, starts, sizeof(double) * 3 * P, hipMemcpyHostToDevice, b->fit));
  int rounds = 0;
  rc = lb_rounds(b, P, n_max, lag_cap, cfg, &rounds);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(theta_out, b->lb_theta_out, sizeof(double) * 4 * B, hipMemcpyDeviceToHost, b->fit));
  HIPCHK(c, gpet_wait(b->fit));
  if (rounds_out) *rounds_out = rounds;
  return check_device_status(b);
}

int gpet_lml_stats(gpet_batch* b, int reset, double* kernel_ms, int64_t* evaluations, int32_t* launches) {
  GPET_BATCH_SCOPE(b);
  if (!b) return GPET_ERR_BAD_ARG;
  if (kernel_ms) *kernel_ms = b->lml_ms;
  if (evaluations) *evaluations = b->lml_evals;
  if (launches) *launches = b->lml_launches;
  if (reset) {
    b->lml_ms = 0.0;
    b->lml_evals = 0;
    b->lml_launches = 0;
  }
  return GPET_OK;
}

}  // extern "C"
