// The converged fit of fit_predict_GP (gpet.py:232-248, 262-266; sklearn_gpr.py:254-295, 475-607) entirely on the device:
// training-set standardisation, the 1 + 12 start points, L-BFGS-B for every (edge, restart) problem, the pick of the
// best restart.  The host only enqueues rounds and reads one counter per round.
//
// The reference calls scipy.optimize.minimize(method="L-BFGS-B") (third party: Byrd, Lu, Nocedal & Zhu 1995, code
// L-BFGS-B 3.0 with the Morales-Nocedal 2011 subspace step and the More-Thuente line search dcsrch/dcstep; scipy's
// defaults m = 10, factr = 1e7, pgtol = 1e-5, maxls = 20).  k_lb_advance restates that published algorithm as a
// reverse-communication state machine, one thread per problem, for the 3 hyper-parameters log(constant, length_scale,
// noise_level): every round is one batched launch of the objective kernel (k_lml / k_lml2) followed by one launch that
// advances every running problem to its next trial point.  With 3 variables the limited-memory matrix is kept as the
// explicit 3 x 3 matrix B that the BFGS updates of the stored (s, y) pairs build from theta I -- the same matrix the
// compact representation W M W^T encodes -- so the generalised Cauchy point and the subspace step are dense 3 x 3
// algebra.  Iterates agree with scipy's to rounding (1e-8 relative where the line search interpolates through huge
// function values); the policy for the result is stated in the tests: final theta within 1e-4 of the reference run
// (noise level compared in linear space), mean / credible interval within 1e-5, edge_trace equal.
#include "gpet_kernels.h"
#include "gpet_options.h"
#include "gpet_lbfgsb_dev.h"

#include <math.h>
#include <stdlib.h>

namespace gpet {

// ---- training set of the converged branch -------------------------------------------------------------------------
// numpy's pairwise summation (np.add.reduce on a contiguous float64 vector), so that mean / std equal the reference's
__device__ double np_pairwise_sum(const double* a, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// one thread per edge: init + observations sorted by x, noise weights, y and x standardised (gpet.py:235-238), y
// standardised again by fit() (sklearn_gpr.py:229-234: normalize_y=False DOES standardise in the reference's copy);
// transforms to fin_par[3..8], theta0 and the 12 restarts (sklearn_gpr.py:283-288, RandomState(seed).uniform) to starts
__global__ void __launch_bounds__(64) k_fin_prepare(EdgeDev* edges, int B, const unsigned int* seeds, double* starts,
                                                    double* scratch, int scratch_stride) {
#pragma clang fp contract(off)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
  EdgeDev& Eg = edges[e];
  const EdgeDev E = Eg;
  const int n_obs = E.sc->n_obs, n = E.n_init + n_obs;
  double* xs = E.fin_x;
  double* ys = E.fin_y;
  double* ws = E.fin_w;
  double* tmp = scratch + (size_t)e * scratch_stride;
  // rank of every point by x (the x are distinct: one observation per bin, end points outside the bins)
  for (int i = 0; i < n; ++i) {
    const long long* pi = i < E.n_init ? E.init_xy + 2 * i : E.obs_xy + 2 * (i - E.n_init);
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const long long* pj = j < E.n_init ? E.init_xy + 2 * j : E.obs_xy + 2 * (j - E.n_init);
      rank += (pj[0] < pi[0]) || (pj[0] == pi[0] && j < i);
    }
    xs[rank] = (double)pi[0];
    ys[rank] = (double)pi[1];
    ws[rank] = i < E.n_init ? (E.fix_endpoints ? 1e-7 : 0.5) : 1.0;
  }
  const double dn = (double)n;
  const double y_m = np_pairwise_sum(ys, n) / dn;
  for (int i = 0; i < n; ++i) {
    const double v = ys[i] - y_m;
    tmp[i] = v * v;
  }
  const double y_s = sqrt(np_pairwise_sum(tmp, n) / dn);
  const double X_m = np_pairwise_sum(xs, n) / dn;
  for (int i = 0; i < n; ++i) {
    const double v = xs[i] - X_m;
    tmp[i] = v * v;
  }
  const double X_s = sqrt(np_pairwise_sum(tmp, n) / dn);
  const double lagmax = xs[n - 1] - xs[0];  // pixel columns: fin_x sits on a lattice of step 1 / X_s (k_lml16 tabulates the correlation per lag)
  for (int i = 0; i < n; ++i) {
    ys[i] = (ys[i] - y_m) / y_s;
    xs[i] = (xs[i] - X_m) / X_s;
  }
  if (n == E.Lg)  // sklearn_gpr.py:673-677: the weights vanish when there are edge_length training rows
    for (int i = 0; i < n; ++i) ws[i] = 0.0;
  const double m2 = np_pairwise_sum(ys, n) / dn;
  for (int i = 0; i < n; ++i) {
    const double v = ys[i] - m2;
    tmp[i] = v * v;
  }
  double s2 = sqrt(np_pairwise_sum(tmp, n) / dn);
  if (s2 == 0.0) s2 = 1.0;
  for (int i = 0; i < n; ++i) ys[i] = (ys[i] - m2) / s2;
  double* par = E.fin_par;
  par[3] = X_m;
  par[4] = X_s;
  par[5] = y_m;
  par[6] = y_s;
  par[7] = m2;
  par[8] = s2;
  par[9] = X_s;
  par[10] = lagmax;
  Eg.fin_n = n;
  // start points: theta of the kernel (gpet.py:244-245), then lo + (hi - lo) * RandomState(seed).uniform(size=(12, 3))
  double lo[3], hi[3];
  lb_bounds(lo, hi);
  double* st = starts + (size_t)e * 13 * 3;
  st[0] = log(5.0);
  st[1] = log(5.0);
  st[2] = log(E.noise_y);
  // MT19937 init_genrand(seed); the first 72 outputs need state words 0..72 and 397..468 of the first twist only
  unsigned int* key = (unsigned int*)tmp;  // (scratch_stride doubles >= 470 words)
  key[0] = seeds[e];
  for (int i = 1; i < 470; ++i) key[i] = 1812433253u * (key[i - 1] ^ (key[i - 1] >> 30)) + (unsigned int)i;
  for (int k = 0; k < 36; ++k) {
    unsigned int wv[2];
    for (int h = 0; h < 2; ++h) {
      const int q = 2 * k + h;
      const unsigned int yv = (key[q] & 0x80000000u) | (key[q + 1] & 0x7FFFFFFFu);
      unsigned int v = key[q + 397] ^ (yv >> 1) ^ ((yv & 1u) ? 0x9908B0DFu : 0u);
      v ^= v >> 11;
      v ^= (v << 7) & 0x9D2C5680u;
      v ^= (v << 15) & 0xEFC60000u;
      v ^= v >> 18;
      wv[h] = v;
    }
    const double uu = ((double)(wv[0] >> 5) * 67108864.0 + (double)(wv[1] >> 6)) / 9007199254740992.0;
    const int c = k % 3;
    st[3 + k] = lo[c] + (hi[c] - lo[c]) * uu;
  }
}

// problem i = (edge i / nstart, start i % nstart): projected start, first evaluation requested
__global__ void __launch_bounds__(256) k_lb_init(LbProb* probs, int P, const double* starts, int* slot_edge,
                                                 double* slot_theta, int* slot_src, LbCfg cfg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  LbProb& p = probs[i];
  const double* lo = cfg.lo;
  const double* hi = cfg.hi;
  for (int c = 0; c < 3; ++c) {
    double v = starts[(size_t)i * 3 + c];
    v = v < lo[c] ? lo[c] : (v > hi[c] ? hi[c] : v);
    p.x[c] = v;
    p.xe[c] = v;
  }
  p.ncorr = 0;
  p.theta = 1.0;
  p.iter = 0;
  p.nfev = 1;
  p.ifun = 0;
  p.task = LB_TASK_FIRST;
  p.why = 0;
  p.edge = i / cfg.nstart;
  p.f = 0.0;
  slot_edge[i] = p.edge;
  slot_src[i] = i;
  for (int c = 0; c < 3; ++c) slot_theta[(size_t)i * 3 + c] = p.xe[c];
}

// slot k of the round just evaluated -> its problem advances; problems still running claim a slot of the next round.
// The 848-byte records of a wave's 64 problems go through LDS: read and written back by all lanes together (two 512-byte
// runs per record) and advanced in place there -- one thread copying its own record field by field made every access of
// the wave 64 cache lines, and the private copy was 1 200 bytes of scratch per thread.
static_assert(sizeof(LbProb) % 8 == 0, "LbProb is copied as doubles");
__global__ void __launch_bounds__(64) k_lb_advance(LbProb* probs, const int* cur_count, const int* slot_src, const double* f,
                                                   const double* g, int* next_count, int* next_edge, double* next_theta,
                                                   int* next_src, LbCfg cfg) {
#pragma clang fp contract(off)
  constexpr int NW = (int)(sizeof(LbProb) / 8);  // doubles per record
  __shared__ LbProb s_p[64];
  const int lane = threadIdx.x;
  const int k = blockIdx.x * blockDim.x + lane;
  // the launch is sized by the host's last KNOWN count (it reads the counter only every few rounds); the true number of
  // running problems is on the device
  const int count = *cur_count;
  if ((int)(blockIdx.x * blockDim.x) >= count) return;  // (the whole wave)
  const bool active = k < count;
  const int pid = active ? slot_src[k] : -1;
#pragma unroll 4
  for (int t = 0; t < 64; ++t) {
    const int pt = __shfl(pid, t, 64);
    if (pt < 0) continue;  // (uniform)
    const double* src = reinterpret_cast<const double*>(probs + pt);
    double* dst = reinterpret_cast<double*>(&s_p[t]);
    dst[lane] = src[lane];
    if (lane + 64 < NW) dst[lane + 64] = src[lane + 64];
  }
  __syncthreads();
  if (active) {
    LbProb& p = s_p[lane];
    double lo[3] = {cfg.lo[0], cfg.lo[1], cfg.lo[2]}, hi[3] = {cfg.hi[0], cfg.hi[1], cfg.hi[2]};
    const double gk[3] = {g[(size_t)k * 3], g[(size_t)k * 3 + 1], g[(size_t)k * 3 + 2]};
    lb_advance(p, f[k], gk, lo, hi);
    if (p.task != LB_TASK_DONE) {
      const int pos = atomicAdd(next_count, 1);
      next_edge[pos] = p.edge;
      next_src[pos] = pid;
      for (int c = 0; c < 3; ++c) next_theta[(size_t)pos * 3 + c] = p.xe[c];
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int t = 0; t < 64; ++t) {
    const int pt = __shfl(pid, t, 64);
    if (pt < 0) continue;
    double* dstg = reinterpret_cast<double*>(probs + pt);
    const double* srcl = reinterpret_cast<const double*>(&s_p[t]);
    dstg[lane] = srcl[lane];
    if (lane + 64 < NW) dstg[lane + 64] = srcl[lane + 64];
  }
}

// best restart of every edge (first minimum, np.argmin in sklearn_gpr.py:292) -> fin_par[0..2] = exp(theta)
__global__ void __launch_bounds__(64) k_lb_pick(EdgeDev* edges, int B, const LbProb* probs, double* theta_out, int nstart) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
  int best = 0;
  double fb = probs[(size_t)e * nstart].f;
  for (int r = 1; r < nstart; ++r) {
    const double fv = probs[(size_t)e * nstart + r].f;
    if (fv < fb) {
      fb = fv;
      best = r;
    }
  }
  const LbProb& p = probs[(size_t)e * nstart + best];
  double* par = edges[e].fin_par;
  for (int c = 0; c < 3; ++c) {
    par[c] = exp(p.x[c]);
    theta_out[(size_t)e * 4 + c] = p.x[c];
  }
  theta_out[(size_t)e * 4 + 3] = fb;
}

// The same with one WAVE per edge: the points in LDS, ranks and the element-wise transforms by all lanes, the sums by
// lane 0 in numpy's order (the same arithmetic on the same operands: bit-identical to the kernel above, which spends
// 2.3 ms per batch on n^2 global loads per thread with 16 waves on the whole GPU -- a tenth of a single edge's trace).
__global__ void __launch_bounds__(64) k_fin_prepare_wave(EdgeDev* edges, int B, const unsigned int* seeds, double* starts,
                                                         int n_cap) {
#pragma clang fp contract(off)
  const int e = blockIdx.x;
  if (e >= B) return;
  EdgeDev& Eg = edges[e];
  const EdgeDev E = Eg;
  extern __shared__ double s_fp[];  // [4][n_cap]: raw x (as integers), sorted x, sorted y, squares
  long long* rx = reinterpret_cast<long long*>(s_fp);
  double* sx = s_fp + n_cap;
  double* sy = sx + n_cap;
  double* tmp = sy + n_cap;
  __shared__ double s_par[6];
  __shared__ unsigned int s_key[470];
  const int lane = threadIdx.x;
  const int n_obs = E.sc->n_obs, n = E.n_init + n_obs;
  for (int j = lane; j < n; j += 64) rx[j] = j < E.n_init ? E.init_xy[2 * j] : E.obs_xy[2 * (j - E.n_init)];
  __syncthreads();
  for (int i = lane; i < n; i += 64) {
    const long long xi = rx[i];
    const long long yi = i < E.n_init ? E.init_xy[2 * i + 1] : E.obs_xy[2 * (i - E.n_init) + 1];
    int rank = 0;
#pragma unroll 8
    for (int j = 0; j < n; ++j) {
      const long long xj = rx[j];
      rank += (xj < xi) || (xj == xi && j < i);
    }
    sx[rank] = (double)xi;
    sy[rank] = (double)yi;
    E.fin_w[rank] = (n == E.Lg) ? 0.0 : (i < E.n_init ? (E.fix_endpoints ? 1e-7 : 0.5) : 1.0);  // sklearn_gpr.py:673-677
  }
  __syncthreads();
  const double dn = (double)n;
  if (lane == 0) s_par[0] = np_pairwise_sum(sy, n) / dn;  // y_m
  __syncthreads();
  const double y_m = s_par[0];
  for (int i = lane; i < n; i += 64) {
    const double v = sy[i] - y_m;
    tmp[i] = v * v;
  }
  __syncthreads();
  if (lane == 0) {
    s_par[1] = sqrt(np_pairwise_sum(tmp, n) / dn);  // y_s
    s_par[2] = np_pairwise_sum(sx, n) / dn;         // X_m
  }
  __syncthreads();
  const double y_s = s_par[1], X_m = s_par[2];
  for (int i = lane; i < n; i += 64) {
    const double v = sx[i] - X_m;
    tmp[i] = v * v;
  }
  __syncthreads();
  if (lane == 0) s_par[3] = sqrt(np_pairwise_sum(tmp, n) / dn);  // X_s
  __syncthreads();
  const double X_s = s_par[3];
  for (int i = lane; i < n; i += 64) {
    sy[i] = (sy[i] - y_m) / y_s;
    E.fin_x[i] = (sx[i] - X_m) / X_s;
  }
  __syncthreads();
  if (lane == 0) s_par[4] = np_pairwise_sum(sy, n) / dn;  // m2
  __syncthreads();
  const double m2 = s_par[4];
  for (int i = lane; i < n; i += 64) {
    const double v = sy[i] - m2;
    tmp[i] = v * v;
  }
  __syncthreads();
  if (lane == 0) {
    double s2 = sqrt(np_pairwise_sum(tmp, n) / dn);
    if (s2 == 0.0) s2 = 1.0;
    s_par[5] = s2;
  }
  __syncthreads();
  const double s2 = s_par[5];
  for (int i = lane; i < n; i += 64) E.fin_y[i] = (sy[i] - m2) / s2;
  if (lane == 0) {
    double* par = E.fin_par;
    par[3] = X_m;
    par[4] = X_s;
    par[5] = y_m;
    par[6] = y_s;
    par[7] = m2;
    par[8] = s2;
    Eg.fin_n = n;
    par[9] = X_s;  // pixel columns: fin_x sits on a lattice of step 1 / X_s (k_lml16 tabulates the correlation per lag)
    par[10] = sx[n - 1] - sx[0];
    // MT19937 init_genrand(seed); the first 72 outputs need state words 0..72 and 397..468 of the first twist only
    s_key[0] = seeds[e];
    for (int i = 1; i < 470; ++i) s_key[i] = 1812433253u * (s_key[i - 1] ^ (s_key[i - 1] >> 30)) + (unsigned int)i;
  }
  __syncthreads();
  // start points: theta of the kernel (gpet.py:244-245), then lo + (hi - lo) * RandomState(seed).uniform(size=(12, 3))
  double lo[3], hi[3];
  lb_bounds(lo, hi);
  double* st = starts + (size_t)e * 13 * 3;
  if (lane == 0) {
    st[0] = log(5.0);
    st[1] = log(5.0);
    st[2] = log(E.noise_y);
  }
  if (lane < 36) {
    const int k = lane;
    unsigned int wv[2];
    for (int h = 0; h < 2; ++h) {
      const int q = 2 * k + h;
      const unsigned int yv = (s_key[q] & 0x80000000u) | (s_key[q + 1] & 0x7FFFFFFFu);
      unsigned int v = s_key[q + 397] ^ (yv >> 1) ^ ((yv & 1u) ? 0x9908B0DFu : 0u);
      v ^= v >> 11;
      v ^= (v << 7) & 0x9D2C5680u;
      v ^= (v << 15) & 0xEFC60000u;
      v ^= v >> 18;
      wv[h] = v;
    }
    const double uu = ((double)(wv[0] >> 5) * 67108864.0 + (double)(wv[1] >> 6)) / 9007199254740992.0;
    const int c = k % 3;
    st[3 + k] = lo[c] + (hi[c] - lo[c]) * uu;
  }
}

size_t lb_prob_bytes() { return sizeof(LbProb); }

hipError_t launch_fin_prepare(hipStream_t st, EdgeDev* d_edges, int B, const unsigned int* d_seeds, double* d_starts,
                              double* d_scratch, int scratch_stride, int n_cap) {
  (void)hipGetLastError();
  const size_t lds = (size_t)4 * n_cap * sizeof(double);
  if (n_cap > 0 && lds <= 60 * 1024 && !option("fin_prepare_serial"))
    hipLaunchKernelGGL(k_fin_prepare_wave, dim3(B), dim3(64), lds, st, d_edges, B, d_seeds, d_starts, n_cap);
  else
    hipLaunchKernelGGL(k_fin_prepare, dim3((B + 63) / 64), dim3(64), 0, st, d_edges, B, d_seeds, d_starts, d_scratch,
                       scratch_stride);
  return hipGetLastError();
}

LbCfg lb_default_cfg() {
  LbCfg c;
  c.lo[0] = log(0.01);  // gpet.py:246-248
  c.hi[0] = log(1e3);
  c.lo[1] = log(0.1);
  c.hi[1] = log(100.0);
  c.lo[2] = log(1e-18);
  c.hi[2] = log(1.0);
  c.nstart = 13;
  return c;
}

hipError_t launch_lb_init(hipStream_t st, void* d_probs, int P, const double* d_starts, int* slot_edge, double* slot_theta,
                          int* slot_src, const LbCfg& cfg) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_lb_init, dim3((P + 255) / 256), dim3(256), 0, st, (LbProb*)d_probs, P, d_starts, slot_edge,
                     slot_theta, slot_src, cfg);
  return hipGetLastError();
}

hipError_t launch_lb_advance(hipStream_t st, void* d_probs, int n_upper, const int* cur_count, const int* slot_src,
                             const double* d_f, const double* d_g, int* next_count, int* next_edge, double* next_theta,
                             int* next_src, const LbCfg& cfg) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_lb_advance, dim3((n_upper + 63) / 64), dim3(64), 0, st, (LbProb*)d_probs, cur_count, slot_src, d_f,
                     d_g, next_count, next_edge, next_theta, next_src, cfg);
  return hipGetLastError();
}

hipError_t launch_lb_pick(hipStream_t st, EdgeDev* d_edges, int B, const void* d_probs, double* d_theta_out, int nstart) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_lb_pick, dim3((B + 63) / 64), dim3(64), 0, st, d_edges, B, (const LbProb*)d_probs, d_theta_out, nstart);
  return hipGetLastError();
}

}  // namespace gpet
