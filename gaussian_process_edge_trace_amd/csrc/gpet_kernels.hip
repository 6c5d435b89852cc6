// Hand-written gfx950 kernels for the GP edge-tracing hot path.
// Wavefront = 64 lanes everywhere.  blockIdx.y selects the edge of the batch.
#include "gpet_kernels.h"
#include "gpet_options.h"
#include "gpet_lbfgsb_dev.h"

#include <atomic>
#include <math.h>
#include <stdlib.h>

namespace gpet {

#define WAVE 64

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

// value of one lane (the same for the whole wave) in every lane: two v_readlane_b32 into scalar registers
__device__ __forceinline__ double wave_bcast(double v, int src_lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, src_lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the whole workgroup; result returned to every thread.  sh: >= 16 doubles.
__device__ __forceinline__ double block_sum(double v, double* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}

// Workgroup id -> (edge, part) such that the parts of one edge share an XCD.  The dispatcher deals consecutive workgroup
// ids round the 8 XCDs (cdna_hip_programming.md T1), so with the plain (part, edge) grid the parts of an edge sit on
// different XCDs and each pulls the edge's shared operand -- or the cache lines it shares with its neighbours -- through
// its own L2.  Bijective for any grid size.  Grid: parts along x (and y), edges along the last used dimension.
__device__ __forceinline__ void xcd_edge_part(int nparts, int& edge, int& part) {
  const int nwg = (int)(gridDim.x * gridDim.y * gridDim.z);
  const int bid = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
#ifdef GPET_NO_XCD_REMAP  // (A/B builds)
  const int wg = bid;
#else
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
#endif
  edge = wg / nparts;
  part = wg - edge * nparts;
}

// General-nu Matern correlation (sklearn kernels.py Matern.__call__, the Bessel-K branch the reference reaches with
// kernel_options = {'kernel': 'Matern', 'nu': ...}, gpet.py:134):
//   rho(r) = 2^(1-nu) / Gamma(nu) (sqrt(2 nu) r)^nu K_nu(sqrt(2 nu) r)  =  1 / Gamma(nu) int exp(nu s - e^s - q e^-s) ds,
// q = nu r^2 / 2  (substitute u = e^s in the Gamma-mixture-of-Gaussians form of x^nu K_nu(x)).  The integrand decays
// doubly exponentially on both sides, so the trapezoid rule converges geometrically: absolute error <= 3e-14 against
// scipy.special.kv for nu in [0.7, 20] with the step below (~60-400 nodes).  Also d rho / d log(length_scale)
// = 2 q / Gamma(nu) int exp((nu - 1) s - e^s - q e^-s) ds -- analytic, where sklearn differentiates numerically.
__device__ double matern_gen(double nu, double inv_gamma, double r, double* dlogl) {
  const double q = 0.5 * nu * r * r;
  double h = 0.45 * rsqrt(nu);
  if (h > 0.2) h = 0.2;
  const double s_hi = log(45.0 + 4.0 * nu) + 0.5;
  double lo_a;
  if (dlogl != nullptr && nu > 1.0) lo_a = -40.0 / fmin(nu, fmax(nu - 1.0, 0.25));
  else lo_a = -40.0 / fmin(nu, 1.0);
  double s_lo = lo_a;
  if (q > 0.0) s_lo = fmax(log(q / 45.0), lo_a);
  const int n0 = (int)floor(s_lo / h), n1 = (int)ceil(s_hi / h);
  double sum = 0.0, dsum = 0.0;
  for (int k = n0; k <= n1; ++k) {
    const double sv = (double)k * h;
    const double es = exp(sv);
    const double w = exp(nu * sv - es - q / es);
    sum += w;
    dsum += w / es;
  }
  if (dlogl != nullptr) *dlogl = 2.0 * q * h * dsum * inv_gamma;
  return h * sum * inv_gamma;
}

// sklearn kernels.py RBF / Matern on pre-scaled 1-D inputs a = x_i / l, b = x_j / l.
__device__ __forceinline__ double corr_fn(const EdgeDev& E, double a, double b) {
  const double d = a - b;
  const double d2 = d * d;
  if (E.kernel_type == GPET_KERNEL_RBF) return exp(-0.5 * d2);
  const double r = sqrt(d2);
  if (E.nu_code == 0) return exp(-r);
  if (E.nu_code == 1) {
    const double k = r * 1.7320508075688772;  // math.sqrt(3)
    return (1.0 + k) * exp(-k);
  }
  if (E.nu_code == 3) return matern_gen(E.nu_gen, E.inv_gamma_nu, r, nullptr);
  const double k = r * 2.23606797749979;  // math.sqrt(5)
  return (1.0 + k + k * k / 3.0) * exp(-k);
}
// The same for two PIXEL coordinates of the loop (integer lags, constructor length scale): general nu reads the lag
// table built at construction instead of integrating again.
__device__ __forceinline__ double corr_px(const EdgeDev& E, double xi, double xj, double length) {
  if (E.kernel_type == GPET_KERNEL_MATERN && E.nu_code == 3 && E.tab_ok) {
    const double lag = fabs(xi - xj);
    return E.rho_tab[(int)(lag + 0.5)];
  }
  return corr_fn(E, xi / length, xj / length);
}

// ---------------------------------------------------------------------------------------
// a1  conv (scipy.ndimage.convolve mode='nearest') + clamp + f32 cast + min/max
//     gpet_utils.py:112-113.  Bit-exact: taps in scipy's order, no FMA contraction.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int f32_order_key(float f) {
  unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_key(unsigned int k) {
  unsigned int u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}

// (a workgroup of 64 x 4 threads owns 64 columns x 16 rows of the output; its (16 + kh - 1) x (64 + kw - 1) patch of the
//  edge-replicated image goes through LDS once -- every thread fetching its kh kw taps from global memory took 1.5 ms per
//  2048 x 2048 image; the products and their order are unchanged)
#define CONV_RY 16
__global__ void k_conv_relu(const double* __restrict__ img, int M, int N, const double* __restrict__ wf,
                            int kh, int kw, int oy, int ox, float* __restrict__ out, unsigned int* minmax) {
#pragma clang fp contract(off)
  extern __shared__ double s_w[];  // [kh * kw] taps, then the patch [CONV_RY + kh - 1][64 + kw - 1]
  const int tid = threadIdx.x + threadIdx.y * blockDim.x, nthr = blockDim.x * blockDim.y;
  for (int i = tid; i < kh * kw; i += nthr) s_w[i] = wf[i];
  const int pw = 64 + kw - 1, ph = CONV_RY + kh - 1;
  double* s_p = s_w + kh * kw;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * CONV_RY;
  for (int e = tid; e < pw * ph; e += nthr) {
    const int py = e / pw, px = e - py * pw;
    int ry = y0 + py - oy, rx = x0 + px - ox;
    ry = ry < 0 ? 0 : (ry > M - 1 ? M - 1 : ry);
    rx = rx < 0 ? 0 : (rx > N - 1 ? N - 1 : rx);
    s_p[e] = img[(size_t)ry * N + rx];
  }
  __syncthreads();
  const int x = x0 + threadIdx.x;
  unsigned int kmin = 0xFFFFFFFFu, kmax = 0u;
  for (int yl = threadIdx.y; yl < CONV_RY; yl += blockDim.y) {
    const int y = y0 + yl;
    if (x < N && y < M) {
      double acc = 0.0;
      for (int a = 0; a < kh; ++a) {
        const double* row = s_p + (yl + a) * pw + threadIdx.x;
        for (int b = 0; b < kw; ++b) {
          const double w = s_w[a * kw + b];
          if (w == 0.0) continue;
          acc = acc + row[b] * w;
        }
      }
      if (acc < 0.0) acc = 0.0;
      const float v = (float)acc;
      out[(size_t)y * N + x] = v;
      const unsigned int key = f32_order_key(v);
      kmin = min(kmin, key);
      kmax = max(kmax, key);
    }
  }
  // workgroup min/max -> one atomic pair per wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (unsigned int)__shfl_xor((int)kmin, o, WAVE));
    kmax = max(kmax, (unsigned int)__shfl_xor((int)kmax, o, WAVE));
  }
  if ((tid & 63) == 0) {
    atomicMin(&minmax[0], kmin);
    atomicMax(&minmax[1], kmax);
  }
}

__global__ void k_minmax_f32(const float* __restrict__ in, size_t count, unsigned int* minmax) {
  unsigned int kmin = 0xFFFFFFFFu, kmax = 0u;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned int k = f32_order_key(in[i]);
    kmin = min(kmin, k);
    kmax = max(kmax, k);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (unsigned int)__shfl_xor((int)kmin, o, WAVE));
    kmax = max(kmax, (unsigned int)__shfl_xor((int)kmax, o, WAVE));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&minmax[0], kmin);
    atomicMax(&minmax[1], kmax);
  }
}

// gpet_utils.py:84-89 in float32:  a = x - min;  a /= max(a);  (x1, +0 are no-ops)
__global__ void k_normalise_f32(const float* __restrict__ in, size_t count, const unsigned int* minmax,
                                float* __restrict__ out) {
  const float mn = f32_from_key(minmax[0]);
  const float mx = f32_from_key(minmax[1]);
  const float span = mx - mn;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const float a = in[i] - mn;
    out[i] = a / span;
  }
}

// ---------------------------------------------------------------------------------------
// a2-a4  training-set assembly + K_obs + Cholesky + alpha   (one workgroup per edge)
//        gpet.py:209-231, sklearn_gpr.py:221-227, 304-320
// ---------------------------------------------------------------------------------------
// K_IN_LDS: the n x n matrix lives in LDS (n_cap <= 128, odd row stride) and only the finished
// factor is written to HBM; otherwise it is factored in place in HBM (large n, config 3).
// FINAL: the converged fit at the optimum found by L-BFGS-B (gpet.py:232-266): training set,
// amplitude, length scale and noise come from fin_x/fin_y/fin_w/fin_par instead of the loop state.
#define FIT_MAXD 136  // 4 * ceil((128 + 1) / 4) + slack
template <bool K_IN_LDS, bool FINAL>
__global__ void __launch_bounds__(576) k_fit(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if (!FINAL && ((sc->done && !sc->force) || sc->status != GPET_OK)) return;
  extern __shared__ double s_dyn[];  // [n_cap] solve vector (+ [n_cap * ld] matrix when K_IN_LDS)
  __shared__ double s_red[16];
  __shared__ double s_diag;
  const int tid = threadIdx.x, bs = blockDim.x;
#ifdef GPET_FIT_PROF
  const long long f0 = clock64();
  long long f1 = f0;
#endif
  const int n_obs = sc->n_obs;
  const int n = FINAL ? E.fin_n : (E.n_init + n_obs);
  const int ld = K_IN_LDS ? (E.n_cap | 1) : E.n_cap;
  double* K = K_IN_LDS ? (s_dyn + E.n_cap) : E.K;
  double amp, length, noise_lvl, jit;
  if (FINAL) {
    for (int i = tid; i < n; i += bs) {
      E.xt[i] = E.fin_x[i];
      E.yt[i] = E.fin_y[i];
      E.wt[i] = E.fin_w[i];
    }
    amp = E.fin_par[0];
    length = E.fin_par[1];
    noise_lvl = E.fin_par[2];
    jit = 1e-6;
    if (tid == 0) {
      sc->amp = amp;
      sc->y_mean = E.fin_par[7];
      sc->y_std = E.fin_par[8];
      sc->n = n;
    }
    __syncthreads();
  } else {

  // 1. gather + stable rank sort by x (np.argsort, gpet.py:212)
  const double w_init = E.fix_endpoints ? 1e-7 : 0.5;  // gpet.py:161
  // (the x values go through LDS -- the solve vector is free until step 5 -- when they fit: the rank count then reads
  //  them as broadcasts instead of paying one global round trip per comparison)
  long long* s_xs = reinterpret_cast<long long*>(s_dyn);
  const bool x_in_lds = n <= E.n_cap;
  if (x_in_lds) {
    for (int j = tid; j < n; j += bs) s_xs[j] = (j < E.n_init) ? E.init_xy[2 * j] : E.obs_xy[2 * (j - E.n_init)];
    __syncthreads();
  }
  for (int i = tid; i < n; i += bs) {
    const long long* pi = (i < E.n_init) ? (E.init_xy + 2 * i) : (E.obs_xy + 2 * (i - E.n_init));
    const long long xi = pi[0], yi = pi[1];
    int r = 0;
    if (x_in_lds) {
#pragma unroll 8
      for (int j = 0; j < n; ++j) {
        const long long xj = s_xs[j];
        r += (xj < xi) || (xj == xi && j < i);
      }
    } else {
      for (int j = 0; j < n; ++j) {
        const long long xj = (j < E.n_init) ? E.init_xy[2 * j] : E.obs_xy[2 * (j - E.n_init)];
        r += (xj < xi) || (xj == xi && j < i);
      }
    }
    E.xt[r] = (double)xi;
    E.yt[r] = (double)yi;
    E.wt[r] = (i < E.n_init) ? w_init : 1.0;
  }
  __syncthreads();

#ifdef GPET_FIT_PROF
  f1 = clock64();
#endif
  // 2. y scaling (gpet.py:228-230) then centring (sklearn_gpr.py:222-227)
  double part = 0.0;
  for (int i = tid; i < n; i += bs) part += E.yt[i];
  const double m1 = block_sum(part, s_red) / n;
  part = 0.0;
  for (int i = tid; i < n; i += bs) {
    const double d = E.yt[i] - m1;
    part += d * d;
  }
  const double y_s = sqrt(block_sum(part, s_red) / n) + 1.0;
  part = 0.0;
  for (int i = tid; i < n; i += bs) {
    const double v = E.yt[i] / y_s;
    E.yt[i] = v;
    part += v;
  }
  const double m2 = block_sum(part, s_red) / n;
  part = 0.0;
  for (int i = tid; i < n; i += bs) {
    const double d = E.yt[i] - m2;
    part += d * d;
  }
  double sd2 = sqrt(block_sum(part, s_red) / n);
  if (sd2 == 0.0) sd2 = 1.0;  // _handle_zeros_in_scale scalar path
  for (int i = tid; i < n; i += bs) E.yt[i] -= m2;
  amp = E.sigma_f * E.sigma_f / (y_s * y_s);
  length = E.length_scale;
  noise_lvl = E.noise_y;
  jit = E.jitter;
  if (tid == 0) {
    sc->y_s = y_s;
    sc->amp = amp;
    sc->y_mean = m2;
    sc->y_std = sd2;
    sc->n = n;
  }
  __syncthreads();
  }  // !FINAL

#ifdef GPET_FIT_PROF
  const long long f2 = clock64();
#endif
  // 3. K = amp * rho + diag(noise_y * w) + jitter     (lower triangle only)
  //    inputs are divided by l exactly as sklearn does (X / length_scale)
  const bool zero_noise = (n == E.Lg);  // sklearn_gpr.py:673-677
  for (int idx = tid; idx < n * n; idx += bs) {
    const int i = idx / n, j = idx - i * n;
    if (j > i) continue;
    double v;
    if (i == j) {
      v = amp;
      v = v + ((zero_noise && !FINAL) ? 0.0 : noise_lvl * E.wt[i]);  // (FINAL: the host already zeroed w)
      v = v + jit;
    } else {
      v = amp * (FINAL ? corr_fn(E, E.xt[i] / length, E.xt[j] / length) : corr_px(E, E.xt[i], E.xt[j], length));
    }
    K[(size_t)i * ld + j] = v;
  }
  __syncthreads();

#ifdef GPET_FIT_PROF
  const long long f3 = clock64();
#endif
  // 4. Cholesky, in place.  K in LDS: right-looking -- per step the pivot column is scaled, then every trailing
  //    entry takes its rank-1 update independently (flat loop over the lower triangle of the trailing block):
  //    two barriers and no serial dot products per step.  K in HBM (n > 128): left-looking (read-mostly).
  bool bad = false;
  bool fwd_done = false;  // the forward solve z = L^-1 y came out of the factorisation (K_IN_LDS)
  if (K_IN_LDS) {
    // Right-looking Cholesky of the BORDERED matrix [K; y^T] in registers: thread t owns the 4 x 4 tile (ti, tj),
    // tj <= ti, of the lower triangle (diagonal tiles keep both halves); row n is y^T and is never a pivot, so when
    // the n pivots are done it holds z^T = (L^-1 y)^T -- the forward solve comes for free.  Per pivot: the owners of
    // column k publish it to LDS (double-buffered), ONE barrier, everyone scales its 4 + 4 column entries by
    // 1 / sqrt(d) and takes its rank-1 update (the structure of k_lml's sweep).  The LDS-resident form it replaces
    // needed three barriers, a column pass and a sqrt-indexed sweep over the trailing triangle per pivot: 2 700 cycles
    // a pivot against ~500 (238 k -> 45 k cycles at n = 89, -DGPET_FIT_PROF).
    __shared__ __attribute__((aligned(16))) double s_col[2][FIT_MAXD];
    const int nbt = (n + 1 + 3) >> 2;
    const int ntile = nbt * (nbt + 1) / 2;
    const bool active = tid < ntile;
    int ti = (int)((sqrt(8.0 * (double)tid + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > tid) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= tid) ++ti;
    const int tj = tid - ti * (ti + 1) / 2;
    if (ntile > bs) {  // (cannot happen: n_cap <= 128 and 576 threads)
      if (tid == 0) sc->status = GPET_ERR_RANK_CAP;
      return;
    }
    double T[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = 4 * ti + a, j = 4 * tj + b;
        double v = 0.0;
        if (active && i <= n && j < n) {
          if (i == n) v = E.yt[j];
          else v = (j <= i) ? K[(size_t)i * ld + j] : K[(size_t)j * ld + i];
        }
        T[a][b] = v;
      }
    bool stop = false;
    for (int kb = 0; kb < nbt && !stop; ++kb) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int k = 4 * kb + kk;
        const int buf = kk & 1;
        if (k >= n) {
          stop = true;
          break;
        }
        if (active && tj == kb) {
#pragma unroll
          for (int a = 0; a < 4; ++a) s_col[buf][4 * ti + a] = T[a][kk];
        }
        __syncthreads();
        const double d = s_col[buf][k];
        if (!(d > 0.0)) {  // not positive definite (the same value in every thread)
          bad = true;
          stop = true;
          break;
        }
        if (active && ti >= kb) {
          // 1 / sqrt(d) from the hardware seed + two Newton steps (full precision), sqrt(d) = d / sqrt(d): a sqrt and
          // a division per pivot would sit on every thread's critical path in front of the next barrier
          double inv = __builtin_amdgcn_rsq(d);
          inv = inv * (1.5 - 0.5 * d * inv * inv);
          inv = inv * (1.5 - 0.5 * d * inv * inv);
          const double dk = d * inv;
          double ci[4], cj[4];
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            ci[a] = s_col[buf][4 * ti + a] * inv;
            cj[a] = s_col[buf][4 * tj + a] * inv;
          }
          if (tj > kb) {  // strictly below and right of the pivot's tile row / column: no masks
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
              for (int b = 0; b < 4; ++b) T[a][b] = fma(-ci[a], cj[b], T[a][b]);
          } else if (tj == kb) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              const int i = 4 * ti + a;
#pragma unroll
              for (int b = 0; b < 4; ++b)
                if (i > k && b > kk) T[a][b] = fma(-ci[a], cj[b], T[a][b]);
              if (i > k) T[a][kk] = ci[a];       // column k of L
              else if (i == k) T[a][kk] = dk;
            }
          }
        }
      }
    }
    if (!bad) {
      // the factor back to LDS (lower triangle) for the backward solve, k_struct_H and the readers; z to the solve vector
      __syncthreads();
      if (active) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int i = 4 * ti + a, j = 4 * tj + b;
            if (i < n && j <= i) K[(size_t)i * ld + j] = T[a][b];
            else if (i == n && j < n) s_dyn[j] = T[a][b];
          }
      }
      __syncthreads();
      fwd_done = true;
    }
  } else {
    for (int j = 0; j < n; ++j) {
      const double* rj = K + (size_t)j * ld;
      for (int i = j + tid; i < n; i += bs) {
        const double* ri = K + (size_t)i * ld;
        double s = ri[j];
        for (int t = 0; t < j; ++t) s -= ri[t] * rj[t];
        if (i == j) s_diag = s; else K[(size_t)i * ld + j] = s;
      }
      __syncthreads();
      const double d = s_diag;
      if (!(d > 0.0)) {
        bad = true;
        break;
      }
      const double dj = sqrt(d);
      for (int i = j + tid; i < n; i += bs) {
        if (i == j) K[(size_t)i * ld + j] = dj; else K[(size_t)i * ld + j] = K[(size_t)i * ld + j] / dj;
      }
      __syncthreads();
    }
  }
  if (bad) {
    if (tid == 0) sc->status = GPET_ERR_NOT_PD;
    return;
  }

#ifdef GPET_FIT_PROF
  const long long f4 = clock64();
#endif
  // 5. alpha = L^-T L^-1 y by one wave in column (axpy) form: the solution vector lives in registers
  //    (rows lane, lane + 64), each step broadcasts one solved component -- no reductions, no barriers
  if (tid < WAVE) {
    const int lane = tid;
    double z0 = (lane < n) ? (fwd_done ? s_dyn[lane] : E.yt[lane]) : 0.0;
    double z1 = (lane + WAVE < n) ? (fwd_done ? s_dyn[lane + WAVE] : E.yt[lane + WAVE]) : 0.0;
    // (n <= 2 * WAVE on this path: n_cap <= 128 whenever K fits LDS; larger n takes the generic loop below)
    if (n <= 2 * WAVE) {
      for (int j = 0; j < n && !fwd_done; ++j) {  // forward: z_j = y_j / l_jj, then y_i -= l_ij z_j for i > j
        const double ljj = K[(size_t)j * ld + j];
        const double zj = __shfl((j < WAVE) ? z0 : z1, j & (WAVE - 1), WAVE) / ljj;
        if (lane == (j & (WAVE - 1))) {
          if (j < WAVE) z0 = zj; else z1 = zj;
        }
        if (lane > j && lane < n) z0 -= K[(size_t)lane * ld + j] * zj;
        if (lane + WAVE > j && lane + WAVE < n) z1 -= K[(size_t)(lane + WAVE) * ld + j] * zj;
      }
      // backward: a_j = z_j / l_jj, then z_i -= l_ji a_j for i < j.  The reciprocals of the diagonal are formed once,
      // before the chain: a step is multiply + shuffle + fused multiply-add instead of carrying a division
      const double ri0 = (lane < n) ? 1.0 / K[(size_t)lane * ld + lane] : 0.0;
      const double ri1 = (lane + WAVE < n) ? 1.0 / K[(size_t)(lane + WAVE) * ld + lane + WAVE] : 0.0;
#pragma unroll 4
      for (int j = n - 1; j >= 0; --j) {
        const double* rj = K + (size_t)j * ld;
        const double aj = wave_bcast((j < WAVE) ? z0 * ri0 : z1 * ri1, j & (WAVE - 1));  // (uniform lane: v_readlane, no LDS permute)
        if (lane == (j & (WAVE - 1))) {
          if (j < WAVE) z0 = aj; else z1 = aj;
        }
        if (lane < j) z0 -= rj[lane] * aj;
        if (lane + WAVE < j) z1 -= rj[lane + WAVE] * aj;
      }
      if (lane < n) E.alpha[lane] = z0;
      if (lane + WAVE < n) E.alpha[lane + WAVE] = z1;
    } else {
      double* z = s_dyn;
      for (int j = 0; j < n; ++j) {
        const double* rj = K + (size_t)j * ld;
        double p = 0.0;
        for (int t = lane; t < j; t += WAVE) p += rj[t] * z[t];
        p = wave_sum(p);
        if (lane == 0) z[j] = (E.yt[j] - p) / rj[j];
      }
      for (int j = n - 1; j >= 0; --j) {
        double p = 0.0;
        for (int t = j + 1 + lane; t < n; t += WAVE) p += K[(size_t)t * ld + j] * z[t];
        p = wave_sum(p);
        if (lane == 0) z[j] = (z[j] - p) / K[(size_t)j * ld + j];
      }
      for (int i = lane; i < n; i += WAVE) E.alpha[i] = z[i];
    }
  }
#ifdef GPET_FIT_PROF
  const long long f5 = clock64();
#endif
  if (K_IN_LDS) {  // publish the factor for k_predict / readers (row stride n_cap in HBM)
    for (int idx = tid; idx < n * n; idx += bs) {
      const int i = idx / n, j = idx - i * n;
      if (j <= i) E.K[(size_t)i * E.n_cap + j] = K[(size_t)i * ld + j];
    }
  }
#ifdef GPET_FIT_PROF
  if (!FINAL && tid == 0 && blockIdx.y == 5)
    printf("k_fit n=%d: sort %lld | scaling %lld | K %lld | cholesky %lld | solves %lld | publish %lld cycles\n", n, f1 - f0, f2 - f1, f3 - f2,
           f4 - f3, f5 - f4, clock64() - f5);
#endif
}

// ---------------------------------------------------------------------------------------
// a2-a4 for MANY training points (n_cap > 128, e.g. config 3's 1500): K and its factor live in HBM and the
// work is spread over the GPU in 64-wide blocks -- head (sort, scaling) -> K tiles -> right-looking blocked
// Cholesky (diagonal block in LDS / row-panel solve / trailing update on the f64 matrix cores) -> blocked solves
// for alpha.  The panel kernels are enqueued for every 64-block of n_cap and return at once past the actual n.
// ---------------------------------------------------------------------------------------
#define CB 64
__global__ void __launch_bounds__(1024) k_fit_head(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  __shared__ double s_red[16];
  extern __shared__ long long s_key[];  // [n_cap] x coordinates (the rank of a point = a pass over all of them: from LDS,
                                        // not n global loads per thread -- 1.0 -> 0.05 ms at 1500 points)
  const int tid = threadIdx.x, bs = blockDim.x;
  const int n = E.n_init + sc->n_obs;
  // 1. gather + stable rank sort by x (np.argsort, gpet.py:212)
  const double w_init = E.fix_endpoints ? 1e-7 : 0.5;  // gpet.py:161
  for (int j = tid; j < n; j += bs) s_key[j] = (j < E.n_init) ? E.init_xy[2 * j] : E.obs_xy[2 * (j - E.n_init)];
  __syncthreads();
  for (int i = tid; i < n; i += bs) {
    const long long xi = s_key[i];
    const long long yi = (i < E.n_init) ? E.init_xy[2 * i + 1] : E.obs_xy[2 * (i - E.n_init) + 1];
    int r = 0;
#pragma unroll 8
    for (int j = 0; j < n; ++j) {
      const long long xj = s_key[j];
      r += (xj < xi) || (xj == xi && j < i);
    }
    E.xt[r] = (double)xi;
    E.yt[r] = (double)yi;
    E.wt[r] = (i < E.n_init) ? w_init : 1.0;
  }
  __syncthreads();
  // 2. y scaling (gpet.py:228-230) then centring (sklearn_gpr.py:222-227) -- same arithmetic as k_fit
  double part = 0.0;
  for (int i = tid; i < n; i += bs) part += E.yt[i];
  const double m1 = block_sum(part, s_red) / n;
  part = 0.0;
  for (int i = tid; i < n; i += bs) {
    const double d = E.yt[i] - m1;
    part += d * d;
  }
  const double y_s = sqrt(block_sum(part, s_red) / n) + 1.0;
  part = 0.0;
  for (int i = tid; i < n; i += bs) {
    const double v = E.yt[i] / y_s;
    E.yt[i] = v;
    part += v;
  }
  const double m2 = block_sum(part, s_red) / n;
  part = 0.0;
  for (int i = tid; i < n; i += bs) {
    const double d = E.yt[i] - m2;
    part += d * d;
  }
  double sd2 = sqrt(block_sum(part, s_red) / n);
  if (sd2 == 0.0) sd2 = 1.0;  // _handle_zeros_in_scale scalar path
  for (int i = tid; i < n; i += bs) E.yt[i] -= m2;
  if (tid == 0) {
    sc->y_s = y_s;
    sc->amp = E.sigma_f * E.sigma_f / (y_s * y_s);
    sc->y_mean = m2;
    sc->y_std = sd2;
    sc->n = n;
  }
}

// K = amp * rho + diag(noise_y * w) + jitter, lower 64x64 tiles (as step 3 of k_fit)
__global__ void __launch_bounds__(256) k_fit_kbuild(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.z];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int n = sc->n, ld = E.n_cap;
  const int i0 = blockIdx.y * CB, j0 = blockIdx.x * CB;
  if (j0 > i0 || i0 >= n) return;
  const double amp = sc->amp, length = E.length_scale;
  const bool zero_noise = (n == E.Lg);  // sklearn_gpr.py:673-677
  for (int e = threadIdx.x; e < CB * CB; e += blockDim.x) {
    const int i = i0 + (e >> 6), j = j0 + (e & 63);
    if (i >= n || j > i) continue;
    double v;
    if (i == j) {
      v = amp;
      v = v + (zero_noise ? 0.0 : E.noise_y * E.wt[i]);
      v = v + E.jitter;
    } else {
      v = amp * corr_px(E, E.xt[i], E.xt[j], length);  // (virtual edges of the blocked objective: tab_ok = 0)
    }
    E.K[(size_t)i * ld + j] = v;
  }
}

// diagonal block k0: Cholesky, written back in place, and -- with_inv -- its INVERSE to E.chol_inv: every triangular
// solve against this block (the rows of L below it, V = L^-1 K_*^T) then becomes a 64x64x64 product on the matrix cores.
// RIGHT-looking, four waves, one LDS barrier per pivot: every wave forms column k (lane i: l_ik = a_ik / sqrt(a_kk),
// through a refined reciprocal square root formed one step ahead) and row k of the inverse (lane c: x_kc = b_kc / l_kk, the
// column sweep of L X = I), then updates its own sixteen columns / rows:
//   a_ij -= l_ik l_jk      (lane i, column j of the wave)          b_jc -= l_jk x_kc      (lane c, row j of the wave)
// History: left-looking with one wave, a dot product of length k per pivot as a chain of LDS round trips: 101 us per block
// (2.4 of the 6.1 ms of config 3's fit); the block in LDS, right-looking on four waves in batches of eight columns: 54 us,
// two thirds of it LDS traffic; the block in registers (below).
// (the body: k_chol_diag is one launch of it; k_chol_syrk runs it on the NEXT diagonal block in the workgroup that has just
//  updated that block -- a launch less per panel in the chain of 24 at n = 1500)
__device__ __forceinline__ void chol_diag_body(const EdgeDev& E, gpet_scalars* sc, int k0, int with_inv) {
  const int n = sc->n, ld = E.n_cap;
  if (k0 >= n) return;
  const int nb = (n - k0) < CB ? (n - k0) : CB;
  // The block lives in REGISTERS: lane i = row i, wave w = columns 16 w .. 16 w + 15 of it (a[16]), and rows 16 w .. of the
  // inverse for column `lane` (x[16]).  Per pivot only column k, row k of the inverse and the next pivot's entry go through
  // LDS (published by their owners at the end of the previous step, two buffers by parity: a slow wave may still read the
  // old one), and the 16 l_jk a wave needs are broadcast reads of that column -- a fifth of the LDS traffic of the form that
  // kept the whole block in LDS (54 us per block, 2/3 of it LDS).
  __shared__ double s_col[2][CB];  // column k: a_ik of the partly factorised block
  __shared__ double s_xr[2][CB];   // row k of the inverse's right-hand side
  __shared__ double s_dn[2];       // a_{k+1,k+1} before the update of step k
  const int tid = threadIdx.x, i = tid & 63, w = tid >> 6;
  double a[16], x[16];
  {
    const double* __restrict__ row = E.K + (size_t)(k0 + i) * ld + k0 + 16 * w;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = 16 * w + u;
      a[u] = (i < nb && j <= i) ? row[u] : 0.0;
      x[u] = (j == i) ? 1.0 : 0.0;
    }
  }
  if (w == 0) {
    s_col[0][i] = a[0];
    s_xr[0][i] = x[0];
    if (i == 1) s_dn[0] = a[1];
  }
  __syncthreads();
  // (global_store, not flat_store: a FLAT instruction also counts in the LDS wait counter the barrier below waits for)
  __attribute__((address_space(1))) double* inv = (__attribute__((address_space(1))) double*)(E.chol_inv + (size_t)(k0 / CB) * CB * CB);
  __attribute__((address_space(1))) double* kcol = (__attribute__((address_space(1))) double*)(E.K + (size_t)(k0 + i) * ld + k0);
  // the pivot and its reciprocal square root are formed one step AHEAD (a_k+1,k+1 - l_k+1,k^2 is all the update does to the
  // next pivot), so the chain rsq + three Newton steps runs under the update
  double d = s_col[0][0];
  double r = __builtin_amdgcn_rsq(d);
#pragma unroll
  for (int it = 0; it < 3; ++it) r = fma(0.5 * r, fma(-d * r, r, 1.0), r);
#ifdef GPET_CD_PROF
  long long cp[5] = {0, 0, 0, 0, 0};
  long long ct = clock64();
#define CD_T(i) { const long long t_ = clock64(); cp[i] += t_ - ct; ct = t_; }
#else
#define CD_T(i)
#endif
  for (int kp = 0; kp < 4; ++kp) {
#pragma unroll
    for (int ku = 0; ku < 16; ++ku) {
      const int k = 16 * kp + ku;
      if (k >= nb) break;  // (uniform)
      if (!(d > 0.0)) {    // (the same value in every thread: the exit is uniform)
        if (tid == 0) sc->status = GPET_ERR_NOT_PD;
        return;
      }
      const int pb = k & 1;
      const double aik = s_col[pb][i], bk = s_xr[pb][i];
      const int kn = (k + 1 < CB) ? k + 1 : k;
      const double lnk = s_col[pb][kn] * r;
      const double dn = fma(-lnk, lnk, s_dn[pb]);
      double rn = __builtin_amdgcn_rsq(dn);
      const double dk = d * r;
      const double lik = (i > k) ? aik * r : (i == k ? dk : 0.0);
      const double xk = bk * r;
      CD_T(0)
      if (w == (k & 3)) {
        if (i >= k && i < nb) kcol[k] = lik;
        if (with_inv) inv[k * CB + i] = xk;
      }
      CD_T(1)
      // this wave's columns j > k (all sixteen behind the pivot's wave, the later ones in it, none before it)
      // (two straight-line forms, all the column reads of a form first: with the test inside the loop every column paid its
      //  own branch and its own LDS round trip, 1 060 of a pivot's 1 990 cycles on the last wave)
      if (w > kp) {
        double lj[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) lj[u] = s_col[pb][16 * w + u];  // (one address for the wave: a broadcast)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const double ljk = lj[u] * r;
          a[u] = fma(-lik, ljk, a[u]);
          x[u] = fma(-ljk, xk, x[u]);
        }
      } else if (w == kp) {
        double lj[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) lj[u] = (u > ku) ? s_col[pb][16 * kp + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (u > ku) {  // (a compile-time test: ku and u are unrolled)
            const double ljk = lj[u] * r;
            a[u] = fma(-lik, ljk, a[u]);
            x[u] = fma(-ljk, xk, x[u]);
          }
        }
      }
      CD_T(2)
#pragma unroll
      for (int it = 0; it < 3; ++it) rn = fma(0.5 * rn, fma(-dn * rn, rn, 1.0), rn);
      d = dn;
      r = rn;
      // what the next step reads: column k + 1, row k + 1 of the inverse, the entry of the pivot after it
      if (k + 1 < CB && w == ((k + 1) >> 4)) {
        s_col[pb ^ 1][i] = a[(ku + 1) & 15];
        s_xr[pb ^ 1][i] = x[(ku + 1) & 15];
      }
      if (k + 2 < CB && w == ((k + 2) >> 4) && i == k + 2) s_dn[pb ^ 1] = a[(ku + 2) & 15];
      // (LDS only: __syncthreads() would also wait for the column and the row just stored to HBM)
      CD_T(3)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      CD_T(4)
    }
  }
#ifdef GPET_CD_PROF
  if ((tid & 63) == 0 && blockIdx.y == 0 && k0 == 0)
    printf("chol_diag wave %d: per pivot: reads + column %lld | stores %lld | update %lld | newton + publish %lld | barrier %lld cycles\n", w,
           cp[0] / nb, cp[1] / nb, cp[2] / nb, cp[3] / nb, cp[4] / nb);
#endif
#undef CD_T
  if (with_inv)  // identity beyond a short last block
    for (int rr = nb + w; rr < CB; rr += 4) inv[rr * CB + i] = (rr == i) ? 1.0 : 0.0;
}
__global__ void __launch_bounds__(256) k_chol_diag(EdgeDev* edges, int k0, int with_inv) {
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  chol_diag_body(E, sc, k0, with_inv);
}

// row blocks below the diagonal block by SUBSTITUTION (one row per lane of the first wave): the blocked objective of
// the converged fits keeps it -- the optimiser visits nearly singular matrices (noise 1e-10 of the amplitude) where a
// product with the explicit inverse of a diagonal block costs digits (2e-8 instead of 1e-9 relative on f = 1.2e8)
__global__ void __launch_bounds__(256) k_chol_trsm_sub(EdgeDev* edges, int k0) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int n = sc->n, ld = E.n_cap;
  const int i0 = k0 + CB * ((int)blockIdx.x + 1);
  if (i0 >= n) return;
  const int nb = CB;  // (a block below exists only under a full diagonal block)
  const int rows = (n - i0) < CB ? (n - i0) : CB;
  __shared__ double sL[CB][CB + 1];
  __shared__ double sA[CB][CB + 1];
  const int tid = threadIdx.x, bs = blockDim.x;
  for (int e = tid; e < CB * CB; e += bs) {
    const int i = e >> 6, j = e & 63;
    sL[i][j] = (j <= i) ? E.K[(size_t)(k0 + i) * ld + k0 + j] : 0.0;
    sA[i][j] = (i < rows) ? E.K[(size_t)(i0 + i) * ld + k0 + j] : 0.0;
  }
  __syncthreads();
  if (tid < rows) {
    for (int j = 0; j < nb; ++j) {
      double acc = sA[tid][j];
      for (int t = 0; t < j; ++t) acc -= sA[tid][t] * sL[j][t];
      sA[tid][j] = acc / sL[j][j];
    }
  }
  __syncthreads();
  for (int e = tid; e < rows * CB; e += bs) {
    const int i = e >> 6, j = e & 63;
    E.K[(size_t)(i0 + i) * ld + k0 + j] = sA[i][j];
  }
}

// row blocks below the diagonal block: X L_kk^T = A_ik, i.e. X = A_ik (L_kk^-1)^T on v_mfma_f64_16x16x4_f64
// (A lane l <- A[16 w + (l & 15)][kk + (l >> 4)], B lane l <- Linv[16 t + (l & 15)][kk + (l >> 4)])
typedef double v4f64c __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_chol_trsm(EdgeDev* edges, int k0) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int n = sc->n, ld = E.n_cap;
  const int i0 = k0 + CB * ((int)blockIdx.x + 1);
  if (i0 >= n) return;
  const int rows = (n - i0) < CB ? (n - i0) : CB;
  __shared__ double sL[CB][CB + 1];
  __shared__ double sA[CB][CB + 1];
  const int tid = threadIdx.x, bs = blockDim.x;
  const double* inv = E.chol_inv + (size_t)(k0 / CB) * CB * CB;
  for (int e = tid; e < CB * CB; e += bs) {
    const int i = e >> 6, j = e & 63;
    sL[i][j] = inv[e];
    sA[i][j] = (i < rows) ? E.K[(size_t)(i0 + i) * ld + k0 + j] : 0.0;
  }
  __syncthreads();
  const int lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  v4f64c acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4f64c){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < CB; kk += 4) {
    const double a = sA[16 * w + li][kk + lq];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sL[16 * t + li][kk + lq], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = 16 * w + lq + 4 * g, j = 16 * t + li;
      if (i < rows) E.K[(size_t)(i0 + i) * ld + k0 + j] = acc[t][g];
    }
}

// trailing update A_ij -= X_i X_j^T for the 64x64 tiles (bj <= bi) behind panel k0, v_mfma_f64_16x16x4_f64
// next_diag: the workgroup of the first tile then factors it (the diagonal block of the next panel, with its inverse)
__global__ void __launch_bounds__(256) k_chol_syrk(EdgeDev* edges, int k0, int next_diag) {
  const EdgeDev E = edges[blockIdx.z];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int n = sc->n, ld = E.n_cap;
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int i0 = k0 + CB * (bi + 1), j0 = k0 + CB * (bj + 1);
  if (i0 >= n) return;
  __shared__ double sI[CB][CB + 1];
  __shared__ double sJ[CB][CB + 1];
  const int tid = threadIdx.x;
  for (int e = tid; e < CB * CB; e += 256) {
    const int i = e >> 6, j = e & 63;
    sI[i][j] = (i0 + i < n) ? E.K[(size_t)(i0 + i) * ld + k0 + j] : 0.0;
    sJ[i][j] = (j0 + i < n) ? E.K[(size_t)(j0 + i) * ld + k0 + j] : 0.0;
  }
  __syncthreads();
  const int lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  v4f64c acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4f64c){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < CB; kk += 4) {
    const double a = sI[16 * w + li][kk + lq];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sJ[16 * t + li][kk + lq], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = i0 + 16 * w + lq + 4 * g, j = j0 + 16 * t + li;
      if (i < n && j <= i) E.K[(size_t)i * ld + j] -= acc[t][g];
    }
  if (next_diag && bi == 0 && bj == 0) {
    __syncthreads();  // (the tile is in memory: hipcc's barrier waits for vmcnt(0))
    chol_diag_body(E, sc, k0 + CB, 1);
  }
}

// alpha = L^-T L^-1 y on the factor in HBM, blocked by 64: diagonal blocks through LDS (column form, one wave).  The
// updates of the rest of the vector read L the way it lies in memory: forward -- the 64 columns of the block, one ROW per
// 64 consecutive lanes' ... per thread group of 16 (a 512-byte run per row); backward -- the 64 ROWS of the block, thread i
// owns vector entry i and walks down the rows (every load a coalesced run).  The vector stays in LDS.
// (One 256-thread workgroup reading the transposed block column by column took 3.2 ms at 1500 points: 5.6 GB/s.)
// the value lane `src` (a compile-time constant at the call sites) holds, in every lane
__device__ __forceinline__ double lane_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__global__ void __launch_bounds__(1024) k_chol_solve(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  extern __shared__ double s_z[];  // [n_cap]
  __shared__ double sL[CB][CB + 1];
  const int n = sc->n, ld = E.n_cap;
  const int tid = threadIdx.x, bs = blockDim.x;
  for (int i = tid; i < n; i += bs) s_z[i] = E.yt[i];
  __syncthreads();
  for (int k0 = 0; k0 < n; k0 += CB) {  // forward
    const int nb = (n - k0) < CB ? (n - k0) : CB;
    for (int e = tid; e < nb * nb; e += bs) {
      const int i = e / nb, j = e - i * nb;
      if (j <= i) sL[i][j] = E.K[(size_t)(k0 + i) * ld + k0 + j];
    }
    __syncthreads();
    if (tid < WAVE) {
      // (the 64 dependent steps of a block: lane j's value through v_readlane -- the loop is unrolled, j a constant -- and times
      //  the reciprocal of the diagonal formed by all lanes at once: a division and a ds_bpermute per step were 10 us per block)
      double z = (tid < nb) ? s_z[k0 + tid] : 0.0;
      const double rd = (tid < nb) ? 1.0 / sL[tid][tid] : 0.0;
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if (j < nb) {
          const double zj = lane_f64(z, j) * lane_f64(rd, j);
          if (tid == j) z = zj;
          if (tid > j && tid < nb) z -= sL[tid][j] * zj;
        }
      }
      if (tid < nb) s_z[k0 + tid] = z;
    }
    __syncthreads();
    // rows below: 16 lanes per row (four columns each, a 512-byte run), summed in the order t = 0..nb-1 per lane and
    // then across the 16 lanes
    for (int i = k0 + nb + (tid >> 4); i < n; i += bs >> 4) {
      const double* ri = E.K + (size_t)i * ld + k0;
      const int t0 = tid & 15;
      // (four loads in flight per lane; entries beyond nb of the last block are not read)
      const double v0 = ri[t0], v1 = t0 + 16 < nb ? ri[t0 + 16] : 0.0, v2 = t0 + 32 < nb ? ri[t0 + 32] : 0.0,
                   v3 = t0 + 48 < nb ? ri[t0 + 48] : 0.0;
      double acc = (t0 < nb ? v0 : 0.0) * s_z[k0 + t0];
      acc += v1 * s_z[k0 + ((t0 + 16) & 63)];
      acc += v2 * s_z[k0 + ((t0 + 32) & 63)];
      acc += v3 * s_z[k0 + ((t0 + 48) & 63)];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 16);
      if ((tid & 15) == 0) s_z[i] -= acc;
    }
    __syncthreads();
  }
  for (int k0 = ((n - 1) / CB) * CB; k0 >= 0; k0 -= CB) {  // backward (L^T)
    const int nb = (n - k0) < CB ? (n - k0) : CB;
    for (int e = tid; e < nb * nb; e += bs) {
      const int i = e / nb, j = e - i * nb;
      if (j <= i) sL[i][j] = E.K[(size_t)(k0 + i) * ld + k0 + j];
    }
    __syncthreads();
    if (tid < WAVE) {
      double z = (tid < nb) ? s_z[k0 + tid] : 0.0;
      const double rd = (tid < nb) ? 1.0 / sL[tid][tid] : 0.0;
#pragma unroll
      for (int j = CB - 1; j >= 0; --j) {
        if (j < nb) {
          const double aj = lane_f64(z, j) * lane_f64(rd, j);
          if (tid == j) z = aj;
          if (tid < j) z -= sL[j][tid] * aj;
        }
      }
      if (tid < nb) s_z[k0 + tid] = z;
    }
    __syncthreads();
    for (int i = tid; i < k0; i += bs) {
      const double* ci = E.K + (size_t)k0 * ld + i;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int t = 0;
      for (; t + 8 <= nb; t += 8) {  // (eight rows in flight, four partial sums)
        const double u0 = ci[(size_t)t * ld], u1 = ci[(size_t)(t + 1) * ld], u2 = ci[(size_t)(t + 2) * ld], u3 = ci[(size_t)(t + 3) * ld];
        const double u4 = ci[(size_t)(t + 4) * ld], u5 = ci[(size_t)(t + 5) * ld], u6 = ci[(size_t)(t + 6) * ld], u7 = ci[(size_t)(t + 7) * ld];
        a0 = fma(u0, s_z[k0 + t], a0);
        a1 = fma(u1, s_z[k0 + t + 1], a1);
        a2 = fma(u2, s_z[k0 + t + 2], a2);
        a3 = fma(u3, s_z[k0 + t + 3], a3);
        a0 = fma(u4, s_z[k0 + t + 4], a0);
        a1 = fma(u5, s_z[k0 + t + 5], a1);
        a2 = fma(u6, s_z[k0 + t + 6], a2);
        a3 = fma(u7, s_z[k0 + t + 7], a3);
      }
      for (; t < nb; ++t) a0 = fma(ci[(size_t)t * ld], s_z[k0 + t], a0);
      s_z[i] -= (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += bs) E.alpha[i] = s_z[i];
}

// alpha = L^-T L^-1 y for many training points on one workgroup PER 64-ROW BLOCK (k_chol_solve above is one workgroup per
// edge: its 2 x 9 MB of L at n = 1500 come through ONE CU's path to L2, 0.66 ms).  Forward (BACK = false): the workgroup
// of block i subtracts L_ij z_j for j < i as the z_j are published (a flag per block in global memory, its value the
// launch number: nothing to reset), runs the 64-step substitution of its diagonal block and publishes z_i.  Backward
// (BACK = true, a second launch): workgroup x owns block nt - 1 - x and subtracts L_ji^T alpha_j for j > i.  In both a
// workgroup waits only for workgroups with a SMALLER blockIdx.x of the same edge, which the dispatcher started before
// it: no residency requirement.  z / alpha cross XCDs: agent-scope accesses (as k_oj_persist, gpet_eig.hip); the waiting
// thread gives up after 4 s of wall time and fails the edge.  The chain per block: flag + 64 values + one 64 x 64 product + the
// substitution, ~2-3 us.
template <bool BACK>
__global__ void __launch_bounds__(256) k_chol_solve_mw(EdgeDev* edges, int epoch) {
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int n = sc->n, ld = E.n_cap;
  const int nt = (n + CB - 1) / CB;
  if ((int)blockIdx.x >= nt) return;
  const int ib = BACK ? nt - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int k0 = ib * CB;
  const int nb = (n - k0) < CB ? (n - k0) : CB;
  __shared__ double sL[CB][CB + 1];
  __shared__ double s_y[CB], s_v[CB];
  __shared__ double s_part[4][CB];
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  int* flag_mine = E.solve_flag + (BACK ? (ld / CB + 1) : 0);
  const double* src = BACK ? E.solve_z : E.yt;
  double* dst = BACK ? E.alpha : E.solve_z;
  if (tid < CB) s_y[tid] = 0.0;
  for (int e = tid; e < nb * nb; e += 256) {
    const int i = e / nb, j = e - i * nb;
    if (j <= i) sL[i][j] = E.K[(size_t)(k0 + i) * ld + k0 + j];
  }
  // the off-diagonal blocks in the order their vectors become available; the next one is in registers before the wait
  //   forward:  block (ib, j), thread = (row r = tid >> 2, columns 16 (tid & 3) ..): rows of 512 contiguous bytes
  //   backward: block (j, ib)^T, thread = (column c = tid & 63, rows 16 (tid >> 6) ..): 512 contiguous bytes per row
  const int nsteps = BACK ? nt - 1 - ib : ib;
  double lb[16];
  auto fetch = [&](int step) {
    const int jb = BACK ? nt - 1 - step : step;
    const int j0 = jb * CB;
    const int njb = (n - j0) < CB ? (n - j0) : CB;
    if (!BACK) {
      const int r = tid >> 2, c0 = 16 * (tid & 3);
      const double* row = E.K + (size_t)(k0 + (r < nb ? r : 0)) * ld + j0 + c0;
#pragma unroll
      for (int u = 0; u < 16; ++u) lb[u] = (r < nb) ? row[u] : 0.0;  // (j < ib: a full block of columns)
    } else {
      const int c = tid & 63, r0 = 16 * (tid >> 6);
#pragma unroll
      for (int u = 0; u < 16; ++u) lb[u] = (r0 + u < njb && c < nb) ? E.K[(size_t)(j0 + r0 + u) * ld + k0 + c] : 0.0;
    }
  };
  if (nsteps > 0) fetch(0);
  bool ok = true;
  for (int step = 0; step < nsteps; ++step) {
    const int jb = BACK ? nt - 1 - step : step;
    const int j0 = jb * CB;
    const int njb = (n - j0) < CB ? (n - j0) : CB;
    if (tid == 0) {
      int good = 1;
      const unsigned long long t0 = wall_clock64();  // (constant 100 MHz)
      // acquire: the vector the flag announces is read after it (pairs with the release store below)
      while (__hip_atomic_load(flag_mine + jb, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 400000000ull) {  // 4 s of wall time: a fault elsewhere, not contention (no residency needed)
          good = 0;
          break;
        }
      }
      s_ok = good;
    }
    __syncthreads();
    if (!s_ok) {
      ok = false;
      break;
    }
    if (tid < CB) s_v[tid] = (tid < njb) ? __hip_atomic_load(dst + j0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    __syncthreads();
    double acc = 0.0;
    if (!BACK) {
      const int c0 = 16 * (tid & 3);
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = fma(lb[u], s_v[c0 + u], acc);
    } else {
      const int r0 = 16 * (tid >> 6);
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = fma(lb[u], s_v[r0 + u], acc);
    }
    if (step + 1 < nsteps) fetch(step + 1);
    if (!BACK) {
      acc += __shfl_xor(acc, 1, 4);
      acc += __shfl_xor(acc, 2, 4);
      if ((tid & 3) == 0) s_y[tid >> 2] += acc;
    } else {
      s_part[tid >> 6][tid & 63] = acc;
    }
    __syncthreads();
    if (BACK && tid < CB) s_y[tid] += (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]);
    // (s_v is rewritten only after the next wait's barrier; s_y is touched by the same threads every step)
  }
  if (!ok) {
    if (tid == 0) sc->status = GPET_ERR_STATE;
    return;
  }
  __syncthreads();
  if (tid < WAVE) {
    // the 64 dependent steps of the diagonal block (k_chol_solve's: lane j's value through v_readlane, reciprocal diagonal)
    double z = (tid < nb) ? src[k0 + tid] - s_y[tid] : 0.0;
    const double rd = (tid < nb) ? 1.0 / sL[tid][tid] : 0.0;
    if (!BACK) {
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if (j < nb) {
          const double zj = lane_f64(z, j) * lane_f64(rd, j);
          if (tid == j) z = zj;
          if (tid > j && tid < nb) z -= sL[tid][j] * zj;
        }
      }
    } else {
#pragma unroll
      for (int j = CB - 1; j >= 0; --j) {
        if (j < nb) {
          const double aj = lane_f64(z, j) * lane_f64(rd, j);
          if (tid == j) z = aj;
          if (tid < j) z -= sL[j][tid] * aj;
        }
      }
    }
    if (tid < nb) __hip_atomic_store(dst + k0 + tid, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();  // (every lane's stores are issued and complete: hipcc's barrier waits for vmcnt(0))
  // release: the flag becomes visible at agent scope only after the vector it announces
  if (tid == 0) __hip_atomic_store(flag_mine + ib, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// gpet_set_option "solve_mw" (default 1): 0 = alpha of the blocked fit by the single-workgroup
// kernel; "diag_in_syrk" (default 1): 0 = a launch of its own for every diagonal block
int& gpet_opt_solve_mw() {
  static int& v = option("solve_mw");
  return v;
}
int& gpet_opt_diag_in_syrk() {
  static int& v = option("diag_in_syrk");
  return v;
}

// the whole fit for n_cap > 128
static void launch_fit_blocked(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd) {
  const int nt = cdiv(bd.n_cap, CB);
  hipLaunchKernelGGL(k_fit_head, dim3(1, B), dim3(1024), (size_t)bd.n_cap * sizeof(long long), st, d_edges);
  hipLaunchKernelGGL(k_fit_kbuild, dim3(nt, nt, B), dim3(256), 0, st, d_edges);
  // per panel: [diagonal block] -> rows below it -> trailing update, whose first workgroup goes on to factor the next
  // diagonal block (option "diag_in_syrk" = 0: a launch of its own per diagonal block)
  const int diag_in_syrk = gpet_opt_diag_in_syrk() ? 1 : 0;
  for (int k0 = 0; k0 < bd.n_cap; k0 += CB) {
    if (k0 == 0 || !diag_in_syrk) hipLaunchKernelGGL(k_chol_diag, dim3(1, B), dim3(256), 0, st, d_edges, k0, 1);
    const int below = cdiv(bd.n_cap - k0 - CB, CB);
    if (below > 0) {
      hipLaunchKernelGGL(k_chol_trsm, dim3(below, B), dim3(256), 0, st, d_edges, k0);
      hipLaunchKernelGGL(k_chol_syrk, dim3(below, below, B), dim3(256), 0, st, d_edges, k0, diag_in_syrk);
    }
  }
  // alpha: one workgroup per 64-row block and direction (option "solve_mw" = 0: the single-workgroup kernel)
  if (gpet_opt_solve_mw()) {
    static std::atomic<int> launch_no{0};
    const int epoch = ++launch_no;  // (flags hold the number of the launch that published them: never reset)
    hipLaunchKernelGGL(k_chol_solve_mw<false>, dim3(nt, B), dim3(256), 0, st, d_edges, epoch);
    hipLaunchKernelGGL(k_chol_solve_mw<true>, dim3(nt, B), dim3(256), 0, st, d_edges, epoch);
  } else {
    hipLaunchKernelGGL(k_chol_solve, dim3(1, B), dim3(1024), (size_t)bd.n_cap * sizeof(double), st, d_edges);
  }
}

// ---------------------------------------------------------------------------------------
// a5  predict: K_* on the fly, mean, V = L^-1 K_*^T, std          (thread per grid point)
//     sklearn_gpr.py:381-394, 414-436
//     One wave per 64 grid points.  V_LDS: the wave keeps its 64 columns of V in LDS
//     ([n][64], conflict-free) and streams the rows of L through LDS, so the forward
//     substitution has no global-load latency chain; the HBM copy of V is write-only here.
// ---------------------------------------------------------------------------------------
// FINAL: prediction of the converged fit on the standardised grid (gpet.py:264-266); results go
// to the batch's contiguous output block as mean in pixels and std in standardised units.
template <bool V_LDS, bool FINAL>
__global__ void __launch_bounds__(64) k_predict(EdgeDev* edges, int out_stride) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if (!FINAL && ((sc->done && !sc->force) || sc->status != GPET_OK)) return;
  extern __shared__ double s_dyn[];
  const int lane = threadIdx.x;
  const int j = blockIdx.x * 64 + lane;
  const int n = sc->n, ld = E.n_cap, Lg = E.Lg;
  const bool live = j < Lg;
  const double amp = sc->amp;
  const double length = FINAL ? E.fin_par[1] : E.length_scale;
  double xq = (double)(E.x_st + (live ? j : 0));
  if (FINAL) xq = (xq - E.fin_par[3]) / E.fin_par[4];  // (x_grid - X_m) / X_s
  xq = xq / length;
  double msum = 0.0, vsum = 0.0;
  if (V_LDS) {
    double* s_v = s_dyn;                      // [n_cap][64]
    double* s_l = s_dyn + (size_t)E.n_cap * 64;  // [n_cap] current row of L
    double* s_x = s_l + E.n_cap;              // [n_cap] x_i / l
    double* s_a = s_x + E.n_cap;              // [n_cap] alpha
    for (int i = lane; i < n; i += 64) {
      s_x[i] = E.xt[i] / length;
      s_a[i] = E.alpha[i];
    }
    for (int i = 0; i < n; ++i) {
      __syncthreads();
      const double* ri = E.K + (size_t)i * ld;
      for (int t = lane; t <= i; t += 64) s_l[t] = ri[t];
      __syncthreads();
      const double ki = amp * ((!FINAL && E.nu_code == 3) ? corr_px(E, (double)(E.x_st + (live ? j : 0)), E.xt[i], length) : corr_fn(E, xq, s_x[i]));
      double acc = ki;
      for (int t = 0; t < i; ++t) acc -= s_l[t] * s_v[t * 64 + lane];
      const double v = acc / s_l[i];
      s_v[i * 64 + lane] = v;
      if (live) E.V[(size_t)i * Lg + j] = v;
      msum += ki * s_a[i];
      vsum += v * v;
    }
  } else {
    if (!live) return;
    for (int i = 0; i < n; ++i) {
      const double ki = amp * ((!FINAL && E.nu_code == 3) ? corr_px(E, (double)(E.x_st + j), E.xt[i], length) : corr_fn(E, xq, E.xt[i] / length));
      const double* ri = E.K + (size_t)i * ld;
      double acc = ki;
      for (int t = 0; t < i; ++t) acc -= ri[t] * E.V[(size_t)t * Lg + j];
      const double v = acc / ri[i];
      E.V[(size_t)i * Lg + j] = v;
      msum += ki * E.alpha[i];
      vsum += v * v;
    }
  }
  if (!live) return;
  const double mean = sc->y_std * msum + sc->y_mean;
  double var = amp - vsum;
  if (var < 0.0) var = 0.0;
  const double sd = sqrt(var * (sc->y_std * sc->y_std));
  if (FINAL) {
    E.fin_out[j] = E.fin_par[6] * mean + E.fin_par[5];  // y_s * y_mean + y_m   (gpet.py:266)
    E.fin_out[out_stride + j] = sd;                      // not rescaled (gpet.py:266, quirk Q10)
  } else {
    E.mean[j] = mean;
    E.std[j] = sd;
  }
}

// ---------------------------------------------------------------------------------------
// a6  factor of the posterior covariance (what numpy's SVD-based multivariate_normal builds)
//     step 1: pivoted Cholesky  cov ~= G^T G  (G rows = columns of the factor), rank-revealing
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_pchol(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  extern __shared__ double s_d[];  // [Lg] remaining diagonal
  __shared__ double s_val[16];
  __shared__ int s_idx[16];
  __shared__ double s_piv;
  __shared__ int s_pidx;
  const int tid = threadIdx.x, bs = blockDim.x, Lg = E.Lg;
  const int lane = tid & 63, w = tid >> 6, nw = bs >> 6;
  for (int i = tid; i < Lg; i += bs) s_d[i] = E.cov[(size_t)i * Lg + i];
  __syncthreads();
  double tol = 0.0;
  int k = 0;
  for (; k < E.r_cap; ++k) {
    // argmax of the remaining diagonal (ties -> smallest index)
    double bv = -1.0;
    int bi = 0x7FFFFFFF;
    for (int i = tid; i < Lg; i += bs) {
      const double v = s_d[i];
      if (v > bv) {
        bv = v;
        bi = i;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(bv, o, WAVE);
      const int oi = __shfl_xor(bi, o, WAVE);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      s_val[w] = bv;
      s_idx[w] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      double v = s_val[0];
      int ix = s_idx[0];
      for (int q = 1; q < nw; ++q)
        if (s_val[q] > v || (s_val[q] == v && s_idx[q] < ix)) {
          v = s_val[q];
          ix = s_idx[q];
        }
      s_piv = v;
      s_pidx = ix;
    }
    __syncthreads();
    const double dp = s_piv;
    const int p = s_pidx;
    if (k == 0) tol = dp * 1e-14;
    if (!(dp > tol) || !(dp > 0.0)) break;
    const double sq = sqrt(dp);
    if (tid == 0) E.perm[k] = p;
    double* s_gp = s_d + Lg;  // [r_cap] the pivot's entries of the previous columns
    for (int t = tid; t < k; t += bs) s_gp[t] = E.G[(size_t)t * Lg + p];
    __syncthreads();
    const double* __restrict__ crow = E.cov + (size_t)p * Lg;  // cov is exactly symmetric: row p == column p
    for (int i = tid; i < Lg; i += bs) {
      double g = 0.0;
      if (s_d[i] >= 0.0) {  // not yet pivoted
        g = crow[i];
        for (int t = 0; t < k; ++t) g -= E.G[(size_t)t * Lg + i] * s_gp[t];
        g = g / sq;
      }
      E.G[(size_t)k * Lg + i] = g;
    }
    __syncthreads();  // all reads of s_d / G row p done before the diagonal is downdated
    for (int i = tid; i < Lg; i += bs) {
      if (i == p) {
        s_d[i] = -1.0;  // mark as used
      } else if (s_d[i] >= 0.0) {
        const double g = E.G[(size_t)k * Lg + i];
        double nd = s_d[i] - g * g;
        s_d[i] = nd > 0.0 ? nd : 0.0;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    sc->rank = k;
    if (k == E.r_cap) {
      // capacity reached: check what is left
      double rem = 0.0;
      for (int i = 0; i < Lg; ++i) rem = s_d[i] > rem ? s_d[i] : rem;
      if (rem > tol * 1e4) sc->status = GPET_ERR_RANK_CAP;
    }
  }
}

// step 1, register-resident variant (Lg <= 512, r_cap <= 96): thread i owns row i of the factor and
// keeps its <= 96 entries in registers (zero until computed), so a step is 96 FMAs against the
// pivot row broadcast from LDS -- no global-load chain.  Same arithmetic order as k_pchol.
#define PCH_R 96
__global__ void __launch_bounds__(512) k_pchol_reg(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  __shared__ double s_d[512];
  __shared__ double s_gp[PCH_R];
  __shared__ double s_val[8];
  __shared__ int s_idx[8];
  __shared__ double s_piv;
  __shared__ int s_pidx;
  const int tid = threadIdx.x, Lg = E.Lg;
  const int lane = tid & 63, w = tid >> 6;
  const bool live = tid < Lg;
  double gown[PCH_R];
#pragma unroll
  for (int t = 0; t < PCH_R; ++t) gown[t] = 0.0;
  double dloc = live ? E.cov[(size_t)tid * Lg + tid] : -1.0;  // remaining diagonal (-1: pivoted / absent)
  s_d[tid] = dloc;
  __syncthreads();
  double tol = 0.0;
  int k = 0;
  const int rcap = E.r_cap < PCH_R ? E.r_cap : PCH_R;
  for (; k < rcap; ++k) {
    double bv = dloc;
    int bi = live ? tid : 0x7FFFFFFF;
    if (!(bv >= 0.0)) {
      bv = -1.0;
      bi = 0x7FFFFFFF;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(bv, o, WAVE);
      const int oi = __shfl_xor(bi, o, WAVE);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      s_val[w] = bv;
      s_idx[w] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      double v = s_val[0];
      int ix = s_idx[0];
      for (int q = 1; q < 8; ++q)
        if (s_val[q] > v || (s_val[q] == v && s_idx[q] < ix)) {
          v = s_val[q];
          ix = s_idx[q];
        }
      s_piv = v;
      s_pidx = ix;
    }
    __syncthreads();
    const double dp = s_piv;
    const int p = s_pidx;
    if (k == 0) tol = dp * 1e-14;
    if (!(dp > tol) || !(dp > 0.0)) break;
    const double sq = sqrt(dp);
    if (tid == p) {
      E.perm[k] = p;
#pragma unroll
      for (int t = 0; t < PCH_R; ++t) s_gp[t] = gown[t];
    }
    __syncthreads();
    double g = 0.0;
    if (live && dloc >= 0.0) {  // not yet pivoted
      g = E.cov[(size_t)p * Lg + tid];  // cov is exactly symmetric: row p == column p
#pragma unroll
      for (int t = 0; t < PCH_R; ++t) g -= gown[t] * s_gp[t];
      g = g / sq;
    }
    if (live) E.G[(size_t)k * Lg + tid] = g;
#pragma unroll
    for (int t = 0; t < PCH_R; ++t) gown[t] = (t == k) ? g : gown[t];
    if (tid == p) {
      dloc = -1.0;
    } else if (dloc >= 0.0) {
      const double nd = dloc - g * g;
      dloc = nd > 0.0 ? nd : 0.0;
    }
    // (s_gp / s_val are rewritten only after the next step's barriers)
  }
  s_d[tid] = dloc;
  __syncthreads();
  if (tid == 0) {
    sc->rank = k;
    if (k == rcap) {
      double rem = 0.0;
      for (int i = 0; i < Lg; ++i) rem = s_d[i] > rem ? s_d[i] : rem;
      if (rem > tol * 1e4) sc->status = GPET_ERR_RANK_CAP;
    }
  }
}

// step 2: Gram matrix C = G G^T (rank x rank)
__global__ void __launch_bounds__(256) k_gram(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.z];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  const int r = sc->rank, Lg = E.Lg;
  const int a0 = blockIdx.y * 16, b0 = blockIdx.x * 16;
  if (a0 >= r || b0 >= r || b0 > a0) return;
  __shared__ double sa[16][65];
  __shared__ double sb[16][65];
  const int ta = threadIdx.x >> 4, tb = threadIdx.x & 15;
  double acc = 0.0;
  for (int i0 = 0; i0 < Lg; i0 += 64) {
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      const int rr = e >> 6, ii = e & 63;
      const int i = i0 + ii;
      sa[rr][ii] = (a0 + rr < r && i < Lg) ? E.G[(size_t)(a0 + rr) * Lg + i] : 0.0;
      sb[rr][ii] = (b0 + rr < r && i < Lg) ? E.G[(size_t)(b0 + rr) * Lg + i] : 0.0;
    }
    __syncthreads();
#pragma unroll 16
    for (int ii = 0; ii < 64; ++ii) acc += sa[ta][ii] * sb[tb][ii];
    __syncthreads();
  }
  const int a = a0 + ta, b = b0 + tb;
  if (a < r && b < r) {
    E.C[(size_t)a * E.r_cap + b] = acc;
    E.C[(size_t)b * E.r_cap + a] = acc;
  }
}

// step 3 (fast path, r_cap <= 96): the same cyclic Jacobi with the Gram matrix and the
// accumulated rotations resident in LDS (m * (m|1) + m * (m+2) doubles <= 146 KB of the CU's 160 KB).
// A round applies its m/2 disjoint rotations as independent 2x2 blocks: block (a, b) holds the
// four entries touched by the row rotation of pair a and the column rotation of pair b, so one
// thread updates it in place (column rotation, then row rotation -- the arithmetic of the
// sequential algorithm) and the whole round needs two barriers: blocks | barrier | accumulated rotations on 15
// waves while the 16th computes the rotation parameters of the next round | barrier.  Odd ranks are padded with a
// decoupled zero row/column.  Row stride ld is odd: conflict-free row and column walks.
__device__ __forceinline__ void rr_pair(int m1, int round, int k, int& p, int& q) {  // m1 = m - 1
  if (k == 0) {
    p = round;
    q = m1;
    return;
  }
  int x = round + k, y = round - k;
  x = x >= m1 ? x - m1 : x;
  y = y < 0 ? y + m1 : y;
  p = x < y ? x : y;
  q = x < y ? y : x;
}

__global__ void __launch_bounds__(1024) k_jacobi_lds(EdgeDev* edges, int scaled_out) {
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  __shared__ double s_red[16];
  __shared__ __attribute__((aligned(16))) double2 s_cs[2][64];  // (c, s) of a round, double-buffered: one 16-byte read per pair
  __shared__ int s_pos[96];
  const int r = sc->rank, ldg = E.r_cap;
  const int m = (r + 1) & ~1;
  const int ld = m | 1;
  // The accumulated rotations are kept TRANSPOSED with an even stride: a column of the eigenvector matrix is a
  // contiguous, 16-byte aligned run, so a thread rotates two rows per 16-byte LDS access (the LDS pipe -- ~8 cycles
  // per wave access whatever its width -- is what bounds this kernel).
  const int ldw = m + 2;
  double* A = s_mem;
  double* W = s_mem + (size_t)m * ld;  // m * ld is even: 16-byte aligned
  const int tid = threadIdx.x, bs = blockDim.x;
  for (int e = tid; e < m * m; e += bs) {
    const int i = e / m, j = e - i * m;
    A[i * ld + j] = (i < r && j < r) ? E.C[(size_t)i * ldg + j] : 0.0;
    W[j * ldw + i] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int half = m >> 1, m1 = m - 1;
  // fixed roles for the whole factorisation.  A: 2x2 blocks (pair a rows) x (pair b columns) of the UPPER
  // triangle of pairs, a <= b (the mirrored block is written, not recomputed): half (half + 1) / 2 blocks,
  // one per thread for m <= 88.  W: (pair b, segment of row PAIRS) items on the first 15 waves; the last wave
  // computes the rotation parameters of the NEXT round meanwhile.
  const int nblk = half * (half + 1) / 2;
  int ba[2], bb[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + u * bs;
    int b_ = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while (b_ * (b_ + 1) / 2 > e) --b_;
    while ((b_ + 1) * (b_ + 2) / 2 <= e) ++b_;
    bb[u] = b_;
    ba[u] = e - b_ * (b_ + 1) / 2;
    if (e >= nblk) ba[u] = -1;
  }
  const int wthreads = bs - 64;  // W workers; threads wthreads .. bs-1 (one wave) own the rotation parameters
  const int nseg = half > 0 ? min(half, wthreads / half) : 1;
  const int wb = tid / nseg, wseg = tid - wb * nseg;
  const bool w_on = half > 0 && tid < wthreads && wb < half;
  const int pk = tid - wthreads;  // pair index of a parameter thread
  // rotation parameters of pair pk for `round`, from the current A, into s_cs[round & 1]
  auto params = [&](int round) {
    if (pk >= 0 && pk < half) {
      int p, q;
      rr_pair(m1, round, pk, p, q);
      double c = 1.0, s = 0.0;
      const double apq = A[p * ld + q];
      const double app = A[p * ld + p], aqq = A[q * ld + q];
      if (fabs(apq) > 1e-300 && apq * apq > 1e-36 * fabs(app * aqq)) {
        // t = sgn(d) h / (|d| + sqrt(d^2 + h^2)): hardware rsqrt / reciprocal + one Newton step (an inexact
        // angle only leaves a ~1e-10 relative residue in a_pq); c = rsqrt(1 + t^2) gets two steps and
        // s = t c, so c^2 + s^2 = 1 to rounding whatever t is
        const double d = aqq - app, hh = 2.0 * apq;
        const double rho2 = d * d + hh * hh;
        double y = __builtin_amdgcn_rsq(rho2);
        y = y * (1.5 - 0.5 * rho2 * y * y);
        const double den = fabs(d) + rho2 * y;
        double iv = __builtin_amdgcn_rcp(den);
        iv = iv * (2.0 - den * iv);
        const double t = (d >= 0.0 ? hh : -hh) * iv;
        const double u = 1.0 + t * t;
        c = __builtin_amdgcn_rsq(u);
        c = c * (1.5 - 0.5 * u * c * c);
        c = c * (1.5 - 0.5 * u * c * c);
        s = t * c;
      }
      s_cs[round & 1][pk] = make_double2(c, s);
    }
  };
  int sweeps = 0;
  if (r >= 2) {
    for (int sweep = 0; sweep < 40; ++sweep) {
      double off = 0.0, dg = 0.0;
      for (int e = tid; e < r * r; e += bs) {
        const int i = e / r, j = e - i * r;
        const double v = A[i * ld + j];
        if (i == j) dg += v * v; else off += v * v;
      }
      off = block_sum(off, s_red);
      dg = block_sum(dg, s_red);
      // quadratic convergence: off^2 <= 1e-24 diag^2 now means <= 1e-48 after one more sweep
      if (off <= 1e-24 * dg || off == 0.0) break;
      ++sweeps;
      params(0);
      __syncthreads();
      for (int round = 0; round < m1; ++round) {
        const double2* cs = s_cs[round & 1];
        // phase 1: the 2x2 blocks of A
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int a_ = ba[u], b_ = bb[u];
          if (a_ < 0) continue;
          const double2 ra = cs[a_], rb = cs[b_];
          const double ca = ra.x, sa = ra.y, cb = rb.x, sb = rb.y;
          if (sa == 0.0 && sb == 0.0) continue;
          int pa, qa, pb, qb;
          rr_pair(m1, round, a_, pa, qa);
          rr_pair(m1, round, b_, pb, qb);
          const double b00 = A[pa * ld + pb], b01 = A[pa * ld + qb];
          const double b10 = A[qa * ld + pb], b11 = A[qa * ld + qb];
          const double t00 = cb * b00 - sb * b01, t01 = sb * b00 + cb * b01;
          const double t10 = cb * b10 - sb * b11, t11 = sb * b10 + cb * b11;
          const double n00 = ca * t00 - sa * t10, n10 = sa * t00 + ca * t10;
          const double n01 = ca * t01 - sa * t11, n11 = sa * t01 + ca * t11;
          A[pa * ld + pb] = n00;
          A[qa * ld + pb] = n10;
          A[pa * ld + qb] = n01;
          A[qa * ld + qb] = n11;
          if (a_ != b_) {
            A[pb * ld + pa] = n00;
            A[pb * ld + qa] = n10;
            A[qb * ld + pa] = n01;
            A[qb * ld + qa] = n11;
          }
        }
        __syncthreads();
        // phase 2: W (column rotations only, two rows per access) on the first 15 waves, while the last wave
        // reads the updated A for the parameters of the next round
        if (w_on) {
          const double2 rb = cs[wb];
          const double cb = rb.x, sb = rb.y;
          if (sb != 0.0) {
            int pb, qb;
            rr_pair(m1, round, wb, pb, qb);
            double2* wp_ = reinterpret_cast<double2*>(W + pb * ldw);
            double2* wq_ = reinterpret_cast<double2*>(W + qb * ldw);
            for (int ip = wseg; ip < half; ip += nseg) {
              const double2 wp = wp_[ip], wq = wq_[ip];
              wp_[ip] = make_double2(cb * wp.x - sb * wq.x, cb * wp.y - sb * wq.y);
              wq_[ip] = make_double2(sb * wp.x + cb * wq.x, sb * wp.y + cb * wq.y);
            }
          }
        }
        if (round + 1 < m1) params(round + 1);
        __syncthreads();
      }
    }
  }
  for (int k = tid; k < r; k += bs) E.theta[k] = A[k * ld + k];
  for (int e = tid; e < r * r; e += bs) {
    const int i = e / r, j = e - i * r;
    E.W[(size_t)i * ldg + j] = W[j * ldw + i];
  }
  __syncthreads();
  for (int k = tid; k < r; k += bs) {
    const double v = A[k * ld + k];
    int pos = 0;
    for (int j = 0; j < r; ++j) {
      const double u = A[j * ld + j];
      pos += (u > v) || (u == v && j < k);
    }
    E.order[pos] = k;
    s_pos[k] = pos;
  }
  __syncthreads();
  // structured path (scaled_out): G (unused there) <- eigenvectors in descending eigenvalue order, scaled by
  // y_std sqrt(theta): column pos of row t is the coefficient of basis vector t in factor row pos -- what
  // k_struct_rows multiplies with Q0.  (C stays intact: repeated launches see the same matrix.)
  if (scaled_out) {
    for (int e = tid; e < r * r; e += bs) {
      const int i = e / r, j = e - i * r;
      const double th = A[j * ld + j];
      E.G[(size_t)i * ldg + s_pos[j]] = W[j * ldw + i] * (sc->y_std * sqrt(th > 0.0 ? th : 0.0));
    }
  }
  if (tid == 0) sc->lml = (double)sweeps;  // diagnostics: Jacobi sweeps of this factorisation
}

// step 3, "seated" form of the same cyclic Jacobi (same pairs in the same order: the circle method of rr_pair).  The
// kernel above addresses the matrix by ROW INDEX, so every round recomputes who meets whom and where their entries
// live, and it writes every off-diagonal block twice.  Here the matrix is addressed by SEAT: the players of pair k always
// sit in slots 2k and 2k+1, and after every round each player of the circle moves one seat back (the pivot of pair 0
// stays; after m - 1 rounds everybody is back where they started).  The 2x2 block (pair a, pair b), a <= b, is then a
// FIXED record, and each of its four updated entries goes to a fixed place of the next round's layout, computed once
// per thread before the first sweep.  Only the upper triangle of blocks exists (the matrix is symmetric): no mirrored
// writes, no per-round index arithmetic, and the matrix takes half the space -- 21 KB + 41 KB of W at m = 72, so that TWO
// workgroups of 512 threads share a CU and one's barriers and rotation parameters hide behind the other's LDS work.
//   layout: four planes [r][c] of nblk doubles, block (a <= b) at index b (b + 1) / 2 + a: consecutive threads read and
//           (mostly) write consecutive doubles -- no bank conflicts; diagonal blocks keep [0][0], [0][1], [1][1].
//   W:      W2[ip][player] as 16-byte pairs (components 2 ip, 2 ip + 1 of the player's eigenvector): the lanes of a wave
//           are consecutive pairs, whose players are consecutive -- contiguous, conflict-free 16-byte accesses.
//   round:  read own block, rotate | barrier | write the four entries to next round's places | barrier |
//           last wave: rotation parameters of the next round; the other seven: W of this round | barrier.
// What bounds it: LDS STORES (VGPR -> LDS transfer, ~80 B/clk per CU: 13 cycles per ds_write_b128, 6 per ds_write_b64),
// 62 KB of them per round, two thirds for W.  Tried and measured slower: two copies of the
// triangle with 2 barriers (85 KB: one workgroup per CU); 768 threads; W in registers moved along the wave with DPP
// shifts, inside the rounds or as a separate pass over logged rotations (tools/ubench/wpass.hip: VALU-bound at the
// same ~0.8 ms per 1 024 factorisations the LDS form of W costs).
__device__ __forceinline__ int seat_player(int slot, int round, int m1) {  // who sits in `slot` in round `round` (< m1)
  const int k = slot >> 1;
  if (slot & 1) {
    if (k == 0) return m1;
    const int v = m1 - k + round;
    return v >= m1 ? v - m1 : v;
  }
  const int v = k + round;
  return v >= m1 ? v - m1 : v;
}
__device__ __forceinline__ int seat_next_slot(int slot, int m1, int half) {  // where the player of `slot` sits next round
  const int k = slot >> 1;
  if (slot == 1) return 1;  // the pivot
  int seat = (slot & 1) ? m1 - k : k;
  seat = seat == 0 ? m1 - 1 : seat - 1;
  if (seat == 0) return 0;
  return seat < half ? 2 * seat : 2 * (m1 - seat) + 1;
}
// offset (in doubles) of entry (slot i, slot j) of the symmetric matrix in the four-plane triangle
__device__ __forceinline__ int seat_offset(int i, int j, int nblk) {
  int a = i >> 1, b = j >> 1, ri = i & 1, cj = j & 1;
  if (a > b || (a == b && ri > cj)) {
    int t = a; a = b; b = t;
    t = ri; ri = cj; cj = t;
  }
  return (2 * ri + cj) * nblk + b * (b + 1) / 2 + a;
}
__device__ __forceinline__ void seat_block_of(int e, int& a_, int& b_) {  // e = b (b + 1) / 2 + a, a <= b
  b_ = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
  while (b_ * (b_ + 1) / 2 > e) --b_;
  while ((b_ + 1) * (b_ + 2) / 2 <= e) ++b_;
  a_ = e - b_ * (b_ + 1) / 2;
}

// the four entries of the 2 x 2 block (pair a, pair b) after this round's rotations (c_a, s_a) and (c_b, s_b); ONE body
// for every kernel that forms them (contraction off)
__device__ __forceinline__ void jac_rot_block(double b00, double b01, double b10, double b11, double ca, double sa, double cb,
                                              double sb, double (&nv)[4]) {
#pragma clang fp contract(off)
  const double t00 = fma(cb, b00, -(sb * b01)), t01 = fma(sb, b00, cb * b01);
  const double t10 = fma(cb, b10, -(sb * b11)), t11 = fma(sb, b10, cb * b11);
  nv[0] = fma(ca, t00, -(sa * t10));
  nv[2] = fma(sa, t00, ca * t10);
  nv[1] = fma(ca, t01, -(sa * t11));
  nv[3] = fma(sa, t01, ca * t11);
}
// rotation (c, s) that annihilates a_pq of [[app, apq], [apq, aqq]]
__device__ __forceinline__ void jac_params(double app, double apq, double aqq, double& c, double& s) {
  c = 1.0;
  s = 0.0;
  if (fabs(apq) > 1e-300 && apq * apq > 1e-36 * fabs(app * aqq)) {
    // t = sgn(d) h / (|d| + sqrt(d^2 + h^2)): hardware rsqrt / reciprocal + one Newton step (an inexact
    // angle only leaves a ~1e-10 relative residue in a_pq); c = rsqrt(1 + t^2) gets two steps and
    // s = t c, so c^2 + s^2 = 1 to rounding whatever t is
    const double d = aqq - app, hh = 2.0 * apq;
    const double rho2 = d * d + hh * hh;
    double y = __builtin_amdgcn_rsq(rho2);
    y = y * (1.5 - 0.5 * rho2 * y * y);
    const double den = fabs(d) + rho2 * y;
    double iv = __builtin_amdgcn_rcp(den);
    iv = iv * (2.0 - den * iv);
    const double t = (d >= 0.0 ? hh : -hh) * iv;
    const double u = 1.0 + t * t;
    c = __builtin_amdgcn_rsq(u);
    c = c * (1.5 - 0.5 * u * c * c);
    c = c * (1.5 - 0.5 * u * c * c);
    s = t * c;
  }
}

#define JS_NT 512
#define JS_LOG_SWEEPS 40  // sweeps a rotation log holds (= the sweep limit of the kernel)
// LOGW (small batches, where the chain of rounds IS the time): the eigenvectors are not accumulated here -- 41 KB of LDS
// stores a round, 1 200-1 600 of its 2 650 cycles -- but the round's rotations (c, s) go to a log in global memory, and
// k_jacobi_wpass applies them to the rows of W afterwards, one wave per row in registers, on as many CUs as there are rows.
template <int NU, bool LOGW>  // NU blocks per thread: half (half + 1) / 2 <= 1024 up to m = 88, 1176 at m = 96
__global__ void __launch_bounds__(JS_NT) k_jacobi_seat(EdgeDev* edges, int scaled_out) {
  constexpr int NT = JS_NT;
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  __shared__ double s_red[16];
  __shared__ __attribute__((aligned(16))) double2 s_cs[2][48];
  __shared__ int s_pos[96];
  __shared__ double s_theta[96];
  const int r = sc->rank, ldg = E.r_cap;
  const int m = (r + 1) & ~1;
  const int half = m >> 1, m1 = m - 1;
  const int nblk = half * (half + 1) / 2;
  double* A0 = s_mem;                                                       // [4][nblk]
  double2* W2 = reinterpret_cast<double2*>(s_mem + 4 * ((nblk + 1) & ~1));  // [half][m]
  const int tid = threadIdx.x;
  if (nblk > NU * NT) {  // (the launcher picks NU from the batch's capacity)
    if (tid == 0) sc->status = GPET_ERR_RANK_CAP;
    return;
  }
  for (int e = tid; e < nblk; e += NT) {
    int a_, b_;
    seat_block_of(e, a_, b_);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = seat_player(2 * a_ + (q >> 1), 0, m1), j = seat_player(2 * b_ + (q & 1), 0, m1);
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      A0[q * nblk + e] = (hi < r) ? E.C[(size_t)lo * ldg + hi] : 0.0;
    }
  }
  if (!LOGW)
    for (int e = tid; e < half * m; e += NT) {
      const int ip = e / m, pl = e - ip * m;
      W2[e] = make_double2(pl == 2 * ip ? 1.0 : 0.0, pl == 2 * ip + 1 ? 1.0 : 0.0);
    }
  double2* jlog = reinterpret_cast<double2*>(E.jlog);
  // fixed roles: blocks tid, tid + NT, ... with the places of their four entries in the next round's layout
  int ba[NU], bb[NU], dst[NU][4];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int e = tid + u * NT;
    seat_block_of(e, ba[u], bb[u]);
    if (e >= nblk) ba[u] = -1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i2 = seat_next_slot(2 * ba[u] + (q >> 1), m1, half), j2 = seat_next_slot(2 * bb[u] + (q & 1), m1, half);
      dst[u][q] = (ba[u] < 0 || (ba[u] == bb[u] && q == 2)) ? -1 : seat_offset(i2, j2, nblk);
    }
  }
  // W workers: the first seven waves, in groups of `half` consecutive threads (thread = pair); group g takes the
  // component pairs g, g + ngrp, ...   The last wave owns the rotation parameters.
  constexpr int WT = NT - 64;
  const int ngrp = half > 0 ? WT / half : 1;
  const int wg = half > 0 ? tid / half : 0, wb = tid - wg * half;
  const bool w_on = tid < WT && wg < ngrp;
  const int pk = tid - WT;
  const int dk = pk * (pk + 1) / 2 + pk;  // diagonal block of pair pk
#ifdef GPET_JAC_TRACE
  double trace_rel = 0.0;
#endif
  auto params = [&](int nxt, int log_round) {
    if (pk >= 0 && pk < half) {
      double c = 1.0, s = 0.0;
      const double app = A0[dk], apq = A0[nblk + dk], aqq = A0[3 * nblk + dk];
#ifdef GPET_JAC_TRACE
      {
        const double den = fabs(app * aqq);
        const double rel2 = den > 0.0 ? apq * apq / den : 0.0;
        trace_rel = rel2 > trace_rel ? rel2 : trace_rel;
      }
#endif
      jac_params(app, apq, aqq, c, s);
      s_cs[nxt][pk] = make_double2(c, s);
      if (LOGW) jlog[(size_t)log_round * half + pk] = make_double2(c, s);
    }
  };
  __syncthreads();
  int sweeps = 0;
#ifdef GPET_JAC_PROF  // cycles per phase of a round, as seen by the first wave and by the parameter wave
  long long pq[7] = {0, 0, 0, 0, 0, 0, 0};
#define JAC_CLK(v) const long long v = clock64()
#else
#define JAC_CLK(v)
#endif
  if (r >= 2) {
    for (int sweep = 0; sweep < 40; ++sweep) {
      double off = 0.0, dg = 0.0;
      for (int e = tid; e < nblk; e += NT) {
        int a_, b_;
        seat_block_of(e, a_, b_);
        const double v00 = A0[e], v01 = A0[nblk + e], v10 = A0[2 * nblk + e], v11 = A0[3 * nblk + e];
        if (a_ == b_) {
          dg += v00 * v00 + v11 * v11;
          off += 2.0 * v01 * v01;
        } else {
          off += 2.0 * ((v00 * v00 + v01 * v01) + (v10 * v10 + v11 * v11));
        }
      }
      off = block_sum(off, s_red);
      dg = block_sum(dg, s_red);
#ifdef GPET_JAC_TRACE
      {
        double tr = trace_rel;
        for (int o = 32; o > 0; o >>= 1) {
          const double ov = __shfl_xor(tr, o, 64);
          tr = ov > tr ? ov : tr;
        }
        if (tid == NT - 64 && (blockIdx.y == 0 || blockIdx.y == 517))
          printf("jac trace blk %d sweep %d: off2/diag2 = %.3e   largest rel2 rotated in the previous sweep = %.3e\n", (int)blockIdx.y, sweep, off / dg, tr);
        trace_rel = 0.0;
      }
#endif
      // quadratic convergence: off^2 <= 1e-24 diag^2 now means <= 1e-48 after one more sweep
      if (off <= 1e-24 * dg || off == 0.0) break;
      ++sweeps;
      params(0, (sweeps - 1) * m1);
      __syncthreads();
      int pb = wb, qb = wb == 0 ? m1 : m1 - wb;  // players of this thread's W pair in round 0
      for (int round = 0; round < m1; ++round) {
        JAC_CLK(q0);
        const double2* cs = s_cs[round & 1];
        double nv[NU][4];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (ba[u] < 0) continue;
          const int e = tid + u * NT;
          const double2 ra = cs[ba[u]], rb = cs[bb[u]];
          const double ca = ra.x, sa = ra.y, cb = rb.x, sb = rb.y;
          const double b00 = A0[e], b01 = A0[nblk + e], b11 = A0[3 * nblk + e];
          const double b10 = (ba[u] == bb[u]) ? b01 : A0[2 * nblk + e];
          jac_rot_block(b00, b01, b10, b11, ca, sa, cb, sb, nv[u]);
        }
        JAC_CLK(q1);
        __syncthreads();  // every block has been read
        JAC_CLK(q2);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (ba[u] >= 0 && dst[u][q] >= 0) A0[dst[u][q]] = nv[u][q];
        JAC_CLK(q3);
        __syncthreads();
        JAC_CLK(q4);
        if (!LOGW && w_on) {  // W: rotate the eigenvector entries of players pb, qb (needs only this round's cs)
          const double2 rb = cs[wb];
          const double cb = rb.x, sb = rb.y;
          if (sb != 0.0) {
            // three component pairs per pass, all loads first: the LDS latency is paid once per pass
            for (int ip0 = wg; ip0 < half; ip0 += 3 * ngrp) {
              double2 wp[3], wq[3];
#pragma unroll
              for (int t = 0; t < 3; ++t) {
                const int ip = ip0 + t * ngrp;
                if (ip < half) {
                  wp[t] = W2[ip * m + pb];
                  wq[t] = W2[ip * m + qb];
                }
              }
#pragma unroll
              for (int t = 0; t < 3; ++t) {
                const int ip = ip0 + t * ngrp;
                if (ip < half) {
                  // (spelled out: k_jacobi_wpass does exactly this arithmetic)
                  W2[ip * m + pb] = make_double2(fma(cb, wp[t].x, -(sb * wq[t].x)), fma(cb, wp[t].y, -(sb * wq[t].y)));
                  W2[ip * m + qb] = make_double2(fma(sb, wp[t].x, cb * wq[t].x), fma(sb, wp[t].y, cb * wq[t].y));
                }
              }
            }
          }
        }
        if (round + 1 < m1) params((round + 1) & 1, (sweeps - 1) * m1 + round + 1);
        JAC_CLK(q5);
        __syncthreads();
#ifdef GPET_JAC_PROF
        const long long q6 = clock64();
        pq[0] += q1 - q0; pq[1] += q2 - q1; pq[2] += q3 - q2; pq[3] += q4 - q3; pq[4] += q5 - q4; pq[5] += q6 - q5; pq[6] += q6 - q0;
#endif
        // next round: everybody on the circle is one player further
        pb = pb + 1 >= m1 ? 0 : pb + 1;
        if (wb != 0) qb = qb + 1 >= m1 ? 0 : qb + 1;
      }
    }
  }
  // back in the initial seats: slot -> player of round 0
  for (int k = tid; k < half; k += NT) {
    const int d = k * (k + 1) / 2 + k;
    s_theta[seat_player(2 * k, 0, m1)] = A0[d];
    s_theta[seat_player(2 * k + 1, 0, m1)] = A0[3 * nblk + d];
  }
  __syncthreads();
  for (int k = tid; k < r; k += NT) E.theta[k] = s_theta[k];
  const double* Wd = reinterpret_cast<const double*>(W2);  // W[j][i] = Wd[((i >> 1) * m + j) * 2 + (i & 1)]
  if (!LOGW)
    for (int e = tid; e < r * r; e += NT) {
      const int i = e / r, j = e - i * r;
      E.W[(size_t)i * ldg + j] = Wd[((i >> 1) * m + j) * 2 + (i & 1)];
    }
  for (int k = tid; k < r; k += NT) {
    const double v = s_theta[k];
    int pos = 0;
    for (int j = 0; j < r; ++j) {
      const double u = s_theta[j];
      pos += (u > v) || (u == v && j < k);
    }
    E.order[pos] = k;
    s_pos[k] = pos;
  }
  __syncthreads();
  // structured path (scaled_out): G (unused there) <- eigenvectors in descending eigenvalue order, scaled by
  // y_std sqrt(theta): column pos of row t is the coefficient of basis vector t in factor row pos
  if (scaled_out && !LOGW) {
    for (int e = tid; e < r * r; e += NT) {
      const int i = e / r, j = e - i * r;
      const double th = s_theta[j];
      E.G[(size_t)i * ldg + s_pos[j]] = Wd[((i >> 1) * m + j) * 2 + (i & 1)] * (sc->y_std * sqrt(th > 0.0 ? th : 0.0));
    }
  }
  if (tid == 0) sc->lml = (double)sweeps;  // diagnostics: Jacobi sweeps of this factorisation
#ifdef GPET_JAC_PROF
  if ((tid == 0 || tid == NT - 64) && (blockIdx.y == 0 || blockIdx.y == 700) && sweeps > 0) {
    const double n = (double)sweeps * m1;
    printf("jac prof blk %d tid %d: per round: read+rotate %.0f | wait %.0f | write %.0f | wait %.0f | W / params %.0f | wait %.0f | total %.0f cycles\n",
           (int)blockIdx.y, tid, pq[0] / n, pq[1] / n, pq[2] / n, pq[3] / n, pq[4] / n, pq[5] / n, pq[6] / n);
  }
#endif
#undef JAC_CLK
}

// The eigenvectors from the rotation log of k_jacobi_seat<.., LOGW>: wave = component i (row i of W over the players), lane
// k = pair k of the seating: x = the entry of the player in slot 2k, y = of the player in slot 2k + 1.  A round rotates
// (x, y) by the pair's (c, s) -- the arithmetic of the LDS form, W[.][p] = c p - s q, W[.][q] = s p + c q -- and then moves
// every player of the circle one seat on: x one lane down, y one lane up, the two ends handed over (slot 0's player goes to
// slot 3, the last odd slot's player to the last even slot; the pivot in slot 1 stays), by whole-wave DPP shifts.  The log
// is read sixteen rounds at a time, the next sixteen requested before these are applied.  After whole sweeps everybody is
// back in the seats of round 0.
__device__ __forceinline__ double wp_lane_from_above(double v, double keep) {  // lane k <- lane k + 1 (the last lane keeps `keep`)
  const long long b = __double_as_longlong(v), o = __double_as_longlong(keep);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)b, 0x130, 0xf, 0xf, false);  // wave_shl:1
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), 0x130, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wp_lane_from_below(double v, double keep) {  // lane k <- lane k - 1 (lane 0 keeps `keep`)
  const long long b = __double_as_longlong(v), o = __double_as_longlong(keep);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)b, 0x138, 0xf, 0xf, false);  // wave_shr:1
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), 0x138, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__global__ void __launch_bounds__(256) k_jacobi_wpass(EdgeDev* edges, int scaled_out) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  const int r = sc->rank, ldg = E.r_cap;
  const int m = (r + 1) & ~1, half = m >> 1, m1 = m - 1;
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  __shared__ int s_pos[96];
  for (int k = threadIdx.x; k < r; k += 256) s_pos[E.order[k]] = k;  // player -> place in descending order (k_jacobi_seat wrote E.order)
  __syncthreads();
  if (i >= r || r < 1) return;
  const int total = (int)sc->lml * m1;  // (sweeps of this factorisation x rounds)
  const double2* jlog = reinterpret_cast<const double2*>(E.jlog);
  const bool on = lane < half;
  const int pe = seat_player(2 * lane, 0, m1), po = seat_player(2 * lane + 1, 0, m1);
  double x = (on && pe == i) ? 1.0 : 0.0, y = (on && po == i) ? 1.0 : 0.0;
  const bool lane0 = lane == 0, lane_last = lane == half - 1;
  constexpr int PF = 16;  // (rounds per block: ~1 800 cycles of work cover the round trip of the next block's loads)
  double2 cur[PF], nxt[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) cur[u] = (on && u < total) ? jlog[(size_t)u * half + lane] : make_double2(1.0, 0.0);
  for (int r0 = 0; r0 < total; r0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) nxt[u] = (on && r0 + PF + u < total) ? jlog[(size_t)(r0 + PF + u) * half + lane] : make_double2(1.0, 0.0);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (r0 + u < total) {  // (uniform)
        const double c = cur[u].x, sn = cur[u].y;
        const double xr = fma(c, x, -(sn * y)), yr = fma(sn, x, c * y);
        const double xd = wp_lane_from_above(xr, xr);
        const double yu = wp_lane_from_below(lane0 ? xr : yr, yr);
        x = lane_last ? yr : xd;
        y = yu;
      }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) cur[u] = nxt[u];
  }
  if (!on) return;
  // lane k holds component i of the eigenvectors of players pe and po
  if (pe < r) E.W[(size_t)i * ldg + pe] = x;
  if (po < r) E.W[(size_t)i * ldg + po] = y;
  if (scaled_out) {
    if (pe < r) {
      const double th = E.theta[pe];
      E.G[(size_t)i * ldg + s_pos[pe]] = x * (sc->y_std * sqrt(th > 0.0 ? th : 0.0));
    }
    if (po < r) {
      const double th = E.theta[po];
      E.G[(size_t)i * ldg + s_pos[po]] = y * (sc->y_std * sqrt(th > 0.0 ? th : 0.0));
    }
  }
}

// ---- structured loop path (training points on the pixel grid, rank(rho) <= 96) ---------------
// The prior covariance on the unit-spaced grid is c * rho with rho Toeplitz and FIXED for the whole
// trace: rho = Q Lam Q^T (rank r0 ~ 2.6 Lg / l for RBF) is factored once per edge at construction.
// With every training x on the grid, K_o* = c rho[obs, :] and the posterior covariance is
//     Sigma / y_std^2 = Q (c Lam - U^T U) Q^T,   U = L^-1 (c Q[obs, :] Lam)       (n x r0)
// so an iteration needs only the r0 x r0 matrix H = c Lam - U^T U and its Jacobi eigen-decomposition
// H = R Theta R^T:  the singular pairs of Sigma are (y_std^2 theta_k, Q r_k) -- no Lg x Lg covariance,
// no pivoted Cholesky, no Gram matrix; the posterior mean is y_std * Q (c Lam Q[obs,:]^T alpha) + m.
__global__ void __launch_bounds__(256) k_rho_fill(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const int Lg = E.Lg;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < (size_t)Lg * Lg; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / Lg), j = (int)(e - (size_t)i * Lg);
    E.cov[e] = (i == j) ? 1.0
                        : corr_fn(E, (double)(E.x_st + i) / E.length_scale,
                                  (double)(E.x_st + j) / E.length_scale);
  }
}

// rows of A = sqrt(lam_a) q_a^T (factor of rho)  ->  Q0 (unit rows) and lam0 = |row|^2
__global__ void __launch_bounds__(256) k_struct_basis(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  const int a = blockIdx.x;
  if (a >= sc->rank) return;
  __shared__ double s_red[16];
  double part = 0.0;
  for (int j = threadIdx.x; j < E.Lg; j += blockDim.x) {
    const double v = E.A[(size_t)a * E.Lg + j];
    part += v * v;
  }
  const double nrm2 = block_sum(part, s_red);
  const double inv = 1.0 / sqrt(nrm2);
  for (int j = threadIdx.x; j < E.Lg; j += blockDim.x) E.Q0[(size_t)a * E.Lg + j] = E.A[(size_t)a * E.Lg + j] * inv;
  if (threadIdx.x == 0) E.lam0[a] = nrm2;
}

// H = c Lam - U^T U (into E.C, row stride r_cap), beta, posterior mean.  One workgroup per edge;
// U (n x r0) lives in LDS, one thread per column for the forward substitution, rows of L streamed
// through LDS.
__global__ void __launch_bounds__(1024) k_struct_H(EdgeDev* edges, int l_in_lds) {
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || !E.structured) return;
  extern __shared__ double s_dyn[];
  const int n = sc->n, r0 = E.r0, Lg = E.Lg, ldk = E.n_cap, ldc = E.r_cap;
  const int ldu = r0 | 1;
  double* U = s_dyn;                                            // [n_cap][ldu]
  double* Lp = U + (size_t)E.n_cap * ldu;                       // l_in_lds: packed lower triangle of L (row i at i (i + 1) / 2);
                                                                // otherwise [n_cap]: the current row of L
  double* s_b = Lp + (l_in_lds ? (size_t)E.n_cap * (E.n_cap + 1) / 2 : (size_t)E.n_cap);  // [r_cap] beta
  const int tid = threadIdx.x, bs = blockDim.x;
  const double c = sc->amp;
#ifdef GPET_SH_PROF
  const long long p0 = clock64();
  const long long w0 = wall_clock64();
#endif
  // right-hand sides  Bo[i][a] = c * lam0[a] * Q0[a][idx_i];  the Cholesky factor comes into LDS once if it fits.
  // The grid indices of the training points go to LDS first (s_b is free until beta is formed): the gathers of Q0 then
  // depend on nothing and overlap with each other and with the copy of L -- two global round trips for the phase
  // instead of two per gathered element
  int* s_idx = reinterpret_cast<int*>(s_b);  // (n <= n_cap <= 128 ints in r_cap >= 4 doubles ... sized below)
  const bool idx_in_lds = (size_t)n * sizeof(int) <= (size_t)E.r_cap * sizeof(double);
  if (idx_in_lds)
    for (int i = tid; i < n; i += bs) s_idx[i] = (int)E.xt[i] - E.x_st;
  if (l_in_lds) {
    for (int e = tid; e < n * n; e += bs) {
      const int i = e / n, t = e - i * n;
      if (t <= i) Lp[i * (i + 1) / 2 + t] = E.K[(size_t)i * ldk + t];
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int e = tid; e < n * r0; e += bs) {
    const int i = e / r0, a = e - i * r0;
    const int idx = idx_in_lds ? s_idx[i] : (int)E.xt[i] - E.x_st;
    U[i * ldu + a] = c * E.lam0[a] * E.Q0[(size_t)a * Lg + idx];
  }
  __syncthreads();
#ifdef GPET_SH_PROF
  const long long p1 = clock64();
#endif
  // beta_a = sum_i Bo[i][a] alpha_i
  for (int a = tid; a < r0; a += bs) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += U[i * ldu + a] * E.alpha[i];
    s_b[a] = acc;
    E.beta[a] = acc;
  }
#ifdef GPET_SH_PROF
  const long long p1b = clock64();
#endif
  if (l_in_lds) {
    // U = L^-1 Bo in blocks of 16 rows: (a) every thread takes one (row of the block, column) entry and subtracts the
    // rows ABOVE the block -- independent inner products, all 16 waves busy; (b) one thread per column finishes the
    // block's 16 rows (<= 120 dependent steps).  The terms of an entry are subtracted in ascending row order, as by
    // the one-thread-per-column loop of round 1 (which left 14 of the 16 waves idle for n^2 / 2 dependent steps: 65 %
    // of this kernel at n = 90), so the result is the same to the bit.
    for (int i0 = 0; i0 < n; i0 += 16) {
      const int nb = (n - i0) < 16 ? (n - i0) : 16;
      if (i0 > 0) {
        for (int o = tid; o < nb * r0; o += bs) {
          const int ri = o / r0, ca = o - ri * r0;
          const int i = i0 + ri;
          const double* li = Lp + i * (i + 1) / 2;
          double acc = U[i * ldu + ca];
#pragma unroll 8
          for (int t = 0; t < i0; ++t) acc -= li[t] * U[t * ldu + ca];
          U[i * ldu + ca] = acc;
        }
        __syncthreads();
      }
      if (tid < r0) {
        // (the block's 16 entries of the column in registers, the triangle fully unrolled: the reads of L carry no
        //  dependence and are issued ahead of the chain instead of costing one LDS round trip per step)
        double u[16];
#pragma unroll
        for (int ri = 0; ri < 16; ++ri) u[ri] = (ri < nb) ? U[(i0 + ri) * ldu + tid] : 0.0;
#pragma unroll
        for (int ri = 0; ri < 16; ++ri) {
          if (ri < nb) {
            const int i = i0 + ri;
            const double* li = Lp + i * (i + 1) / 2 + i0;
            double acc = u[ri];
#pragma unroll
            for (int t = 0; t < ri; ++t) acc -= li[t] * u[t];
            u[ri] = acc / li[ri];
          }
        }
#pragma unroll
        for (int ri = 0; ri < 16; ++ri)
          if (ri < nb) U[(i0 + ri) * ldu + tid] = u[ri];
      }
      __syncthreads();
    }
  } else {
    // (many training points: L does not fit next to U; its rows are streamed through LDS one at a time)
    for (int i = 0; i < n; ++i) {
      __syncthreads();
      const double* ri = E.K + (size_t)i * ldk;
      for (int t = tid; t <= i; t += bs) Lp[t] = ri[t];
      __syncthreads();
      if (tid < r0) {
        double acc = U[i * ldu + tid];
        for (int t = 0; t < i; ++t) acc -= Lp[t] * U[t * ldu + tid];
        U[i * ldu + tid] = acc / Lp[i];
      }
    }
  }
  __syncthreads();
#ifdef GPET_SH_PROF
  const long long p2 = clock64();
#endif
  // H[a][b] = c lam0[a] delta_ab - sum_i U[i][a] U[i][b]: U^T U on the matrix cores, one 16 x 16 tile of the upper
  // triangle per wave (v_mfma_f64_16x16x4: A lane l <- U[4 q + (l >> 4)][16 ta + (l & 15)], B the same with tb; both
  // operands straight from the LDS copy of U), mirrored on the way out
  {
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lq = lane >> 4;
    const int nt = (r0 + 15) >> 4, ntile = nt * (nt + 1) / 2, nwave = bs >> 6;
    for (int tile = wv; tile < ntile; tile += nwave) {
      int tb = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
      while (tb * (tb + 1) / 2 > tile) --tb;
      while ((tb + 1) * (tb + 2) / 2 <= tile) ++tb;
      const int ta = tile - tb * (tb + 1) / 2;  // ta <= tb
      const int ca = 16 * ta + li, cb = 16 * tb + li;
      v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
      for (int i0 = 0; i0 < n; i0 += 4) {
        const int i = i0 + lq;
        const double av = (i < n && ca < r0) ? U[i * ldu + ca] : 0.0;
        const double bv = (i < n && cb < r0) ? U[i * ldu + cb] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int a = 16 * ta + lq + 4 * g, b = cb;
        if (a < r0 && b < r0 && b >= a) {
          const double v = ((a == b) ? c * E.lam0[a] : 0.0) - acc[g];
          E.C[(size_t)a * ldc + b] = v;
          E.C[(size_t)b * ldc + a] = v;
        }
      }
    }
  }
#ifdef GPET_SH_PROF
  const long long p3 = clock64();
#endif
  // posterior mean on the grid: y_std * (Q beta) + y_mean        sklearn_gpr.py:382-385
  {  // (two threads per grid point, even and odd basis vectors, loads unrolled: the 72 reads of a thread were one chain)
    const int hh = tid & 1;
    const double y_std = sc->y_std, y_mean = sc->y_mean;
    for (int j = tid >> 1; j < Lg; j += bs >> 1) {
      double acc = 0.0;
#pragma unroll 6
      for (int a = hh; a < r0; a += 2) acc += E.Q0[(size_t)a * Lg + j] * s_b[a];
      acc += __shfl_xor(acc, 1, 2);
      if (hh == 0) E.mean[j] = y_std * acc + y_mean;
    }
  }
  if (tid == 0) sc->rank = r0;
#ifdef GPET_SH_PROF
  if (tid == 0 && blockIdx.y == 5) printf("k_struct_H n=%d: build+L %lld | beta %lld | substitution %lld | H %lld | mean %lld ticks; whole kernel %lld ticks = %lld wall ticks of 10 ns\n", n, p1 - p0, p1b - p1, p2 - p1b, p3 - p2, clock64() - p3, clock64() - p0, wall_clock64() - w0);
#endif
}

// ---- structured path with MANY training points (n_cap > 128): U lives in HBM (the V buffer, row stride r_cap) ----
// B rows -> blocked forward substitution U = L^-1 B (left-looking: block k subtracts the blocks before it, then
// solves against its diagonal block in LDS; the r0 columns are split over workgroups of 16) -> H, beta, mean.
#define SB_COLS 16
__global__ void __launch_bounds__(256) k_structB_build(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || !E.structured) return;
  const int n = sc->n, r0 = E.r0, Lg = E.Lg, ldu = E.r_cap;
  const double c = sc->amp;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * r0; e += gridDim.x * blockDim.x) {
    const int i = e / r0, a = e - i * r0;
    const int idx = (int)E.xt[i] - E.x_st;
    E.V[(size_t)i * ldu + a] = c * E.lam0[a] * E.Q0[(size_t)a * Lg + idx];
  }
}

// beta_a = sum_i B[i][a] alpha_i  (before the substitution overwrites B)
__global__ void __launch_bounds__(128) k_struct_beta(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || !E.structured) return;
  const int n = sc->n, r0 = E.r0, ldu = E.r_cap;
  for (int a = threadIdx.x; a < r0; a += blockDim.x) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += E.V[(size_t)i * ldu + a] * E.alpha[i];
    E.beta[a] = acc;
  }
}

// block k0 of U = L^-1 B for the 16 columns of this workgroup
// (predict != 0: the same substitution on the Lg columns of V = K_*^T, row stride Lg -- the generic path's
//  V = L^-1 K_*^T for many training points)
__global__ void __launch_bounds__(256) k_struct_trsm(EdgeDev* edges, int k0, int predict) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || (!predict && !E.structured)) return;
  const int n = sc->n, r0 = predict ? E.Lg : E.r0, ld = E.n_cap, ldu = predict ? E.Lg : E.r_cap;
  const int a0 = blockIdx.x * SB_COLS;
  if (k0 >= n || a0 >= r0) return;
  const int nb = (n - k0) < CB ? (n - k0) : CB;
  __shared__ double sL[CB][CB + 1];       // L[k0 + i][j0 + t] of the block being subtracted, then the diagonal block
  __shared__ double sU[CB][SB_COLS + 1];  // U[j0 + t][a0 + a]
  __shared__ double sX[CB][SB_COLS + 1];  // the block being solved
  const int tid = threadIdx.x;
  const int ca = tid & (SB_COLS - 1), ri = tid >> 4;  // thread tile: column ca, rows ri, ri + 16, ri + 32, ri + 48
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int j0 = 0; j0 < k0; j0 += CB) {
    __syncthreads();
    for (int e = tid; e < CB * CB; e += 256) {
      const int i = e >> 6, t = e & 63;
      sL[i][t] = (i < nb) ? E.K[(size_t)(k0 + i) * ld + j0 + t] : 0.0;
    }
    for (int e = tid; e < CB * SB_COLS; e += 256) {
      const int t = e >> 4, a = e & 15;
      sU[t][a] = (a0 + a < r0) ? E.V[(size_t)(j0 + t) * ldu + a0 + a] : 0.0;
    }
    __syncthreads();
#pragma unroll 8
    for (int t = 0; t < CB; ++t) {
      const double u = sU[t][ca];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += sL[ri + 16 * q][t] * u;
    }
  }
  __syncthreads();
  for (int e = tid; e < CB * CB; e += 256) {
    const int i = e >> 6, t = e & 63;
    sL[i][t] = (i < nb && t <= i) ? E.K[(size_t)(k0 + i) * ld + k0 + t] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = ri + 16 * q;
    sX[i][ca] = (i < nb && a0 + ca < r0) ? E.V[(size_t)(k0 + i) * ldu + a0 + ca] - acc[q] : 0.0;
  }
  __syncthreads();
  if (tid < SB_COLS) {  // forward substitution against the diagonal block, one column per thread
    for (int i = 0; i < nb; ++i) {
      double x = sX[i][tid];
      for (int t = 0; t < i; ++t) x -= sL[i][t] * sX[t][tid];
      sX[i][tid] = x / sL[i][i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = ri + 16 * q;
    if (i < nb && a0 + ca < r0) E.V[(size_t)(k0 + i) * ldu + a0 + ca] = sX[i][ca];
  }
}

// H = c Lam - U^T U (lower 16x16 tiles, mirrored) from U in HBM
__global__ void __launch_bounds__(256) k_struct_Hbig(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.z];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || !E.structured) return;
  const int n = sc->n, r0 = E.r0, ldu = E.r_cap, ldc = E.r_cap;
  const int a0 = blockIdx.y * 16, b0 = blockIdx.x * 16;
  if (b0 > a0 || a0 >= r0) return;
  __shared__ double sA[64][17], sB[64][17];
  const int tid = threadIdx.x, ta = tid >> 4, tb = tid & 15;
  double acc = 0.0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * 16; e += 256) {
      const int i = e >> 4, q = e & 15;
      const bool in = i0 + i < n;
      sA[i][q] = (in && a0 + q < r0) ? E.V[(size_t)(i0 + i) * ldu + a0 + q] : 0.0;
      sB[i][q] = (in && b0 + q < r0) ? E.V[(size_t)(i0 + i) * ldu + b0 + q] : 0.0;
    }
    __syncthreads();
#pragma unroll 16
    for (int i = 0; i < 64; ++i) acc += sA[i][ta] * sB[i][tb];
  }
  const int a = a0 + ta, b = b0 + tb;
  if (a < r0 && b < r0 && b <= a) {
    const double v = ((a == b) ? sc->amp * E.lam0[a] : 0.0) - acc;
    E.C[(size_t)a * ldc + b] = v;
    E.C[(size_t)b * ldc + a] = v;
  }
}

// posterior mean on the grid: y_std * (Q beta) + y_mean, and the rank of the factorisation to come
__global__ void __launch_bounds__(256) k_struct_mean(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || !E.structured) return;
  const int r0 = E.r0, Lg = E.Lg;
  __shared__ double s_b[96];
  for (int a = threadIdx.x; a < r0; a += blockDim.x) s_b[a] = E.beta[a];
  __syncthreads();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < Lg) {
    double acc = 0.0;
    for (int a = 0; a < r0; ++a) acc += E.Q0[(size_t)a * Lg + j] * s_b[a];
    E.mean[j] = sc->y_std * acc + sc->y_mean;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) sc->rank = r0;
}

// factor rows of the structured path: A[k, :] = y_std * sqrt(theta_k) * (Q r_k)^T as a column-tiled product
// (r0 x r0) . (r0 x 64) on v_mfma_f64_16x16x4_f64: the Q0 tile and the scaled, ordered eigenvectors sit in
// LDS, wave w owns 16 of the 64 columns and all MT row tiles, so Q0 is read from HBM once per iteration.
// The sign convention needs whole-row sums: per-tile partials here, the flip in k_struct_sign (fixed order).
// MT (row tiles of 16) is a template parameter for the same reason as the K extent of the sample GEMM.
#define SR_TJ 64
typedef double v4f64_ __attribute__((ext_vector_type(4)));
template <int MT, int KS>
__device__ __forceinline__ void struct_rows_body(const EdgeDev& E, double* s_w, int tile) {
  const int r = E.r0, Lg = E.Lg;
  const int j0 = tile * SR_TJ;
  constexpr int kpad = 4 * KS;
  constexpr int ldw = 16 * MT + 1;  // [kpad][ldw]: s_w[t][k] = scaled coefficient of basis vector t in row k
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int j = j0 + 16 * w + li;
#ifdef GPET_SR_PROF
  const long long c0 = clock64();
#endif
  // B operand (Q0[t][j], this wave's 16 columns) straight into registers; A operand (E.G, written by the Jacobi
  // kernel) through LDS, shared by the four waves
  double breg[KS];
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const int t = 4 * q + lq;
    breg[q] = (t < r && j < Lg) ? E.Q0[(size_t)t * Lg + j] : 0.0;
  }
  // (all the loads of the coefficient matrix in flight at once -- the trip count is a compile-time constant --
  //  instead of six dependent batches of four: the staging was two thirds of this kernel's time)
  {
    constexpr int NE = kpad * 16 * MT, NL = (NE + 255) / 256;
    double gv[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int e = tid + 256 * u;
      const int t = e / (16 * MT), k = e - t * (16 * MT);
      gv[u] = (e < NE && t < r && k < r) ? E.G[(size_t)t * E.r_cap + k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int e = tid + 256 * u;
      const int t = e / (16 * MT), k = e - t * (16 * MT);
      if (e < NE) s_w[t * ldw + k] = gv[u];
    }
  }
  __syncthreads();
#ifdef GPET_SR_PROF
  const long long c1 = clock64();
#endif
  v4f64_ acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f64_){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const double* wr = s_w + (4 * q + lq) * ldw + li;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[16 * mt], breg[q], acc[mt], 0, 0, 0);
  }
#ifdef GPET_SR_PROF
  const long long c2 = clock64();
#endif
  __syncthreads();  // s_w is reused, as [r][64], for the products A[k][j] / (j + 1)
  double* s_p = s_w;
  const double inv_j = 1.0 / (double)(j + 1);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int k = 16 * mt + lq + 4 * g;
      if (k < r) {
        const double v = acc[mt][g];
        if (j < Lg) E.A[(size_t)k * Lg + j] = v;
        s_p[k * 64 + 16 * w + li] = (j < Lg) ? v * inv_j : 0.0;
      }
    }
  __syncthreads();
#ifdef GPET_SR_PROF
  const long long c3 = clock64();
#endif
  const int ntile = (Lg + SR_TJ - 1) / SR_TJ;
  // row sums of the tile: 4 independent accumulators per row (fixed order), rows strided over the threads
  for (int k = tid; k < r; k += 256) {
    const double* row = s_p + k * 64;
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
    for (int c = 0; c < 64; c += 4) {
      d0 += row[(c + k) & 63];
      d1 += row[(c + 1 + k) & 63];
      d2 += row[(c + 2 + k) & 63];
      d3 += row[(c + 3 + k) & 63];
    }
    E.row_part[(size_t)k * ntile + tile] = (d0 + d1) + (d2 + d3);
  }
#ifdef GPET_SR_PROF
  if (tid == 0 && blockIdx.y == 3 && blockIdx.x == 2)
    printf("k_struct_rows: loads + staging %lld | mfma %lld | stores + products %lld | row sums %lld cycles\n", c1 - c0, c2 - c1, c3 - c2, clock64() - c3);
#endif
}

__global__ void __launch_bounds__(256) k_struct_rows(EdgeDev* edges) {
  int edge, tile;  // the column tiles of an edge on one XCD: its r x r coefficient matrix comes out of HBM once
  xcd_edge_part((int)gridDim.x, edge, tile);
  const EdgeDev E = edges[edge];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || !E.structured) return;
  if (tile * SR_TJ >= E.Lg || E.r0 < 1) return;
  extern __shared__ __attribute__((aligned(16))) double s_rows[];
  const int r = E.r0;  // uniform over the workgroup
  if (r <= 32) struct_rows_body<2, 8>(E, s_rows, tile);
  else if (r <= 48) struct_rows_body<3, 12>(E, s_rows, tile);
  else if (r <= 64) struct_rows_body<4, 16>(E, s_rows, tile);
  else if (r <= 72) struct_rows_body<5, 18>(E, s_rows, tile);
  else if (r <= 80) struct_rows_body<5, 20>(E, s_rows, tile);
  else struct_rows_body<6, 24>(E, s_rows, tile);
}

// sign convention sum_j A[k][j] / (j + 1) >= 0 from the per-tile partials
__global__ void __launch_bounds__(128) k_struct_sign(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || !E.structured) return;
  const int k = blockIdx.x, Lg = E.Lg;
  if (k >= E.r0) return;
  const int ntile = (Lg + SR_TJ - 1) / SR_TJ;
  double d = 0.0;
  for (int c = 0; c < ntile; ++c) d += E.row_part[(size_t)k * ntile + c];
  if (d < 0.0)
    for (int j = threadIdx.x; j < Lg; j += blockDim.x) E.A[(size_t)k * Lg + j] = -E.A[(size_t)k * Lg + j];
}

// ---- large ranks (r_cap > 96: Matern spectra, short RBF length scales) ---------------------
// Cyclic Jacobi applied directly to the Lg x Lg posterior covariance, spread over the whole GPU:
// one parameter kernel + one apply kernel per round (the round's Lg/2 disjoint rotations touch
// disjoint 2x2 blocks), matrices in HBM/L2.  No pivoted Cholesky / Gram step: the eigenvectors
// are the singular vectors numpy's SVD returns (up to sign), singular values = |eigenvalues|.
__device__ __forceinline__ void jb_pair(int m, int round, int k, int& p, int& q) {
  if (k == 0) {
    p = m - 1;
    q = round;
  } else {
    p = (round + k) % (m - 1);
    q = (round - k + (m - 1)) % (m - 1);
  }
  if (p > q) {
    const int t = p;
    p = q;
    q = t;
  }
}

__global__ void __launch_bounds__(256) k_jb_init(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  const int r = E.Lg, ld = E.r_cap;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < (size_t)r * r; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / r), j = (int)(e - (size_t)i * r);
    E.C[(size_t)i * ld + j] = E.cov[(size_t)i * r + j];
    E.W[(size_t)i * ld + j] = (i == j) ? 1.0 : 0.0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sc->rank = r;
    E.jb_norm[0] = 1.0;  // "not converged yet"
    E.jb_norm[1] = 1.0;
  }
}

__global__ void __launch_bounds__(256) k_jb_params(EdgeDev* edges, int round) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  if (E.jb_norm[0] <= 1e-24 * E.jb_norm[1]) return;  // converged: remaining launches are no-ops
  const int r = E.Lg, ld = E.r_cap;
  const int m = (r + 1) & ~1, half = m >> 1;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= half) return;
  int p, q;
  jb_pair(m, round, k, p, q);
  double c = 1.0, s = 0.0;
  if (q < r) {
    const double apq = E.C[(size_t)p * ld + q];
    const double app = E.C[(size_t)p * ld + p], aqq = E.C[(size_t)q * ld + q];
    if (fabs(apq) > 1e-300 && fabs(apq) > 1e-18 * sqrt(fabs(app * aqq))) {
      const double tau = (aqq - app) / (2.0 * apq);
      const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
      c = 1.0 / sqrt(1.0 + t * t);
      s = t * c;
    }
  }
  E.jb_cs[2 * k] = c;
  E.jb_cs[2 * k + 1] = s;
}

// blockIdx.x enumerates (a, b-tile) for the 2x2 blocks, then the W items
__global__ void __launch_bounds__(256) k_jb_apply(EdgeDev* edges, int round) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  if (E.jb_norm[0] <= 1e-24 * E.jb_norm[1]) return;
  const int r = E.Lg, ld = E.r_cap;
  const int m = (r + 1) & ~1, half = m >> 1;
  const long long nblk = (long long)half * half, nw = (long long)half * r;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < nblk + nw;
       e += (long long)gridDim.x * blockDim.x) {
    if (e < nblk) {
      const int a = (int)(e / half), b = (int)(e - (long long)a * half);
      const double sa = E.jb_cs[2 * a + 1], sb = E.jb_cs[2 * b + 1];
      if (sa == 0.0 && sb == 0.0) continue;
      const double ca = E.jb_cs[2 * a], cb = E.jb_cs[2 * b];
      int pa, qa, pb, qb;
      jb_pair(m, round, a, pa, qa);
      jb_pair(m, round, b, pb, qb);
      if (qa >= r || qb >= r) continue;  // the padding index of an odd size never rotates
      double* A = E.C;
      const double b00 = A[(size_t)pa * ld + pb], b01 = A[(size_t)pa * ld + qb];
      const double b10 = A[(size_t)qa * ld + pb], b11 = A[(size_t)qa * ld + qb];
      const double t00 = cb * b00 - sb * b01, t01 = sb * b00 + cb * b01;
      const double t10 = cb * b10 - sb * b11, t11 = sb * b10 + cb * b11;
      A[(size_t)pa * ld + pb] = ca * t00 - sa * t10;
      A[(size_t)qa * ld + pb] = sa * t00 + ca * t10;
      A[(size_t)pa * ld + qb] = ca * t01 - sa * t11;
      A[(size_t)qa * ld + qb] = sa * t01 + ca * t11;
    } else {
      const long long f = e - nblk;
      const int b = (int)(f / r), i = (int)(f - (long long)b * r);
      const double sb = E.jb_cs[2 * b + 1];
      if (sb == 0.0) continue;
      const double cb = E.jb_cs[2 * b];
      int pb, qb;
      jb_pair(m, round, b, pb, qb);
      if (qb >= r) continue;
      const double wp = E.W[(size_t)i * ld + pb], wq = E.W[(size_t)i * ld + qb];
      E.W[(size_t)i * ld + pb] = cb * wp - sb * wq;
      E.W[(size_t)i * ld + qb] = sb * wp + cb * wq;
    }
  }
}

// off-diagonal / diagonal square sums of the current matrix, in two steps: 64 workgroups write partial sums
// (into the rotation-parameter buffer, idle between sweeps), one wave adds them in a fixed order
#define JBN_PARTS 64
__global__ void __launch_bounds__(1024) k_jb_norms(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  __shared__ double s_red[16];
  const int r = E.Lg, ld = E.r_cap;
  double off = 0.0, dg = 0.0;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < (size_t)r * r; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / r), j = (int)(e - (size_t)i * r);
    const double v = E.C[(size_t)i * ld + j];
    if (i == j) dg += v * v; else off += v * v;
  }
  off = block_sum(off, s_red);
  dg = block_sum(dg, s_red);
  if (threadIdx.x == 0) {
    E.jb_cs[2 * blockIdx.x] = off;
    E.jb_cs[2 * blockIdx.x + 1] = dg;
  }
}
__global__ void __launch_bounds__(64) k_jb_norms_fin(EdgeDev* edges, int parts) {
  const EdgeDev E = edges[blockIdx.x];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  if (threadIdx.x == 0) {
    double off = 0.0, dg = 0.0;
    for (int p = 0; p < parts; ++p) {
      off += E.jb_cs[2 * p];
      dg += E.jb_cs[2 * p + 1];
    }
    E.jb_norm[0] = off;
    E.jb_norm[1] = dg;
  }
}

__global__ void __launch_bounds__(1024) k_jb_order(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  const int r = E.Lg, ld = E.r_cap;
  for (int k = threadIdx.x; k < r; k += blockDim.x) E.theta[k] = fabs(E.C[(size_t)k * ld + k]);
  __syncthreads();
  for (int k = threadIdx.x; k < r; k += blockDim.x) {
    const double v = E.theta[k];
    int pos = 0;
    for (int j = 0; j < r; ++j) {
      const double u = E.theta[j];
      pos += (u > v) || (u == v && j < k);
    }
    E.order[pos] = k;
  }
}

__global__ void __launch_bounds__(256) k_jb_rows(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  const int r = E.Lg, Lg = E.Lg, ld = E.r_cap, k = blockIdx.x;
  if (k >= r) return;
  __shared__ double s_red[16];
  const int col = E.order[k];
  const double sv = sqrt(E.theta[col]);
  double part = 0.0;
  for (int j = threadIdx.x; j < Lg; j += blockDim.x) {
    const double v = sv * E.W[(size_t)j * ld + col];
    E.A[(size_t)k * Lg + j] = v;
    part += v / (double)(j + 1);
  }
  const double dot = block_sum(part, s_red);
  if (dot < 0.0)
    for (int j = threadIdx.x; j < Lg; j += blockDim.x) E.A[(size_t)k * Lg + j] = -E.A[(size_t)k * Lg + j];
}

// step 4: factor rows  A[k, :] = sum_t W[t, order[k]] * G[t, :]  (= sqrt(s_k) v_k up to sign).
// Sign convention (LAPACK's is implementation-defined): sum_j A[k, j] / (j + 1) >= 0.
__global__ void __launch_bounds__(256) k_factor_rows(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected) return;
  const int r = sc->rank, Lg = E.Lg, k = blockIdx.x;
  if (k >= r) return;
  __shared__ double s_red[16];
  extern __shared__ double s_w[];  // [r]
  const int col = E.order[k];
  for (int t = threadIdx.x; t < r; t += blockDim.x) s_w[t] = E.W[(size_t)t * E.r_cap + col];
  __syncthreads();
  double part = 0.0;
  for (int j = threadIdx.x; j < Lg; j += blockDim.x) {
    double acc = 0.0;
    for (int t = 0; t < r; ++t) acc += s_w[t] * E.G[(size_t)t * Lg + j];
    E.A[(size_t)k * Lg + j] = acc;
    part += acc / (double)(j + 1);
  }
  const double dot = block_sum(part, s_red);
  if (dot < 0.0)
    for (int j = threadIdx.x; j < Lg; j += blockDim.x) E.A[(size_t)k * Lg + j] = -E.A[(size_t)k * Lg + j];
}

// ---------------------------------------------------------------------------------------
// K5  numpy legacy RandomState stream on the device: MT19937 (init_genrand) + legacy_gauss.
//     One workgroup per edge; 624-word blocks = 156 polar attempts (4 words each).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int mt_temper(unsigned int y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9D2C5680u;
  y ^= (y << 15) & 0xEFC60000u;
  y ^= y >> 18;
  return y;
}
__device__ __forceinline__ unsigned int mt_mix(unsigned int a, unsigned int b, unsigned int far) {
  const unsigned int y = (a & 0x80000000u) | (b & 0x7FFFFFFFu);
  return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
}

// (Tried and dropped, bit-identical but no faster: software-pipelining the blocks so that a third of the next block's
// twist shares each barrier phase with the attempts of the current one.)
// blockIdx.x = how many iterations ahead of the edge's current one this stream belongs to: the
// seeds of future iterations are known a priori (gpet.py:839), so a whole ring of them is
// generated by one launch, one workgroup per (iteration, edge).
#define MTQ_CAP 256  // ring of pending (r2, x1, x2, destinations) records: < 64 left over + 156 per block = 219 at most
#include "gpet_mt_jump.inc"  // MTJ_CB, MTJ_LEVELS, mtj_poly: jump-ahead polynomials (tools/gen_mt_jump.py)
// workspace of the chunked generator, per stream (= iteration ahead x edge): chunk states, accepted pairs per chunk and
// their exclusive prefix; per jump of a level: the 33 blocks of raw words its convolution reads
struct MtjWork {
  unsigned int* T;     // [streams][nc][624]
  int* cnt;            // [streams][nc]
  long long* offs;     // [streams][nc]
  unsigned int* src;   // [streams][nc / 2 + 1][MTJ_SRC]
  int nc;              // chunks per stream
};
#define MTJ_SRC (33 * 624)
// CHUNKED = false: one workgroup generates a whole stream from its seed (blockIdx = (iterations ahead, edge)).
// CHUNKED = true: blockIdx = (chunk, iterations ahead, edge): the workgroup starts from the state of its chunk (k_mtj_*
// below found it by jumping ahead) at the stream position the chunks before it have filled, and stops after MTJ_CB
// blocks -- except the last chunk, which runs until the stream is complete (the chunk count is an estimate).
// NW = waves per workgroup: 4 (round 1-2: 227 lanes twist in three single passes, the fourth wave drains the queue during
// the attempts) or 3: the twist's first two phases then take two passes on wave 0 only (lanes 192..226 of a phase), the
// last wave makes its 28 attempts and drains the queue afterwards -- fewer waves executing each block's instructions.
template <bool CHUNKED, int NW>
__global__ void __launch_bounds__(64 * NW) k_mt_normals(EdgeDev* edges, const unsigned int* seeds, int add_iter,
                                                       int iter_abs, int z_store, MtjWork wk) {
#pragma clang fp contract(off)
  constexpr int NT = 64 * NW;
  const int e_idx = CHUNKED ? (int)blockIdx.z : (int)blockIdx.y, ahead = CHUNKED ? (int)blockIdx.y : (int)blockIdx.x;
  const EdgeDev E = edges[e_idx];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  // iter_abs >= 0: the host names the iteration (the RNG stream runs ahead of the loop, so the
  // device counter is not meaningful here); otherwise relative to the edge's current iteration
  const int iter_idx = (iter_abs >= 0 ? iter_abs : sc->iter) + ahead;
  double* __restrict__ Zs = E.Z + (size_t)(iter_idx % E.z_ring) * ((size_t)E.S * E.z_cols);
  const int chunk = CHUNKED ? (int)blockIdx.x : 0;
  const size_t stream = CHUNKED ? (size_t)e_idx * gridDim.y + blockIdx.y : 0;
  __shared__ unsigned int s_mt[2][624];
  __shared__ int s_cnt[2][4];  // (accepted attempts per wave; NW = 3: the fourth entry stays 0)
  __shared__ int s_qtail;  // records queued so far (monotonic; slots are taken with one LDS atomic per wave)
  __shared__ double q_r2[MTQ_CAP], q_x1[MTQ_CAP], q_x2[MTQ_CAP];
  __shared__ int q_d0[MTQ_CAP], q_d1[MTQ_CAP];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) s_qtail = 0;
  if (tid < 8) (&s_cnt[0][0])[tid] = 0;
  if (CHUNKED) {
    const unsigned int* st0 = wk.T + (stream * wk.nc + chunk) * 624;
    for (int i = tid; i < 624; i += NT) s_mt[0][i] = st0[i];
  } else if (tid == 0) {
    // seed of iteration k (0-based) = base + k + 1 (gpet.py:839)
    unsigned int p = seeds[e_idx] + (add_iter ? (unsigned int)(iter_idx + 1) : 0u);
    s_mt[0][0] = p;
    for (int i = 1; i < 624; ++i) {
      p = 1812433253u * (p ^ (p >> 30)) + (unsigned int)i;
      s_mt[0][i] = p;
    }
  }
  __syncthreads();
  // zc: row stride of the stored block; zs: how many leading normals of a row are stored (the structured loop knows
  // its factors have at most r0 rows, so the host asks for r0 rounded up to 4 instead of the whole capacity)
  const int Lg = E.Lg, zc = E.z_cols, zs = (z_store > 0 && z_store < zc) ? z_store : zc;
  const long long total = (long long)E.S * Lg;
  const long long need_pairs = (total + 1) / 2;
  // Only the first z_cols normals of every sample row are stored, but every attempt's accept/reject decision is
  // needed (it positions the rest of the stream).  When few are stored (z_cols << Lg: the structured loop path keeps
  // ~1 in 7) the log/sqrt of the stored ones would still run in every wave of every block, so those pairs are queued
  // in LDS and the fourth wave -- idle during the attempts -- evaluates them 64 at a time; when most are stored
  // they are evaluated in place.
  const bool queued = 2 * zs <= Lg;
  long long done_pairs = CHUNKED ? wk.offs[stream * wk.nc + chunk] : 0;
  // (row, column) of the normal at stream position 2 * done_pairs
  int row0 = (int)((2 * done_pairs) / Lg), col0 = (int)((2 * done_pairs) - (long long)row0 * Lg);
  const bool last_chunk = !CHUNKED || chunk == wk.nc - 1;
  int q_popped = 0;  // records evaluated so far (same value in every thread)
  int cur = 0, it = 0;
  auto emit = [&](double r2, double x1, double x2, int d0, int d1) {
    const double f = sqrt(-2.0 * log(r2) / r2);
    if (d0 >= 0) Zs[d0] = f * x2;
    if (d1 >= 0) Zs[d1] = f * x1;
  };
  while (done_pairs < need_pairs && (last_chunk || it < MTJ_CB)) {
    unsigned int* o = s_mt[cur];
    unsigned int* nw = s_mt[cur ^ 1];
#pragma unroll
    for (int i = tid; i < 227; i += NT) nw[i] = mt_mix(o[i], o[i + 1], o[i + 397]);
    __syncthreads();
#pragma unroll
    for (int i = tid; i < 227; i += NT) nw[227 + i] = mt_mix(o[227 + i], o[228 + i], nw[i]);
    __syncthreads();
    if (tid < 169) nw[454 + tid] = mt_mix(o[454 + tid], o[455 + tid], nw[227 + tid]);
    if (tid == NT - 1) nw[623] = mt_mix(o[623], nw[0], nw[396]);
    __syncthreads();
    // 156 polar attempts on waves 0-2; the last wave drains the queue of the previous blocks (NW = 4: meanwhile).
    // Accept / reject needs r2 = x1^2 + x2^2 against 1 -- in double, as numpy decides it -- but only 1 pair in 7 is
    // stored and needs the doubles themselves.  x1 = (a - 2^26) / 2^26 + b / 2^52 with a the 27 high bits: a float32
    // estimate from a and c alone is within 4e-7 of r2, so an estimate farther than 1e-5 from the boundary (and from 0)
    // decides exactly what the double comparison decides; the others (1 attempt in ~1e5) and the stored pairs take the
    // double path (two more words tempered, the arithmetic of legacy_gauss).
    bool ok = false, exact = false;
    double x1 = 0.0, x2 = 0.0, r2 = 1.0;
    unsigned int wa = 0u, wc = 0u;
    auto exact_pair = [&]() {
      const unsigned int b = mt_temper(nw[4 * tid + 1]) >> 6, d = mt_temper(nw[4 * tid + 3]) >> 6;
      const double u1 = ((double)wa * 67108864.0 + (double)b) / 9007199254740992.0;
      const double u2 = ((double)wc * 67108864.0 + (double)d) / 9007199254740992.0;
      x1 = 2.0 * u1 - 1.0;
      x2 = 2.0 * u2 - 1.0;
      r2 = x1 * x1 + x2 * x2;
      exact = true;
    };
    if (tid < 156) {
      wa = mt_temper(nw[4 * tid]) >> 5;
      wc = mt_temper(nw[4 * tid + 2]) >> 5;
      const float xf = (float)((int)wa - 67108864) * 1.4901161193847656e-08f;  // 2^-26
      const float yf = (float)((int)wc - 67108864) * 1.4901161193847656e-08f;
      const float rf = xf * xf + yf * yf;
      ok = rf < 1.0f;
      if (!(fabsf(rf - 1.0f) > 1e-5f && rf > 1e-5f)) {
        exact_pair();
        ok = !(r2 >= 1.0 || r2 == 0.0);
      }
    }
    if (queued) {
      // (s_qtail is only written after the barrier below, so every thread reads the same value here)
      const int pops = (s_qtail - q_popped) >> 6;  // whole groups of 64 pending records
      if (w == NW - 1) {
        for (int g = 0; g < pops; ++g) {
          const int i = (q_popped + 64 * g + lane) & (MTQ_CAP - 1);
          emit(q_r2[i], q_x1[i], q_x2[i], q_d0[i], q_d1[i]);
        }
      }
      q_popped += 64 * pops;
    }
    // position of this attempt's pair in the stream and whether any of its two normals is stored
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) s_cnt[it & 1][w] = __popcll(bal);
    __syncthreads();
    const int c0 = s_cnt[it & 1][0], c1 = s_cnt[it & 1][1], c2 = s_cnt[it & 1][2], c3 = s_cnt[it & 1][3];
    const int tot = (c0 + c1) + (c2 + c3);
    bool need = false;
    int d0 = -1, d1 = -1;
    // (pairs still wanted, as a 32-bit number: the 64-bit stream position stays in wave-uniform registers)
    const long long rem64 = need_pairs - done_pairs;
    const int rem = rem64 > 1024 ? 1024 : (int)rem64;
    // a block whose normals all fall between the stored columns of one row (uniform test) stores nothing: no positions --
    // six blocks in seven when 72 of 500 columns are kept, so everything that only the positions need stays inside the branch
    const bool none_stored = (col0 >= zs) && (col0 + 2 * (tot < rem ? tot : rem) <= Lg);
    if (!none_stored && ok) {
      const bool odd_total = (total & 1LL) != 0;
      const int before = __popcll(bal & ((1ull << lane) - 1ull));
      const int base = w == 0 ? 0 : (w == 1 ? c0 : (w == 2 ? c0 + c1 : (c0 + c1) + c2));
      const int k = base + before;  // k-th accepted pair of this block
      if (k < rem) {
        int col = col0 + 2 * k, row = row0;
        while (col >= Lg) {
          col -= Lg;
          ++row;
        }
        int row1 = row, col1 = col + 1;
        if (col1 == Lg) {
          col1 = 0;
          row1 = row + 1;
        }
        const bool last_odd = odd_total && (k == rem - 1) && rem64 <= 1024;  // the stream ends on the first normal of the pair
        if (col < zs) d0 = row * zc + col;
        if (!last_odd && col1 < zs) d1 = row1 * zc + col1;
        need = (d0 >= 0) || (d1 >= 0);
      }
    }
    if (need && !exact) exact_pair();
    if (!queued) {
      if (need) emit(r2, x1, x2, d0, d1);
    } else {
      const unsigned long long nb = __ballot(need);
      if (nb != 0ull) {  // (wave-uniform)
        const int nbefore = __popcll(nb & ((1ull << lane) - 1ull));
        int wbase = 0;
        if (lane == 0) wbase = atomicAdd(&s_qtail, __popcll(nb));
        wbase = __shfl(wbase, 0);
        if (need) {
          const int i = (wbase + nbefore) & (MTQ_CAP - 1);
          q_r2[i] = r2;
          q_x1[i] = x1;
          q_x2[i] = x2;
          q_d0[i] = d0;
          q_d1[i] = d1;
        }
      }
    }
    done_pairs += tot;
    col0 += 2 * tot;
    while (col0 >= Lg) {
      col0 -= Lg;
      ++row0;
    }
    cur ^= 1;
    ++it;
  }
  if (queued) {  // what is left in the queue (fewer than 64 + 156 records)
    __syncthreads();
    const int left = s_qtail - q_popped;
    for (int i = tid; i < left; i += blockDim.x) {
      const int j = (q_popped + i) & (MTQ_CAP - 1);
      emit(q_r2[j], q_x1[j], q_x2[j], q_d0[j], q_d1[j]);
    }
  }
}

// ---------------------------------------------------------------------------------------
// K5, one LONG stream in parallel (BASELINE config 3: RandomState(seed).standard_normal((4000, 2048)) is 21 M words of ONE
// MT19937 stream; a single edge of the bench: 1.3 M).  The stream is cut into chunks of MTJ_CB blocks of 624 words.
// The raw state words x_k obey a linear recurrence over GF(2) with characteristic polynomial phi (degree 19937); with
// g = x^J mod phi:  x_{k+J} = XOR_{i : g_i = 1} x_{k+i}  (tools/gen_mt_jump.py computes phi by Berlekamp-Massey and the
// tables g for J = 624 MTJ_CB 2^m, and checks them on the sequence).  So the state J words ahead of a known state is
// a binary convolution of 19937 + 623 consecutive words: 33 blocks of the recurrence (k_mtj_src) and ~10 k x 624 word
// XORs that parallelise trivially (k_mtj_conv).  Doubling (states of chunks c known for c = 0 mod 2^(m+1) -> chunks
// c + 2^m) reaches every chunk in log2(chunks) levels of two launches.  Then every chunk counts its accepted polar
// attempts (k_mtj_count), a scan turns the counts into stream positions (k_mtj_scan), and k_mt_normals<true> emits.
// The numbers are those of the sequential generator bit for bit: same words, same attempts, same arithmetic.
// (The low 31 bits of x_0 are not part of the generator's state; they reach only the low bits of the first word of a
// jumped state, which the recurrence never reads.)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mtj_init(EdgeDev* edges, const unsigned int* seeds, int add_iter, int iter_abs,
                                                  MtjWork wk) {
  const int e_idx = blockIdx.y, ahead = blockIdx.x;
  const EdgeDev E = edges[e_idx];
  const gpet_scalars* sc = E.sc;
  const size_t stream = (size_t)e_idx * gridDim.x + ahead;
  unsigned int* T = wk.T + stream * wk.nc * 624;
  for (size_t i = 624 + threadIdx.x; i < (size_t)wk.nc * 624; i += blockDim.x) T[i] = 0u;  // (the convolutions XOR into them)
  if (threadIdx.x == 0) {
    const int iter_idx = (iter_abs >= 0 ? iter_abs : sc->iter) + ahead;
    unsigned int p = seeds[e_idx] + (add_iter ? (unsigned int)(iter_idx + 1) : 0u);  // gpet.py:839
    T[0] = p;
    for (int i = 1; i < 624; ++i) {
      p = 1812433253u * (p ^ (p >> 30)) + (unsigned int)i;
      T[i] = p;
    }
  }
}

// level m, jump jj: the state of chunk c = jj 2^(m+1) and the 32 blocks after it -> src[jj] (20592 raw words)
__global__ void __launch_bounds__(256) k_mtj_src(MtjWork wk, int m) {
  const int jj = blockIdx.x;
  const size_t stream = blockIdx.y;
  const int c = jj << (m + 1);
  if (c + (1 << m) >= wk.nc) return;
  __shared__ unsigned int s_mt[2][624];
  const int tid = threadIdx.x;
  const unsigned int* st0 = wk.T + (stream * wk.nc + c) * 624;
  unsigned int* dst = wk.src + (stream * (wk.nc / 2 + 1) + jj) * MTJ_SRC;
  for (int i = tid; i < 624; i += 256) {
    const unsigned int v = st0[i];
    s_mt[0][i] = v;
    dst[i] = v;
  }
  __syncthreads();
  int cur = 0;
  for (int b = 1; b < 33; ++b) {
    unsigned int* o = s_mt[cur];
    unsigned int* nw = s_mt[cur ^ 1];
    if (tid < 227) nw[tid] = mt_mix(o[tid], o[tid + 1], o[tid + 397]);
    __syncthreads();
    if (tid < 227) nw[227 + tid] = mt_mix(o[227 + tid], o[228 + tid], nw[tid]);
    __syncthreads();
    if (tid < 169) nw[454 + tid] = mt_mix(o[454 + tid], o[455 + tid], nw[227 + tid]);
    if (tid == 255) nw[623] = mt_mix(o[623], nw[0], nw[396]);
    __syncthreads();
    for (int i = tid; i < 624; i += 256) dst[b * 624 + i] = nw[i];
    cur ^= 1;
  }
}

// level m, jump jj, part p: taps 640 p .. 640 p + 639 of g_m; XORs its share into the state of chunk c + 2^m
#define MTJ_TAPS 640
#define MTJ_PARTS 32  // 32 x 640 >= 19968
__global__ void __launch_bounds__(256) k_mtj_conv(MtjWork wk, int m, const unsigned int* __restrict__ poly) {
  const int p = blockIdx.x, jj = blockIdx.y;
  const size_t stream = blockIdx.z;
  const int c = jj << (m + 1), tgt = c + (1 << m);
  if (tgt >= wk.nc) return;
  __shared__ unsigned int s_w[MTJ_TAPS + 624];
  const int tid = threadIdx.x;
  const unsigned int* src = wk.src + (stream * (wk.nc / 2 + 1) + jj) * MTJ_SRC + MTJ_TAPS * p;
  const int avail = MTJ_SRC - MTJ_TAPS * p;  // (the last part's window ends with the source)
  for (int i = tid; i < MTJ_TAPS + 624; i += 256) s_w[i] = i < avail ? src[i] : 0u;
  __syncthreads();
  unsigned int a0 = 0u, a1 = 0u, a2 = 0u;
  const int j2 = tid + 512 < 624 ? tid + 512 : tid;  // (threads 112..255 recompute a word they already own: no branch)
  const unsigned int* g = poly + (size_t)m * 624 + (MTJ_TAPS / 32) * p;
  for (int w = 0; w < MTJ_TAPS / 32; ++w) {
    unsigned int bits = (MTJ_TAPS / 32) * p + w < 624 ? g[w] : 0u;  // (uniform)
    while (bits) {
      const int i = 32 * w + __builtin_ctz(bits);
      bits &= bits - 1;
      a0 ^= s_w[i + tid];
      a1 ^= s_w[i + tid + 256];
      a2 ^= s_w[i + j2];
    }
  }
  unsigned int* T = wk.T + (stream * wk.nc + tgt) * 624;
  atomicXor(&T[tid], a0);
  atomicXor(&T[tid + 256], a1);
  if (tid + 512 < 624) atomicXor(&T[tid + 512], a2);
}

// accepted polar attempts of every chunk (the decisions of k_mt_normals, nothing else of it)
__global__ void __launch_bounds__(256) k_mtj_count(MtjWork wk) {
#pragma clang fp contract(off)
  const int chunk = blockIdx.x;
  const size_t stream = blockIdx.y;
  __shared__ unsigned int s_mt[2][624];
  __shared__ int s_part[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned int* st0 = wk.T + (stream * wk.nc + chunk) * 624;
  for (int i = tid; i < 624; i += 256) s_mt[0][i] = st0[i];
  __syncthreads();
  int cur = 0, acc = 0;
  for (int it = 0; it < MTJ_CB; ++it) {
    unsigned int* o = s_mt[cur];
    unsigned int* nw = s_mt[cur ^ 1];
    if (tid < 227) nw[tid] = mt_mix(o[tid], o[tid + 1], o[tid + 397]);
    __syncthreads();
    if (tid < 227) nw[227 + tid] = mt_mix(o[227 + tid], o[228 + tid], nw[tid]);
    __syncthreads();
    if (tid < 169) nw[454 + tid] = mt_mix(o[454 + tid], o[455 + tid], nw[227 + tid]);
    if (tid == 255) nw[623] = mt_mix(o[623], nw[0], nw[396]);
    __syncthreads();
    bool ok = false;
    if (tid < 156) {
      const unsigned int wa = mt_temper(nw[4 * tid]) >> 5, wc = mt_temper(nw[4 * tid + 2]) >> 5;
      const float xf = (float)((int)wa - 67108864) * 1.4901161193847656e-08f;  // 2^-26
      const float yf = (float)((int)wc - 67108864) * 1.4901161193847656e-08f;
      const float rf = xf * xf + yf * yf;
      ok = rf < 1.0f;
      if (!(fabsf(rf - 1.0f) > 1e-5f && rf > 1e-5f)) {
        const unsigned int b = mt_temper(nw[4 * tid + 1]) >> 6, d = mt_temper(nw[4 * tid + 3]) >> 6;
        const double u1 = ((double)wa * 67108864.0 + (double)b) / 9007199254740992.0;
        const double u2 = ((double)wc * 67108864.0 + (double)d) / 9007199254740992.0;
        const double x1 = 2.0 * u1 - 1.0, x2 = 2.0 * u2 - 1.0;
        const double r2 = x1 * x1 + x2 * x2;
        ok = !(r2 >= 1.0 || r2 == 0.0);
      }
    }
    acc += __popcll(__ballot(ok));
    cur ^= 1;
  }
  if (lane == 0) s_part[w] = acc;
  __syncthreads();
  if (tid == 0) wk.cnt[stream * wk.nc + chunk] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// exclusive prefix of the counts: the stream position (in pairs) where every chunk starts
__global__ void __launch_bounds__(64) k_mtj_scan(MtjWork wk) {
  const size_t stream = blockIdx.x;
  const int lane = threadIdx.x;
  const int* cnt = wk.cnt + stream * wk.nc;
  long long* offs = wk.offs + stream * wk.nc;
  long long base = 0;
  for (int c0 = 0; c0 < wk.nc; c0 += 64) {
    const int c = c0 + lane;
    const long long v = c < wk.nc ? (long long)cnt[c] : 0;
    long long incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (c < wk.nc) offs[c] = base + incl - v;
    base += __shfl(incl, 63, 64);
  }
}

size_t mtj_work_bytes(int streams, int nc) {
  return (size_t)streams * ((size_t)nc * 624 * 4 + (size_t)nc * 4 + (size_t)nc * 8 + (size_t)(nc / 2 + 1) * MTJ_SRC * 4) + 1024;
}

// chunks for a stream of S x Lg normals: expected attempts + 6 sigma (the last chunk runs on if that was too few)
int mtj_chunks(long long normals) {
  const double pairs = 0.5 * (double)(normals + 1);
  const double attempts = pairs / 0.7853981633974483;
  const double blocks = (attempts + 6.0 * 0.523 * sqrt(attempts)) / 156.0 + 1.0;
  long long nc = (long long)ceil(blocks / (double)MTJ_CB);
  if (nc > (1LL << MTJ_LEVELS)) return 0;  // (longer than the jump tables reach: the sequential generator)
  return (int)nc;
}

hipError_t launch_normals_chunked(hipStream_t st, EdgeDev* d_edges, int B, const unsigned int* d_seeds, int add_iter,
                                  int iter_abs, int n_ahead, int z_store, void* work, int nc, const unsigned int* d_poly) {
  (void)hipGetLastError();
  const int streams = B * n_ahead;
  MtjWork wk;
  char* base = (char*)work;
  wk.nc = nc;
  wk.T = (unsigned int*)base;
  base += (size_t)streams * nc * 624 * 4;
  wk.offs = (long long*)base;
  base += (size_t)streams * nc * 8;
  wk.cnt = (int*)base;
  base += (((size_t)streams * nc * 4 + 15) / 16) * 16;
  wk.src = (unsigned int*)base;
  hipLaunchKernelGGL(k_mtj_init, dim3(n_ahead, B), dim3(256), 0, st, d_edges, d_seeds, add_iter, iter_abs, wk);
  int top = 0;
  while ((2 << top) < nc) ++top;  // jumps of 2^top chunks from chunk 0 first
  for (int m = top; m >= 0; --m) {
    const int jumps = (nc + (2 << m) - 1) / (2 << m);
    hipLaunchKernelGGL(k_mtj_src, dim3(jumps, streams), dim3(256), 0, st, wk, m);
    hipLaunchKernelGGL(k_mtj_conv, dim3(MTJ_PARTS, jumps, streams), dim3(256), 0, st, wk, m, d_poly);
  }
  hipLaunchKernelGGL(k_mtj_count, dim3(nc, streams), dim3(256), 0, st, wk);
  hipLaunchKernelGGL(k_mtj_scan, dim3(streams), dim3(64), 0, st, wk);
  hipLaunchKernelGGL((k_mt_normals<true, 4>), dim3(nc, n_ahead, B), dim3(256), 0, st, d_edges, d_seeds, add_iter, iter_abs, z_store, wk);
  return hipGetLastError();
}

const unsigned int* mtj_poly_host() { return &mtj_poly[0][0]; }
size_t mtj_poly_bytes() { return sizeof(mtj_poly); }

// K6 on the matrix cores: v_mfma_f64_16x16x4_f64.  64x64 output tile per workgroup, 4 waves, wave w
// owns rows 16w..16w+15 and all 64 columns (4 accumulators of 4 f64 per lane); K streamed through
// LDS in chunks of 32.  Operand maps (cdna_hip_programming.md section 3): A lane l <- A[l&15][l>>4],
// B lane l <- B[l>>4][l&15]; D register g of lane l -> row (l>>4) + 4g, column l&15.
typedef double v4f64 __attribute__((ext_vector_type(4)));
// The samples are f64 (the reference's; every parity tier is stated on them).  Opt-in (gpet_batch_set_sample_dtype, BASELINE
// config 2's "fp32 posterior samples"): the GEMM rounds each sample to f32 when it stores it and every consumer widens
// it again -- all arithmetic stays f64, the buffer and its HBM traffic halve.  y_ld / y_st: the sites that are not hot.
__device__ __forceinline__ double y_ld(const EdgeDev& E, size_t idx) {
  return E.y_f32 ? (double)reinterpret_cast<const float*>(E.Y)[idx] : E.Y[idx];
}
__device__ __forceinline__ void y_st(const EdgeDev& E, size_t idx, double v) {
  if (E.y_f32) reinterpret_cast<float*>(E.Y)[idx] = (float)v;
  else E.Y[idx] = v;
}
template <bool F32>
struct YT {
  typedef double type;
};
template <>
struct YT<true> {
  typedef float type;
};

__global__ void __launch_bounds__(256) k_sample_gemm_mfma(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.z];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int Lg = E.Lg, S = E.S, zc = E.z_cols;
  const int s0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  if (s0 >= S || j0 >= Lg) return;
  const int rows = sc->rank;
  const double* __restrict__ Zs = E.Z + (size_t)(sc->iter % E.z_ring) * ((size_t)S * zc);
  __shared__ double sz[64][33];  // [s][k]
  __shared__ double sa[32][65];  // [k][j]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  v4f64 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < rows; k0 += 32) {
    for (int e = tid; e < 64 * 32; e += 256) {
      const int kk = e & 31, ss = e >> 5;
      const int k = k0 + kk, sidx = s0 + ss;
      sz[ss][kk] = (k < rows && sidx < S) ? Zs[(size_t)sidx * zc + k] : 0.0;
    }
    for (int e = tid; e < 32 * 64; e += 256) {
      const int jj = e & 63, kk = e >> 6;
      const int k = k0 + kk, j = j0 + jj;
      sa[kk][jj] = (k < rows && j < Lg) ? E.A[(size_t)k * Lg + j] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; kk += 4) {
      const double a = sz[16 * w + li][kk + lq];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double b = sa[kk + lq][16 * t + li];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const double y_s = sc->y_s;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = j0 + 16 * t + li;
    if (j >= Lg) continue;
    const double mu = E.mean[j];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int sidx = s0 + 16 * w + lq + 4 * g;
      if (sidx < S) y_st(E, (size_t)sidx * Lg + j, (acc[t][g] + mu) * y_s);
    }
  }
}

// K6, rank <= 96 (the production case): each wave keeps its 16 rows of Z -- the whole K extent,
// 24 f64 per lane -- in registers for the entire sweep over the columns, so Z is read exactly
// once.  A workgroup is 8 waves = 128 sample rows; the K x 64 chunk of the factor for the
// current column tile sits in LDS (shared by the 8 waves) while the next chunk is already in
// flight from HBM/L2 into registers, so the matrix pipe does not wait for the staging.
// The MFMA loop only runs over the actual rank (rounded up to 4).
// The K extent is a template parameter (KS steps of 4, rank rounded up): a run-time bound inside the
// unrolled MFMA chain makes the compiler copy the accumulators around every step and drain the pipe.
// What bounds it: the 4.1 GB of samples it has to STORE per launch of 1 024 edges.  tools/ubench/gemm_pipe.hip rebuilds
// the loop piece by piece: matrix pipe + LDS operand + barriers + staging run at 68-70 TFLOP/s (1.13 ms); with the
// stores 1.56 ms, and the stores with 1/18 of the MFMAs still 1.51 ms (2.7 TB/s; 2.85 TB/s with 512-byte runs per
// instruction).  Measured slower or equal, and dropped: streaming (nontemporal) stores (+35 %), the chunk by LDS-DMA
// with four 4-wave workgroups per CU (+8 %), a half-tile phase offset between the workgroups of a CU (+-0).
#define GEMM_KMAX 96
// row stride of the factor chunk in LDS (doubles): the operand read of a matrix instruction takes 16 consecutive columns of
// FOUR rows (k = 4 q + lq); with 80 (= 32 dwords mod 64) the rows of each half-wave fall on disjoint bank halves -- 65 put
// rows 2 banks apart and every read was a 2-way conflict
#ifndef GEMM_LDA
#define GEMM_LDA 80
#endif
#define GEMM_LDS_MAX (150 * 1024)
// A pointer the compiler knows to be GLOBAL memory.  The buffers of an edge are pointers read from its EdgeDev record, which the
// compiler can only treat as generic: every access becomes a FLAT instruction, and a FLAT instruction counts in BOTH wait
// counters (it might be an LDS access), so a wait for an LDS operand (lgkmcnt) can end up waiting for global stores issued
// before it.  Tried on the sample GEMM as the explanation of its store cost: global_store / global_load instead of flat_*
// take 0-5 % (1.92 -> 1.82-1.91 ms per 1 024 edges) -- kept, but not the explanation.
#define GPET_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ GPET_GLOBAL T* as_global(T* p) {
  return (GPET_GLOBAL T*)p;
}
template <typename T>
__device__ __forceinline__ const GPET_GLOBAL T* as_global(const T* p) {
  return (const GPET_GLOBAL T*)p;
}
// MU_LDS (the posterior mean behind the chunk in LDS) is a COMPILE-TIME switch: as a run-time one the mean had two producers --
// an LDS read and a global load into the same registers -- and at their join the compiler waits for BOTH counters: every
// store of a column group then waited (vmcnt(0)) for all the stores before it.
template <int KS, bool F32, bool MU_LDS>
__device__ __forceinline__ void sample_gemm_body(const EdgeDev& E, const gpet_scalars* sc, double* s_fa, int part, int cpart, int ncs) {
  constexpr bool mu_lds = MU_LDS;
  typedef typename YT<F32>::type yt;
  GPET_GLOBAL yt* __restrict__ Yo = as_global(reinterpret_cast<yt*>(E.Y));
  const GPET_GLOBAL double* __restrict__ Ag = as_global(E.A);
  const GPET_GLOBAL double* __restrict__ meang = as_global(E.mean);
  constexpr int PF = (KS * 4 * 64) / 512;  // prefetch registers per thread (KS even)
  const int Lg = E.Lg, S = E.S, zc = E.z_cols;
  const int s0 = part * 128;
  // the 64-column tiles [jlo, jhi) of this workgroup: all of them, or one of ncs runs of tiles when few edges leave the
  // GPU empty (a single edge: 8 row blocks x 8 column runs instead of 8 workgroups sweeping 500 columns each)
  const int tpc = ((Lg + 63) / 64 + ncs - 1) / ncs;
  const int jlo = cpart * tpc * 64;
  const int jhi = (jlo + tpc * 64 < Lg) ? (jlo + tpc * 64) : Lg;
  if (jlo >= Lg) return;
  const int rows = sc->rank;
  const GPET_GLOBAL double* __restrict__ Zs = as_global(E.Z) + (size_t)(sc->iter % E.z_ring) * ((size_t)S * zc);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int srow = s0 + 16 * w + li;
  double areg[KS];
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const int k = 4 * q + lq;
    areg[q] = (k < rows && srow < S) ? Zs[(size_t)srow * zc + k] : 0.0;
  }
  const double y_s = sc->y_s;
  // the posterior mean sits in LDS behind the chunk: a global load in the epilogue would make every 16-column group
  // wait (vmcnt counts in order) for ALL the stores issued before it
  double* s_mu = s_fa + 4 * KS * GEMM_LDA;
  if (mu_lds)
    for (int j = tid; j < Lg; j += 512) s_mu[j] = meang[j];
  double pf[PF];
  // element e = tid + 512 * u of the [4 KS][64] chunk: row kk = e >> 6, column jj = e & 63 (zero beyond the rank)
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int e = tid + 512 * u;
    const int kk = e >> 6, j = jlo + (e & 63);
    pf[u] = (kk < rows && j < Lg) ? Ag[(size_t)kk * Lg + j] : 0.0;
  }
  for (int j0 = jlo; j0 < jhi; j0 += 64) {
    __syncthreads();  // previous tile's LDS reads are done
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int e = tid + 512 * u;
      s_fa[(e >> 6) * GEMM_LDA + (e & 63)] = pf[u];
    }
    __syncthreads();
    if (j0 + 64 < jhi) {  // next tile's chunk: loads stay in flight during the MFMAs below
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int e = tid + 512 * u;
        const int kk = e >> 6, j = j0 + 64 + (e & 63);
        pf[u] = (kk < rows && j < Lg) ? Ag[(size_t)kk * Lg + j] : 0.0;
      }
    }
    // one 16-column group at a time: its 4 stores go out while the matrix pipe works on the next group (and the
    // accumulators take 8 registers instead of 32)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (j0 + 16 * t >= Lg) continue;  // (an empty group: uniform over the workgroup)
      v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < KS; ++q)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[q], s_fa[(4 * q + lq) * GEMM_LDA + li + 16 * t], acc, 0, 0, 0);
      const int j = j0 + 16 * t + li;
      if (j >= Lg) continue;
      const double mu = mu_lds ? s_mu[j] : meang[j];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int sidx = s0 + 16 * w + lq + 4 * g;
        // (streaming stores, __builtin_nontemporal_store, were measured 35 % slower here)
        if (sidx < S) Yo[(size_t)sidx * Lg + j] = (yt)((acc[g] + mu) * y_s);
      }
    }
  }
}

// One kernel per K extent (the launcher picks it from the batch's largest possible rank; an edge of smaller
// rank multiplies a few zero rows): register allocation is per kernel, and the variants up to K = 72 fit the
// 128 VGPRs that let two workgroups share a CU, so one's staging and stores overlap the other's MFMAs.
template <int KS, bool F32, bool MU_LDS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) k_sample_gemm_mfma_r(EdgeDev* edges, int ncs) {
  int edge, part;  // the row blocks of an edge on one XCD: its factor comes out of HBM once, not once per row block
  xcd_edge_part((int)gridDim.x, edge, part);
  const EdgeDev E = edges[edge];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int rparts = (int)gridDim.x / ncs, rp = part % rparts, cp = part / rparts;
  if (rp * 128 >= E.S) return;
  extern __shared__ double s_fa[];  // [4 KS][GEMM_LDA]
  sample_gemm_body<KS, F32, MU_LDS>(E, sc, s_fa, rp, cp, ncs);
}
template <int KS, bool F32, bool MU_LDS>
__global__ void __launch_bounds__(512) k_sample_gemm_mfma_rl(EdgeDev* edges, int ncs) {  // (K > 72: one workgroup per CU)
  int edge, part;
  xcd_edge_part((int)gridDim.x, edge, part);
  const EdgeDev E = edges[edge];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int rparts = (int)gridDim.x / ncs, rp = part % rparts, cp = part / rparts;
  if (rp * 128 >= E.S) return;
  extern __shared__ double s_fa[];
  sample_gemm_body<KS, F32, MU_LDS>(E, sc, s_fa, rp, cp, ncs);
}

// cov = (amp*rho(x*,x*) - V^T V) * y_std^2 on the matrix cores (same tiling as k_sample_gemm_mfma):
// both operands are 32-row chunks of V; only tiles on or above the diagonal are computed and mirrored.
// final_mode: covariance of the converged fit (GaussianProcessRegressor.predict(return_cov=True) on the standardised
// grid, sklearn_gpr.py:398-403) from the V that k_predict<.., FINAL> left behind.
__global__ void __launch_bounds__(256) k_cov_mfma(EdgeDev* edges, int final_mode) {
  const EdgeDev E = edges[blockIdx.z];
  const gpet_scalars* sc = E.sc;
  if (!final_mode && ((sc->done && !sc->force) || sc->status != GPET_OK)) return;
  const int bx = blockIdx.x, by = blockIdx.y;
  if (bx < by) return;
  const int Lg = E.Lg;
  const int r0 = by * 64, c0 = bx * 64;
  if (r0 >= Lg || c0 >= Lg) return;
  const int n = sc->n;
  __shared__ double sr[32][65];  // [i][row j]
  __shared__ double sc_[32][65]; // [i][col j']
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  v4f64 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
  for (int i0 = 0; i0 < n; i0 += 32) {
    for (int e = tid; e < 32 * 64; e += 256) {
      const int jj = e & 63, ii = e >> 6;
      const int i = i0 + ii;
      sr[ii][jj] = (i < n && r0 + jj < Lg) ? E.V[(size_t)i * Lg + r0 + jj] : 0.0;
      sc_[ii][jj] = (i < n && c0 + jj < Lg) ? E.V[(size_t)i * Lg + c0 + jj] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; kk += 4) {
      const double a = sr[kk + lq][16 * w + li];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double b = sc_[kk + lq][16 * t + li];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const double amp = sc->amp, s2 = sc->y_std * sc->y_std;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = c0 + 16 * t + li;
    if (c >= Lg) continue;
    const double xoff = final_mode ? E.fin_par[3] : 0.0, xsc = final_mode ? E.fin_par[4] : 1.0;
    const double length = final_mode ? E.fin_par[1] : E.length_scale;
    const double xc = (((double)(E.x_st + c) - xoff) / xsc) / length;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int r = r0 + 16 * w + lq + 4 * g;
      if (r >= Lg) continue;
      if (bx == by && c < r) continue;  // diagonal tile: the mirror write below covers it
      const double k = (r == c) ? amp
                              : amp * ((!final_mode && E.nu_code == 3) ? corr_px(E, (double)r, (double)c, length)
                                                                        : corr_fn(E, (((double)(E.x_st + r) - xoff) / xsc) / length, xc));
      const double val = (k - acc[t][g]) * s2;
      E.cov[(size_t)r * Lg + c] = val;
      E.cov[(size_t)c * Lg + r] = val;
    }
  }
}

// ---------------------------------------------------------------------------------------
// a7  cost of every sampled curve (gpet.py:391-408): one wave per curve, Simpson pairs
//     spread over the lanes.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double grad_at(const float* __restrict__ grad, int M, int N, double y, int x) {
  // RectBivariateSpline(kx=ky=1)(y, x) with integer x: clamped linear interpolation along y
  y = y < 0.0 ? 0.0 : (y > (double)(M - 1) ? (double)(M - 1) : y);
  int iy = (int)floor(y);
  if (iy > M - 2) iy = M - 2;
  if (iy < 0) iy = 0;
  const double w1 = y - (double)iy, w0 = ((double)iy + 1.0) - y;
  const int iy1 = (iy + 1 < M) ? iy + 1 : M - 1;
  return (double)grad[(size_t)iy * N + x] * w0 + (double)grad[(size_t)iy1 * N + x] * w1;
}

// Even number of Simpson samples (odd edge length): scipy >= 1.11 adds Cartwright's correction for the
// last interval, alpha*y[-1] + beta*y[-2] - eta*y[-3] with the last two spacings h0, h1
// (scipy/integrate/_quadrature.py simpson; version dependent in the reference).
template <typename T>
__device__ __forceinline__ void simpson_tail(const EdgeDev& E, const T* __restrict__ row, double& al, double& li) {
  const int Lg = E.Lg;
  if ((Lg & 1) == 0 || Lg < 5) return;
  const int k = Lg - 2;  // index of the last sample (N-1); samples are points 0..Lg-2
  const double ya = (double)row[k - 2], yb = (double)row[k - 1], yc = (double)row[k], yd = (double)row[k + 1];
  const double da = yb - ya, db = yc - yb, dc = yd - yc;
  const double l3 = sqrt(1.0 + da * da), l2 = sqrt(1.0 + db * db), l1 = sqrt(1.0 + dc * dc);  // l_{N-3}, l_{N-2}, l_{N-1}
  // arc length: unit spacing h0 = h1 = 1
  al += (5.0 / 12.0) * l1 + (2.0 / 3.0) * l2 - (1.0 / 12.0) * l3;
  // line integral: abscissae are the cumulative lengths -> h0 = l_{N-2}, h1 = l_{N-1}
  const double h0 = l2, h1 = l1;
  const double alpha = (2.0 * h1 * h1 + 3.0 * h0 * h1) / (6.0 * (h1 + h0));
  const double beta = (h1 * h1 + 3.0 * h0 * h1) / (6.0 * h0);
  const double eta = (h1 * h1 * h1) / (6.0 * h0 * (h0 + h1));
  const double g1 = grad_at(E.grad, E.M, E.N, yc, E.x_st + k) + 1e-3;
  const double g2 = grad_at(E.grad, E.M, E.N, yb, E.x_st + k - 1) + 1e-3;
  const double g3 = grad_at(E.grad, E.M, E.N, ya, E.x_st + k - 2) + 1e-3;
  li += alpha * g1 + beta * g2 - eta * g3;
}

// Lane i of a 64-pair chunk owns curve points 2i and 2i+1 (one 16-byte load, fully coalesced);
// the third point of its Simpson pair, that point's gradient value and the following segment
// length are the next lane's own values and arrive by wave shuffle -- each sample, each gather and
// each square root is done exactly once per curve.  (Lane 63 fetches its successor's data itself.)
typedef double v2f64 __attribute__((ext_vector_type(2)));
template <bool F32>
__global__ void __launch_bounds__(256) k_score(EdgeDev* edges) {
  typedef typename YT<F32>::type yt;
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= E.S) return;
  const int Lg = E.Lg;
  const yt* __restrict__ row = reinterpret_cast<const yt*>(E.Y) + (size_t)s * Lg;
  const bool aligned = !F32 && ((((size_t)row) & 15) == 0);
  const int npair = (Lg - 2) / 2;  // Simpson over Lg-1 samples (odd count)
  double al = 0.0, li = 0.0;
  for (int i0 = 0; i0 < npair; i0 += WAVE) {
    const int i = i0 + lane;
    const int k = 2 * i;
    // own points: valid while k + 1 < Lg (the lane after the last pair still supplies its data)
    double y0 = 0.0, y1 = 0.0;
    if (k + 1 < Lg) {
      if (aligned) {
        const v2f64 v = *reinterpret_cast<const v2f64*>(row + k);
        y0 = v[0];
        y1 = v[1];
      } else {
        y0 = (double)row[k];
        y1 = (double)row[k + 1];
      }
    }
    // segment length l = sqrt(1 + d^2) and its reciprocal from ONE rsqrt each (the Simpson weights
    // need h1/h0, h0/h1 and hsum^2/(h0 h1): products of the reciprocals instead of three divisions)
    const double d0 = y1 - y0;
    const double q0 = 1.0 + d0 * d0;
    const double r0 = rsqrt(q0), l0 = q0 * r0;
    const double g0 = (k + 1 < Lg) ? grad_at(E.grad, E.M, E.N, y0, E.x_st + k) + 1e-3 : 0.0;
    const double g1 = (k + 1 < Lg) ? grad_at(E.grad, E.M, E.N, y1, E.x_st + k + 1) + 1e-3 : 0.0;
    double y2 = __shfl_down(y0, 1, WAVE), l2 = __shfl_down(l0, 1, WAVE), r2 = __shfl_down(r0, 1, WAVE);
    double g2 = __shfl_down(g0, 1, WAVE);
    if (lane == 63 && i < npair) {  // successor lives in the next chunk
      y2 = (double)row[k + 2];
      const double y3 = (double)row[k + 3];
      const double d2 = y3 - y2;
      const double q2 = 1.0 + d2 * d2;
      r2 = rsqrt(q2);
      l2 = q2 * r2;
      g2 = grad_at(E.grad, E.M, E.N, y2, E.x_st + k + 2) + 1e-3;
    }
    if (i < npair) {
      const double d1 = y2 - y1;
      const double q1 = 1.0 + d1 * d1;
      const double r1 = rsqrt(q1), l1 = q1 * r1;
      al += (2.0 / 6.0) * (l0 + 4.0 * l1 + l2);
      const double h0 = l1, h1 = l2, ih0 = r1, ih1 = r2;
      const double hsum = h0 + h1;
      li += hsum * (1.0 / 6.0) *
            (g0 * (2.0 - h1 * ih0) + g1 * (hsum * hsum * (ih0 * ih1)) + g2 * (2.0 - h0 * ih1));
    }
  }
  al = wave_sum(al);
  li = wave_sum(li);
  if (lane == 0) {
    simpson_tail(E, row, al, li);
    E.costs[s] = al / li;
  }
}

// a7, tiled variant (M <= 1100): a workgroup owns 15 Simpson pairs (32 image columns with the next pair's first) of up to 1024
// curves.  The 32 x M slab of the gradient image is staged in LDS (column-major, odd stride), so the
// 4-tap bilinear gathers -- the L1/TA-bound part of the wave-per-curve kernel -- become LDS reads.
// 16 consecutive lanes share a curve: one coalesced 256-byte read of its samples, successor data
// by shuffle inside the group, group reduction by shuffle, and one (arc, integral) partial per
// (tile, curve); k_score_combine adds the partials in tile order (deterministic) and divides.
#define SC_PAIRS 15  // Simpson pairs per tile: lanes 0..14 of a 16-lane row; lane 15 only supplies the next pair's data
#define SC_CURVES 1024
#define SC_THREADS 1024
// cross-lane moves of a double inside rows of 16 lanes by DPP (two v_mov_b32_dpp; __shfl_xor / __shfl_down with width 16
// go through ds_bpermute and recompute the lane index every time): CTRL = row_ror:8 / row_ror:4 / quad_perm for the
// butterfly, row_shl:1 for "the lane above" (lane 15 of a row has no source and reads 0: its callers replace the value)
template <int CTRL>
__device__ __forceinline__ double dpp_row(double v) {
  const long long b = __double_as_longlong(v);
  // (no "old" operand: a lane without a source -- lane 15 of a row under row_shl:1 -- reads 0 and its callers do not use
  //  the value; with old = the source the compiler copies every operand into the destination first: 24 moves per pair)
  const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sum over the 16 lanes of a row, in every lane; the operands of every addition are those of the xor-8, 4, 2, 1 butterfly
// (after the first step lanes i and i ^ 8 hold the same value, so a rotation by 4 brings what xor 4 would).
// (The exchanges through the LDS crossbar -- ds_swizzle instead of two v_mov_b32_dpp per step -- were measured for every
// subset of the four steps: 1.21-1.22 against 1.20 ms per 1 024 edges; DPP stays.)
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_row<0x128>(v);  // row_ror:8
  v += dpp_row<0x124>(v);  // row_ror:4
  v += dpp_row<0x4E>(v);   // quad_perm:[2,3,0,1]
  v += dpp_row<0xB1>(v);   // quad_perm:[1,0,3,2]
  return v;
}
__device__ __forceinline__ double grad_lds(const float* __restrict__ col, int M, double y) {
#pragma clang fp contract(off)
  y = y < 0.0 ? 0.0 : (y > (double)(M - 1) ? (double)(M - 1) : y);
  // (0 <= y <= M - 1 from here on, M >= 2: the row below is min(floor(y), M - 2) >= 0 and the row above is always inside
  //  the column -- no further clamps; the kernel is bound by vector-ALU issue and this runs 2-3 times per curve pair)
  int iy = (int)y;  // = floor(y) for y >= 0
  iy = iy > M - 2 ? M - 2 : iy;
  const double w1 = y - (double)iy, w0 = ((double)iy + 1.0) - y;
  return fma((double)col[iy + 1], w1, (double)col[iy] * w0);
}

// One Simpson pair of G curves, as lane `pl` of each curve's row of sixteen lanes sees it: own points (y0, y1) at the
// slab columns col0 / col1; the third point, its gradient value and the following segment come from the lane above by
// DPP; (arc length, line integral) summed over the row.  ONE body for k_score_tile (samples from memory, G = 1) and
// k_sample_score (samples from the accumulators, G = 4), every fused multiply-add spelled out and contraction off: the
// two kernels give the same bits for the same samples.  No branches: the G curves are independent chains in one basic
// block, which the scheduler interleaves (k_sample_score runs two or three waves per SIMD and has nothing else to hide
// the reciprocal square roots and the LDS gathers behind).
// 1 / sqrt(x) for x >= 1 (a segment's 1 + d^2): the arithmetic of the device library's rsqrt -- hardware estimate, then
// y + y e (0.5 + 0.375 e) with e = 1 - x y^2 -- without its test for zero / infinite arguments (a v_cmp_class and two
// selects per call in a kernel bound by vector-ALU issue); the same bits for every finite x > 0.
__device__ __forceinline__ double sc_rsqrt(double x) {
#pragma clang fp contract(off)
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-y0 * x, y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}
template <int G>
__device__ __forceinline__ void score_pairs_row(const double (&y0)[G], const double (&y1)[G], const float* __restrict__ col0,
                                                const float* __restrict__ col1, int M, bool on, double (&al)[G], double (&li)[G]) {
#pragma clang fp contract(off)
  double l0[G], r0[G], g0[G], g1[G];
  // the 4 G taps of the bilinear lookups first (grad_lds, split: all the LDS reads are out before anything needs them),
  // the segment lengths under their latency, then the interpolation
  double yc0[G], yc1[G];
  float ta0[G], tb0[G], ta1[G], tb1[G];
  int iy0[G], iy1[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    // (v_max_f64 / v_min_f64 instead of two compares and four selects per clamp)
    yc0[g] = fmin(fmax(y0[g], 0.0), (double)(M - 1));
    yc1[g] = fmin(fmax(y1[g], 0.0), (double)(M - 1));
    iy0[g] = (int)yc0[g];
    iy1[g] = (int)yc1[g];
    iy0[g] = iy0[g] > M - 2 ? M - 2 : iy0[g];
    iy1[g] = iy1[g] > M - 2 ? M - 2 : iy1[g];
    ta0[g] = col0[iy0[g]];
    tb0[g] = col0[iy0[g] + 1];
    ta1[g] = col1[iy1[g]];
    tb1[g] = col1[iy1[g] + 1];
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const double d0 = y1[g] - y0[g];
    const double q0 = fma(d0, d0, 1.0);
    r0[g] = sc_rsqrt(q0);
    l0[g] = q0 * r0[g];
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {  // (= grad_lds(col, M, y) + 1e-3)
    const double w01 = yc0[g] - (double)iy0[g], w00 = ((double)iy0[g] + 1.0) - yc0[g];
    const double w11 = yc1[g] - (double)iy1[g], w10 = ((double)iy1[g] + 1.0) - yc1[g];
    g0[g] = fma((double)tb0[g], w01, (double)ta0[g] * w00) + 1e-3;
    g1[g] = fma((double)tb1[g], w11, (double)ta1[g] * w10) + 1e-3;
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const double y2 = dpp_row<0x101>(y0[g]), l2 = dpp_row<0x101>(l0[g]), r2 = dpp_row<0x101>(r0[g]);  // row_shl:1: the pair above
    const double g2 = dpp_row<0x101>(g0[g]);
    const double d1 = y2 - y1[g];
    const double q1 = fma(d1, d1, 1.0);
    const double r1 = sc_rsqrt(q1), l1 = q1 * r1;
    const double a_ = (2.0 / 6.0) * (fma(4.0, l1, l0[g]) + l2);
    const double h0 = l1, h1 = l2, ih0 = r1, ih1 = r2;
    const double hsum = h0 + h1;
    const double t0 = g0[g] * fma(-h1, ih0, 2.0), t1 = g1[g] * ((hsum * hsum) * (ih0 * ih1)), t2 = g2 * fma(-h0, ih1, 2.0);
    const double l_ = (hsum * (1.0 / 6.0)) * ((t0 + t1) + t2);
    al[g] = on ? a_ : 0.0;
    li[g] = on ? l_ : 0.0;
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    al[g] = row16_sum(al[g]);
    li[g] = row16_sum(li[g]);
  }
}

template <bool F32>
__global__ void __launch_bounds__(SC_THREADS) k_score_tile(EdgeDev* edges, int curves_per_wg) {
  typedef typename YT<F32>::type yt;
  // the tiles of an edge on one XCD: neighbouring tiles split cache lines of the sample rows (a tile's 256-byte runs
  // start on 32-byte boundaries), which then come out of HBM once instead of once per L2
  int edge, part;
  xcd_edge_part((int)(gridDim.x * gridDim.y), edge, part);
  const int bx = part % (int)gridDim.x, byy = part / (int)gridDim.x;
  const EdgeDev E = edges[edge];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  extern __shared__ float s_img[];  // [2 * SC_PAIRS + 2][ldm]
  const int M = E.M, N = E.N, Lg = E.Lg, S = E.S;
  const int npair = (Lg - 2) / 2;
  const int p0 = bx * SC_PAIRS;
  if (p0 >= npair) return;
  const int c0 = E.x_st + 2 * p0;   // first image column of the slab
  const int ncol = 2 * SC_PAIRS + 2;
  const int ldm = M | 1;
  const int tid = threadIdx.x;
  for (int e = tid; e < ncol * M; e += SC_THREADS) {
    const int y = e / ncol, c = e - y * ncol;
    const int x = c0 + c;
    s_img[c * ldm + y] = (x < N) ? as_global(E.grad)[(size_t)y * N + x] : 0.f;
  }
  __syncthreads();
  const int pl = tid & 15;  // pair within the tile (15: the first pair of the next tile, as a source of data only)
  const int i = p0 + pl;
  const int k = 2 * i;
  // (curves_per_wg: SC_CURVES, or fewer when few edges leave the GPU empty -- a multiple of the 64 curves of a pass)
  const int s_lo = byy * curves_per_wg;
  const int s_hi = (s_lo + curves_per_wg < S) ? (s_lo + curves_per_wg) : S;
  if (s_lo >= S) return;
  GPET_GLOBAL double* __restrict__ cpart = as_global(E.cost_part) + ((size_t)bx * S) * 2;
  // the samples of the NEXT group of curves are requested before this group is worked on: the loop is bound by the
  // latency of these loads (8 waves per SIMD do not cover an HBM round trip per 600 cycles of work on their own)
  // (Round 2 had 16 pairs per tile and let lane 15 fetch and evaluate its successor itself: a divergent path of ~40
  //  instructions that every wave executed for one lane in sixteen -- a fifth of the loop.  A sixteenth lane that computes
  //  the next pair's first point like everybody else costs 1/15 more lanes and no extra path.)
  auto fetch = [&](int s0, double& a0, double& a1) {
    const int s = s0 + (tid >> 4);
    const GPET_GLOBAL yt* __restrict__ row = as_global(reinterpret_cast<const yt*>(E.Y)) + (size_t)(s < s_hi ? s : s_lo) * Lg;
    a0 = a1 = 0.0;
    if (k + 1 < Lg) {
      a0 = (double)row[k];
      a1 = (double)row[k + 1];
    }
  };
  double y0, y1;
  fetch(s_lo, y0, y1);
  for (int s0 = s_lo; s0 < s_hi; s0 += SC_THREADS / 16) {
    const int s = s0 + (tid >> 4);
    const bool live = s < s_hi;
    double n0 = 0.0, n1 = 0.0;
    if (s0 + SC_THREADS / 16 < s_hi) fetch(s0 + SC_THREADS / 16, n0, n1);
    const double ya[1] = {y0}, yb[1] = {y1};
    double al[1], li[1];
    score_pairs_row<1>(ya, yb, s_img + (2 * pl) * ldm, s_img + (2 * pl + 1) * ldm, M, pl < SC_PAIRS && i < npair, al, li);
    if (pl == 0 && live) {
      cpart[2 * s] = al[0];
      cpart[2 * s + 1] = li[0];
    }
    y0 = n0;
    y1 = n1;
  }
}

__global__ void __launch_bounds__(256) k_score_combine(EdgeDev* edges, int n_tiles) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= E.S) return;
  (void)n_tiles;
  const int my_tiles = ((E.Lg - 2) / 2 + SC_PAIRS - 1) / SC_PAIRS;  // this edge's own tile count
  double al = 0.0, li = 0.0;
  for (int t = 0; t < my_tiles; ++t) {
    al += E.cost_part[((size_t)t * E.S + s) * 2];
    li += E.cost_part[((size_t)t * E.S + s) * 2 + 1];
  }
  if (E.y_f32) simpson_tail(E, reinterpret_cast<const float*>(E.Y) + (size_t)s * E.Lg, al, li);
  else simpson_tail(E, E.Y + (size_t)s * E.Lg, al, li);
  E.costs[s] = al / li;
}

// a6 + a7 in one kernel (the device loop; sklearn_gpr.py:440-473 + gpet.py:391-408): the sample matrix of an iteration is
// written once (8 S Lg bytes per edge, what bounds k_sample_gemm_mfma_r) and read once (k_score_tile) although only the
// n_keep best curves are ever looked at again.  Here the workgroup of a scorer tile (15 Simpson pairs = 32 grid columns,
// their image slab in LDS) forms the samples of its columns itself on the matrix cores and scores them out of the
// accumulators:
//   * B operands: the 2 x KS factor entries of a lane's two columns (even / odd point of pair `pl`) stay in registers for
//     the whole kernel; A operands: 16 rows of Z (the normals) per wave and group, KS loads per lane, the next group's
//     requested as soon as the last MFMA has read this group's (they arrive under the scoring);
//   * two accumulator tiles (even points, odd points): register g of lane (pl, lq) is curve s0 + lq + 4 g at the two
//     points of pair pl -- exactly the (y0, y1) of k_score_tile's lane, with the sixteen pairs of a curve in one DPP row;
//     the MFMA chain runs over k in the order of k_sample_gemm_mfma_r, so a sample has the same bits in both kernels
//     and the scoring arithmetic below is k_score_tile's: costs, best_idx and traces are identical to the unfused path.
// The kept curves are then formed once more by k_sample_keep_rows (n_keep rows instead of S) where the KDE and the pixel
// kernels look for them.  Needs an even grid length (an odd one has Simpson's tail correction, which reads sample rows).
#ifndef GPET_SS_WAVES
#define GPET_SS_WAVES 2  // waves per SIMD the register allocation aims at (experiments: 3 with 384-thread workgroups)
#endif
template <int KS, bool F32>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(GPET_SS_WAVES, 3))) k_sample_score(EdgeDev* edges) {
  int edge, part;
  xcd_edge_part((int)(gridDim.x * gridDim.y), edge, part);
  const int bx = part % (int)gridDim.x, byy = part / (int)gridDim.x;
  const EdgeDev E = edges[edge];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  extern __shared__ float s_img[];  // [2 * SC_PAIRS + 2][ldm]
  const int M = E.M, N = E.N, Lg = E.Lg, S = E.S, zc = E.z_cols;
  const int npair = (Lg - 2) / 2;
  const int p0 = bx * SC_PAIRS;
  if (p0 >= npair) return;
  const int c0 = E.x_st + 2 * p0;  // first image column of the slab
  const int ncol = 2 * SC_PAIRS + 2;
  const int ldm = M | 1;
  const int tid = threadIdx.x, nthr = blockDim.x;
  {  // the slab, eight loads per thread in flight (four waves stage what sixteen stage in k_score_tile)
    const GPET_GLOBAL float* __restrict__ gimg = as_global(E.grad);
    const int tot = ncol * M;
    for (int e0 = tid; e0 < tot; e0 += 8 * nthr) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * nthr;
        const int y = e / ncol, c = e - y * ncol;
        v[u] = (e < tot && c0 + c < N) ? gimg[(size_t)y * N + c0 + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * nthr;
        const int y = e / ncol, c = e - y * ncol;
        if (e < tot) s_img[c * ldm + y] = v[u];
      }
    }
  }
  const int lane = tid & 63, w = tid >> 6, nw = nthr >> 6;
  const int pl = lane & 15, lq = lane >> 4;
  const int i = p0 + pl;
  const int k = 2 * i;
  const bool valid = k + 1 < Lg;  // (the lane after the last pair still supplies its data)
  const int rows = sc->rank;
  const GPET_GLOBAL double* __restrict__ Ag = as_global(E.A);
  const GPET_GLOBAL double* __restrict__ Zs = as_global(E.Z) + (size_t)(sc->iter % E.z_ring) * ((size_t)S * zc);
  double bE[KS], bO[KS];
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const int kk = 4 * q + lq;
    const bool in = kk < rows && valid;
    bE[q] = in ? Ag[(size_t)kk * Lg + k] : 0.0;
    bO[q] = in ? Ag[(size_t)kk * Lg + k + 1] : 0.0;
  }
  const double mu0 = valid ? as_global(E.mean)[k] : 0.0, mu1 = valid ? as_global(E.mean)[k + 1] : 0.0;
  const double y_s = sc->y_s;
  // curves of this workgroup: gridDim.y equal parts, rounded up to whole groups of 16
  const int per = (((S + (int)gridDim.y - 1) / (int)gridDim.y) + 15) & ~15;
  const int s_lo = byy * per;
  const int s_hi = (s_lo + per < S) ? (s_lo + per) : S;
  if (s_lo >= S) return;  // (uniform over the workgroup)
  GPET_GLOBAL double* __restrict__ cpart = as_global(E.cost_part) + ((size_t)bx * S) * 2;
  const float* col0 = s_img + (2 * pl) * ldm;
  const float* col1 = col0 + ldm;
  const bool on = pl < SC_PAIRS && i < npair;
  double a[KS];
  int s0 = s_lo + 16 * w;
  // (rows of Z: every load is issued, at a clamped address where the entry does not exist, and its value used as it is --
  //  eighteen predicated loads are eighteen branches, and a select on the loaded value makes every load wait for itself.
  //  Beyond the rank the factor entries bE / bO are exact zeros and a clamped address holds a finite normal, so the
  //  product is zero; a row beyond the workgroup's curves repeats row s_lo and its scores are never stored.)
  const int kmax = rows > 0 ? rows - 1 : 0;
  auto load_a = [&](int sbase) {
    const int srow = sbase + pl;
    const GPET_GLOBAL double* __restrict__ zr = Zs + (size_t)(srow < s_hi ? srow : s_lo) * zc;
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int kk = 4 * q + lq;
      a[q] = zr[kk < kmax ? kk : kmax];
    }
  };
  load_a(s0);
#ifdef GPET_SS_PROF  // cycles per phase of a group, as wave 0 of workgroup 0 sees them
  long long pf[5] = {0, 0, 0, 0, 0};
  const long long t_in = clock64();
#define SS_STAMP(i) { const long long t_ = clock64(); pf[i] += t_ - tl; tl = t_; }
#else
#define SS_STAMP(i)
#endif
  __syncthreads();  // the slab is staged
#ifdef GPET_SS_PROF
  long long tl = clock64();
  const long long t_loop = tl;
  int ngroups = 0;
#endif
  for (; s0 < s_hi; s0 += 16 * nw) {
    v4f64 accE = (v4f64){0.0, 0.0, 0.0, 0.0}, accO = (v4f64){0.0, 0.0, 0.0, 0.0};
#ifdef GPET_SS_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SS_STAMP(0)
    ++ngroups;
#endif
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      accE = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bE[q], accE, 0, 0, 0);
      accO = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bO[q], accO, 0, 0, 0);
    }
#ifdef GPET_SS_PROF
    asm volatile("s_nop 0" ::"v"(accE[0]), "v"(accO[0]));
    SS_STAMP(1)
#endif
    if (s0 + 16 * nw < s_hi) load_a(s0 + 16 * nw);  // the next group's rows of Z (wave-uniform condition)
    SS_STAMP(2)
    double ya[4], yb[4], al[4], li[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // (a lane beyond the grid has zero factor entries and a zero mean: its points are 0, as in k_score_tile)
      ya[g] = (accE[g] + mu0) * y_s;
      yb[g] = (accO[g] + mu1) * y_s;
      if (F32) {  // (gpet_batch_set_sample_dtype: what the f32 store of the GEMM and the widening load of the scorer give)
        ya[g] = (double)(float)ya[g];
        yb[g] = (double)(float)yb[g];
      }
    }
    score_pairs_row<4>(ya, yb, col0, col1, M, on, al, li);
#ifdef GPET_SS_PROF
    asm volatile("s_nop 0" ::"v"(al[3]), "v"(li[3]));
    SS_STAMP(3)
#endif
    if (pl == 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int s = s0 + lq + 4 * g;
        if (s < s_hi) {
          cpart[2 * s] = al[g];
          cpart[2 * s + 1] = li[g];
        }
      }
    }
    SS_STAMP(4)
  }
#ifdef GPET_SS_PROF
  if (blockIdx.x + blockIdx.y + blockIdx.z == 0 && tid == 0 && ngroups > 0)
    printf("ss prof (%d threads, %d groups): staging+B %lld | per group: wait Z %lld, mfma %lld, issue loads %lld, scoring %lld, stores %lld | loop %lld cycles\n",
           nthr, ngroups, (long long)(t_loop - t_in), pf[0] / ngroups, pf[1] / ngroups, pf[2] / ngroups, pf[3] / ngroups, pf[4] / ngroups,
           (long long)(clock64() - t_loop));
#endif
#undef SS_STAMP
}

// a6 + a7 in one kernel, the OTHER way round (round 4): k_sample_score above is stationary in the COLUMN tile -- its workgroup
// keeps one image slab and streams the normals of all curves through it, so the 17 tiles of an edge each pull the edge's
// 0.58 MB of normals through their CU again, with 64 different rows per load instruction: 2 000 cycles to issue and 1 000 to
// wait for 18 loads per group of 16 curves, 3.2 ms per 1 024 edges against 3.05 for the separate kernels.  Here the workgroup
// is stationary in the CURVES, like the sample GEMM: 16 waves x 16 rows of Z in registers for the whole kernel (read once),
// and it sweeps the column tiles: per tile the 32 factor columns (even / odd points de-interleaved: the B operands of the
// two accumulator tiles, conflict-free 128-byte rows) and the 32 x M image slab are staged in LDS -- fully coalesced
// 128-byte rows of the image, which all edges of a batch share and which therefore lives in L2.  Four workgroups per edge
// stage the whole image once each (4 MB per edge through L2 -> LDS) where the separate kernels write and re-read 8 MB of
// samples in HBM.  The MFMA chain, the epilogue and the scoring arithmetic are those of k_sample_score / k_score_tile
// (score_pairs_row), the per-tile partials go where k_score_combine expects them: costs, best_idx, kept rows and traces
// are identical to the unfused path bit for bit.
// Measured (1 024 edges of the bench shape, tools/time_fused_score.py): 3.05 ms against 3.15-3.17 for the column-tile form
// and 2.98-3.03 for the separate GEMM (1.79) + scorer (1.20) -- plus 0.38 ms for the kept rows in both fused forms.  The
// floor of ANY fused form is the matrix instructions (0.94 ms at peak) plus the scoring's vector work (1.1 ms): an f64
// matrix instruction holds the vector unit's FMA lanes, the two do not overlap.  What this kernel adds to that
// floor is the staging of slab and factor tile between two barriers per tile (16 image loads per thread in four dependent
// batches).  Staging tile t + 1 under tile t's matrix instructions into a second pair of buffers was built as well
// (161 KB of LDS: two 31-column slabs, two factor tiles): with 1 024 threads the staged values do not fit the 128
// registers four waves per SIMD leave (the rows of Z spill and are reloaded per tile: 4.6 ms), with 768 threads (three
// waves per SIMD, no spill, six workgroups per edge) 3.69 ms -- both slower than this plain form, which is the one kept.
template <int KS, bool F32>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) k_sample_score2(EdgeDev* edges) {
  int edge, part;  // the curve blocks of an edge on one XCD: its factor and its slabs come out of HBM / the other L2s once
  xcd_edge_part((int)gridDim.x, edge, part);
  const EdgeDev E = edges[edge];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int M = E.M, N = E.N, Lg = E.Lg, S = E.S, zc = E.z_cols;
  const int s_blk = part * 256;
  if (s_blk >= S) return;
  extern __shared__ float s_img2[];  // [32][ldm] slab, then the factor tile and the tile's posterior mean
  constexpr int NCOL = 2 * SC_PAIRS + 2;  // 32
  const int ldm = M | 1;
  double* s_fe = reinterpret_cast<double*>(s_img2 + (((size_t)NCOL * ldm + 3) & ~(size_t)3));  // [4 KS][16] even points
  double* s_fo = s_fe + 4 * KS * 16;                                                                // [4 KS][16] odd points
  double* s_mu = s_fo + 4 * KS * 16;                                                                // [32]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int pl = lane & 15, lq = lane >> 4;
  const int npair = (Lg - 2) / 2;
  const int n_tiles = (npair + SC_PAIRS - 1) / SC_PAIRS;
  const int rows = sc->rank;
  const GPET_GLOBAL double* __restrict__ Ag = as_global(E.A);
  const GPET_GLOBAL double* __restrict__ Zs = as_global(E.Z) + (size_t)(sc->iter % E.z_ring) * ((size_t)S * zc);
  const GPET_GLOBAL float* __restrict__ gimg = as_global(E.grad);
  const GPET_GLOBAL double* __restrict__ meang = as_global(E.mean);
  const double y_s = sc->y_s;
  // the wave's 16 rows of Z: A operand lane (pl, lq) <- Z[row pl][4 q + lq]
  const int s0 = s_blk + 16 * w;
  double a[KS];
  {
    const int srow = s0 + pl;
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int kk = 4 * q + lq;
      a[q] = (kk < rows && srow < S) ? Zs[(size_t)srow * zc + kk] : 0.0;
    }
  }
  for (int bx = 0; bx < n_tiles; ++bx) {
    const int p0 = bx * SC_PAIRS;
    const int c0 = E.x_st + 2 * p0;  // first image column of the slab
    __syncthreads();                 // (the previous tile's operands and slab are no longer read)
    for (int e0 = tid; e0 < NCOL * M; e0 += 4 * 1024) {  // the slab: four 128-byte image rows per wave instruction
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + 1024 * u;
        const int y = e >> 5, c = e & 31;
        v[u] = (e < NCOL * M && c0 + c < N) ? gimg[(size_t)y * N + c0 + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + 1024 * u;
        if (e < NCOL * M) s_img2[(e & 31) * ldm + (e >> 5)] = v[u];
      }
    }
    for (int e = tid; e < 4 * KS * NCOL; e += 1024) {  // the tile's 32 factor columns, zero beyond rank and grid
      const int kk = e >> 5, c = e & 31;
      const int j = 2 * p0 + c;
      const double v = (kk < rows && j < Lg) ? Ag[(size_t)kk * Lg + j] : 0.0;
      ((c & 1) ? s_fo : s_fe)[kk * 16 + (c >> 1)] = v;
    }
    if (tid < NCOL) s_mu[tid] = (2 * p0 + tid < Lg) ? meang[2 * p0 + tid] : 0.0;
    __syncthreads();
    v4f64 accE = (v4f64){0.0, 0.0, 0.0, 0.0}, accO = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      accE = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], s_fe[(4 * q + lq) * 16 + pl], accE, 0, 0, 0);
      accO = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], s_fo[(4 * q + lq) * 16 + pl], accO, 0, 0, 0);
    }
    const double mu0 = s_mu[2 * pl], mu1 = s_mu[2 * pl + 1];
    const float* col0 = s_img2 + (2 * pl) * ldm;
    const float* col1 = col0 + ldm;
    const bool on = pl < SC_PAIRS && (p0 + pl) < npair;
    GPET_GLOBAL double* __restrict__ cpart = as_global(E.cost_part) + ((size_t)bx * S) * 2;
    // two curves of the lane's four at a time (the scoring of four needs ~150 registers; two fit the 128 that let four
    // waves share a SIMD): register g of lane (pl, lq) = curve s0 + lq + 4 g at the two points of pair p0 + pl
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double ya[2], yb[2], al[2], li[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        ya[g] = (accE[2 * h + g] + mu0) * y_s;
        yb[g] = (accO[2 * h + g] + mu1) * y_s;
        if (F32) {
          ya[g] = (double)(float)ya[g];
          yb[g] = (double)(float)yb[g];
        }
      }
      score_pairs_row<2>(ya, yb, col0, col1, M, on, al, li);
      if (pl == 0) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int sidx = s0 + lq + 4 * (2 * h + g);
          if (sidx < S) {
            cpart[2 * sidx] = al[g];
            cpart[2 * sidx + 1] = li[g];
          }
        }
      }
    }
  }
}

// The n_keep best curves of an iteration whose samples were scored out of the accumulators (k_sample_score): rows
// best_idx[0 .. n_keep) of the sample matrix, formed by the MFMA chain of the sample GEMM (same bits) and stored where the
// KDE and the pixel kernels read them.  One wave per (group of 16 kept rows, 16 columns).
template <bool F32>
__global__ void __launch_bounds__(256) k_sample_keep_rows(EdgeDev* edges) {
  typedef typename YT<F32>::type yt;
  constexpr int KS = 18;  // (the fused kernel serves ranks up to 72; steps beyond the rank multiply zeros)
  int edge, rg;  // (the row groups of an edge on one XCD: its factor comes out of HBM once, not once per L2)
  xcd_edge_part((int)gridDim.x, edge, rg);
  const EdgeDev E = edges[edge];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int Lg = E.Lg, S = E.S, zc = E.z_cols, n_keep = E.n_keep;
  if (16 * rg >= n_keep) return;
  const int rows = sc->rank;
  const GPET_GLOBAL double* __restrict__ Ag = as_global(E.A);
  const GPET_GLOBAL double* __restrict__ Zs = as_global(E.Z) + (size_t)(sc->iter % E.z_ring) * ((size_t)S * zc);
  const GPET_GLOBAL int* __restrict__ bidx = as_global(E.best_idx);
  GPET_GLOBAL yt* __restrict__ Yo = as_global(reinterpret_cast<yt*>(E.Y));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  const int nct = (Lg + 15) >> 4;
  const double y_s = sc->y_s;
  const int bi = 16 * rg + li;
  int arow = bi < n_keep ? bidx[bi] : -1;
  if (arow >= S) arow = -1;
  double a[KS];
  {
    const GPET_GLOBAL double* __restrict__ zr = Zs + (size_t)(arow >= 0 ? arow : 0) * zc;
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int kk = 4 * q + lq;
      const bool in = arow >= 0 && kk < rows;
      const double v = zr[in ? kk : 0];
      a[q] = in ? v : 0.0;
    }
  }
  int orow[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int bo = 16 * rg + lq + 4 * g;
    orow[g] = bo < n_keep ? bidx[bo] : -1;
    if (orow[g] >= S) orow[g] = -1;
  }
  // two buffers of factor entries (and the tile's posterior mean): the loads of the tile after next are in flight while
  // this one multiplies and stores
  const int kmax = rows > 0 ? rows - 1 : 0;
  auto load_b = [&](double (&b)[KS], double& mu, int ct) {
    // (clamped addresses, values used as loaded: beyond the rank the normals a[] are exact zeros, beyond the grid
    //  nothing is stored -- a select on the loaded value would make every load wait for itself)
    const int j = 16 * ct + li;
    const GPET_GLOBAL double* __restrict__ ac = Ag + (j < Lg ? j : 0);
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int kk = 4 * q + lq;
      b[q] = ac[(size_t)(kk < kmax ? kk : kmax) * Lg];
    }
    mu = as_global(E.mean)[j < Lg ? j : 0];
  };
  auto tile = [&](const double (&b)[KS], double mu, int ct) {
    v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < KS; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
    const int j = 16 * ct + li;
    if (j < Lg) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (orow[g] >= 0) Yo[(size_t)orow[g] * Lg + j] = (yt)((acc[g] + mu) * y_s);
    }
  };
  double b0[KS], b1[KS], mu0 = 0.0, mu1 = 0.0;
  if (w < nct) load_b(b0, mu0, w);
  for (int ct = w; ct < nct; ct += 8) {  // (wave-uniform)
    if (ct + 4 < nct) load_b(b1, mu1, ct + 4);
    tile(b0, mu0, ct);
    if (ct + 8 < nct) load_b(b0, mu0, ct + 8);
    if (ct + 4 < nct) tile(b1, mu1, ct + 4);
  }
}

// argsort(costs)[:n_keep] for S <= 1024 by a bitonic sort of (cost, index) in LDS: 55 compare-exchange steps of 512
// pairs instead of S^2 comparisons (rank counting, below: 0.20 ms per 1 024 edges).  Equal costs keep index order (the
// index is the second key); -0.0 and +0.0 compare equal, as `<` on doubles has it.
__global__ void __launch_bounds__(512) k_topk_sort(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  __shared__ double s_k[1024];
  __shared__ int s_i[1024];
  const int tid = threadIdx.x, S = E.S;
  for (int e = tid; e < 1024; e += 512) {
    s_k[e] = (e < S) ? E.costs[e] : INFINITY;  // (the padding sorts behind every finite cost, and behind +inf by index)
    s_i[e] = e;
  }
  __syncthreads();
  for (int k = 2; k <= 1024; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int i = ((tid & ~(j - 1)) << 1) | (tid & (j - 1)), l = i | j;
      const double a = s_k[i], b = s_k[l];
      const int ia = s_i[i], ib = s_i[l];
      const bool a_first = (a < b) || (a == b && ia < ib);
      const bool up = (i & k) == 0;
      if (a_first != up) {
        s_k[i] = b;
        s_k[l] = a;
        s_i[i] = ib;
        s_i[l] = ia;
      }
      __syncthreads();
    }
  for (int b = tid; b < E.n_keep; b += 512) {
    E.best_idx[b] = s_i[b];
    E.best_costs[b] = s_k[b];
  }
}

// argsort(costs)[:n_keep] by rank counting (ties -> lower index first)       gpet.py:443
__global__ void __launch_bounds__(256) k_topk(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  __shared__ __attribute__((aligned(16))) double s_c[1024];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double ci = (i < E.S) ? E.costs[i] : 0.0;
  int rank = 0;
  for (int j0 = 0; j0 < E.S; j0 += 1024) {
    __syncthreads();
    // (entries beyond S are +inf: they never rank before a finite cost, and a tie on +inf needs a smaller index)
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) s_c[e] = (j0 + e < E.S) ? E.costs[j0 + e] : INFINITY;
    __syncthreads();
    const int lim = (E.S - j0) < 1024 ? (E.S - j0) : 1024;
    const int lim2 = (lim + 1) & ~1;  // two costs per (broadcast) LDS read
#pragma unroll 4
    for (int e = 0; e < lim2; e += 2) {
      const double2 cj = *reinterpret_cast<const double2*>(&s_c[e]);
      rank += (cj.x < ci) || (cj.x == ci && (j0 + e) < i);
      rank += (cj.y < ci) || (cj.y == ci && (j0 + e + 1) < i && (j0 + e + 1) < E.S);
    }
  }
  if (i < E.S && rank < E.n_keep) {
    E.best_idx[rank] = i;
    E.best_costs[rank] = ci;
  }
}


// ---------------------------------------------------------------------------------------
// f1  KDE of the best curves + pixel scoring / binning / argmax        gpet.py:455-662
//     KDEpy.FFTKDE restated: linear binning on the unit grid x=-1..N, y=-1..M, 9x9 Gaussian
//     (bw=1, support 4), crop, float32 min-max.  (PARITY UNPINNED: KDEpy is not available.)
//     Grid layout is x-major like KDEpy's: cell (gx, gy) at gx*(M+2)+gy.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_kde_clear(EdgeDev* edges, int mode) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if (mode == 0 && ((sc->done && !sc->force) || sc->status != GPET_OK)) return;
  const size_t cells = (size_t)(E.N + 2) * (E.M + 2);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < cells; i += (size_t)gridDim.x * blockDim.x)
    E.bins[i] = 0.0;
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      E.mm[0] = 0xFFFFFFFFu;
      E.mm[1] = 0u;
    }
    for (int i = threadIdx.x; i < E.n_bins; i += blockDim.x) {
      E.binbest[i] = 0ull;
      E.binarg[i] = 0x7FFFFFFFFFFFFFFFll;
    }
    for (int i = threadIdx.x; i < E.N; i += blockDim.x) E.colsum[i] = 0.0;
  }
}

// gradient KDE (ctor, gpet.py:505-509): points = pixels with grad > 1e-3, weight = grad
__global__ void __launch_bounds__(256) k_kde_bin_gradient(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const int x = blockIdx.x;
  if (x >= E.N) return;
  __shared__ double s_red[16];
  double part = 0.0;
  for (int y = threadIdx.x; y < E.M; y += blockDim.x) {
    const double g = (double)E.grad[(size_t)y * E.N + x];
    if (g > 1e-3) {
      E.bins[(size_t)(x + 1) * (E.M + 2) + (y + 1)] = g;
      part += g;
    }
  }
  part = block_sum(part, s_red);
  if (threadIdx.x == 0) E.colsum[x] = part;
}

__constant__ double c_gauss9[9] = {3.3546262790251185e-04, 1.1108996538242306e-02, 1.3533528323661270e-01,
                                   6.0653065971263342e-01, 1.0,
                                   6.0653065971263342e-01, 1.3533528323661270e-01, 1.1108996538242306e-02,
                                   3.3546262790251185e-04};  // exp(-t^2/2), t=-4..4

// total weight W = sum of the column sums in index order, once per edge (every workgroup of the vertical pass used to
// walk the N column sums on its thread 0 before its 256 outputs: 1.4 ms per 2048 x 2048 image)
__global__ void __launch_bounds__(64) k_kde_wsum(EdgeDev* edges, int mode) {
  const EdgeDev E = edges[blockIdx.x];
  const gpet_scalars* sc = E.sc;
  if (mode == 0 && ((sc->done && !sc->force) || sc->status != GPET_OK)) return;
  if (threadIdx.x != 0) return;
  double w = 0.0;
#pragma unroll 8
  for (int i = 0; i < E.N; ++i) w += E.colsum[i];
  E.kde_wsum[0] = w;
}

// vertical pass (along y, contiguous): tmp = (bins / W) (*) g
__global__ void __launch_bounds__(256) k_kde_conv_y(EdgeDev* edges, int mode) {
  const EdgeDev E = edges[blockIdx.z];
  const gpet_scalars* sc = E.sc;
  if (mode == 0 && ((sc->done && !sc->force) || sc->status != GPET_OK)) return;
  const int H = E.M + 2, Wd = E.N + 2;
  const int gy = blockIdx.x * blockDim.x + threadIdx.x;
  const int gx = blockIdx.y;
  if (gy >= H || gx >= Wd) return;
  const double s_w = E.kde_wsum[0];
  const double* col = E.bins + (size_t)gx * H;
  double acc = 0.0;
#pragma unroll
  for (int t = -4; t <= 4; ++t) {
    const int yy = gy + t;
    if (yy >= 0 && yy < H) acc += (col[yy] / s_w) * c_gauss9[t + 4];
  }
  E.tmpk[(size_t)gx * H + gy] = acc;
}

// horizontal pass + crop + transpose to (M, N) float32 + min/max.  A workgroup owns a tile of 32 image rows x 32 image
// columns: the 40 grid columns it needs are read along y (the contiguous direction of the x-major grid) into LDS, the
// outputs are written along x (the contiguous direction of the image) -- one thread per image row writing one float per
// column made every store its own cache line: 1.5 ms per 2048 x 2048 image.  Same nine products in the same order.
#define KCX_T 32
__global__ void __launch_bounds__(256) k_kde_conv_x(EdgeDev* edges, int mode) {
  const EdgeDev E = edges[blockIdx.z];
  const gpet_scalars* sc = E.sc;
  if (mode == 0 && ((sc->done && !sc->force) || sc->status != GPET_OK)) return;
  const int H = E.M + 2, Wd = E.N + 2;
  const int y0 = blockIdx.x * KCX_T, x0 = blockIdx.y * KCX_T;  // first image row / column of the tile
  __shared__ double s_t[KCX_T + 8][KCX_T + 1];  // [grid column x0 + c - 3][image row]
  const int tid = threadIdx.x;
  for (int e = tid; e < (KCX_T + 8) * KCX_T; e += 256) {
    const int c = e / KCX_T, r = e - c * KCX_T;
    const int xx = x0 + 1 + c - 4, gy = y0 + r + 1;  // grid coordinates (image pixel (x, y) sits at grid (x + 1, y + 1))
    s_t[c][r] = (xx >= 0 && xx < Wd && gy < H) ? E.tmpk[(size_t)xx * H + gy] : 0.0;
  }
  __syncthreads();
  float* dst = (mode == 0) ? E.kde : (float*)E.grad_kde;
  unsigned int kmin = 0xFFFFFFFFu, kmax = 0u;
  const int xl = tid & (KCX_T - 1);
  for (int yl = tid / KCX_T; yl < KCX_T; yl += 256 / KCX_T) {
    const int x = x0 + xl, y = y0 + yl;
    if (x < E.N && y < E.M) {
      double acc = 0.0;
#pragma unroll
      for (int t = -4; t <= 4; ++t) {
        const int xx = x + 1 + t;
        if (xx >= 0 && xx < Wd) acc += s_t[xl + t + 4][yl] * c_gauss9[t + 4];
      }
      acc *= 0.15915494309189535;  // 1 / (2 pi): Gaussian pdf normalisation in 2-D
      const float v = (float)acc;
      dst[(size_t)y * E.N + x] = v;
      const unsigned int key = f32_order_key(v);
      kmin = min(kmin, key);
      kmax = max(kmax, key);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (unsigned int)__shfl_xor((int)kmin, o, WAVE));
    kmax = max(kmax, (unsigned int)__shfl_xor((int)kmax, o, WAVE));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&E.mm[0], kmin);
    atomicMax(&E.mm[1], kmax);
  }
}

__global__ void __launch_bounds__(256) k_kde_normalise(EdgeDev* edges, int mode) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if (mode == 0 && ((sc->done && !sc->force) || sc->status != GPET_OK)) return;
  const float mn = f32_from_key(E.mm[0]);
  const float span = f32_from_key(E.mm[1]) - mn;
  float* a = (mode == 0) ? E.kde : (float*)E.grad_kde;
  const size_t px = (size_t)E.M * E.N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < px; i += (size_t)gridDim.x * blockDim.x)
    a[i] = (a[i] - mn) / span;
}

// ---- fused curve KDE (per-iteration path) ----------------------------------------------
// k_kde_prep: total kept weight W (KDEpy normalises the weights by their sum), points removed
// for lying outside the image (gpet.py:498-500), and the per-iteration resets.
#define KDE_PREP_MAXB 1024
__global__ void __launch_bounds__(1024) k_kde_prep(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  __shared__ double s_red[16];
  __shared__ double s_inv;
  __shared__ double s_ic[KDE_PREP_MAXB];   // 1 / cost of the kept curves, then their weights
  __shared__ int s_row[KDE_PREP_MAXB];     // their sample rows
  const int tid = threadIdx.x, nk = E.n_keep;
  const bool staged = nk <= KDE_PREP_MAXB;
  // the reciprocals in parallel, their sum by one thread in index order (the order of the sequential loop this
  // replaces, which paid one global round trip and one division per curve on a single thread)
  if (staged)
    for (int b = tid; b < nk; b += blockDim.x) {
      s_ic[b] = 1.0 / E.best_costs[b];
      s_row[b] = E.best_idx[b];
    }
  for (int i = tid; i < E.n_bins; i += blockDim.x) {
    E.binbest[i] = 0ull;
    E.binarg[i] = 0x7FFFFFFFFFFFFFFFll;
  }
  __syncthreads();
  if (tid == 0) {
    double inv_sum = 0.0;
    if (staged)
      for (int b = 0; b < nk; ++b) inv_sum += s_ic[b];
    else
      for (int b = 0; b < nk; ++b) inv_sum += 1.0 / E.best_costs[b];
    s_inv = inv_sum;
    E.mm[0] = 0xFFFFFFFFu;
    E.mm[1] = 0u;
  }
  __syncthreads();
  const double inv_sum = s_inv, ymax = (double)(E.M - 1);
  if (staged) {
    for (int b = tid; b < nk; b += blockDim.x) s_ic[b] = s_ic[b] / inv_sum;  // weight of curve b
    __syncthreads();
  }
  double wsum = 0.0;
  int removed = 0;
  if (staged) {
    // wave w takes curves w, w + 16, ...; its lanes walk the columns: coalesced rows, no division per point
    const int lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    for (int b = wv; b < nk; b += nw) {
      const size_t row = (size_t)s_row[b] * E.Lg;
      const double wb = s_ic[b];
#pragma unroll 4
      for (int k = lane; k < E.Lg; k += 64) {
        const double y = y_ld(E, row + k);
        if (y < 0.0 || y > ymax) ++removed; else wsum += wb;
      }
    }
  } else {
    const int total = nk * E.Lg;
#pragma unroll 4
    for (int e = tid; e < total; e += blockDim.x) {
      const int b = e / E.Lg, k = e - b * E.Lg;
      const double y = y_ld(E, (size_t)E.best_idx[b] * E.Lg + k);
      const double wb = (1.0 / E.best_costs[b]) / inv_sum;
      if (y < 0.0 || y > ymax) ++removed; else wsum += wb;
    }
  }
  wsum = block_sum(wsum, s_red);
  const double rem = block_sum((double)removed, s_red);
  if (tid == 0) {
    E.colsum[0] = wsum;     // W
    E.colsum[1] = inv_sum;  // sum of 1/cost over the kept curves
    sc->n_removed = (int)rem;
  }
}

#define KDE_TX 16
#define KDE_H 128    // image rows per LDS row-chunk
#define KDE_NB 128   // curves staged per pass
#define KDE_THREADS 512

// One workgroup per (16-column tile, edge).  The tile's curve points are staged in LDS once; the
// rows that can receive weight are the band [ymin-4, ymax+5] of those points, and only that band is
// processed, in chunks of KDE_H rows through ONE (KDE_TX+8) x (KDE_H+8) f64 LDS tile (39 KB per
// workgroup with the staging -> 4 workgroups per CU):
//   linear binning   one (curve, column) point per thread, 64-bit fixed-point LDS atomics (order-independent
//                    sums; KDEpy itself convolves by FFT, so no summation order is "the" reference);
//   vertical 9 taps  in place, register sliding window (halo rows saved before the barrier);
//   horizontal 9 taps + crop + f32 cast + min/max, straight to HBM.
// Rows outside the band are written as zeros.  No global binning grid, no boundary tests: rows and
// columns outside the padded grid never receive weight.
// (Round 4: clearing, conversion and the vertical pass per wave on its own three columns -- four barriers per chunk instead
//  of eight, the vertical pass on 63 lanes of every wave -- was built and is bit-identical: 0.85-0.88 against 0.89-0.90 ms
//  per 1 024 edges on a batch alone, 15.2 against 14.6 ms per step inside the bench; binning per wave as well: 0.90-0.92.
//  The kernel waits for its binning atomics and the gathers of the staged points, not for its barriers.  Dropped.)
__global__ void __launch_bounds__(KDE_THREADS) k_kde_fused(EdgeDev* edges, int raw_band) {
  int edge, tile;  // the column tiles of an edge on one XCD: neighbours share 8 of their 24 staged columns of every curve
  xcd_edge_part((int)gridDim.x, edge, tile);
  const EdgeDev E = edges[edge];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  extern __shared__ double s_a[];
  __shared__ int s_band[2];
  __shared__ int s_brow[KDE_NB];
  const int M = E.M, N = E.N;
  const int x0 = tile * KDE_TX;
  if (x0 >= N) return;
  const int NC = KDE_TX + 8;
  const int ld = (KDE_H + 8) | 1;
  double* s_y = s_a + NC * ld;        // [KDE_NB][NC] staged points (-1: none)
  double* s_wt = s_y + KDE_NB * NC;   // [KDE_NB] staged weights
  const int tid = threadIdx.x;
  const bool band = (x0 + KDE_TX + 4 > E.x_st) && (x0 - 4 <= E.x_en);
  float* __restrict__ out = E.kde;
  unsigned int kmin = 0xFFFFFFFFu, kmax = 0u;
  int y_lo = M, y_hi = -1;  // band of image rows with possibly non-zero density
  if (band) {
    const double W = E.colsum[0], inv_sum = E.colsum[1], ymax = (double)(M - 1);
    double g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = c_gauss9[t];
    const bool single = (E.n_keep <= KDE_NB);  // all kept curves fit one staging pass (the usual case)
    // fixed-point scale of the binning: a bin holds at most one column's weight, <= 1 / W, so
    // 2^(62 + floor(log2 W)) keeps every sum below 2^62 with an lsb of ~2^-70 (contributions are ~2^-17)
    const int kexp = 62 + ilogb(W > 0.0 ? W : 1.0);
    const double fscale = ldexp(1.0, kexp), finv = ldexp(1.0, -kexp);
    // stage curves [b0, b0+nb) of this tile's columns; returns the rows they touch through (lo, hi)
    auto stage = [&](int b0, int nb, int& lo, int& hi) {
      // (the sample rows of the curves of this pass through LDS first: the loads of the points then depend on nothing
      //  and are all in flight together, instead of one index load + one dependent sample load per point)
      for (int e = tid; e < nb; e += KDE_THREADS) s_brow[e] = E.best_idx[b0 + e];
      __syncthreads();
#pragma unroll 6
      for (int e = tid; e < nb * NC; e += KDE_THREADS) {
        const int bb = e / NC, c = e - bb * NC;
        const int xc = x0 + c - 4;
        double y = -1.0;
        if (xc >= E.x_st && xc <= E.x_en) y = y_ld(E, (size_t)s_brow[bb] * E.Lg + (xc - E.x_st));
        if (y < 0.0 || y > ymax) y = -1.0;  // gpet.py:498-500
        s_y[e] = y;
        if (y >= 0.0) {
          const int iy = (int)floor(y);
          lo = min(lo, iy);
          hi = max(hi, iy + 1);
        }
      }
      for (int e = tid; e < nb; e += KDE_THREADS) s_wt[e] = ((1.0 / E.best_costs[b0 + e]) / inv_sum) / W;
    };
    // phase A: band of rows that receive weight
    if (tid == 0) {
      s_band[0] = 0x7FFFFFFF;
      s_band[1] = -1;
    }
    __syncthreads();
    {
      int mylo = 0x7FFFFFFF, myhi = -1;
      for (int b0 = 0; b0 < E.n_keep; b0 += KDE_NB) {
        const int nb = (E.n_keep - b0) < KDE_NB ? (E.n_keep - b0) : KDE_NB;
        if (b0 > 0) __syncthreads();
        stage(b0, nb, mylo, myhi);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        mylo = min(mylo, __shfl_xor(mylo, o, WAVE));
        myhi = max(myhi, __shfl_xor(myhi, o, WAVE));
      }
      if ((tid & 63) == 0) {
        atomicMin(&s_band[0], mylo);
        atomicMax(&s_band[1], myhi);
      }
    }
    __syncthreads();
    if (s_band[1] >= 0) {
      y_lo = max(0, s_band[0] - 4);
      y_hi = min(M - 1, s_band[1] + 4);
    }
    // phase B: the band, KDE_H image rows at a time
    for (int r0 = y_lo; r0 <= y_hi; r0 += KDE_H) {
      const int nrow = (y_hi + 1 - r0) < KDE_H ? (y_hi + 1 - r0) : KDE_H;
      // LDS row l <-> padded-grid row gy = r0 - 3 + l  (image row y sits at l = y - r0 + 4)
      for (int i = tid; i < NC * ld; i += KDE_THREADS) s_a[i] = 0.0;
      __syncthreads();
      for (int b0 = 0; b0 < E.n_keep; b0 += KDE_NB) {
        const int nb = (E.n_keep - b0) < KDE_NB ? (E.n_keep - b0) : KDE_NB;
        if (!single) {
          int dl = 0, dh = 0;
          __syncthreads();
          stage(b0, nb, dl, dh);
          __syncthreads();
        }
        // linear binning, one (curve, column) point per thread: 64-bit fixed-point LDS atomics, so the sums do
        // not depend on the order the points arrive in (deterministic), at the resolution of f64 arithmetic
        unsigned long long* s_bits = reinterpret_cast<unsigned long long*>(s_a);
        const int lmax = nrow + 8;
        for (int e = tid; e < nb * NC; e += KDE_THREADS) {
          const double y = s_y[e];
          if (y < 0.0) continue;
          const int bb = e / NC, bc = e - bb * NC;
          const double gy = y + 1.0;
          const int iy = (int)floor(gy);
          const int l = iy - (r0 - 3);
          if (l + 1 < 0 || l >= lmax) continue;
          const double w = s_wt[bb] * fscale;
          const double fy = gy - (double)iy;
          unsigned long long* col = s_bits + bc * ld;
          if (l >= 0) atomicAdd(&col[l], (unsigned long long)__double2ll_rn((1.0 - fy) * w));
          if (l + 1 < lmax) atomicAdd(&col[l + 1], (unsigned long long)__double2ll_rn(fy * w));
        }
      }
      __syncthreads();
      for (int i = tid; i < NC * ld; i += KDE_THREADS)
        s_a[i] = (double)(long long)reinterpret_cast<unsigned long long*>(s_a)[i] * finv;
      __syncthreads();
      {  // vertical pass in place: filtered value of chunk row q is written to LDS row q + 4
        int nseg = KDE_THREADS / NC;
        if (nseg > nrow / 12) nseg = (nrow / 12 > 0) ? nrow / 12 : 1;
        const int seg = tid / NC, c = tid % NC;
        const int R = (nrow + nseg - 1) / nseg;
        const int q0 = seg * R, q1 = (q0 + R < nrow) ? (q0 + R) : nrow;
        double* col = s_a + c * ld;
        double head[4], tail[4];
        const bool active = (seg < nseg) && (q0 < nrow);
        if (active) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            head[t] = col[q0 + t];
            tail[t] = col[q1 + 4 + t];
          }
        }
        __syncthreads();
        if (active) {
          double w0 = head[0], w1 = head[1], w2 = head[2], w3 = head[3];
          double w4 = col[q0 + 4], w5 = col[q0 + 5], w6 = col[q0 + 6], w7 = col[q0 + 7];
          for (int q = q0; q < q1; ++q) {
            const int lnew = q + 8;
            const double w8 = (lnew >= q1 + 4) ? tail[lnew - (q1 + 4)] : col[lnew];
            const double acc = w0 * g[0] + w1 * g[1] + w2 * g[2] + w3 * g[3] + w4 * g[4] + w5 * g[5] + w6 * g[6] +
                               w7 * g[7] + w8 * g[8];
            col[q + 4] = acc;
            w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = w6; w6 = w7; w7 = w8;
          }
        }
      }
      __syncthreads();
      for (int idx = tid; idx < KDE_TX * nrow; idx += KDE_THREADS) {
        const int xl = idx % KDE_TX, yl = idx / KDE_TX;
        const int x = x0 + xl;
        if (x >= N) continue;
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc += s_a[(xl + t) * ld + yl + 4] * g[t];
        acc *= 0.15915494309189535;  // 1 / (2 pi): 2-D Gaussian pdf normalisation
        const float v = (float)acc;
        out[(size_t)(r0 + yl) * N + x] = v;
        const unsigned int key = f32_order_key(v);
        kmin = min(kmin, key);
        kmax = max(kmax, key);
      }
      __syncthreads();
    }
  }
  // rows outside the band (all rows for tiles away from the edge): zeros.  raw_band (the loop form): they are
  // not written -- the pixel kernels take the band from kde_band and treat everything outside it as zero.
  bool wrote_zero = false;
  if (raw_band) {
    if (tid == 0) {
      E.kde_band[2 * tile] = y_lo;
      E.kde_band[2 * tile + 1] = y_hi;
    }
    wrote_zero = (tid == 0) && (y_hi - y_lo + 1 < M);
  } else {
    for (int idx = tid; idx < KDE_TX * M; idx += KDE_THREADS) {
      const int xl = idx % KDE_TX, y = idx / KDE_TX;
      if (y >= y_lo && y <= y_hi) continue;
      if (x0 + xl < N) {
        out[(size_t)y * N + x0 + xl] = 0.f;
        wrote_zero = true;
      }
    }
  }
  if (wrote_zero) {
    const unsigned int kz = f32_order_key(0.f);
    kmin = min(kmin, kz);
    kmax = max(kmax, kz);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (unsigned int)__shfl_xor((int)kmin, o, WAVE));
    kmax = max(kmax, (unsigned int)__shfl_xor((int)kmax, o, WAVE));
  }
  if ((tid & 63) == 0) {
    atomicMin(&E.mm[0], kmin);
    atomicMax(&E.mm[1], kmax);
  }
}

// normalised curve KDE at a pixel.  raw_band: E.kde holds the raw density of the band rows only; the min-max
// normalisation of k_kde_normalise ((v - min) / span in float32, gpet_utils.py:81-91) is applied here instead.
__device__ __forceinline__ double kde_at(const EdgeDev& E, int x, int y, int raw_band, float mn, float span) {
  if (!raw_band) return (double)E.kde[(size_t)y * E.N + x];
  const int t = x / KDE_TX;
  const float raw = (y >= E.kde_band[2 * t] && y <= E.kde_band[2 * t + 1]) ? E.kde[(size_t)y * E.N + x] : 0.f;
  return (double)((raw - mn) / span);
}

// pixel_scores = 1/3 * (i*g + i + g) with numpy's rounding order (gpet.py:582)
__device__ __forceinline__ double pixel_score(double iv, double gv) {
#pragma clang fp contract(off)
  double t = iv * gv;
  t = t + iv;
  t = t + gv;
  return (1.0 / 3.0) * t;
}

__device__ __forceinline__ int bin_of(const EdgeDev& E, int x) {
  return (int)rint((double)(x - E.x_st) / (double)E.delta_x) - E.bin_lo;  // np.round = half-to-even
}

// best new candidate of every admissible column (first in row-major order among equals).
// 32 columns x 8 row-lanes per workgroup: each thread scans every 8th row of its column (independent loads; a wave reads
// two rows of 128 contiguous bytes per instruction: whole cache lines), then the 8 lanes of a column reduce in LDS.
#define PIX_CX 32
#define PIX_RY 8
__global__ void __launch_bounds__(256) k_pix_columns(EdgeDev* edges, int raw_band) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  __shared__ double s_best[PIX_RY][PIX_CX + 1];
  __shared__ int s_by[PIX_RY][PIX_CX + 1];
  const int cx = threadIdx.x & (PIX_CX - 1), ry = threadIdx.x / PIX_CX;
  const int x = blockIdx.x * PIX_CX + cx;
  double best = -1.0;
  int by = -1;
  const bool admissible = (x < E.N) && (E.fix_endpoints ? (x > E.x_st && x < E.x_en) : true);  // gpet.py:655-657
  const float mn = f32_from_key(E.mm[0]);
  const float span = f32_from_key(E.mm[1]) - mn;
  // raw_band: rows outside the band of the column's KDE tile hold density 0 -> normalised 0 <= 1e-3 unless the minimum
  // of the whole image is negative (it is not: densities are sums of non-negative terms), so only the band is scanned
  int y_first = 0, y_last = E.M - 1;
  if (raw_band && x < E.N) {
    y_first = E.kde_band[2 * (x / KDE_TX)];
    y_last = E.kde_band[2 * (x / KDE_TX) + 1];
  }
  if (admissible) {
    // eight rows per pass, their densities requested together (a row at a time the loop paid one memory round trip
    // per row: 62 dependent trips for a 500-row band); the gradient KDE only where the density passes the threshold
    for (int y0 = y_first + ry; y0 <= y_last; y0 += 8 * PIX_RY) {
      float kv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int y = y0 + u * PIX_RY;
        kv[u] = (y <= y_last) ? E.kde[(size_t)y * E.N + x] : -1.0f;
      }
      double ivs[8];
      float gk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int y = y0 + u * PIX_RY;
        ivs[u] = (y <= y_last) ? (raw_band ? (double)((kv[u] - mn) / span) : (double)kv[u]) : 0.0;
        gk[u] = (ivs[u] > 1e-3) ? E.grad_kde[(size_t)y * E.N + x] : 0.f;  // gpet.py:651
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (ivs[u] > 1e-3) {
          const double sv = pixel_score(ivs[u], (double)gk[u]);
          if (sv > best) {  // rows visited in increasing order: first maximum kept
            best = sv;
            by = y0 + u * PIX_RY;
          }
        }
      }
    }
  }
  s_best[ry][cx] = best;
  s_by[ry][cx] = by;
  __syncthreads();
  if (ry == 0 && x < E.N) {
    for (int q = 1; q < PIX_RY; ++q) {
      const double v = s_best[q][cx];
      const int yy = s_by[q][cx];
      if (yy >= 0 && (v > best || (v == best && yy < by))) {
        best = v;
        by = yy;
      }
    }
    E.colbest[x] = best;
    E.colbest_y[x] = by;
    if (by >= 0) atomicMax(&E.binbest[bin_of(E, x)], (unsigned long long)__double_as_longlong(best));
  }
}

// previously accepted observations are re-scored and compete first (gpet.py:568-579)
__global__ void __launch_bounds__(256) k_pix_old(EdgeDev* edges, int raw_band) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sc->n_obs) return;
  const long long x = E.obs_xy[2 * i], y = E.obs_xy[2 * i + 1];
  const float mn = f32_from_key(E.mm[0]);
  const double iv = kde_at(E, (int)x, (int)y, raw_band, mn, f32_from_key(E.mm[1]) - mn);
  if (iv > 1e-3) {
    const double sv = pixel_score(iv, (double)E.grad_kde[(size_t)y * E.N + x]);
    atomicMax(&E.binbest[bin_of(E, (int)x)], (unsigned long long)__double_as_longlong(sv));
  }
}

// among the candidates that reach their bin's best score, the first in the reference's candidate
// order wins (np.argmax): old observations in their order, then new pixels in row-major order.
__global__ void __launch_bounds__(256) k_pix_argbest(EdgeDev* edges, int raw_band) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_obs = sc->n_obs;
  if (t < n_obs) {
    const long long x = E.obs_xy[2 * t], y = E.obs_xy[2 * t + 1];
    const float mn = f32_from_key(E.mm[0]);
    const double iv = kde_at(E, (int)x, (int)y, raw_band, mn, f32_from_key(E.mm[1]) - mn);
    if (iv > 1e-3) {
      const double sv = pixel_score(iv, (double)E.grad_kde[(size_t)y * E.N + x]);
      const int b = bin_of(E, (int)x);
      if ((unsigned long long)__double_as_longlong(sv) == E.binbest[b])
        atomicMin(&E.binarg[b], (long long)t - (long long)(1ll << 40));  // old: before every new pixel
    }
  }
  if (t < E.N) {
    const int by = E.colbest_y[t];
    if (by >= 0) {
      const int b = bin_of(E, t);
      if ((unsigned long long)__double_as_longlong(E.colbest[t]) == E.binbest[b])
        atomicMin(&E.binarg[b], (long long)by * E.N + t);
    }
  }
}

// adaptive threshold (gpet.py:589-609), one pixel per bin (gpet.py:613-616), loop bookkeeping
// (gpet.py:861-865).  One wave per edge.
__global__ void __launch_bounds__(64) k_pix_select(EdgeDev* edges) {
#pragma clang fp contract(off)
  const EdgeDev E = edges[blockIdx.y];
  gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int lane = threadIdx.x;
  const int n_pre = sc->n_obs;
  double thresh = sc->score_thresh;
  int n_pix = n_pre;
  int it = 0;
  bool stuck = false;
  while ((n_pix - n_pre < E.pixel_thresh) && (n_pix < E.algo_thresh)) {
    if (it > 0) {
      if (thresh == 0.0) {  // the reference would spin forever here (latent hang 1)
        stuck = true;
        break;
      }
      thresh = thresh * 0.95;
    }
    int cnt = 0;
    for (int b = lane; b < E.n_bins; b += WAVE) {
      const unsigned long long bits = E.binbest[b];
      cnt += (E.binarg[b] != 0x7FFFFFFFFFFFFFFFll) && (__longlong_as_double((long long)bits) >= thresh);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, WAVE);
    n_pix = cnt;
    ++it;
  }
  if (stuck) {
    if (lane == 0) sc->status = GPET_ERR_ITER_CAP;
    return;
  }
  // compact the winning pixels in ascending bin order (np.unique order)
  int base = 0;
  for (int b0 = 0; b0 < E.n_bins; b0 += WAVE) {
    const int b = b0 + lane;
    bool take = false;
    long long key = 0;
    if (b < E.n_bins) {
      key = E.binarg[b];
      take = (key != 0x7FFFFFFFFFFFFFFFll) && (__longlong_as_double((long long)E.binbest[b]) >= thresh);
    }
    const unsigned long long bal = __ballot(take);
    if (take) {
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      long long x, y;
      if (key < 0) {  // an old observation
        const long long oi = key + (1ll << 40);
        x = E.obs_xy[2 * oi];
        y = E.obs_xy[2 * oi + 1];
      } else {
        y = key / E.N;
        x = key - y * E.N;
      }
      E.obs_new[2 * pos] = x;
      E.obs_new[2 * pos + 1] = y;
    }
    base += __popcll(bal);
  }
  // (same wave: obs_new fully written before it is copied back)
  for (int i = lane; i < 2 * n_pix; i += WAVE) E.obs_xy[i] = E.obs_new[i];
  if (lane == 0) {
    sc->score_thresh = thresh;
    sc->n_obs = n_pix;
    sc->iter = sc->iter + 1;
    sc->done = (n_pix >= E.algo_thresh) ? 1 : 0;
  }
}


// ---------------------------------------------------------------------------------------
// f2  converged fit: -log marginal likelihood and its gradient wrt theta = log(c, l, noise)
//     sklearn_gpr.py:512-585 on the standardised training set (gpet.py:235-248).
//     One workgroup per (edge, restart) problem; L and L^-1 packed-lower in LDS (n <= 128).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void corr_and_dlog(const EdgeDev& E, double a, double b, double& R, double& dR) {
  const int kernel_type = E.kernel_type, nu_code = E.nu_code;
  const double d = a - b;
  const double D = d * d;
  if (kernel_type == GPET_KERNEL_RBF) {
    R = exp(-0.5 * D);
    dR = R * D;
    return;
  }
  const double r = sqrt(D);
  if (nu_code == 0) {
    R = exp(-r);
    dR = (r > 0.0) ? R * D / r : 0.0;
  } else if (nu_code == 3) {
    R = matern_gen(E.nu_gen, E.inv_gamma_nu, r, &dR);
  } else if (nu_code == 1) {
    const double k = r * 1.7320508075688772;
    R = (1.0 + k) * exp(-k);
    dR = 3.0 * D * exp(-sqrt(3.0 * D));
  } else {
    const double k = r * 2.23606797749979;
    const double t = sqrt(5.0 * D);
    R = (1.0 + k + k * k / 3.0) * exp(-k);
    dR = 5.0 / 3.0 * D * (t + 1.0) * exp(-t);
  }
}

// Symmetric sweep operator on the bordered matrix  A = [[0, y^T], [y, K]]  (index 0 = the y border,
// 1..n = K): sweeping the pivots 1..n in place leaves  A = [[-y^T K^-1 y, alpha^T], [alpha, -K^-1]]
// and the pivots d_k are the squared Cholesky diagonal (log|K| = sum log d_k).  The matrix lives in
// REGISTERS: thread t owns the 4x4 tile (ti, tj), tj <= ti, of the lower triangle (diagonal tiles
// keep both halves), so a step is: owners of column k publish it to LDS (double-buffered, n+1
// doubles), ONE barrier, everyone reads its 4+4 column entries and does 16 FMAs.  n^3/2 FMAs in
// total and n barriers, against the ~3n barriers and the LDS-resident L / L^-1 (2 x 66 KB, one
// workgroup per CU) of a Cholesky + triangular inverse; LDS use is ~3 KB so several problems share a CU.
#define LML_MAXD 136  // 4 * ceil((128 + 1) / 4) + slack
__global__ void __launch_bounds__(576) __attribute__((amdgpu_waves_per_eu(5, 5))) k_lml(EdgeDev* edges, const int* edge_of, const double* theta, double* f_out,
                                             double* g_out, const int* count) {
  const int pb = blockIdx.x;
  if (count != nullptr && pb >= *count) return;  // (launches are sized by the host's last KNOWN number of running problems)
  const EdgeDev E = edges[edge_of[pb]];
  const int n = E.fin_n;
  const int nb = (n + 1 + 3) >> 2;  // 4x4 tiles per side of the bordered matrix
  const int ntile = nb * (nb + 1) / 2;
  __shared__ __attribute__((aligned(16))) double s_col[2][LML_MAXD];
  __shared__ double s_x[LML_MAXD], s_y[LML_MAXD], s_w[LML_MAXD];  // x / l, y, noise weights at index 1..n
  __shared__ double s_piv[LML_MAXD];
  __shared__ double s_red[16];
  const int tid = threadIdx.x, bs = blockDim.x;
  const bool active = tid < ntile;
  int ti = (int)((sqrt(8.0 * (double)tid + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tid) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tid) ++ti;
  const int tj = tid - ti * (ti + 1) / 2;
  const double c = exp(theta[3 * pb]), ell = exp(theta[3 * pb + 1]), nl = exp(theta[3 * pb + 2]);
  for (int i = tid; i < 4 * nb; i += bs) {
    const bool in = (i >= 1 && i <= n);
    s_x[i] = in ? E.fin_x[i - 1] / ell : 0.0;
    s_y[i] = in ? E.fin_y[i - 1] : 0.0;
    s_w[i] = in ? E.fin_w[i - 1] : 0.0;
    s_piv[i] = 1.0;
  }
  __syncthreads();
  double T[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = 4 * ti + a, j = 4 * tj + b;
      double v = 0.0;
      if (active && i <= n && j <= n) {
        if (i == 0 || j == 0) {
          v = (i == j) ? 0.0 : s_y[i + j];
        } else if (i == j) {
          v = c + nl * s_w[i];
          v = v + 1e-6;
        } else {
          v = c * corr_fn(E, s_x[i], s_x[j]);
        }
      }
      T[a][b] = v;
    }
  bool bad = false, stop = false;
  for (int kb = 0; kb < nb && !stop; ++kb) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = 4 * kb + kk;
      const int buf = kk & 1;  // (= k & 1: a constant once the loop is unrolled -- no double-buffer address arithmetic per step)
      if (k == 0) continue;  // the y border is not a pivot
      if (k > n) {
        stop = true;
        break;
      }
      if (active) {
        if (tj == kb) {
#pragma unroll
          for (int a = 0; a < 4; ++a) s_col[buf][4 * ti + a] = T[a][kk];
          if (ti == kb) s_piv[k] = T[kk][kk];
        } else if (ti == kb) {
#pragma unroll
          for (int b = 0; b < 4; ++b) s_col[buf][4 * tj + b] = T[kk][b];
        }
      }
      __syncthreads();
      const double d = s_col[buf][k];
      if (!(d > 0.0)) {  // not positive definite (same for every thread)
        bad = true;
        stop = true;
        break;
      }
      if (active) {
        // 1 / d by every thread from the hardware reciprocal + two Newton steps: keeps the division off the
        // critical path in front of the barrier (the pivot owner would otherwise hold everyone up)
        double inv = __builtin_amdgcn_rcp(d);
        inv = inv * (2.0 - d * inv);
        inv = inv * (2.0 - d * inv);
        double ci[4], cj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          ci[a] = s_col[buf][4 * ti + a];
          cj[a] = s_col[buf][4 * tj + a];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const double ma = -(ci[a] * inv);
#pragma unroll
          for (int b = 0; b < 4; ++b) T[a][b] = fma(ma, cj[b], T[a][b]);
        }
        if (ti == kb) {
#pragma unroll
          for (int b = 0; b < 4; ++b) T[kk][b] = cj[b] * inv;
        }
        if (tj == kb) {
#pragma unroll
          for (int a = 0; a < 4; ++a) T[a][kk] = ci[a] * inv;
          if (ti == kb) T[kk][kk] = -inv;
        }
      }
    }
  }
  if (bad) {  // sklearn returns (-inf, 0) -> objective (+inf, -0)
    if (tid == 0) {
      f_out[pb] = INFINITY;
      g_out[3 * pb] = g_out[3 * pb + 1] = g_out[3 * pb + 2] = 0.0;
    }
    return;
  }
  // alpha = column 0 of the swept matrix; its corner is -y^T alpha
  const int fbuf = (n + 1) & 1;  // (the last pivot n used buffer n & 1)
  if (active && tj == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a) s_col[fbuf][4 * ti + a] = T[a][0];
  }
  __syncthreads();
  const double* al = s_col[fbuf];
  double ld = 0.0;
  for (int k = 1 + tid; k <= n; k += bs) ld += log(sqrt(s_piv[k]));
  const double logdet = block_sum(ld, s_red);
  // gradient: 0.5 * sum_ij (alpha_i alpha_j - Kinv_ij) dK_ij; off-diagonal tiles stand for both triangles
  double gc = 0.0, gl = 0.0, gn = 0.0;
  if (active) {
    const double wt = (ti == tj) ? 1.0 : 2.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = 4 * ti + a, j = 4 * tj + b;
        if (i >= 1 && j >= 1 && i <= n && j <= n) {
          const double inner = al[i] * al[j] + T[a][b];  // T = -Kinv
          if (i == j) {
            gc += inner * c;
            gn += inner * (nl * s_w[i]);
          } else {
            double R, dR;
            corr_and_dlog(E, s_x[i], s_x[j], R, dR);
            gc += wt * inner * (c * R);
            gl += wt * inner * (c * dR);
          }
        }
      }
  }
  gc = block_sum(gc, s_red);
  gl = block_sum(gl, s_red);
  gn = block_sum(gn, s_red);
  if (tid == 0) {
    const double yta = -al[0];
    const double lml = -0.5 * yta - logdet - 0.5 * (double)n * 1.8378770664093453;  // log(2 pi)
    f_out[pb] = -lml;
    g_out[3 * pb] = -0.5 * gc;
    g_out[3 * pb + 1] = -0.5 * gl;
    g_out[3 * pb + 2] = -0.5 * gn;
  }
}

// The same kernel for 128 < n <= 250 training points (small delta_x): two tiles per thread (t and t + blockDim), up to
// 63 x 64 / 2 = 2016 tiles on 1024 threads.  Kept separate so that the common sizes keep their lean single-tile code.
#define LML2_MAXD 260
__global__ void __launch_bounds__(1024) k_lml2(EdgeDev* edges, const int* edge_of, const double* theta, double* f_out,
                                               double* g_out, const int* count) {
  constexpr int SLOTS = 2;  // (three and four tiles per thread were measured: 163 / 203 VGPRs, 6-40 % slower)
  const int pb = blockIdx.x;
  if (count != nullptr && pb >= *count) return;
  const EdgeDev E = edges[edge_of[pb]];
  const int n = E.fin_n;
  const int nb = (n + 1 + 3) >> 2;
  const int ntile = nb * (nb + 1) / 2;
  __shared__ __attribute__((aligned(16))) double s_col[2][LML2_MAXD];
  __shared__ double s_x[LML2_MAXD], s_y[LML2_MAXD], s_w[LML2_MAXD];  // x / l, y, noise weights at index 1..n
  __shared__ double s_piv[LML2_MAXD];
  __shared__ double s_red[16];
  const int tid = threadIdx.x, bs = blockDim.x;
  bool active[SLOTS];
  int ti[SLOTS], tj[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int t = tid + s * bs;
    int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (r * (r + 1) / 2 > t) --r;
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    ti[s] = r;
    tj[s] = t - r * (r + 1) / 2;
    active[s] = t < ntile;
  }
  const double c = exp(theta[3 * pb]), ell = exp(theta[3 * pb + 1]), nl = exp(theta[3 * pb + 2]);
  for (int i = tid; i < 4 * nb; i += bs) {
    const bool in = (i >= 1 && i <= n);
    s_x[i] = in ? E.fin_x[i - 1] / ell : 0.0;
    s_y[i] = in ? E.fin_y[i - 1] : 0.0;
    s_w[i] = in ? E.fin_w[i - 1] : 0.0;
    s_piv[i] = 1.0;
  }
  __syncthreads();
  double T[SLOTS][4][4];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = 4 * ti[s] + a, j = 4 * tj[s] + b;
        double v = 0.0;
        if (active[s] && i <= n && j <= n) {
          if (i == 0 || j == 0) {
            v = (i == j) ? 0.0 : s_y[i + j];
          } else if (i == j) {
            v = c + nl * s_w[i];
            v = v + 1e-6;
          } else {
            v = c * corr_fn(E, s_x[i], s_x[j]);
          }
        }
        T[s][a][b] = v;
      }
  bool bad = false, stop = false;
  for (int kb = 0; kb < nb && !stop; ++kb) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = 4 * kb + kk;
      const int buf = kk & 1;
      if (k == 0) continue;  // the y border is not a pivot
      if (k > n) {
        stop = true;
        break;
      }
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        if (active[s]) {
          if (tj[s] == kb) {
#pragma unroll
            for (int a = 0; a < 4; ++a) s_col[buf][4 * ti[s] + a] = T[s][a][kk];
            if (ti[s] == kb) s_piv[k] = T[s][kk][kk];
          } else if (ti[s] == kb) {
#pragma unroll
            for (int b = 0; b < 4; ++b) s_col[buf][4 * tj[s] + b] = T[s][kk][b];
          }
        }
      }
      __syncthreads();
      const double d = s_col[buf][k];
      if (!(d > 0.0)) {  // not positive definite (same for every thread)
        bad = true;
        stop = true;
        break;
      }
      double inv = __builtin_amdgcn_rcp(d);
      inv = inv * (2.0 - d * inv);
      inv = inv * (2.0 - d * inv);
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        if (active[s]) {
          double ci[4], cj[4];
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            ci[a] = s_col[buf][4 * ti[s] + a];
            cj[a] = s_col[buf][4 * tj[s] + a];
          }
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const double ma = -(ci[a] * inv);
#pragma unroll
            for (int b = 0; b < 4; ++b) T[s][a][b] = fma(ma, cj[b], T[s][a][b]);
          }
          if (ti[s] == kb) {
#pragma unroll
            for (int b = 0; b < 4; ++b) T[s][kk][b] = cj[b] * inv;
          }
          if (tj[s] == kb) {
#pragma unroll
            for (int a = 0; a < 4; ++a) T[s][a][kk] = ci[a] * inv;
            if (ti[s] == kb) T[s][kk][kk] = -inv;
          }
        }
      }
    }
  }
  if (bad) {  // sklearn returns (-inf, 0) -> objective (+inf, -0)
    if (tid == 0) {
      f_out[pb] = INFINITY;
      g_out[3 * pb] = g_out[3 * pb + 1] = g_out[3 * pb + 2] = 0.0;
    }
    return;
  }
  const int fbuf = (n + 1) & 1;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s)
    if (active[s] && tj[s] == 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a) s_col[fbuf][4 * ti[s] + a] = T[s][a][0];
    }
  __syncthreads();
  const double* al = s_col[fbuf];
  double ld = 0.0;
  for (int k = 1 + tid; k <= n; k += bs) ld += log(sqrt(s_piv[k]));
  const double logdet = block_sum(ld, s_red);
  double gc = 0.0, gl = 0.0, gn = 0.0;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    if (active[s]) {
      const double wt = (ti[s] == tj[s]) ? 1.0 : 2.0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int i = 4 * ti[s] + a, j = 4 * tj[s] + b;
          if (i >= 1 && j >= 1 && i <= n && j <= n) {
            const double inner = al[i] * al[j] + T[s][a][b];  // T = -Kinv
            if (i == j) {
              gc += inner * c;
              gn += inner * (nl * s_w[i]);
            } else {
              double R, dR;
              corr_and_dlog(E, s_x[i], s_x[j], R, dR);
              gc += wt * inner * (c * R);
              gl += wt * inner * (c * dR);
            }
          }
        }
    }
  }
  gc = block_sum(gc, s_red);
  gl = block_sum(gl, s_red);
  gn = block_sum(gn, s_red);
  if (tid == 0) {
    const double yta = -al[0];
    const double lml = -0.5 * yta - logdet - 0.5 * (double)n * 1.8378770664093453;  // log(2 pi)
    f_out[pb] = -lml;
    g_out[3 * pb] = -0.5 * gc;
    g_out[3 * pb + 1] = -0.5 * gl;
    g_out[3 * pb + 2] = -0.5 * gn;
  }
}

// ---------------------------------------------------------------------------------------
// f2 on the matrix cores (round 3; n <= 108 training points on a lattice): the same symmetric sweep of the bordered
// matrix, FOUR pivots at a time, on 16x16 tiles of v_mfma_f64_16x16x4_f64.
//
// Layout of the bordered matrix (size n4 + 1 <= 112, n4 = n rounded up to a multiple of 4): indices 0..n-1 = K,
// n..n4-1 = identity padding (pivots 1: log 1 = 0, no coupling), n4 = the y border (never a pivot), the rest zero.
// The lower triangle of 16x16 tiles (diagonal tiles whole) lives in the accumulator registers of TWO waves: wave
// `role` owns the tile rows {6, 3, 1, 0} (role 0) and {5, 4, 2} (role 1): 14 tiles each.  ONE instruction stream serves
// both: 16 static accumulator slots (g, J), J < 7 - 2 g, whose tile row l16_row(g, role) is a scalar of the wave; slot
// (g, J) is in use when J <= that row -- 12 slots in both waves, two in one of them each.  (Two separate streams were
// built first: the register allocator kept both sets of accumulators apart -- 224 + 165 registers, one wave per SIMD.
// On this chip the f64 MFMA runs on the vector unit's own FMA lanes -- SQ_VALU_MFMA_BUSY_CYCLES + the VALU issue cycles
// add up to the kernel's time, nothing overlaps between the waves of a SIMD -- so what counts is the SUM of MFMA and
// VALU cycles of the busier wave: the split 16 + 12 of {6, 4, 2, 0} / {5, 3, 1} cost 128 cycles a step more.)
// f64 C/D layout: lane (q = l >> 4, col = l & 15), register r holds element (row q + 4 r, column col) of its tile;
// A operand: lane holds A[row = col][k = q], B operand: B[k = q][column = col].
//
// A block step for the pivots k0..k0+3 (tile Jp, sub-block B; W = columns k0..k0+3 of the matrix = the PANEL, in LDS
// k-major, copied out of the tiles at the end of the previous step; P = its rows k0..k0+3):
//   P = L D L^T in every lane (unit lower L; the four pivots d are the squared Cholesky diagonal in the scalar order);
//   Y = W L^-T row by row, straight from the panel (each lane the rows 16 J + col of all seven tile columns: six FMAs
//   each; component q of them are its B operands, four of them -- its own tile rows -- its A operands);
//   the four rank-one updates of the scalar sweep as ONE MFMA per tile:  T -= (Y D^-1) Y^T,  A operand -Y[row][q] / d_q,
//   B operand Y[column][q] -- the scalar algorithm's own arithmetic, no product with an explicit P^-1;
//   the pivot rows and columns must become (W P^-1)^T and W P^-1: the A operand of the four pivot rows is
//   Linv[q][i'] / d_q - L[i'][q] instead, the B operand of the four pivot columns Y[j'][q] - Linv[q][j']: the MFMA
//   then yields T_pj - W_jp + (W P^-1)_jp with the first two cancelling to the rounding of Y -- no accumulator touched;
//   the 4x4 block of the pivots gets both and becomes 2 I - P^-1: the epilogue takes the 2 off the diagonal;
//   the next panel is copied into the other panel buffer (one scalar jump on its tile column); ONE barrier per step.
// n^3 / 2 FMAs of the scalar sweep become 25 x 28 MFMAs (1024 FMAs each, 68 % useful at n = 98).  The training x sit on
// a lattice (pixel columns), so the correlation and its length-scale derivative are tabulated at the integer lags once
// per problem: lagmax + 1 transcendentals instead of n^2 for the set-up and n^2 more for the gradient.
// ---------------------------------------------------------------------------------------
typedef double v4d __attribute__((ext_vector_type(4)));
#define L16_WS 112    // row stride of the panel planes in doubles: = 16 mod 32, so the four k-planes of an operand read fall on disjoint banks
#define L16_MAXN 108  // 4 * ceil(n / 4) + 1 <= 112
#define L16_NT 7
#define L16_LAG_MAX 4096  // longest correlation table (2 x 32 KB of LDS per problem)
#define L16_NG 4
#define L16_NS 16
__host__ __device__ constexpr int l16_sg(int t) { return t < 7 ? 0 : (t < 12 ? 1 : (t < 15 ? 2 : 3)); }  // slot -> group
__host__ __device__ constexpr int l16_sj(int t) { return t < 7 ? t : (t < 12 ? t - 7 : (t < 15 ? t - 12 : 0)); }  // slot -> tile column
// tile row of group g of wave `role`: {6, 3, 1, 0} and {5, 4, 2} -- 14 tiles each (-1: role 1 has no fourth tile row)
__host__ __device__ constexpr int l16_row(int g, int role) { return g == 0 ? 6 - role : (g == 1 ? 3 + role : (g == 2 ? 1 + role : -role)); }
// slot (g, J) belongs to both waves / to one of them only (the widest tile of the group's lower tile row)
__host__ __device__ constexpr int l16_owner(int g, int J) {  // 2: both, 0 / 1: that role only
  return g == 0 ? (J == 6 ? 0 : 2) : (g == 1 ? (J == 4 ? 1 : 2) : (g == 2 ? (J == 2 ? 1 : 2) : 0));
}

struct L16Shared {
  double Wt[2][4][L16_WS];  // panel of the current / next block step, k-major
  double al[L16_WS];        // alpha (row n4 of the swept matrix)
  double y[L16_WS], w[L16_WS];
  double piv[L16_WS];
  int m[L16_WS];            // lattice coordinate of training point i
  double red[8];
};

__device__ __forceinline__ double l16_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = r * (2.0 - d * r);
  r = r * (2.0 - d * r);
  return r;
}

__device__ __forceinline__ double l16_sel(const v4d& a, int r) {
  return r == 0 ? a[0] : (r == 1 ? a[1] : (r == 2 ? a[2] : a[3]));
}

struct L16Acc {
  v4d v[L16_NS];
};

// copy the panel of pivot block (Jn, BN) out of the tiles: W[row][a] = A[row][16 Jn + 4 BN + a].  Per tile row of the wave at
// most ONE tile has the panel's columns (tile column Jn: a scalar jump picks its slot) and at most one tile row has its
// rows (the tile row Jn itself, whose register BN holds the panel rows of the tile columns left of it, by symmetry).
// (the empty asm with a different immediate per case keeps the cases apart: merged, they become ONE store sequence behind
// a phi of accumulator ADDRESSES, and those accumulators then live in scratch memory)
#define L16_COLPART(slot)                                       \
  if (pcn) {                                                    \
    const double t0_ = A.v[slot][0], t1_ = A.v[slot][1], t2_ = A.v[slot][2], t3_ = A.v[slot][3]; \
    asm volatile("; l16 panel from slot %0" ::"n"(slot));      \
    wc[0] = t0_;                                                \
    wc[4] = t1_;                                                \
    wc[8] = t2_;                                                \
    wc[12] = t3_;                                               \
  }
// ONE scalar jump on the tile column Jn of the next pivots; inside a case everything but the wave's role is a constant:
// the tiles (I, Jn) of the tile rows I >= Jn hold the panel's columns, and the tile row Jn itself -- which exactly one
// of the two waves owns -- holds its rows for the tile columns left of it.
template <int BN, int JN, int G>
__device__ __forceinline__ void l16_extract_col(const L16Acc& A, double* Wn, int role, int q, int col) {
  constexpr int base = G == 0 ? 0 : (G == 1 ? 7 : (G == 2 ? 12 : 15));
  constexpr int lo = l16_row(G, 0) < l16_row(G, 1) ? l16_row(G, 0) : l16_row(G, 1);
  constexpr int hi = l16_row(G, 0) < l16_row(G, 1) ? l16_row(G, 1) : l16_row(G, 0);
  if constexpr (JN <= hi) {
    if (JN <= lo || role == (l16_row(G, 0) == hi ? 0 : 1)) {  // this wave's tile row of the group is not above the pivots
      const bool pcn = (col >> 2) == BN;
      double* wc = Wn + (col & 3) * L16_WS + 16 * l16_row(G, role) + q;
      L16_COLPART(base + JN)
    }
  }
}
template <int BN, int JN>
__device__ __forceinline__ void l16_extract_case(const L16Acc& A, double* Wn, int role, int q, int col) {
  l16_extract_col<BN, JN, 0>(A, Wn, role, q, col);
  l16_extract_col<BN, JN, 1>(A, Wn, role, q, col);
  l16_extract_col<BN, JN, 2>(A, Wn, role, q, col);
  l16_extract_col<BN, JN, 3>(A, Wn, role, q, col);
  // tile row JN = l16_row(G, par)
  constexpr int par = (JN == 5 || JN == 4 || JN == 2) ? 1 : 0, G = JN >= 5 ? 0 : (JN >= 3 ? 1 : (JN >= 1 ? 2 : 3));
  static_assert(l16_row(G, par) == JN, "owner of the tile row");
  constexpr int base = G == 0 ? 0 : (G == 1 ? 7 : (G == 2 ? 12 : 15));
  if (JN > 0 && role == par) {
    double* wr = Wn + q * L16_WS + col;
#pragma unroll
    for (int J = 0; J < JN; ++J) {
      const double t_ = A.v[base + J][BN];
      asm volatile("; l16 panel row from slot %0" ::"n"(base + J));
      wr[16 * J] = t_;
    }
  }
}
template <int BN>
__device__ __forceinline__ void l16_extract(const L16Acc& A, double* Wn, int Jn, int role, int q, int col) {
  switch (Jn) {
    case 0: l16_extract_case<BN, 0>(A, Wn, role, q, col); break;
    case 1: l16_extract_case<BN, 1>(A, Wn, role, q, col); break;
    case 2: l16_extract_case<BN, 2>(A, Wn, role, q, col); break;
    case 3: l16_extract_case<BN, 3>(A, Wn, role, q, col); break;
    case 4: l16_extract_case<BN, 4>(A, Wn, role, q, col); break;
    case 5: l16_extract_case<BN, 5>(A, Wn, role, q, col); break;
    default: l16_extract_case<BN, 6>(A, Wn, role, q, col); break;
  }
}

#ifdef GPET_L16_PROF  // cycles per phase of a block step (wave 0 of workgroup 0 prints them)
#define L16_STAMP(i) { const long long t_ = clock64(); prof[i] += t_ - tl; tl = t_; }
#else
#define L16_STAMP(i)
#endif
// one block step (pivots 16 Jp + 4 B ..); false: a pivot was not positive
template <int B>
__device__ __forceinline__ bool l16_step(L16Acc& A, L16Shared& S, int Jp, int n4, int role, int q, int col
#ifdef GPET_L16_PROF
                                         , long long* prof, long long& tl
#endif
) {
  constexpr int buf = B & 1;  // (four steps per tile: the parity of the step is the parity of B)
  const int k0 = 16 * Jp + 4 * B;
  const double* W = &S.Wt[buf][0][0];
  // P = L D L^T (uniform: every lane reads the same ten entries), the four pivots one after the other: d_k are the
  // squared Cholesky diagonal in the scalar order.  (Elimination by 2x2 blocks with adjugates halves this dependent
  // chain -- 740 instead of ~1100 cycles of a 3500-cycle step -- but forms X = Pb Pa^-1 and the Schur complement from
  // differences of products: 6e-7 relative on the objective at c / noise = 4e4 where this form gives 6e-12.  Dropped.)
  const double p00 = W[k0], p10 = W[k0 + 1], p20 = W[k0 + 2], p30 = W[k0 + 3];
  const double p11 = W[L16_WS + k0 + 1], p21 = W[L16_WS + k0 + 2], p31 = W[L16_WS + k0 + 3];
  const double p22 = W[2 * L16_WS + k0 + 2], p32 = W[2 * L16_WS + k0 + 3];
  const double p33 = W[3 * L16_WS + k0 + 3];
  const double i0 = l16_rcp(p00);
  const double l10 = p10 * i0, l20 = p20 * i0, l30 = p30 * i0;
  const double u11 = fma(-p10, l10, p11), u21 = fma(-p20, l10, p21), u31 = fma(-p30, l10, p31);
  const double i1 = l16_rcp(u11);
  const double l21 = u21 * i1, l31 = u31 * i1;
  const double u22 = fma(-u21, l21, fma(-p20, l20, p22));
  const double u32 = fma(-u31, l21, fma(-p30, l20, p32));
  const double i2 = l16_rcp(u22);
  const double l32 = u32 * i2;
  const double u33 = fma(-u32, l32, fma(-u31, l31, fma(-p30, l30, p33)));
  const double i3 = l16_rcp(u33);
  const bool ok = p00 > 0.0 && u11 > 0.0 && u22 > 0.0 && u33 > 0.0;  // (the same in every lane of both waves)
  if (role == 0 && (q | col) == 0) {
    S.piv[k0] = p00;
    S.piv[k0 + 1] = u11;
    S.piv[k0 + 2] = u22;
    S.piv[k0 + 3] = u33;
  }
  const bool pc = (col >> 2) == B;  // this lane's row (A operand) / column (tile) index is one of the pivots
  const int kc = col & 3;
  const int lb = q * L16_WS + col;
  // Per lane, from the uniform factors: row q of L^-1 (component q of a row of Y = W L^-T is its product with the four
  // panel entries of that row),  ykq = Linv[q][kc],  lkq = L[kc][q],  iq = 1 / d_q.
  const double m20 = fma(l21, l10, -l20), m31 = fma(l32, l21, -l31);  // L^-1 below its first subdiagonal (uniform)
  const double m30 = fma(-l32, m20, fma(l31, l10, -l30));
  const double cq0 = q == 0 ? 1.0 : (q == 1 ? -l10 : (q == 2 ? m20 : m30));
  const double cq1 = q == 1 ? 1.0 : (q == 2 ? -l21 : (q == 3 ? m31 : 0.0));
  const double cq2 = q == 2 ? 1.0 : (q == 3 ? -l32 : 0.0);
  const double cq3 = q == 3 ? 1.0 : 0.0;
  const double iq = q == 0 ? i0 : (q == 1 ? i1 : (q == 2 ? i2 : i3));
  const double ykq = kc == 0 ? cq0 : (kc == 1 ? cq1 : (kc == 2 ? cq2 : cq3));
  const double lrow1 = q == 0 ? l10 : (q == 1 ? 1.0 : 0.0);                                      // L[1][q]
  const double lrow2 = q == 0 ? l20 : (q == 1 ? l21 : (q == 2 ? 1.0 : 0.0));                   // L[2][q]
  const double lrow3 = q == 0 ? l30 : (q == 1 ? l31 : (q == 2 ? l32 : 1.0));                   // L[3][q]
  const double lkq = kc == 0 ? (q == 0 ? 1.0 : 0.0) : (kc == 1 ? lrow1 : (kc == 2 ? lrow2 : lrow3));
  L16_STAMP(0)
  // Y = W L^-T row by row (the scalar sweep's own arithmetic: four rank-one updates in a row), component q of it is what
  // a lane contributes: the rank-4 update is  T -= (Y D^-1) Y^T  -- A operand -Y[row][q] / d_q, B operand Y[column][q].
  // (The first form of this kernel multiplied W by the explicit P^-1: entries of size 1 / lambda_min(P) against W's
  // 433 cancel to O(1) -- 1e-6 relative on the objective at c / noise = 4e4 where the scalar sweep gives 6e-12.)
  // Pivot rows and columns through the operands: pivot row i' carries A = Linv[q][i'] / d_q - L[i'][q], so the MFMA
  // yields T_pj - W_jp + (W P^-1)_jp (the first two cancel to the rounding of Y); pivot column j' carries
  // B = Y[j'][q] - Linv[q][j'] for the mirror image.  The 4x4 block of the pivots gets both: with Y = L D on the
  // pivot rows the product is 2 I - P^-1 - P, so the block becomes 2 I - P^-1 (P cancels to its own rounding, 1e-16 of
  // entries a hundred times smaller than those of P^-1); the epilogue takes the 2 off the diagonal of the swept
  // matrix.  No accumulator is written outside the MFMAs.
  // Every wave forms ALL the rows of Y it needs from the panel itself: its B operands are component q of the rows of
  // all seven tile columns, its A operands those of its own four tile rows among them -- 7 instead of 4 short chains
  // per lane, and no plane of Y through LDS, no second barrier in the step.
  double yv[L16_NT];
#pragma unroll
  for (int J = 0; J < L16_NT; ++J) {
    const double* wr = W + 16 * J + col;
    yv[J] = fma(cq0, wr[0], fma(cq1, wr[L16_WS], fma(cq2, wr[2 * L16_WS], cq3 * wr[3 * L16_WS])));
  }
  double aop[L16_NG];
  {
    const double niq = -iq;
    aop[0] = (role ? yv[l16_row(0, 1)] : yv[l16_row(0, 0)]) * niq;
    aop[1] = (role ? yv[l16_row(1, 1)] : yv[l16_row(1, 0)]) * niq;
    aop[2] = (role ? yv[l16_row(2, 1)] : yv[l16_row(2, 0)]) * niq;
    aop[3] = yv[0] * niq;  // (role 1 has no fourth tile row: its slot 15 is never multiplied)
    const double apiv = fma(ykq, iq, -lkq);
#pragma unroll
    for (int g = 0; g < L16_NG; ++g) aop[g] = ((role ? l16_row(g, 1) : l16_row(g, 0)) == Jp && pc) ? apiv : aop[g];
  }
  L16_STAMP(1)
  double bop[L16_NT];
  {
    const double bdel = pc ? ykq : 0.0;
#pragma unroll
    for (int J = 0; J < L16_NT; ++J) bop[J] = fma(-bdel, (J == Jp) ? 1.0 : 0.0, yv[J]);  // (the factor is a scalar of the wave)
  }
  // one MFMA per tile
#pragma unroll
  for (int t = 0; t < L16_NS; ++t) {
    const int g = l16_sg(t), J = l16_sj(t);
    if (l16_owner(g, J) != 2) continue;
    A.v[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[g], bop[J], A.v[t], 0, 0, 0);
  }
  if (role == 0) {
#pragma unroll
    for (int t = 0; t < L16_NS; ++t) {
      const int g = l16_sg(t), J = l16_sj(t);
      if (l16_owner(g, J) != 0) continue;
      A.v[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[g], bop[J], A.v[t], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int t = 0; t < L16_NS; ++t) {
      const int g = l16_sg(t), J = l16_sj(t);
      if (l16_owner(g, J) != 1) continue;
      A.v[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[g], bop[J], A.v[t], 0, 0, 0);
    }
  }
  L16_STAMP(2)
  L16_STAMP(3)
  // next panel
  if (k0 + 4 < n4) l16_extract<(B + 1) & 3>(A, &S.Wt[buf ^ 1][0][0], B == 3 ? Jp + 1 : Jp, role, q, col);
  L16_STAMP(4)
  return ok;
}

__device__ __forceinline__ void l16_run(L16Shared& S, const double2* tab, int n, int n4, int nt, double c, double nl, int pb,
                                        double* f_out, double* g_out) {
  const int lane = threadIdx.x & 63, q = lane >> 4, col = lane & 15;
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  L16Acc A;
#ifdef GPET_L16_PROF
  const long long t_entry = clock64();
#endif
  // tiles.  First every slot as if it were an off-diagonal tile of training points: correlation from the table only, all
  // loads independent of each other (the lattice coordinates are zero beyond the training set: any lag is inside the
  // table).  Then the special tiles again, properly: diagonal tiles and the tile row with the padding and the border.
  {
    int mi[L16_NG][4], mjv[L16_NT];
#pragma unroll
    for (int g = 0; g < L16_NG; ++g) {
      const int I = l16_row(g, role) < 0 ? 0 : l16_row(g, role);
#pragma unroll
      for (int r = 0; r < 4; ++r) mi[g][r] = S.m[16 * I + q + 4 * r];
    }
#pragma unroll
    for (int J = 0; J < L16_NT; ++J) mjv[J] = S.m[16 * J + col];
#pragma unroll
    for (int t = 0; t < L16_NS; ++t) {
      const int g = l16_sg(t), J = l16_sj(t);
#pragma unroll
      for (int r = 0; r < 4; ++r) A.v[t][r] = c * tab[abs(mi[g][r] - mjv[J])].x;
    }
  }
#pragma unroll
  for (int t = 0; t < L16_NS; ++t) {
    const int I = l16_row(l16_sg(t), role), J = l16_sj(t);
    const int j = 16 * J + col;
    if (J > I) {  // (a slot this wave does not use)
      A.v[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    } else if (J < I && 16 * I + 16 > n) {
      // off-diagonal tile of a tile row that reaches beyond the training set: kernel entries where both indices are
      // training points, y in the border row, zero elsewhere (no diagonal, no padding pivot in it)
      const double yj = S.y[j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = 16 * I + q + 4 * r;
        A.v[t][r] = (ii < n && j < n) ? A.v[t][r] : ((ii == n4 && j < n) ? yj : 0.0);
      }
    } else if (J == I) {
      // diagonal tile: every candidate value is loaded (the arrays are zero beyond the training set) and the right one
      // selected -- no divergent branches
      const double yj = S.y[j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = 16 * I + q + 4 * r;
        const int ic = (ii >= 0 && ii < L16_WS) ? ii : 0;
        const double dg = (c + nl * S.w[ic]) + 1e-6;
        const double yi = S.y[ic];
        const bool in = ii < n && j < n;
        double v = in ? (ii == j ? dg : A.v[t][r]) : 0.0;
        v = (!in && ii == j && ii < n4) ? 1.0 : v;
        v = (ii == n4 && j < n) ? yj : v;
        v = (j == n4 && ii < n) ? yi : v;
        A.v[t][r] = v;
      }
    }
  }
  l16_extract<0>(A, &S.Wt[0][0][0], 0, role, q, col);
  __syncthreads();
  bool ok = true;
#ifdef GPET_L16_PROF
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tl = clock64();
  const long long t_begin = tl;
#define L16_PROF_ARGS , prof, tl
#else
#define L16_PROF_ARGS
#endif
  for (int Jp = 0; 16 * Jp < n4; ++Jp) {
    ok = l16_step<0>(A, S, Jp, n4, role, q, col L16_PROF_ARGS) && ok;
    __syncthreads();
    L16_STAMP(5)
    if (16 * Jp + 4 < n4) ok = l16_step<1>(A, S, Jp, n4, role, q, col L16_PROF_ARGS) && ok;
    __syncthreads();
    L16_STAMP(5)
    if (16 * Jp + 8 < n4) ok = l16_step<2>(A, S, Jp, n4, role, q, col L16_PROF_ARGS) && ok;
    __syncthreads();
    L16_STAMP(5)
    if (16 * Jp + 12 < n4) ok = l16_step<3>(A, S, Jp, n4, role, q, col L16_PROF_ARGS) && ok;
    __syncthreads();
    L16_STAMP(5)
  }
#ifdef GPET_L16_PROF
  const long long t_loop_end = clock64();
#endif
  if (!ok) {  // sklearn returns (-inf, 0) -> objective (+inf, -0)
    if (threadIdx.x == 0) {
      f_out[pb] = INFINITY;
      g_out[3 * pb] = g_out[3 * pb + 1] = g_out[3 * pb + 2] = 0.0;
    }
    return;
  }
  // alpha = row n4 of the swept matrix, its corner = -y^T alpha
  const int It = n4 >> 4, rq = n4 & 15, qs = rq & 3, rs = rq >> 2;
#pragma unroll
  for (int t = 0; t < L16_NS; ++t) {
    const int I = l16_row(l16_sg(t), role), J = l16_sj(t);
    if (I == It && J <= I) {
      const double v = l16_sel(A.v[t], rs);
      if (q == qs) S.al[16 * J + col] = v;
    }
  }
  __syncthreads();
  const double* al = S.al;
  double ld = 0.0;
  for (int k = (int)threadIdx.x; k < n; k += 128) ld += log(sqrt(S.piv[k]));  // (both waves: a log is ~80 instructions)
  // gradient: 0.5 * sum_ij (alpha_i alpha_j - Kinv_ij) dK_ij; off-diagonal tiles stand for both triangles.
  // sR, sD: sums of inner_ij R_ij and inner_ij dR_ij over i != j (c applied at the end); gd, gn: the diagonal terms.
  // Same two passes as the set-up: every slot that is an off-diagonal tile of training points without a branch inside,
  // then the special tiles element by element with selects.
  double sR = 0.0, sD = 0.0, gd = 0.0, gn = 0.0;
  double ai[L16_NG][4];
  int mi[L16_NG][4], mjv[L16_NT];
#pragma unroll
  for (int g = 0; g < L16_NG; ++g) {
    const int I = l16_row(g, role) < 0 ? 0 : l16_row(g, role);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ai[g][r] = al[16 * I + q + 4 * r];
      mi[g][r] = S.m[16 * I + q + 4 * r];
    }
  }
#pragma unroll
  for (int J = 0; J < L16_NT; ++J) mjv[J] = S.m[16 * J + col];
#pragma unroll
  for (int t = 0; t < L16_NS; ++t) {
    const int g = l16_sg(t), J = l16_sj(t);
    const int I = l16_row(g, role);
    const int j = 16 * J + col;
    const double aj = al[j];
    if (J < I && 16 * I + 16 <= n) {  // (all rows and columns are training points, none on the diagonal)
      double tR = 0.0, tD = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double inner = fma(ai[g][r], aj, A.v[t][r]);  // T = -Kinv
        const double2 rd = tab[abs(mi[g][r] - mjv[J])];
        tR = fma(inner, rd.x, tR);
        tD = fma(inner, rd.y, tD);
      }
      sR = fma(2.0, tR, sR);
      sD = fma(2.0, tD, sD);
    } else if (J < I) {  // (off-diagonal tile of a tile row that reaches beyond the training set)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = 16 * I + q + 4 * r;
        const double inner = fma(ai[g][r], aj, A.v[t][r]);
        const double2 rd = tab[abs(mi[g][r] - mjv[J])];
        const bool od = ii < n && j < n;
        sR += od ? 2.0 * inner * rd.x : 0.0;
        sD += od ? 2.0 * inner * rd.y : 0.0;
      }
    } else if (J == I) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = 16 * I + q + 4 * r;
        const int ic = (ii >= 0 && ii < L16_WS) ? ii : 0;
        const double inner = fma(ai[g][r], aj, A.v[t][r] - ((ii == j) ? 2.0 : 0.0));  // (the pivot blocks hold 2 I - P^-1: l16_step)
        const double2 rd = tab[abs(mi[g][r] - mjv[J])];
        const bool in = ii < n && j < n;
        const bool dg = in && ii == j, od = in && ii != j;
        gd += dg ? inner : 0.0;
        gn += dg ? inner * (nl * S.w[ic]) : 0.0;
        sR += od ? inner * rd.x : 0.0;
        sD += od ? inner * rd.y : 0.0;
      }
    }
  }
  double gc = c * (sR + gd), gl = c * sD;
  ld = wave_sum(ld);
  gc = wave_sum(gc);
  gl = wave_sum(gl);
  gn = wave_sum(gn);
  if (role == 1 && lane == 0) {
    S.red[0] = gc;
    S.red[1] = gl;
    S.red[2] = gn;
    S.red[3] = ld;
  }
  __syncthreads();
#ifdef GPET_L16_PROF
  if (blockIdx.x == 0 && lane == 0)
    printf("l16 role %d: set-up %lld, loop %lld (solve %lld, operands %lld, mfma %lld, diag %lld, extract %lld, barrier %lld), epilogue %lld cycles\n",
           role, (long long)(t_begin - t_entry), (long long)(t_loop_end - t_begin), prof[0], prof[1], prof[2], prof[3], prof[4], prof[5],
           (long long)(clock64() - t_loop_end));
#endif
  if (threadIdx.x == 0) {
    gc += S.red[0];
    gl += S.red[1];
    gn += S.red[2];
    ld += S.red[3];
    const double yta = -al[n4];
    const double lml = -0.5 * yta - ld - 0.5 * (double)n * 1.8378770664093453;  // log(2 pi)
    f_out[pb] = -lml;
    g_out[3 * pb] = -0.5 * gc;
    g_out[3 * pb + 1] = -0.5 * gl;
    g_out[3 * pb + 2] = -0.5 * gn;
  }
}

#ifndef GPET_L16_WAVES
#define GPET_L16_WAVES 2  // waves per SIMD (3 spills 44 doubles in the block step: tools/build_instrumented.sh -DGPET_L16_WAVES=3 to try)
#endif
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(GPET_L16_WAVES, GPET_L16_WAVES))) k_lml16(EdgeDev* edges, const int* edge_of, const double* theta, double* f_out,
                                               double* g_out, const int* count, int lag_cap) {
  const int pb = blockIdx.x;
#ifdef GPET_L16_PROF
  const long long t_kernel = clock64();
#endif
  if (count != nullptr && pb >= *count) return;  // (launches are sized by the host's last KNOWN number of running problems)
  const EdgeDev E = edges[edge_of[pb]];
  const int n = E.fin_n;
  const int n4 = (n + 3) & ~3;
  const int nt = (n4 + 1 + 15) >> 4;
  __shared__ L16Shared S;
  extern __shared__ double2 l16_tab[];  // [lag_cap]: (correlation, d/dlog(length scale)) at the lattice lags
  const int tid = threadIdx.x;
  const double c = exp(theta[3 * pb]), ell = exp(theta[3 * pb + 1]), nl = exp(theta[3 * pb + 2]);
  const double hinv = E.fin_par[9];
  const int lagmax = (int)E.fin_par[10];
  if (!(hinv > 0.0) || lagmax < 0 || lagmax >= lag_cap || n > L16_MAXN) {  // the host routes such sets to k_lml / k_lml2: fail loudly
    if (tid == 0) {
      f_out[pb] = NAN;
      g_out[3 * pb] = g_out[3 * pb + 1] = g_out[3 * pb + 2] = 0.0;
      E.sc->status = GPET_ERR_STATE;
    }
    return;
  }
  const double x0 = E.fin_x[0];
  for (int i = tid; i < L16_WS; i += 128) {
    const bool in = i < n;
    S.y[i] = in ? E.fin_y[i] : 0.0;
    S.w[i] = in ? E.fin_w[i] : 0.0;
    S.m[i] = in ? (int)rint((E.fin_x[i] - x0) * hinv) : 0;
    S.piv[i] = 1.0;
    S.al[i] = 0.0;
  }
  // the zero fill of both panel buffers: rows beyond the matrix are never written but are read as operands of tiles
  // nobody uses
  for (int i = tid; i < 2 * 4 * L16_WS; i += 128) (&S.Wt[0][0][0])[i] = 0.0;
  // the correlation as sklearn evaluates it for two inputs a lag apart: a = x_i / l, b = x_j / l, d = a - b
  for (int m = tid; m <= lagmax; m += 128) {
    double R, dR;
    corr_and_dlog(E, ((double)m / hinv) / ell, 0.0, R, dR);
    l16_tab[m] = make_double2(R, dR);
  }
  __syncthreads();
#ifdef GPET_L16_PROF
  if (blockIdx.x == 0 && tid == 0) printf("l16 staging + table: %lld cycles\n", (long long)(clock64() - t_kernel));
#endif
  l16_run(S, l16_tab, n, n4, nt, c, nl, pb, f_out, g_out);
}

// The converged fit of one (edge, restart) problem from its start point to its optimum in ONE workgroup: objective
// (the block sweep above) -> L-BFGS-B state machine (lb_advance, thread 0, the problem's state in LDS) -> next trial
// point, until the machine says done.  Problems do not talk to each other, so there is nothing to synchronise across
// workgroups and no residency requirement: a launch of P workgroups of which only some fit the GPU just runs them in
// generations.  Against the round-based driver (one objective launch + one k_lb_advance launch per round of ALL running
// problems, a counter read-back every fourth round) this takes the launch gaps and the host round trips out of a
// problem's chain of ~50 evaluations and stages its training set once instead of once per evaluation.
// counters[0] += evaluations, counters[1] = max evaluations of a problem, counters[2] += problems cut off at max_evals.
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(GPET_L16_WAVES, GPET_L16_WAVES))) k_lml16_fit(EdgeDev* edges, LbProb* probs, LbCfg cfg, int lag_cap,
                                                                                                                 int max_evals, int* counters) {
  const int pb = blockIdx.x;
  __shared__ L16Shared S;
  __shared__ LbProb sp;
  __shared__ double s_fg[4];
  extern __shared__ double2 l16_tab[];
  const int tid = threadIdx.x;
  {
    const int* src = reinterpret_cast<const int*>(&probs[pb]);
    int* dst = reinterpret_cast<int*>(&sp);
    for (int i = tid; i < (int)(sizeof(LbProb) / sizeof(int)); i += 128) dst[i] = src[i];
  }
  __syncthreads();
  // (only what the loop needs of the edge's record.  The kernel still spills -- 114 vector registers around the set-up,
  //  the gradient epilogue and the call of the state machine, none inside the block steps)
  const EdgeDev* Ep = &edges[sp.edge];
  EdgeDev E = {};
  E.kernel_type = Ep->kernel_type;
  E.nu_code = Ep->nu_code;
  E.nu_gen = Ep->nu_gen;
  E.inv_gamma_nu = Ep->inv_gamma_nu;
  E.fin_n = Ep->fin_n;
  E.fin_par = Ep->fin_par;
  E.fin_x = Ep->fin_x;
  E.fin_y = Ep->fin_y;
  E.fin_w = Ep->fin_w;
  E.sc = Ep->sc;
  const int n = E.fin_n;
  const int n4 = (n + 3) & ~3;
  const int nt = (n4 + 1 + 15) >> 4;
  const double hinv = E.fin_par[9];
  const int lagmax = (int)E.fin_par[10];
  if (!(hinv > 0.0) || lagmax < 0 || lagmax >= lag_cap || n > L16_MAXN) {  // (the host routes such sets to the round-based driver)
    if (tid == 0) {
      E.sc->status = GPET_ERR_STATE;
      atomicAdd(&counters[2], 1);
    }
    return;
  }
  const double x0 = E.fin_x[0];
  for (int i = tid; i < L16_WS; i += 128) {
    const bool in = i < n;
    S.y[i] = in ? E.fin_y[i] : 0.0;
    S.w[i] = in ? E.fin_w[i] : 0.0;
    S.m[i] = in ? (int)rint((E.fin_x[i] - x0) * hinv) : 0;
  }
  int evals = 0;
  for (;;) {
    const double c = exp(sp.xe[0]), ell = exp(sp.xe[1]), nl = exp(sp.xe[2]);
    for (int i = tid; i < L16_WS; i += 128) {
      S.piv[i] = 1.0;
      S.al[i] = 0.0;
    }
    for (int i = tid; i < 2 * 4 * L16_WS; i += 128) (&S.Wt[0][0][0])[i] = 0.0;
    for (int m = tid; m <= lagmax; m += 128) {
      double R, dR;
      corr_and_dlog(E, ((double)m / hinv) / ell, 0.0, R, dR);
      l16_tab[m] = make_double2(R, dR);
    }
    __syncthreads();
    l16_run(S, l16_tab, n, n4, nt, c, nl, 0, &s_fg[0], &s_fg[1]);  // (thread 0 leaves f and the gradient in s_fg)
    ++evals;
    if (tid == 0) {
      double lo[3] = {cfg.lo[0], cfg.lo[1], cfg.lo[2]}, hi[3] = {cfg.hi[0], cfg.hi[1], cfg.hi[2]};
      const double gk[3] = {s_fg[1], s_fg[2], s_fg[3]};
      lb_advance(sp, s_fg[0], gk, lo, hi);
    }
    __syncthreads();
    if (sp.task == LB_TASK_DONE || evals >= max_evals) break;
  }
  {
    int* dst = reinterpret_cast<int*>(&probs[pb]);
    const int* src = reinterpret_cast<const int*>(&sp);
    for (int i = tid; i < (int)(sizeof(LbProb) / sizeof(int)); i += 128) dst[i] = src[i];
  }
  if (tid == 0) {
    atomicAdd(&counters[0], evals);
    atomicMax(&counters[1], evals);
    if (sp.task != LB_TASK_DONE) atomicAdd(&counters[2], 1);
  }
}

bool lml16_fit_applies(int n_max, int lag_cap) {
  return n_max <= L16_MAXN && lag_cap > 0 && lag_cap <= L16_LAG_MAX && gpet_opt_lml_mfma() != 0;
}

hipError_t launch_lml16_fit(hipStream_t st, EdgeDev* d_edges, void* d_probs, int P, const LbCfg& cfg, int lag_cap, int max_evals,
                            int* d_counters) {
  (void)hipGetLastError();
  const size_t dyn = (size_t)2 * lag_cap * sizeof(double);
  hipLaunchKernelGGL(k_lml16_fit, dim3(P), dim3(128), dyn, st, d_edges, static_cast<LbProb*>(d_probs), cfg, lag_cap, max_evals,
                     d_counters);
  return hipGetLastError();
}

hipError_t launch_lml(hipStream_t st, EdgeDev* d_edges, int P, int n_max, const int* d_edge_of, const double* d_theta,
                      double* d_f, double* d_g, const int* d_count, int lag_cap) {
  (void)hipGetLastError();
  if (n_max > 250) return hipErrorInvalidValue;
  // up to 108 training points on a lattice (lag_cap > 0: every training set of the launch sits on one with fewer than
  // lag_cap points; the caller knows): the block sweep on the matrix cores, two waves per problem, correlation tables
  // of lag_cap entries in dynamic LDS
  if (n_max <= L16_MAXN && lag_cap > 0 && lag_cap <= L16_LAG_MAX && gpet_opt_lml_mfma()) {
    const size_t dyn = (size_t)2 * lag_cap * sizeof(double);
    hipLaunchKernelGGL(k_lml16, dim3(P), dim3(128), dyn, st, d_edges, d_edge_of, d_theta, d_f, d_g, d_count, lag_cap);
    return hipGetLastError();
  }
  // two tiles per thread: the only form above 128 training points, and the faster one for big launches (fewer
  // instructions per problem: 109 instead of 134 us at 900 problems of 98 points, 1.07 instead of 1.41 ms at 13312) --
  // small launches are latency-bound and keep one tile per thread (64 instead of 80 us at 256 problems)
  if (n_max > 128 || gpet_opt_lml_two_tiles_from() <= P) {
    const int nb2 = (n_max + 1 + 3) >> 2;
    const int tiles = nb2 * (nb2 + 1) / 2;
    int threads = (((tiles + 1) / 2 + 63) / 64) * 64;
    if (threads > 1024) threads = 1024;
    hipLaunchKernelGGL(k_lml2, dim3(P), dim3(threads), 0, st, d_edges, d_edge_of, d_theta, d_f, d_g, d_count);
    return hipGetLastError();
  }
  const int nb = (n_max + 1 + 3) >> 2;
  const int threads = ((nb * (nb + 1) / 2 + 63) / 64) * 64;  // one thread per 4x4 tile of the lower triangle
  hipLaunchKernelGGL(k_lml, dim3(P), dim3(threads), 0, st, d_edges, d_edge_of, d_theta, d_f, d_g, d_count);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// f2 for MORE than 250 training points (wide edges with a small delta_x): the objective of one L-BFGS-B problem is
// evaluated with the blocked HBM kernels of the many-point fit -- every problem becomes a VIRTUAL edge of a scratch
// table (training set = the edge's standardised set, K / L / alpha / L^-1 in per-problem scratch, amplitude, length
// scale and noise level = the problem's theta):
//   K tiles -> blocked Cholesky (MFMA SYRK) -> alpha -> X = L^-1 (blocked substitution on the identity) ->
//   tiles of K^-1 = X^T X on the matrix cores, contracted on the fly with dK/dtheta (k_lmlbig_grad) -> f, g.
// ~2.5 n^3 flops per evaluation instead of n^3, ~100 launches per round: a rare path, kept simple.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_lmlbig_setup(EdgeDev* edges, int P, const int* edge_of, const double* theta,
                                                     EdgeDev* vedges, gpet_scalars* vsc, double* scratch, size_t per_prob,
                                                     int ncap_v) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const EdgeDev E = edges[edge_of[p]];
  EdgeDev V = E;
  const double c = exp(theta[3 * p]), l = exp(theta[3 * p + 1]), nl = exp(theta[3 * p + 2]);
  double* base = scratch + (size_t)p * per_prob;
  V.xt = E.fin_x;
  V.yt = E.fin_y;
  V.wt = E.fin_w;
  V.K = base;
  V.V = base + (size_t)ncap_v * ncap_v;
  V.alpha = V.V + (size_t)ncap_v * ncap_v;
  V.chol_inv = nullptr;  // (the blocked objective solves by substitution: k_chol_trsm_sub, k_struct_trsm)
  V.n_cap = ncap_v;
  V.Lg = ncap_v;  // column count / stride of V in the blocked substitution; != n, so the weights are not zeroed again
  V.length_scale = l;
  V.noise_y = nl;
  V.jitter = 1e-6;
  V.structured = 0;
  V.tab_ok = 0;  // standardised coordinates, the problem's own length scale
  V.sc = vsc + p;
  gpet_scalars s;
  s.y_s = 1.0;
  s.amp = c;
  s.y_mean = 0.0;
  s.y_std = 1.0;
  s.score_thresh = 0.0;
  s.lml = 0.0;
  s.n = E.fin_n;
  s.n_obs = 0;
  s.rank = 0;
  s.status = GPET_OK;
  s.iter = 0;
  s.done = 0;
  s.n_removed = 0;
  s.force = 1;
  vsc[p] = s;
  vedges[p] = V;
}

__global__ void __launch_bounds__(256) k_lmlbig_identity(EdgeDev* vedges) {
  const EdgeDev E = vedges[blockIdx.y];
  const int n = E.sc->n, ldu = E.Lg;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < (size_t)n * ldu; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / ldu), j = (int)(e - (size_t)i * ldu);
    E.V[e] = (i == j) ? 1.0 : 0.0;
  }
}

// tile (bi >= bj) of K^-1 = X^T X, X = L^-1 rows in V, contracted with the three dK/dtheta: partial sums per tile
__global__ void __launch_bounds__(256) k_lmlbig_grad(EdgeDev* vedges, double* part, int nt) {
  const EdgeDev E = vedges[blockIdx.z];
  const gpet_scalars* sc = E.sc;
  const int bi = blockIdx.y, bj = blockIdx.x;
  double* out = part + ((size_t)blockIdx.z * nt * nt + (size_t)bi * nt + bj) * 3;
  const int tid = threadIdx.x;
  if (tid < 3) out[tid] = 0.0;
  if (sc->status != GPET_OK || bj > bi) return;
  const int n = sc->n, ldu = E.Lg;
  const int i0 = bi * 64, j0 = bj * 64;
  if (i0 >= n) return;
  __shared__ double sr[32][65];
  __shared__ double sc_[32][65];
  __shared__ double s_red[16];
  const int lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  v4f64 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
  // X is lower triangular: rows k < i0 contribute nothing to columns >= i0
  for (int k0 = (i0 / 32) * 32; k0 < n; k0 += 32) {
    for (int e = tid; e < 32 * 64; e += 256) {
      const int jj = e & 63, kk = e >> 6;
      const int k = k0 + kk;
      sr[kk][jj] = (k < n && i0 + jj < n) ? E.V[(size_t)k * ldu + i0 + jj] : 0.0;
      sc_[kk][jj] = (k < n && j0 + jj < n) ? E.V[(size_t)k * ldu + j0 + jj] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; kk += 4) {
      const double a = sr[kk + lq][16 * w + li];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sc_[kk + lq][16 * t + li], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }
  const double c = sc->amp, length = E.length_scale, nl = E.noise_y;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = j0 + 16 * t + li;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = i0 + 16 * w + lq + 4 * g;
      if (i >= n || j > i) continue;
      const double inner = E.alpha[i] * E.alpha[j] - acc[t][g];
      if (i == j) {
        s1 += inner * c;            // d K / d log c on the diagonal: c * 1
        s3 += inner * nl * E.wt[i]; // d K / d log noise_level
      } else {
        double R, dR;
        corr_and_dlog(E, E.xt[i] / length, E.xt[j] / length, R, dR);
        s1 += 2.0 * inner * (c * R);
        s2 += 2.0 * inner * (c * dR);
      }
    }
  }
  s1 = block_sum(s1, s_red);
  s2 = block_sum(s2, s_red);
  s3 = block_sum(s3, s_red);
  if (tid == 0) {
    out[0] = s1;
    out[1] = s2;
    out[2] = s3;
  }
}

// f = -log marginal likelihood, g = its gradient (sklearn_gpr.py:521-585); non-PD -> +inf, 0
__global__ void __launch_bounds__(256) k_lmlbig_finish(EdgeDev* vedges, const double* part, int nt, double* f_out, double* g_out) {
  const int p = blockIdx.x;
  const EdgeDev E = vedges[p];
  const gpet_scalars* sc = E.sc;
  __shared__ double s_red[16];
  const int tid = threadIdx.x, n = sc->n, ld = E.n_cap;
  if (sc->status != GPET_OK) {
    if (tid == 0) {
      f_out[p] = INFINITY;
      g_out[3 * p] = g_out[3 * p + 1] = g_out[3 * p + 2] = 0.0;
    }
    return;
  }
  double ya = 0.0, ld_ = 0.0;
  for (int i = tid; i < n; i += blockDim.x) {
    ya += E.yt[i] * E.alpha[i];
    ld_ += log(E.K[(size_t)i * ld + i]);
  }
  ya = block_sum(ya, s_red);
  ld_ = block_sum(ld_, s_red);
  if (tid == 0) {
    double s[3] = {0.0, 0.0, 0.0};
    const double* pp = part + (size_t)p * nt * nt * 3;
    for (int t = 0; t < nt * nt; ++t)  // fixed order: reproducible sums
      for (int q = 0; q < 3; ++q) s[q] += pp[3 * t + q];
    f_out[p] = 0.5 * ya + ld_ + 0.5 * (double)n * 1.8378770664093453;  // log(2 pi)
    for (int q = 0; q < 3; ++q) g_out[3 * p + q] = -0.5 * s[q];
  }
}

size_t lmlbig_scratch_doubles(int ncap_v) { return 2 * (size_t)ncap_v * ncap_v + ncap_v; }

hipError_t launch_lml_big(hipStream_t st, EdgeDev* d_edges, int P, int n_max, const int* d_edge_of, const double* d_theta,
                          double* d_f, double* d_g, void* d_vedges, void* d_vsc, double* d_scratch, double* d_part,
                          int ncap_v) {
  (void)hipGetLastError();
  EdgeDev* ve = (EdgeDev*)d_vedges;
  const int nt = ncap_v / CB;
  hipLaunchKernelGGL(k_lmlbig_setup, dim3(cdiv(P, 64)), dim3(64), 0, st, d_edges, P, d_edge_of, d_theta, ve,
                     (gpet_scalars*)d_vsc, d_scratch, lmlbig_scratch_doubles(ncap_v), ncap_v);
  const int ntn = cdiv(n_max, CB);  // tiles that can hold training points
  hipLaunchKernelGGL(k_fit_kbuild, dim3(ntn, ntn, P), dim3(256), 0, st, ve);
  for (int k0 = 0; k0 < n_max; k0 += CB) {
    hipLaunchKernelGGL(k_chol_diag, dim3(1, P), dim3(256), 0, st, ve, k0, 0);
    const int below = cdiv(n_max - k0 - CB, CB);
    if (below > 0) {
      hipLaunchKernelGGL(k_chol_trsm_sub, dim3(below, P), dim3(256), 0, st, ve, k0);
      hipLaunchKernelGGL(k_chol_syrk, dim3(below, below, P), dim3(256), 0, st, ve, k0, 0);
    }
  }
  hipLaunchKernelGGL(k_chol_solve, dim3(1, P), dim3(256), (size_t)ncap_v * sizeof(double), st, ve);
  hipLaunchKernelGGL(k_lmlbig_identity, dim3(256, P), dim3(256), 0, st, ve);
  const int cgroups = cdiv(n_max, SB_COLS);
  for (int k0 = 0; k0 < n_max; k0 += CB) hipLaunchKernelGGL(k_struct_trsm, dim3(cgroups, P), dim3(256), 0, st, ve, k0, 1);
  hipLaunchKernelGGL(k_lmlbig_grad, dim3(nt, nt, P), dim3(256), 0, st, ve, d_part, nt);
  hipLaunchKernelGGL(k_lmlbig_finish, dim3(P), dim3(256), 0, st, ve, d_part, nt, d_f, d_g);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------

hipError_t launch_conv(hipStream_t st, const double* d_img, int M, int N, const double* d_wf, int kh, int kw, int oy,
                       int ox, float* d_tmp, unsigned int* d_minmax) {
  (void)hipGetLastError();  // drop stale errors: report only these launches
  dim3 bs(64, 4), gs(cdiv(N, 64), cdiv(M, CONV_RY));
  const size_t lds = ((size_t)kh * kw + (size_t)(CONV_RY + kh - 1) * (64 + kw - 1)) * sizeof(double);
  if (lds > 64 * 1024) return hipErrorInvalidValue;  // (a kernel of hundreds of taps per side: not the reference's use)
  hipLaunchKernelGGL(k_conv_relu, gs, bs, lds, st, d_img, M, N, d_wf, kh, kw, oy, ox, d_tmp, d_minmax);
  return hipGetLastError();
}
hipError_t launch_minmax(hipStream_t st, const float* d_in, size_t count, unsigned int* d_minmax) {
  (void)hipGetLastError();  // drop stale errors: report only these launches
  int blocks = (int)((count + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_minmax_f32, dim3(blocks), dim3(256), 0, st, d_in, count, d_minmax);
  return hipGetLastError();
}
hipError_t launch_normalise(hipStream_t st, const float* d_in, size_t count, const unsigned int* d_minmax,
                            float* d_out) {
  (void)hipGetLastError();  // drop stale errors: report only these launches
  int blocks = (int)((count + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_normalise_f32, dim3(blocks), dim3(256), 0, st, d_in, count, d_minmax, d_out);
  return hipGetLastError();
}

// ---- a5 for many training points (generic path): K_*^T rows into V, mean, blocked V = L^-1 K_*^T, std ----
__global__ void __launch_bounds__(256) k_kstar_build(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.z];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int n = sc->n, Lg = E.Lg;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const double amp = sc->amp, length = E.length_scale;
  if (j >= Lg) return;
  const double xq = (double)(E.x_st + j) / length;
  for (int i = blockIdx.y * 4 + (threadIdx.x >> 6); i < n; i += gridDim.y * 4)
    E.V[(size_t)i * Lg + j] = amp * corr_px(E, (double)(E.x_st + j), E.xt[i], length);
}
// block k0 of V = L^-1 K_*^T for 32 columns of the grid per workgroup, on the matrix cores:
//   X = K_*^T[k0.., cols] - sum_{j0 < k0} L[k0.., j0..] V[j0.., cols]   (64 x 64 by 64 x 32 products, left-looking),
//   V[k0.., cols] = L_kk^-1 X   (the inverse of the diagonal block from k_chol_diag).
// Wave w owns rows 16 w .. 16 w + 15 of the block and both 16-column halves.
#ifndef VS_COLS
#define VS_COLS 16  // grid columns per workgroup (16, 32 or 64 by -DVS_COLS=..; at n = 1500, Lg = 2048: 3.42 / 3.62 / 4.09 ms per fit + predict + covariance)
#endif
#define VS_NH (VS_COLS / 16)          // 16-column halves per wave
#define VS_PU (CB * VS_COLS / 256)    // prefetch registers of the V block per thread
__global__ void __launch_bounds__(256) k_vsolve_mfma(EdgeDev* edges, int k0) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int n = sc->n, Lg = E.Lg, ld = E.n_cap;
  const int a0 = blockIdx.x * VS_COLS;
  if (k0 >= n || a0 >= Lg) return;
  const int nb = (n - k0) < CB ? (n - k0) : CB;
  __shared__ double sL[CB][CB + 1];
  __shared__ double sU[CB][VS_COLS + 1];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  v4f64c acc[VS_NH];
#pragma unroll
  for (int h = 0; h < VS_NH; ++h) acc[h] = (v4f64c){0.0, 0.0, 0.0, 0.0};
  // the blocks of the next j0 are loaded into registers while the matrix cores work on the current ones (a launch has
  // only Lg / 32 workgroups: nothing else hides the two dependent round trips per block otherwise -- 85 -> ~25 us)
  double pl[16], pu[VS_PU];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = tid + 256 * u, i = e >> 6, t = e & 63;
      pl[u] = (i < nb) ? E.K[(size_t)(k0 + i) * ld + j0 + t] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < VS_PU; ++u) {
      const int e = tid + 256 * u, t = e / VS_COLS, a = e % VS_COLS;
      pu[u] = (a0 + a < Lg) ? E.V[(size_t)(j0 + t) * Lg + a0 + a] : 0.0;
    }
  };
  if (k0 > 0) fetch(0);
  for (int j0 = 0; j0 < k0; j0 += CB) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = tid + 256 * u;
      sL[e >> 6][e & 63] = pl[u];
    }
#pragma unroll
    for (int u = 0; u < VS_PU; ++u) {
      const int e = tid + 256 * u;
      sU[e / VS_COLS][e % VS_COLS] = pu[u];
    }
    __syncthreads();
    if (j0 + CB < k0) fetch(j0 + CB);
#pragma unroll
    for (int kk = 0; kk < CB; kk += 4) {
      const double a = sL[16 * w + li][kk + lq];
#pragma unroll
      for (int h = 0; h < VS_NH; ++h) acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sU[kk + lq][16 * h + li], acc[h], 0, 0, 0);
    }
  }
  __syncthreads();
  // X = B - acc into sU (rows of the block beyond n: zero), the inverse into sL
  const double* inv = E.chol_inv + (size_t)(k0 / CB) * CB * CB;
  for (int e = tid; e < CB * CB; e += 256) sL[e >> 6][e & 63] = inv[e];
#pragma unroll
  for (int h = 0; h < VS_NH; ++h)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = 16 * w + lq + 4 * g, a = 16 * h + li;
      sU[i][a] = (i < nb && a0 + a < Lg) ? E.V[(size_t)(k0 + i) * Lg + a0 + a] - acc[h][g] : 0.0;
    }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < VS_NH; ++h) acc[h] = (v4f64c){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < CB; kk += 4) {
    const double a = sL[16 * w + li][kk + lq];
#pragma unroll
    for (int h = 0; h < VS_NH; ++h) acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sU[kk + lq][16 * h + li], acc[h], 0, 0, 0);
  }
#pragma unroll
  for (int h = 0; h < VS_NH; ++h)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = 16 * w + lq + 4 * g, a = 16 * h + li;
      if (i < nb && a0 + a < Lg) E.V[(size_t)(k0 + i) * Lg + a0 + a] = acc[h][g];
    }
}

// Column sums over the n rows of V for many training points: a workgroup owns 16 grid columns, its 16 row-lanes take every
// sixteenth row each (independent loads, a wave reads four rows of 128 contiguous bytes per instruction) and their
// partial sums are added in row-lane order.  (One thread per column walking all n rows -- Lg / 256 workgroups on the
// whole GPU -- took 0.39 + 0.36 ms at n = 1500, Lg = 2048.)
//   STD = false: mean_j = y_std * sum_i K_*[i][j] alpha_i + y_mean   (sklearn_gpr.py:381-385; before V is overwritten)
//   STD = true:  std_j = sqrt(max(amp - sum_i V[i][j]^2, 0) * y_std^2)   (sklearn_gpr.py:414-436)
#define PB_COLS 16
#define PB_LANES 16
template <bool STD>
__global__ void __launch_bounds__(PB_COLS * PB_LANES) k_pred_colsum_big(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  __shared__ double s_part[PB_LANES][PB_COLS + 1];
  const int n = sc->n, Lg = E.Lg;
  const int c = threadIdx.x & (PB_COLS - 1), r = threadIdx.x / PB_COLS;
  const int j = blockIdx.x * PB_COLS + c;
  const GPET_GLOBAL double* __restrict__ Vg = as_global(E.V);
  const GPET_GLOBAL double* __restrict__ al = as_global(E.alpha);
  double sum = 0.0;
  if (j < Lg) {
#pragma unroll 8
    for (int i = r; i < n; i += PB_LANES) {
      const double v = Vg[(size_t)i * Lg + j];
      sum += STD ? v * v : v * al[i];
    }
  }
  s_part[r][c] = sum;
  __syncthreads();
  if (r == 0 && j < Lg) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < PB_LANES; ++q) t += s_part[q][c];
    if (STD) {
      double var = sc->amp - t;
      if (var < 0.0) var = 0.0;
      E.std[j] = sqrt(var * (sc->y_std * sc->y_std));
    } else {
      E.mean[j] = sc->y_std * t + sc->y_mean;
    }
  }
}

// hipFuncSetAttribute applies to the CURRENT device: remember per device what has been raised, so that one process
// may hold contexts on several GPUs (the launchers run with the context's device current).
struct PerDeviceOnce {
  bool done[64] = {};
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
};

static void fit_predict_attrs() {
  static PerDeviceOnce once;
  if (!once.first()) return;
  (void)hipFuncSetAttribute((const void*)k_fit<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  (void)hipFuncSetAttribute((const void*)k_fit<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  (void)hipFuncSetAttribute((const void*)k_predict<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  (void)hipFuncSetAttribute((const void*)k_predict<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
}

hipError_t launch_fit_predict(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, int want_cov, unsigned parts) {
  (void)hipGetLastError();  // drop stale errors: report only these launches
  fit_predict_attrs();
  if (!(parts & 1u)) {
  } else if (bd.n_cap <= 128)
    hipLaunchKernelGGL((k_fit<true, false>), dim3(1, B), dim3(576),
                       ((size_t)bd.n_cap * (bd.n_cap | 1) + bd.n_cap) * sizeof(double), st, d_edges);
  else
    launch_fit_blocked(st, d_edges, B, bd);
  const size_t plds = ((size_t)bd.n_cap * 64 + 3 * (size_t)bd.n_cap) * sizeof(double);
  if (!(parts & 2u)) {
  } else if (bd.n_cap > 128 && plds > 150 * 1024) {
    // many training points: V through HBM, blocked substitution with the panel kernel of the structured path
    hipLaunchKernelGGL(k_kstar_build, dim3(cdiv(bd.Lg, 64), 64, B), dim3(256), 0, st, d_edges);
    hipLaunchKernelGGL(k_pred_colsum_big<false>, dim3(cdiv(bd.Lg, PB_COLS), B), dim3(PB_COLS * PB_LANES), 0, st, d_edges);
    for (int k0 = 0; k0 < bd.n_cap; k0 += CB)
      hipLaunchKernelGGL(k_vsolve_mfma, dim3(cdiv(bd.Lg, VS_COLS), B), dim3(256), 0, st, d_edges, k0);
    hipLaunchKernelGGL(k_pred_colsum_big<true>, dim3(cdiv(bd.Lg, PB_COLS), B), dim3(PB_COLS * PB_LANES), 0, st, d_edges);
  } else if (plds <= 150 * 1024)
    hipLaunchKernelGGL((k_predict<true, false>), dim3(cdiv(bd.Lg, 64), B), dim3(64), plds, st, d_edges, 0);
  else
    hipLaunchKernelGGL((k_predict<false, false>), dim3(cdiv(bd.Lg, 64), B), dim3(64), 0, st, d_edges, 0);
  if (want_cov && (parts & 4u)) {
    const int t = cdiv(bd.Lg, 64);
    hipLaunchKernelGGL(k_cov_mfma, dim3(t, t, B), dim3(256), 0, st, d_edges, 0);
  }
  return hipGetLastError();
}

hipError_t launch_final_cov(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd) {
  (void)hipGetLastError();
  const int t = cdiv(bd.Lg, 64);
  hipLaunchKernelGGL(k_cov_mfma, dim3(t, t, B), dim3(256), 0, st, d_edges, 1);
  return hipGetLastError();
}

// converged fit at the optimum: factor + alpha, then mean/std on the standardised grid
hipError_t launch_final_predict(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd) {
  (void)hipGetLastError();
  fit_predict_attrs();
  if (bd.n_cap <= 128)
    hipLaunchKernelGGL((k_fit<true, true>), dim3(1, B), dim3(576),
                       ((size_t)bd.n_cap * (bd.n_cap | 1) + bd.n_cap) * sizeof(double), st, d_edges);
  else
    hipLaunchKernelGGL((k_fit<false, true>), dim3(1, B), dim3(256), (size_t)bd.n_cap * sizeof(double), st, d_edges);
  const size_t plds = ((size_t)bd.n_cap * 64 + 3 * (size_t)bd.n_cap) * sizeof(double);
  if (plds <= 150 * 1024)
    hipLaunchKernelGGL((k_predict<true, true>), dim3(cdiv(bd.Lg, 64), B), dim3(64), plds, st, d_edges, bd.Lg);
  else
    hipLaunchKernelGGL((k_predict<false, true>), dim3(cdiv(bd.Lg, 64), B), dim3(64), 0, st, d_edges, bd.Lg);
  return hipGetLastError();
}

int& gpet_opt_lml_two_tiles_from() {
  static int& v = option("lml_two_tiles_from");
  return v;
}

int& gpet_opt_lml_mfma() {
  static int& v = option("lml_mfma");
  return v;
}

int& gpet_opt_rng_lookahead() {
  static int& v = option("rng_lookahead");  // -1: by batch size
  return v;
}

// LDS Jacobi of ranks <= 96: 1 (default) = seated (k_jacobi_seat), 0 = by row index (k_jacobi_lds, the round-1 form)
int& gpet_opt_jacobi_variant() {
  static int& v = option("jacobi_variant");
  return v;
}
// rotation log + separate eigenvector pass for batches that have a log (gpet_batch_create: up to 16 edges): 1 = on (default)
int& gpet_opt_jacobi_logw() {
  static int& v = option("jacobi_logw");
  return v;
}
static void launch_jacobi_small(hipStream_t st, EdgeDev* d_edges, int B, int rank_max, int scaled_out, bool logw) {
  const int mm = (rank_max + 1) & ~1;
  static PerDeviceOnce once;
  if (once.first()) {
    (void)hipFuncSetAttribute((const void*)k_jacobi_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)k_jacobi_seat<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)k_jacobi_seat<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)k_jacobi_seat<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)k_jacobi_seat<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  }
  if (gpet_opt_jacobi_variant() == 0) {
    hipLaunchKernelGGL(k_jacobi_lds, dim3(1, B), dim3(1024), ((size_t)mm * (mm | 1) + (size_t)mm * (mm + 2)) * sizeof(double), st,
                       d_edges, scaled_out);
  } else {
    const size_t nblk = (size_t)(mm / 2) * (mm / 2 + 1) / 2;
    const size_t lds = (4 * ((nblk + 1) & ~(size_t)1) + (size_t)mm * mm) * sizeof(double);
    if (logw) {
      // small batch: rotations logged, eigenvectors by a second kernel (one wave per row of W)
      if (nblk <= 2 * JS_NT) hipLaunchKernelGGL((k_jacobi_seat<2, true>), dim3(1, B), dim3(JS_NT), lds, st, d_edges, scaled_out);
      else hipLaunchKernelGGL((k_jacobi_seat<3, true>), dim3(1, B), dim3(JS_NT), lds, st, d_edges, scaled_out);
      hipLaunchKernelGGL(k_jacobi_wpass, dim3((rank_max + 3) / 4, B), dim3(256), 0, st, d_edges, scaled_out);
    } else if (nblk <= 2 * JS_NT) {
      hipLaunchKernelGGL((k_jacobi_seat<2, false>), dim3(1, B), dim3(JS_NT), lds, st, d_edges, scaled_out);
    } else {
      hipLaunchKernelGGL((k_jacobi_seat<3, false>), dim3(1, B), dim3(JS_NT), lds, st, d_edges, scaled_out);
    }
  }
}

int& gpet_opt_scalar_jacobi() {
  static int& v = option("scalar_jacobi");
  return v;
}

hipError_t launch_factor(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, unsigned parts, const EdgeDev* h_edges) {
  (void)hipGetLastError();  // drop stale errors: report only these launches
  if (bd.r_cap > 96 && !gpet_opt_scalar_jacobi()) return launch_factor_big(st, d_edges, B, bd, h_edges);
  if (bd.r_cap > 96) {
    // gpet_set_option("scalar_jacobi", 1): the round-1 solver, kept as an independent cross-check of the one above --
    // cyclic two-sided Jacobi directly on the covariance, one parameter + one apply kernel per round.  A fixed budget
    // of sweeps is enqueued; once the device-side convergence test passes the remaining launches return at once.
    const int r = bd.Lg, m = (r + 1) & ~1, half = m >> 1;
    const long long items = (long long)half * half + (long long)half * r;
    int ablocks = (int)((items + 255) / 256);
    if (ablocks > 4096) ablocks = 4096;
    hipLaunchKernelGGL(k_jb_init, dim3(512, B), dim3(256), 0, st, d_edges);
    const int jbn_parts = JBN_PARTS;  // (partial sums of the norms live at the head of jb_cs, sized for them)
    // (graded spectra need up to ~26 sweeps from a cold start: linear phase, then quadratic)
    for (int sweep = 0; sweep < 30; ++sweep) {
      hipLaunchKernelGGL(k_jb_norms, dim3(jbn_parts, B), dim3(1024), 0, st, d_edges);
      hipLaunchKernelGGL(k_jb_norms_fin, dim3(B), dim3(64), 0, st, d_edges, jbn_parts);
      for (int round = 0; round < m - 1; ++round) {
        hipLaunchKernelGGL(k_jb_params, dim3(cdiv(half, 256), B), dim3(256), 0, st, d_edges, round);
        hipLaunchKernelGGL(k_jb_apply, dim3(ablocks, B), dim3(256), 0, st, d_edges, round);
      }
    }
    hipLaunchKernelGGL(k_jb_order, dim3(1, B), dim3(1024), 0, st, d_edges);
    hipLaunchKernelGGL(k_jb_rows, dim3(r, B), dim3(256), 0, st, d_edges);
    return hipGetLastError();
  }
  if (!(parts & 1u)) {
  } else if (bd.Lg <= 512 && bd.r_cap <= PCH_R) {
    hipLaunchKernelGGL(k_pchol_reg, dim3(1, B), dim3(512), 0, st, d_edges);
  } else {
    int pth = bd.Lg >= 1024 ? 1024 : (bd.Lg > 512 ? 1024 : (bd.Lg > 256 ? 512 : 256));
    hipLaunchKernelGGL(k_pchol, dim3(1, B), dim3(pth), (size_t)(bd.Lg + bd.r_cap) * sizeof(double), st, d_edges);
  }
  const int t = cdiv(bd.r_cap, 16);
  if (parts & 2u) hipLaunchKernelGGL(k_gram, dim3(t, t, B), dim3(256), 0, st, d_edges);
  if (parts & 4u) launch_jacobi_small(st, d_edges, B, bd.r_cap, 0, bd.jlog != 0 && gpet_opt_jacobi_logw() != 0);
  if (parts & 8u)
    hipLaunchKernelGGL(k_factor_rows, dim3(bd.r_cap, B), dim3(256), (size_t)bd.r_cap * sizeof(double), st, d_edges);
  return hipGetLastError();
}

// structured loop path: fit -> (U, H, mean) -> Jacobi (the LDS kernel, on E.C) -> factor rows
hipError_t launch_struct_iteration(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, unsigned parts) {
  (void)hipGetLastError();
  fit_predict_attrs();
  static PerDeviceOnce once;
  if (once.first()) {
    (void)hipFuncSetAttribute((const void*)k_struct_H, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)k_jacobi_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)k_struct_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  }
  if (!(parts & 1u)) {
  } else if (bd.n_cap <= 128)
    hipLaunchKernelGGL((k_fit<true, false>), dim3(1, B), dim3(576),
                       ((size_t)bd.n_cap * (bd.n_cap | 1) + bd.n_cap) * sizeof(double), st, d_edges);
  else  // (more possible training points than K fits LDS for: blocked factorisation in HBM)
    launch_fit_blocked(st, d_edges, B, bd);
  if ((parts & 2u) && bd.n_cap > 128) {
    // many training points: U through HBM, blocked substitution
    hipLaunchKernelGGL(k_structB_build, dim3(64, B), dim3(256), 0, st, d_edges);
    hipLaunchKernelGGL(k_struct_beta, dim3(1, B), dim3(128), 0, st, d_edges);
    const int cgroups = cdiv(bd.r0_max, SB_COLS);
    for (int k0 = 0; k0 < bd.n_cap; k0 += CB)
      hipLaunchKernelGGL(k_struct_trsm, dim3(cgroups, B), dim3(256), 0, st, d_edges, k0, 0);
    const int t16 = cdiv(bd.r0_max, 16);
    hipLaunchKernelGGL(k_struct_Hbig, dim3(t16, t16, B), dim3(256), 0, st, d_edges);
    hipLaunchKernelGGL(k_struct_mean, dim3(cdiv(bd.Lg, 256), B), dim3(256), 0, st, d_edges);
  } else if (parts & 2u) {
    const size_t full = ((size_t)bd.n_cap * (bd.r0_max | 1) + (size_t)bd.n_cap * (bd.n_cap + 1) / 2 + bd.r_cap) * sizeof(double);
    const size_t rowm = ((size_t)bd.n_cap * (bd.r0_max | 1) + bd.n_cap + bd.r_cap) * sizeof(double);
    const int l_in_lds = full <= (size_t)STRUCT_H_LDS_MAX ? 1 : 0;  // (gpet_batch_create checked that `rowm` fits)
    hipLaunchKernelGGL(k_struct_H, dim3(1, B), dim3(1024), l_in_lds ? full : rowm, st, d_edges, l_in_lds);
  }
  if (parts & 4u) launch_jacobi_small(st, d_edges, B, bd.r0_max > 0 ? bd.r0_max : bd.r_cap, 1, bd.jlog != 0 && gpet_opt_jacobi_logw() != 0);
  if (parts & 8u) {
    // the variant k_struct_rows picks for r0_max: [4 KS][16 MT + 1] eigenvector tile, reused as [r][64] products
    const int rm = bd.r0_max;
    const int mt = rm <= 32 ? 2 : rm <= 48 ? 3 : rm <= 64 ? 4 : rm <= 80 ? 5 : 6;
    const int ks = rm <= 32 ? 8 : rm <= 48 ? 12 : rm <= 64 ? 16 : rm <= 72 ? 18 : rm <= 80 ? 20 : 24;
    size_t words = (size_t)4 * ks * (16 * mt + 1);
    if (words < (size_t)4 * ks * 64) words = (size_t)4 * ks * 64;
    const size_t lds = words * sizeof(double);
    hipLaunchKernelGGL(k_struct_rows, dim3(cdiv(bd.Lg, SR_TJ), B), dim3(256), lds, st, d_edges);
    hipLaunchKernelGGL(k_struct_sign, dim3(bd.r0_max, B), dim3(128), 0, st, d_edges);
  }
  return hipGetLastError();
}

// construction: eigenbasis of the grid's correlation matrix through the generic factor pipeline
hipError_t launch_struct_basis(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_rho_fill, dim3(256, B), dim3(256), 0, st, d_edges);
  hipError_t e = launch_set_force(st, d_edges, B, 1);
  if (e != hipSuccess) return e;
  e = launch_factor(st, d_edges, B, bd, ~0u);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_struct_basis, dim3(bd.r_cap, B), dim3(256), 0, st, d_edges);
  return launch_set_force(st, d_edges, B, 0);
}

// ---- opt-in counter-based normals (SURVEY K5 "fast Philox mode"; gpet_batch_set_rng) ---------------------------------
// Philox4x32-10 (Salmon, Moraes, Dror & Shaw, SC'11; the Random123 known-answer vectors are in the tests) + Box-Muller.
// The normal of (sample row s, column j) is a pure function of (seed of the iteration, s, j): counter = (j / 2, s, 0, 0),
// key = (seed, "Phlx"); its four words give two 53-bit uniforms u1, u2 in (0, 1) and the pair
// sqrt(-2 ln u1) (cos 2 pi u2, sin 2 pi u2) for columns j, j + 1.  No stream to walk: every thread writes its own pair, and
// only the columns the factor multiplies are generated at all.  NOT the reference's numbers (sklearn_gpr.py:464 draws from
// RandomState(seed)): a mode of its own with its own CPU twin in the tests (philox_standard_normal), never the default.
__device__ __forceinline__ void philox4x32_10(unsigned int c[4], unsigned int k0, unsigned int k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (unsigned int)p1;
    c[3] = (unsigned int)p0;
    c[0] = n0;
    c[2] = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
__global__ void __launch_bounds__(256) k_philox_normals(EdgeDev* edges, const unsigned int* seeds, int add_iter, int iter_abs,
                                                        int z_store) {
  const EdgeDev E = edges[blockIdx.z];
  const gpet_scalars* sc = E.sc;
  if ((sc->done && !sc->force) || sc->status != GPET_OK) return;
  const int iter_idx = (iter_abs >= 0 ? iter_abs : sc->iter) + (int)blockIdx.y;
  double* __restrict__ Zs = E.Z + (size_t)(iter_idx % E.z_ring) * ((size_t)E.S * E.z_cols);
  const int zc = E.z_cols, zs = (z_store > 0 && z_store < zc) ? z_store : zc;
  const int hp = (zs + 1) >> 1;  // column pairs per row
  const unsigned int key = seeds[blockIdx.z] + (add_iter ? (unsigned int)(iter_idx + 1) : 0u);  // (the loop's seed rule, gpet.py:839)
  const long long total = (long long)E.S * hp;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int srow = (int)(e / hp), jp = (int)(e - (long long)srow * hp);
    unsigned int c[4] = {(unsigned int)jp, (unsigned int)srow, 0u, 0u};
    philox4x32_10(c, key, 0x50686c78u);
    const double u1 = ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(c[2] >> 5) * 67108864.0 + (double)(c[3] >> 6) + 0.5) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    double* o = Zs + (size_t)srow * zc + 2 * jp;
    o[0] = r * cs;
    if (2 * jp + 1 < zs) o[1] = r * sn;
  }
}
hipError_t launch_normals_philox(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, const unsigned int* d_seeds, int add_iter,
                                 int iter_abs, int n_ahead, int z_store) {
  (void)hipGetLastError();
  const int zs = (z_store > 0 && z_store < bd.z_cols) ? z_store : bd.z_cols;
  const long long pairs = (long long)bd.S * ((zs + 1) / 2);
  int gx = (int)((pairs + 256 * 4 - 1) / (256 * 4));  // four pairs per thread
  gx = gx < 1 ? 1 : (gx > 4096 ? 4096 : gx);
  hipLaunchKernelGGL(k_philox_normals, dim3(gx, n_ahead, B), dim3(256), 0, st, d_edges, d_seeds, add_iter, iter_abs, z_store);
  return hipGetLastError();
}

hipError_t launch_normals(hipStream_t st, EdgeDev* d_edges, int B, const unsigned int* d_seeds, int add_iter,
                          int iter_abs, int n_ahead, int z_store) {
  (void)hipGetLastError();  // drop stale errors: report only these launches
  // (three waves per workgroup: four were measured 4 % slower)
  hipLaunchKernelGGL((k_mt_normals<false, 3>), dim3(n_ahead, B), dim3(192), 0, st, d_edges, d_seeds, add_iter, iter_abs, z_store, MtjWork{});
  return hipGetLastError();
}

// mode 0: KDE of the best curves -> E.kde ; mode 1: KDE of the gradient image -> E.grad_kde
hipError_t launch_kde(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, int mode, unsigned parts,
                      int raw_band) {
  (void)hipGetLastError();  // drop stale errors: report only these launches
  if (mode == 0) {
    // per-iteration path: one prep kernel + one fused bin/convolve kernel + normalise
    const size_t lds = ((size_t)(KDE_TX + 8) * ((KDE_H + 8) | 1) + (size_t)KDE_NB * (KDE_TX + 8) + KDE_NB) * sizeof(double);
    if (parts & 1u) hipLaunchKernelGGL(k_kde_prep, dim3(1, B), dim3(1024), 0, st, d_edges);
    if (parts & 2u)
      hipLaunchKernelGGL(k_kde_fused, dim3(cdiv(bd.N, KDE_TX), B), dim3(KDE_THREADS), lds, st, d_edges, raw_band);
    if ((parts & 4u) && !raw_band) hipLaunchKernelGGL(k_kde_normalise, dim3(64, B), dim3(256), 0, st, d_edges, mode);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_kde_clear, dim3(64, B), dim3(256), 0, st, d_edges, mode);
  hipLaunchKernelGGL(k_kde_bin_gradient, dim3(bd.N, B), dim3(256), 0, st, d_edges);
  hipLaunchKernelGGL(k_kde_wsum, dim3(B), dim3(64), 0, st, d_edges, mode);
  hipLaunchKernelGGL(k_kde_conv_y, dim3(cdiv(bd.M + 2, 256), bd.N + 2, B), dim3(256), 0, st, d_edges, mode);
  hipLaunchKernelGGL(k_kde_conv_x, dim3(cdiv(bd.M, KCX_T), cdiv(bd.N, KCX_T), B), dim3(256), 0, st, d_edges, mode);
  hipLaunchKernelGGL(k_kde_normalise, dim3(64, B), dim3(256), 0, st, d_edges, mode);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_pix_reset(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  for (int i = threadIdx.x; i < E.n_bins; i += blockDim.x) {
    E.binbest[i] = 0ull;
    E.binarg[i] = 0x7FFFFFFFFFFFFFFFll;
  }
}

__global__ void k_set_force(EdgeDev* edges, int v) { edges[blockIdx.x].sc->force = v; }

// training sets of the converged fits: staged as three [B][stride] blocks (x | y | w), scattered to the edges
__global__ void __launch_bounds__(128) k_fin_scatter(EdgeDev* edges, const double* stage, const int* n, int stride, int B) {
  const int e = blockIdx.x;
  EdgeDev& E = edges[e];
  const int ne = n[e];
  for (int i = threadIdx.x; i < ne; i += blockDim.x) {
    E.fin_x[i] = stage[(size_t)e * stride + i];
    E.fin_y[i] = stage[((size_t)B + e) * stride + i];
    E.fin_w[i] = stage[((size_t)2 * B + e) * stride + i];
  }
  if (threadIdx.x == 0) E.fin_n = ne;
}

hipError_t launch_fin_scatter(hipStream_t st, EdgeDev* d_edges, int B, const double* d_stage, const int* d_n, int stride) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_fin_scatter, dim3(B), dim3(128), 0, st, d_edges, d_stage, d_n, stride, B);
  return hipGetLastError();
}

// general-nu Matern: correlation at the integer lags 0..N-1 of the pixel grid, once per edge at construction
__global__ void __launch_bounds__(256) k_rho_tab(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  if (E.kernel_type != GPET_KERNEL_MATERN || E.nu_code != 3) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= E.N) return;
  E.rho_tab[k] = matern_gen(E.nu_gen, E.inv_gamma_nu, (double)k / E.length_scale, nullptr);
}
hipError_t launch_rho_tab(hipStream_t st, EdgeDev* d_edges, int B, int N) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_rho_tab, dim3(cdiv(N, 256), B), dim3(256), 0, st, d_edges);
  return hipGetLastError();
}

hipError_t launch_set_force(hipStream_t st, EdgeDev* d_edges, int B, int v) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_set_force, dim3(B), dim3(1), 0, st, d_edges, v);
  return hipGetLastError();
}

hipError_t launch_pixels_reset(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd) {
  (void)hipGetLastError();  // drop stale errors: report only these launches
  (void)bd;
  hipLaunchKernelGGL(k_pix_reset, dim3(1, B), dim3(256), 0, st, d_edges);
  return hipGetLastError();
}

hipError_t launch_pixels(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, int raw_band, unsigned parts) {
  (void)hipGetLastError();  // drop stale errors: report only these launches
  if (parts & 1u) hipLaunchKernelGGL(k_pix_columns, dim3(cdiv(bd.N, PIX_CX), B), dim3(256), 0, st, d_edges, raw_band);
  const int nt = bd.N > bd.obs_cap ? bd.N : bd.obs_cap;
  if (parts & 2u) hipLaunchKernelGGL(k_pix_old, dim3(cdiv(bd.obs_cap, 256), B), dim3(256), 0, st, d_edges, raw_band);
  if (parts & 4u) hipLaunchKernelGGL(k_pix_argbest, dim3(cdiv(nt, 256), B), dim3(256), 0, st, d_edges, raw_band);
  if (parts & 8u) hipLaunchKernelGGL(k_pix_select, dim3(1, B), dim3(64), 0, st, d_edges);
  return hipGetLastError();
}

hipError_t launch_sample(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, int rank_max) {
  (void)hipGetLastError();  // drop stale errors: report only these launches
  // rank <= 96 everywhere in the batch (factor capacity): Z rows stay in registers; otherwise
  // (full factors injected by tests, Matern ranks) the K-chunked kernel.  rank_max: the largest rank any
  // edge can have in this launch (r0_max inside the structured loop, else the factor capacity).
  if (bd.r_cap <= GEMM_KMAX && bd.a_rows_cap <= GEMM_KMAX) {
    const int rm = rank_max > 0 && rank_max <= bd.r_cap ? rank_max : (bd.r_cap > bd.a_rows_cap ? bd.r_cap : bd.a_rows_cap);
    const int ks = (rm + 3) >> 2;
    // column runs per row block: while the row blocks alone leave CUs empty (small batches are latency chains)
    const int rparts = cdiv(bd.S, 128), ctiles = cdiv(bd.Lg, 64);
    int ncs = cdiv(256, B * rparts);
    ncs = ncs > ctiles ? ctiles : (ncs < 1 ? 1 : ncs);
    if (ncs > 8) ncs = 8;
    const dim3 grid(rparts * ncs, B), block(512);
    {  // (the chunk plus the posterior mean of a wide edge exceed the 64 KB a kernel gets without asking: 66 KB at K = 96, Lg = 2048)
      static PerDeviceOnce once;
      if (once.first()) {
#define GPET_GEMM_ATTR(KERNEL, KS_)                                                                                       \
  (void)hipFuncSetAttribute((const void*)KERNEL<KS_, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_MAX);  \
  (void)hipFuncSetAttribute((const void*)KERNEL<KS_, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_MAX);   \
  (void)hipFuncSetAttribute((const void*)KERNEL<KS_, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_MAX); \
  (void)hipFuncSetAttribute((const void*)KERNEL<KS_, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_MAX)
        GPET_GEMM_ATTR(k_sample_gemm_mfma_r, 8);
        GPET_GEMM_ATTR(k_sample_gemm_mfma_r, 12);
        GPET_GEMM_ATTR(k_sample_gemm_mfma_r, 16);
        GPET_GEMM_ATTR(k_sample_gemm_mfma_r, 18);
        GPET_GEMM_ATTR(k_sample_gemm_mfma_rl, 20);
        GPET_GEMM_ATTR(k_sample_gemm_mfma_rl, 24);
#undef GPET_GEMM_ATTR
      }
    }
    // (an edge too wide for its posterior mean to sit behind the chunk reads it from global memory in the epilogue)
    const int mu_in_lds = ((size_t)4 * 24 * GEMM_LDA + bd.Lg) * sizeof(double) <= (size_t)GEMM_LDS_MAX ? 1 : 0;
#define GPET_GEMM_LAUNCH(KERNEL, KS_)                                                                                                  \
  do {                                                                                                                                 \
    const size_t lds_ = ((size_t)4 * KS_ * GEMM_LDA + (mu_in_lds ? bd.Lg : 0)) * sizeof(double);                                            \
    if (bd.y_f32 && mu_in_lds) hipLaunchKernelGGL((KERNEL<KS_, true, true>), grid, block, lds_, st, d_edges, ncs);                        \
    else if (bd.y_f32) hipLaunchKernelGGL((KERNEL<KS_, true, false>), grid, block, lds_, st, d_edges, ncs);                               \
    else if (mu_in_lds) hipLaunchKernelGGL((KERNEL<KS_, false, true>), grid, block, lds_, st, d_edges, ncs);                              \
    else hipLaunchKernelGGL((KERNEL<KS_, false, false>), grid, block, lds_, st, d_edges, ncs);                                            \
  } while (0)
    if (ks <= 8) GPET_GEMM_LAUNCH(k_sample_gemm_mfma_r, 8);
    else if (ks <= 12) GPET_GEMM_LAUNCH(k_sample_gemm_mfma_r, 12);
    else if (ks <= 16) GPET_GEMM_LAUNCH(k_sample_gemm_mfma_r, 16);
    else if (ks <= 18) GPET_GEMM_LAUNCH(k_sample_gemm_mfma_r, 18);
    else if (ks <= 20) GPET_GEMM_LAUNCH(k_sample_gemm_mfma_rl, 20);
    else GPET_GEMM_LAUNCH(k_sample_gemm_mfma_rl, 24);
#undef GPET_GEMM_LAUNCH
  } else {
    hipLaunchKernelGGL(k_sample_gemm_mfma, dim3(cdiv(bd.Lg, 64), cdiv(bd.S, 64), B), dim3(256), 0, st, d_edges);
  }
  return hipGetLastError();
}

hipError_t launch_score(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, unsigned parts) {
  (void)hipGetLastError();  // drop stale errors: report only these launches
  if (parts & 1u) {
    const size_t lds = (size_t)(2 * SC_PAIRS + 2) * (bd.M | 1) * sizeof(float);
    if (lds <= 150 * 1024 && B * 1 > 0 && bd.S >= 64) {
      static PerDeviceOnce once;
      if (once.first())
      {
        (void)hipFuncSetAttribute((const void*)k_score_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)k_score_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      }
      const int n_tiles = cdiv((bd.Lg - 2) / 2, SC_PAIRS);
      // curves per workgroup: 1024, or down to 128 while the tiles alone leave CUs empty (every part stages the slab again)
      int cpw = SC_CURVES;
      while (cpw > 128 && B * n_tiles * cdiv(bd.S, cpw) < 256) cpw >>= 1;
      if (bd.y_f32) hipLaunchKernelGGL(k_score_tile<true>, dim3(n_tiles, cdiv(bd.S, cpw), B), dim3(SC_THREADS), lds, st, d_edges, cpw);
      else hipLaunchKernelGGL(k_score_tile<false>, dim3(n_tiles, cdiv(bd.S, cpw), B), dim3(SC_THREADS), lds, st, d_edges, cpw);
      hipLaunchKernelGGL(k_score_combine, dim3(cdiv(bd.S, 256), B), dim3(256), 0, st, d_edges, n_tiles);
    } else {
      if (bd.y_f32) hipLaunchKernelGGL(k_score<true>, dim3(cdiv(bd.S, 4), B), dim3(256), 0, st, d_edges);
      else hipLaunchKernelGGL(k_score<false>, dim3(cdiv(bd.S, 4), B), dim3(256), 0, st, d_edges);
    }
  }
  if (parts & 2u) {
    if (bd.S <= 1024 && !option("topk_rank")) hipLaunchKernelGGL(k_topk_sort, dim3(1, B), dim3(512), 0, st, d_edges);
    else hipLaunchKernelGGL(k_topk, dim3(cdiv(bd.S, 256), B), dim3(256), 0, st, d_edges);
  }
  return hipGetLastError();
}

// gpet_set_option "fused_score" (default 0; environment GPET_FUSED_SCORE): 1 = where it applies the device loop scores the
// samples out of the matrix-core accumulators (k_sample_score) and stores only the kept rows, instead of writing the whole
// sample matrix (k_sample_gemm_mfma_r) and scoring it from memory (k_score_tile).  Identical results; measured SLOWER at
// the bench shape (3.2 + 0.4 ms against 1.8 + 1.25 ms per 1 024 edges: every column tile streams the edge's normals
// through its CU's L1 again, 13 GB per launch), so it is off by default.
int& gpet_opt_fused_score() {
  static int& v = option("fused_score");
  return v;
}

static int sample_score_ks(const BatchDims& bd, int rank_max) {
  const int rm = rank_max > 0 && rank_max <= bd.r_cap ? rank_max : (bd.r_cap > bd.a_rows_cap ? bd.r_cap : bd.a_rows_cap);
  return (rm + 3) >> 2;
}

bool sample_score_fused_applies(const BatchDims& bd, int rank_max) {
  if (!bd.lg_even || bd.Lg < 4 || bd.S < 64) return false;
  if (bd.r_cap > GEMM_KMAX || bd.a_rows_cap > GEMM_KMAX) return false;
  if (sample_score_ks(bd, rank_max) > 18) return false;  // (two waves per SIMD hold 2 x 18 factor entries and 18 normals per lane)
  if (bd.z_cols < 4) return false;
  const size_t lds = (size_t)(2 * SC_PAIRS + 2) * (bd.M | 1) * sizeof(float);
  return lds <= 150 * 1024;
}

// parts: 1 = samples scored out of the accumulators + the per-tile partials combined, 2 = the best n_keep, 4 = their rows
hipError_t launch_sample_score(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, int rank_max, unsigned parts) {
  (void)hipGetLastError();
  const int ks = sample_score_ks(bd, rank_max);
  const size_t lds = (size_t)(2 * SC_PAIRS + 2) * (bd.M | 1) * sizeof(float);
  {
    static PerDeviceOnce once;
    if (once.first()) {
#define GPET_SS_ATTR(KS_)                                                                                                  \
  (void)hipFuncSetAttribute((const void*)k_sample_score<KS_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
  (void)hipFuncSetAttribute((const void*)k_sample_score<KS_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)
      GPET_SS_ATTR(8);
      GPET_SS_ATTR(12);
      GPET_SS_ATTR(16);
      GPET_SS_ATTR(18);
#undef GPET_SS_ATTR
    }
  }
  if (parts & 1u) {
    const int n_tiles = cdiv((bd.Lg - 2) / 2, SC_PAIRS);
    // two workgroups of four waves per CU while two slabs fit its LDS (one stages while the other works), else one of eight
    const int threads = 2 * lds + 4096 <= 160 * 1024 ? 256 : 512;  // (256 / 384 / 512 were measured: 256 x 2 per CU best)
    // curves split over workgroups only while the tiles alone do not fill the GPU (every part stages the slab again)
    int ny = cdiv(768, B * n_tiles);
    const int ny_max = cdiv(bd.S, threads / 4);
    ny = ny > ny_max ? ny_max : (ny < 1 ? 1 : ny);
    const dim3 grid(n_tiles, ny, B), block(threads);
#define GPET_SS_LAUNCH(KS_)                                                                                  \
  do {                                                                                                       \
    if (bd.y_f32) hipLaunchKernelGGL((k_sample_score<KS_, true>), grid, block, lds, st, d_edges);  \
    else hipLaunchKernelGGL((k_sample_score<KS_, false>), grid, block, lds, st, d_edges);          \
  } while (0)
    // option "fused_score" = 2: the curve-stationary form (k_sample_score2: 256 curves per workgroup, slab and factor tile
    // re-staged per column tile), while its LDS fits one workgroup per CU
    const size_t lds2 = (((size_t)(2 * SC_PAIRS + 2) * (bd.M | 1) + 3) & ~(size_t)3) * sizeof(float) +
                        ((size_t)2 * 4 * (ks <= 8 ? 8 : ks <= 12 ? 12 : ks <= 16 ? 16 : 18) * 16 + 32) * sizeof(double);
    if (gpet_opt_fused_score() == 2 && lds2 <= 160 * 1024) {
      static PerDeviceOnce once2;
      if (once2.first()) {
#define GPET_SS2_ATTR(KS_)                                                                                                      \
  (void)hipFuncSetAttribute((const void*)k_sample_score2<KS_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  (void)hipFuncSetAttribute((const void*)k_sample_score2<KS_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        GPET_SS2_ATTR(8);
        GPET_SS2_ATTR(12);
        GPET_SS2_ATTR(16);
        GPET_SS2_ATTR(18);
#undef GPET_SS2_ATTR
      }
      const dim3 grid2(cdiv(bd.S, 256), B), block2(1024);
#define GPET_SS2_LAUNCH(KS_)                                                                            \
  do {                                                                                                  \
    if (bd.y_f32) hipLaunchKernelGGL((k_sample_score2<KS_, true>), grid2, block2, lds2, st, d_edges);   \
    else hipLaunchKernelGGL((k_sample_score2<KS_, false>), grid2, block2, lds2, st, d_edges);           \
  } while (0)
      if (ks <= 8) GPET_SS2_LAUNCH(8);
      else if (ks <= 12) GPET_SS2_LAUNCH(12);
      else if (ks <= 16) GPET_SS2_LAUNCH(16);
      else GPET_SS2_LAUNCH(18);
#undef GPET_SS2_LAUNCH
    } else if (ks <= 8) GPET_SS_LAUNCH(8);
    else if (ks <= 12) GPET_SS_LAUNCH(12);
    else if (ks <= 16) GPET_SS_LAUNCH(16);
    else GPET_SS_LAUNCH(18);
#undef GPET_SS_LAUNCH
    hipLaunchKernelGGL(k_score_combine, dim3(cdiv(bd.S, 256), B), dim3(256), 0, st, d_edges, n_tiles);
  }
  if (parts & 2u) {
    if (bd.S <= 1024 && !option("topk_rank")) hipLaunchKernelGGL(k_topk_sort, dim3(1, B), dim3(512), 0, st, d_edges);
    else hipLaunchKernelGGL(k_topk, dim3(cdiv(bd.S, 256), B), dim3(256), 0, st, d_edges);
  }
  if (parts & 4u) {
    const int gx = cdiv(bd.n_keep, 16);
    if (bd.y_f32) hipLaunchKernelGGL(k_sample_keep_rows<true>, dim3(gx, B), dim3(256), 0, st, d_edges);
    else hipLaunchKernelGGL(k_sample_keep_rows<false>, dim3(gx, B), dim3(256), 0, st, d_edges);
  }
  return hipGetLastError();
}

}  // namespace gpet
