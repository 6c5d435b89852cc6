// Factor of a posterior covariance of ANY rank (sklearn_gpr.py:464 -> numpy legacy multivariate_normal: rows
// sqrt(s_k) v_k of the symmetric SVD), for the cases the LDS-resident path (rank <= 96) does not cover: Matern
// kernels (full rank: the reference's default, gpet.py:22-35,139-151) and short-length-scale RBF.
//
//   1. k_pcx_*   rank-revealing pivoted Cholesky  Sigma ~= G^T G  spread over the GPU: one launch per pivot step,
//                every workgroup owns 32 columns of G (its slab stays in its XCD's L2 across the launches);
//                stops at 1e-14 of the largest diagonal entry like the single-workgroup kernels.
//   2. k_oj_*    one-sided block Jacobi on the ROWS of G: rows are rotated until they are mutually orthogonal, and
//                then they ARE the factor rows sqrt(s_k) v_k^T (G^T G = Sigma is invariant under row rotations) --
//                no eigenvector matrix is accumulated and nothing is divided by a small singular value.  The pivoted
//                Cholesky is the preconditioner that makes this converge in ~10 sweeps instead of ~26 (Drmac/Veselic);
//                the relative stopping test |g_p.g_q| <= tol |g_p||g_q| resolves the small, clustered end of a
//                Matern spectrum that an absolute off-norm test leaves unconverged.
//                Blocks of 8 rows are paired round-robin; one workgroup per pair: 16x16 Gram matrix on the f64
//                matrix cores -> one cyclic Jacobi sweep on it inside a single wave (LDS, no workgroup barriers)
//                -> the accumulated 16x16 rotation applied to the 16 rows on the matrix cores.
//   3. k_oj_norms / k_oj_order / k_oj_rows: singular values = row norms, descending order, sign convention.
#include "gpet_kernels.h"
#include "gpet_options.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace gpet {

#define WAVE 64
#define PCX_COLS 32
#define OJ_B 8
#define OJ_M 16
#define OJ_ARGS_MAXB 8     // batches up to this size get their per-edge pointers in the kernel arguments
#define OJ_STAGE_MAX 1024  // widest edge whose 16-row panel (16 x Lg doubles) is staged in LDS
// a pair of blocks is left alone when its largest relative coupling is this far (squared) below the stopping tolerance:
// 1e-4 of it, i.e. 1e-12 at the default 1e-8 -- what one more rotation would make of it is below f64 resolution anyway
#define OJ_SKIP_REL2 1e-8
#define OJ_PF 16  // 16-column tiles per wave whose operands are prefetched into registers (4 waves x 16 x 16 = 1024 columns)

static inline int cdiv_h(int a, int b) { return (a + b - 1) / b; }

typedef double v4f64e __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool eig_skip(const EdgeDev& E) {
  const gpet_scalars* sc = E.sc;
  return (sc->done && !sc->force) || sc->status != GPET_OK || E.factor_injected;
}

// round-robin tournament: pair k of `round` among m1 + 1 players (m1 odd), p < q
__device__ __forceinline__ void oj_rr_pair(int m1, int round, int k, int& p, int& q) {
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

// ---- 1. pivoted Cholesky over the whole GPU ------------------------------------------------------------------------
__device__ __forceinline__ void pcx_argmax_wave(double& bv, int& bi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(bv, o, WAVE);
    const int oi = __shfl_xor(bi, o, WAVE);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
}

// Pivot candidates: one (value, index) pair per WAVE (8 columns), two halves of pcx_cand used alternately: the
// candidates of step t live in half (t & 1), a step writes those of step t + 1 into the other half.
__global__ void __launch_bounds__(256) k_pcx_init(EdgeDev* edges, int nw_max) {
  const EdgeDev E = edges[blockIdx.y];
  if (eig_skip(E)) return;
  const int Lg = E.Lg, j0 = blockIdx.x * PCX_COLS, tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0) {
    EigState* st = E.eig;
    st->tol = 0.0;
    st->stopped = 0;
    st->rank = 0;
    st->maxrel_bits = 0ull;
    st->converged = 0;
    st->sweeps = 0;
    st->bar = 0u;
    st->ticket = 0u;
    st->verdicts = 0;
  }
  if (j0 >= Lg) return;
  const int lane = tid & 63, w = tid >> 6;
  const int j = j0 + w * 8 + lane;
  double bv = -1.0;
  int bi = 0x7FFFFFFF;
  if (lane < 8 && j < Lg) {
    const double d = E.cov[(size_t)j * Lg + j];
    E.pcx_d[j] = d;
    bv = d;
    bi = j;
  }
  pcx_argmax_wave(bv, bi);
  if (lane == 0) {
    E.pcx_cand[2 * (blockIdx.x * 4 + w)] = bv;
    E.pcx_cand[2 * (blockIdx.x * 4 + w) + 1] = (double)bi;
  }
  (void)nw_max;
}

// Pivot step t.  A kernel starts with cold caches (data written by the previous launch comes from the Infinity Cache),
// so what bounds a step is its chain of DEPENDENT global round trips, not its arithmetic.  There are three: the edge
// table; {state, candidates, this wave's remaining diagonals}; {the pivot's previous entries, the pivot's covariance
// row, the previous entries of this wave's 8 columns}.  No LDS and no workgroup barrier: every wave finds the pivot
// itself and reads the pivot's entries itself, a wave owns 8 columns, lane = previous row index s (mod 64).
struct PcxEdge {
  double *G, *Gt, *pcx_d, *pcx_cand;
  const double* cov;
  int* perm;
  EigState* eig;
  const gpet_scalars* sc;
  int Lg, r_cap, factor_injected, pad;
};
struct PcxArgs {
  PcxEdge e[OJ_ARGS_MAXB];
};

template <bool ARGS>
__global__ void __launch_bounds__(256) k_pcx_step(PcxArgs args, EdgeDev* edges, int t, int nw_max) {
  PcxEdge E;
  if (ARGS) {
    E = args.e[blockIdx.y];
  } else {
    const EdgeDev& Et = edges[blockIdx.y];
    E.G = Et.G;
    E.Gt = Et.Gt;
    E.pcx_d = Et.pcx_d;
    E.pcx_cand = Et.pcx_cand;
    E.cov = Et.cov;
    E.perm = Et.perm;
    E.eig = Et.eig;
    E.sc = Et.sc;
    E.Lg = Et.Lg;
    E.r_cap = Et.r_cap;
    E.factor_injected = Et.factor_injected;
  }
  const int Lg = E.Lg, j0 = blockIdx.x * PCX_COLS, tid = threadIdx.x;
  if (j0 >= Lg || t >= E.r_cap) return;
  const int lane = tid & 63, w = tid >> 6;
  EigState* st = E.eig;
  const gpet_scalars* sc = E.sc;
  // -- round trip 2: everything that does not depend on the pivot
  const int s_done = sc->done, s_force = sc->force, s_status = sc->status, s_stopped = st->stopped;
  const double tol_prev = st->tol;
  const int ncand = 4 * ((Lg + PCX_COLS - 1) / PCX_COLS);
  const double* cand = E.pcx_cand + (size_t)(t & 1) * 8 * nw_max;
  double bv = -1.0;
  int bi = 0x7FFFFFFF;
  for (int i = lane; i < ncand; i += WAVE) {
    const double v = cand[2 * i];
    const int ix = (int)cand[2 * i + 1];
    if (v > bv || (v == bv && ix < bi)) {
      bv = v;
      bi = ix;
    }
  }
  const int jw = j0 + w * 8;  // this wave's 8 columns
  double dcol[8];
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) dcol[cc] = (jw + cc < Lg) ? E.pcx_d[jw + cc] : -1.0;  // (< 0: pivoted / no such column)
  if ((s_done && !s_force) || s_status != GPET_OK || E.factor_injected || s_stopped) return;
  pcx_argmax_wave(bv, bi);
  const double dp = bv;
  const int p = bi;
  const double tol = (t == 0) ? dp * 1e-14 : tol_prev;
  if (!(dp > tol) || !(dp > 0.0)) {
    if (blockIdx.x == 0 && tid == 0) {
      st->stopped = 1;
      st->rank = t;
    }
    return;
  }
  if (t == 0 && blockIdx.x == 0 && tid == 0) st->tol = tol;
  // -- round trip 3: the transposed copy Gt[j][s] = G[s][j] makes the pivot's previous entries and every column's
  //    dot product contiguous
  const int ld = E.r_cap;
  const double* __restrict__ gp = E.Gt + (size_t)p * ld;
  double cvv = 0.0;
  if (lane < 8 && jw + lane < Lg) cvv = E.cov[(size_t)p * Lg + jw + lane];  // (cov is exactly symmetric: row p == column p)
  double acc[8];
  const double* __restrict__ gj[8];
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    gj[cc] = E.Gt + (size_t)(jw + cc < Lg ? jw + cc : 0) * ld;
    acc[cc] = 0.0;
  }
  for (int s0 = 0; s0 < t; s0 += 4 * WAVE) {
    double gpv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = s0 + i * WAVE + lane;
      gpv[i] = s < t ? gp[s] : 0.0;
    }
#pragma unroll
    for (int cc = 0; cc < 8; ++cc)
      if (dcol[cc] >= 0.0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int s = s0 + i * WAVE + lane;
          if (s < t) acc[cc] += gj[cc][s] * gpv[i];
        }
      }
  }
  double nb = -1.0;
  int ni = 0x7FFFFFFF;
  const double isq = sqrt(dp);
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const int j = jw + cc;
    if (j >= Lg) continue;
    double sum = 0.0;
    if (dcol[cc] >= 0.0) {
      sum = acc[cc];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, WAVE);
    }
    const double cj = __shfl(cvv, cc, WAVE);
    if (lane == 0) {
      double dj = dcol[cc], gv = 0.0;
      if (dj >= 0.0) gv = (cj - sum) / isq;
      E.G[(size_t)t * Lg + j] = gv;
      E.Gt[(size_t)j * ld + t] = gv;
      if (j == p) {
        dj = -1.0;  // used
        E.perm[t] = p;
      } else if (dj >= 0.0) {
        const double nd = dj - gv * gv;
        dj = nd > 0.0 ? nd : 0.0;
      }
      E.pcx_d[j] = dj;
      if (dj > nb || (dj == nb && j < ni)) {
        nb = dj;
        ni = j;
      }
    }
  }
  if (lane == 0) {
    double* nxt = E.pcx_cand + (size_t)((t + 1) & 1) * 8 * nw_max;
    nxt[2 * (blockIdx.x * 4 + w)] = nb;
    nxt[2 * (blockIdx.x * 4 + w) + 1] = (double)ni;
  }
}

// ---- the same factorisation, PCB_NB pivots per launch --------------------------------------------------------------
// One launch per pivot is bound by launch + round-trip latency (~11 us x 1024 pivots).  A block of up to 16 pivots per
// launch: every 32-column workgroup offers its largest remaining diagonal as a candidate (so candidates are spread over
// the edge: neighbouring columns of a smooth covariance would knock each other out), the 16 largest are taken; every
// workgroup computes the 16 panel rows for its own columns AND, redundantly, the 16 x 16 Schur block of the candidates,
// factors that block with greedy pivoting (identical arithmetic in every workgroup; a candidate whose pivot has fallen
// under the tolerance is dropped and may come back in a later block), and finishes its columns of the new rows by
// forward substitution.  Any pivot order gives G^T G = Sigma; NumPy emulation: same Jacobi sweep count as the exact
// greedy order, no rejected candidates on the Matern frames.
#define PCB_NB 16
#define PCB_CH 128
template <bool ARGS>
__global__ void __launch_bounds__(256) k_pcb_block(PcxArgs args, EdgeDev* edges, int blk, int nw_max, double accept_ratio) {
  PcxEdge E;
  if (ARGS) {
    E = args.e[blockIdx.y];
  } else {
    const EdgeDev& Et = edges[blockIdx.y];
    E.G = Et.G;
    E.Gt = Et.Gt;
    E.pcx_d = Et.pcx_d;
    E.pcx_cand = Et.pcx_cand;
    E.cov = Et.cov;
    E.perm = Et.perm;
    E.eig = Et.eig;
    E.sc = Et.sc;
    E.Lg = Et.Lg;
    E.r_cap = Et.r_cap;
    E.factor_injected = Et.factor_injected;
  }
  const int Lg = E.Lg, j0 = blockIdx.x * PCX_COLS, tid = threadIdx.x;
  if (j0 >= Lg) return;
  const int lane = tid & 63, w = tid >> 6;
  EigState* st = E.eig;
  const gpet_scalars* sc = E.sc;
  // (state of THIS block lives in slot blk & 1, written by the previous launch; this launch writes the other slot)
  const int cur = blk & 1, nxt = cur ^ 1, half = cur;
  const int s_done = sc->done, s_force = sc->force, s_status = sc->status, s_stopped = st->stop_slot[cur];
  const int t0 = st->t_slot[cur];
  const double tol_prev = st->tol;
  const int nwe = (Lg + PCX_COLS - 1) / PCX_COLS;  // <= 64 (the launcher takes the one-pivot path for wider edges)
  const double* cand = E.pcx_cand + (size_t)half * 2 * nw_max;
  double cv = -2.0;
  int ci = 0x7FFFFFFF;
  if (lane < nwe) {
    cv = cand[2 * lane];
    ci = (int)cand[2 * lane + 1];
  }
  const int cc = tid & 31, kg = tid >> 5;
  const int j = j0 + cc;
  double dj = (j < Lg) ? E.pcx_d[j] : -1.0;
  if ((s_done && !s_force) || s_status != GPET_OK || E.factor_injected) return;
  if (s_stopped) {
    if (blockIdx.x == 0 && tid == 0) {
      st->t_slot[nxt] = t0;
      st->stop_slot[nxt] = 1;
    }
    return;
  }
  __shared__ int s_cand[PCB_NB];
  __shared__ double s_cval[PCB_NB];
  __shared__ int s_nc;
  __shared__ double s_gp[PCB_NB][PCB_CH + 1];
  __shared__ double s_P[PCB_NB][PCX_COLS + 1];
  __shared__ double s_S[PCB_NB][PCB_NB + 1];
  __shared__ double s_l[PCB_NB][PCB_NB + 1];  // s_l[i][a]: coefficient of candidate i on accepted pivot a
  __shared__ int s_ord[PCB_NB];
  __shared__ int s_na;
  __shared__ double s_bv[4];
  __shared__ int s_bi[4];
  if (w == 0) {
    // the PCB_NB largest candidates, ties to the smaller column
    for (int k = 0; k < PCB_NB; ++k) {
      double bv = cv;
      int bi = ci;
      pcx_argmax_wave(bv, bi);
      if (lane == 0) {
        s_cand[k] = bi;
        s_cval[k] = bv;
      }
      if (ci == bi) cv = -2.0;  // taken
    }
  }
  __syncthreads();
  const double tol = (blk == 0 && t0 == 0) ? s_cval[0] * 1e-14 : tol_prev;
  int nc = 0;
  for (int k = 0; k < PCB_NB; ++k) nc += (s_cval[k] > tol && s_cval[k] > 0.0) ? 1 : 0;  // (sorted: the first nc)
  if (nc > E.r_cap - t0) nc = E.r_cap - t0;
  if (nc <= 0) {
    if (blockIdx.x == 0 && tid == 0) {
      st->t_slot[nxt] = t0;
      st->stop_slot[nxt] = 1;
    }
    return;
  }
  if (blk == 0 && t0 == 0 && blockIdx.x == 0 && tid == 0) st->tol = tol;
  // panel rows of this workgroup's columns and the Schur block of the candidates, previous rows streamed in chunks
  const int ld = E.r_cap;
  double accP0 = 0.0, accP1 = 0.0, accS = 0.0;
  const int sk = tid >> 4, sm = tid & 15;
  for (int s0 = 0; s0 < t0; s0 += PCB_CH) {
    const int len = (t0 - s0) < PCB_CH ? (t0 - s0) : PCB_CH;
    __syncthreads();
    for (int e = tid; e < PCB_NB * PCB_CH; e += 256) {
      const int k = e / PCB_CH, s = e - k * PCB_CH;
      s_gp[k][s] = (k < nc && s < len) ? E.Gt[(size_t)s_cand[k] * ld + s0 + s] : 0.0;
    }
    __syncthreads();
    if (j < Lg) {
      const double* __restrict__ gcol = E.G + (size_t)s0 * Lg + j;
      for (int s = 0; s < len; ++s) {
        const double gv = gcol[(size_t)s * Lg];
        accP0 += gv * s_gp[2 * kg][s];
        accP1 += gv * s_gp[2 * kg + 1][s];
      }
    }
    for (int s = 0; s < len; ++s) accS += s_gp[sk][s] * s_gp[sm][s];
  }
  if (j < Lg) {
    s_P[2 * kg][cc] = (2 * kg < nc) ? E.cov[(size_t)s_cand[2 * kg] * Lg + j] - accP0 : 0.0;
    s_P[2 * kg + 1][cc] = (2 * kg + 1 < nc) ? E.cov[(size_t)s_cand[2 * kg + 1] * Lg + j] - accP1 : 0.0;
  } else {
    s_P[2 * kg][cc] = 0.0;
    s_P[2 * kg + 1][cc] = 0.0;
  }
  s_S[sk][sm] = (sk < nc && sm < nc) ? E.cov[(size_t)s_cand[sk] * Lg + s_cand[sm]] - accS : 0.0;
  __syncthreads();
  // greedy pivoted Cholesky of the nc x nc Schur block: lanes 0..15 of wave 0 hold one candidate each (its row of the
  // factor in registers, the pivot's row broadcast by shuffles); every workgroup gets the same result
  if (w == 0) {
    const int i = lane & 15;
    double Sd = s_S[i][i];
    bool used = i >= nc;
    double li[PCB_NB];
#pragma unroll
    for (int m = 0; m < PCB_NB; ++m) li[m] = 0.0;
    int na = 0;
    bool going = true;
    double first_piv = 0.0;
#pragma unroll
    for (int a = 0; a < PCB_NB; ++a) {
      if (going && a < nc) {
        double bv = used ? -1.0 : Sd;
        int bi = i;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          const double ov = __shfl_xor(bv, o, WAVE);
          const int oi = __shfl_xor(bi, o, WAVE);
          if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
          }
        }
        const int k = bi;
        // (accept_ratio > 0: a block ends where the next pivot falls under that fraction of its first -- the global maximum --
        //  so that the order stays within a constant of the greedy one: factors that end at the rank cap)
        if (a == 0) first_piv = bv;
        if (!(bv > tol) || !(bv > 0.0) || bv < accept_ratio * first_piv) {
          going = false;
        } else {
          const double piv = sqrt(bv);
          double v = s_S[i][k];
#pragma unroll
          for (int m = 0; m < PCB_NB; ++m)
            if (m < a) v -= li[m] * __shfl(li[m], k, WAVE);
          const double la = (i == k) ? piv : v / piv;
          li[a] = la;
          if (i == k) used = true;
          else if (!used) Sd -= la * la;
          if (lane == 0) s_ord[a] = k;
          na = a + 1;
        }
      }
    }
    if (lane < PCB_NB) {
#pragma unroll
      for (int m = 0; m < PCB_NB; ++m) s_l[i][m] = li[m];
    }
    if (lane == 0) s_na = na;
  }
  __syncthreads();
  const int na = s_na;
  if (na == 0) {  // (cannot happen while the first candidate is above the tolerance; defensive)
    if (blockIdx.x == 0 && tid == 0) {
      st->t_slot[nxt] = t0;
      st->stop_slot[nxt] = 1;
    }
    return;
  }
  // this workgroup's columns of the new rows: r_a = (P[k_a] - sum_{m<a} l[k_a][m] r_m) / l[k_a][a]
  double nb_v = -1.0;
  int nb_i = 0x7FFFFFFF;
  if (tid < PCX_COLS && j < Lg) {
    double r[PCB_NB];
    for (int a = 0; a < na; ++a) {
      const int k = s_ord[a];
      double v = s_P[k][cc];
      for (int m = 0; m < a; ++m) v -= s_l[k][m] * r[m];
      v = v / s_l[k][a];
      if (!(dj >= 0.0)) v = 0.0;  // a column pivoted earlier has no entries in later rows
      r[a] = v;
      E.G[(size_t)(t0 + a) * Lg + j] = v;
      E.Gt[(size_t)j * ld + t0 + a] = v;
      if (j == s_cand[k]) {
        dj = -1.0;
        E.perm[t0 + a] = j;
      } else if (dj >= 0.0) {
        const double nd = dj - v * v;
        dj = nd > 0.0 ? nd : 0.0;
      }
    }
    E.pcx_d[j] = dj;
    nb_v = dj;
    nb_i = j;
  }
  if (tid < WAVE) {
    pcx_argmax_wave(nb_v, nb_i);
    if (tid == 0) {
      double* cnx = E.pcx_cand + (size_t)nxt * 2 * nw_max;
      cnx[2 * blockIdx.x] = nb_v;
      cnx[2 * blockIdx.x + 1] = (double)nb_i;
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    st->t_slot[nxt] = t0 + na;
    st->stop_slot[nxt] = (t0 + na >= E.r_cap) ? 1 : 0;
    st->rank = t0 + na;     // (read by k_pcb_fin and the Jacobi kernels: later launches)
    st->cand_half = nxt;
  }
}

// per-workgroup candidates for the first block
__global__ void __launch_bounds__(64) k_pcb_init(EdgeDev* edges, int nw_max) {
  const EdgeDev E = edges[blockIdx.y];
  if (eig_skip(E)) return;
  const int Lg = E.Lg, j0 = blockIdx.x * PCX_COLS, tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0) {
    EigState* st = E.eig;
    st->tol = 0.0;
    st->stopped = 0;
    st->rank = 0;
    st->maxrel_bits = 0ull;
    st->converged = 0;
    st->sweeps = 0;
    st->bar = 0u;
    st->ticket = 0u;
    st->verdicts = 0;
    st->cand_half = 0;
    st->warm = 0;
    st->t_slot[0] = 0;
    st->stop_slot[0] = 0;
    st->t_slot[1] = 0;
    st->stop_slot[1] = 0;
  }
  if (j0 >= Lg) return;
  const int j = j0 + tid;
  double bv = -1.0;
  int bi = 0x7FFFFFFF;
  if (tid < PCX_COLS && j < Lg) {
    const double d = E.cov[(size_t)j * Lg + j];
    E.pcx_d[j] = d;
    bv = d;
    bi = j;
  }
  pcx_argmax_wave(bv, bi);
  if (tid == 0) {
    E.pcx_cand[2 * blockIdx.x] = bv;
    E.pcx_cand[2 * blockIdx.x + 1] = (double)bi;
  }
  (void)nw_max;
}

__global__ void __launch_bounds__(64) k_pcb_fin(EdgeDev* edges, int nw_max) {
  const EdgeDev E = edges[blockIdx.x];
  if (eig_skip(E)) return;
  EigState* st = E.eig;
  gpet_scalars* sc = E.sc;
  if (threadIdx.x != 0) return;
  const int rank = st->rank;
  if (rank < E.Lg) {  // stopped short of full rank: by the tolerance (fine) or by the capacity / launch budget
    const int nwe = (E.Lg + PCX_COLS - 1) / PCX_COLS;
    const double* cand = E.pcx_cand + (size_t)st->cand_half * 2 * nw_max;
    double rem = 0.0;
    for (int i = 0; i < nwe; ++i) rem = cand[2 * i] > rem ? cand[2 * i] : rem;
    if (rem > st->tol * 1e4) sc->status = GPET_ERR_RANK_CAP;
  }
  st->stopped = 1;
  sc->rank = rank;
}

__global__ void __launch_bounds__(64) k_pcx_fin(EdgeDev* edges, int steps, int nw_max) {
  const EdgeDev E = edges[blockIdx.x];
  if (eig_skip(E)) return;
  EigState* st = E.eig;
  gpet_scalars* sc = E.sc;
  if (threadIdx.x != 0) return;
  if (!st->stopped) {
    const int cap = steps < E.r_cap ? steps : E.r_cap;
    st->rank = cap;
    st->stopped = 1;
    if (cap < E.Lg) {  // capacity reached before the tolerance: what is left must be negligible
      const int ncand = 4 * ((E.Lg + PCX_COLS - 1) / PCX_COLS);
      const double* cand = E.pcx_cand + (size_t)(cap & 1) * 8 * nw_max;
      double rem = 0.0;
      for (int i = 0; i < ncand; ++i) rem = cand[2 * i] > rem ? cand[2 * i] : rem;
      if (rem > st->tol * 1e4) sc->status = GPET_ERR_RANK_CAP;
    }
  }
  sc->rank = st->rank;
}


// ---- 1b. warm start: rows that are nearly orthogonal before the first rotation ----------------------------------------------
// The posterior covariance of an iteration differs from the last one's by the few observations the iteration added.  With
// the previous factor rows A (A^T A = Sigma_prev, rows orthogonal, |A_k|^2 = s_k) the matrix M' = A Sigma A^T is close to
// diag(s_k^2), its Cholesky factor L' (M' = L' L'^T) close to diagonal, and
//       X = L'^T diag(1 / s_k) A        satisfies        X^T X = A^T S^-1 (A Sigma A^T) S^-1 A = Sigma
// (A^T S^-1 A = I for rows of full rank) with rows that are ALREADY nearly orthogonal: the one-sided Jacobi then needs about
// half the sweeps the pivoted-Cholesky rows need (tests/analysis/onesided_warm_start.py: 10 -> 5 at 256 columns).  Three
// products on the f64 matrix cores (k_ojw_gemm: 64 x 64 tiles, K in chunks of 32 through LDS), a blocked Cholesky
// (k_ojw_chol_*: 64-wide panels), one copy.  Only for rows of full rank (Matern); a non-positive pivot falls back to the
// pivoted Cholesky, whose launches are no-ops otherwise.  T = A Sigma lives in Gt, M' and L' in G, X in Gt and then G.
int& gpet_opt_oj_warm() {
  static const int i_ = option_index("oj_warm");
  int& v = option_at(i_);
  return v;
}
__device__ __forceinline__ const double* ojw_source(const EdgeDev& E, int warm) {
  const gpet_scalars* sc = E.sc;
  const int k = sc->iter;
  if (!warm || E.Lg > E.r_cap) return nullptr;
  if (k < 1) {  // a trace's first factor: the previous FRAME's last rows, if the host carried them over (gpet_batch_set_images
                // with GPET_IMAGES_NEXT_FRAME; every other restart clears the tag)
    const int carry = E.ap_tag[2];
    return carry > 0 ? E.Ap + (size_t)(carry - 1) * E.r_cap * E.Lg : nullptr;
  }
  const int slot = (k - 1) & 1;
  if (E.ap_tag[slot] != k) return nullptr;
  return E.Ap + (size_t)slot * E.r_cap * E.Lg;
}
// 1 / |A_k|^2 into theta, and the decision (after k_pcb_init / k_pcx_init, which reset the state)
__global__ void __launch_bounds__(256) k_ojw_begin(EdgeDev* edges, int warm) {
  const EdgeDev E = edges[blockIdx.y];
  if (eig_skip(E)) return;
  const double* Ap = ojw_source(E, warm);
  if (!Ap) return;
  const int Lg = E.Lg, k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0) E.eig->warm = 1;
  if (k >= Lg) return;
  const double* __restrict__ row = Ap + (size_t)k * Lg;
  double s = 0.0;
  for (int j = lane; j < Lg; j += WAVE) s += row[j] * row[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, WAVE);
  if (lane == 0) E.theta[k] = s > 0.0 ? 1.0 / s : 0.0;
}
// MODE 0: T = A Sigma (into Gt)   1: M' = T A^T (lower tiles, into G)   2: X = (diag(1 / s) L')^T A, L' the lower triangle of G (into Gt)
template <int MODE>
__global__ void __launch_bounds__(256) k_ojw_gemm(EdgeDev* edges, int warm) {
  const EdgeDev E = edges[blockIdx.z];
  if (eig_skip(E) || E.eig->warm != 1) return;
  const double* __restrict__ Ap = ojw_source(E, warm);
  if (!Ap) return;
  const int n = E.Lg;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  if (i0 >= n || j0 >= n) return;
  if (MODE == 1 && j0 > i0) return;
  const double* __restrict__ Am = MODE == 0 ? Ap : (MODE == 1 ? E.Gt : E.G);
  const double* __restrict__ Bm = MODE == 0 ? E.cov : Ap;
  double* __restrict__ Cm = MODE == 1 ? E.G : E.Gt;
  const double* __restrict__ inv_s = E.theta;
  __shared__ double sA[64][33];  // [row of C][k]
  __shared__ double sB[32][65];  // [k][column of C]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  v4f64e acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4f64e){0.0, 0.0, 0.0, 0.0};
  for (int k0 = (MODE == 2 ? i0 : 0); k0 < n; k0 += 32) {
    double va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + 256 * u;
      if (MODE == 2) {  // A(i, l) = L'[l][i] / s_l for l >= i
        const int kk = e >> 6, ss = e & 63, l = k0 + kk, i = i0 + ss;
        va[u] = (l < n && i < n && l >= i) ? Am[(size_t)l * n + i] * inv_s[l] : 0.0;
      } else {
        const int ss = e >> 5, kk = e & 31, i = i0 + ss, k = k0 + kk;
        va[u] = (i < n && k < n) ? Am[(size_t)i * n + k] : 0.0;
      }
      if (MODE == 1) {  // B(k, j) = A[j][k]
        const int jj = e >> 5, kk = e & 31, j = j0 + jj, k = k0 + kk;
        vb[u] = (j < n && k < n) ? Bm[(size_t)j * n + k] : 0.0;
      } else {
        const int kk = e >> 6, jj = e & 63, j = j0 + jj, k = k0 + kk;
        vb[u] = (j < n && k < n) ? Bm[(size_t)k * n + j] : 0.0;
      }
    }
    __syncthreads();  // (the previous chunk's operands have been read)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + 256 * u;
      if (MODE == 2) sA[e & 63][e >> 6] = va[u];
      else sA[e >> 5][e & 31] = va[u];
      if (MODE == 1) sB[e & 31][e >> 5] = vb[u];
      else sB[e >> 6][e & 63] = vb[u];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; kk += 4) {
      const double a = sA[16 * w + li][kk + lq];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sB[kk + lq][16 * t + li], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = i0 + 16 * w + lq + 4 * g, j = j0 + 16 * t + li;
      if (i < n && j < n) Cm[(size_t)i * n + j] = acc[t][g];
    }
}
// blocked Cholesky of the lower triangle of G (n x n, row stride n), 64-wide panels: diagonal block, rows below it, trailing update
// (both kernels: thread = (row r = tid & 63, column class tid >> 6: columns class, class + 4, ...): the other factor of an
//  update, A[cc][c], is the same address for the 64 lanes of a wave -- an LDS broadcast --, rows are 65 doubles apart -- no bank
//  conflicts --, and no index needs a division; ONE barrier per column.  The first forms (element e -> e / width, e % width with
//  a run-time width; a thread per row in the substitution) took 62 and 56 us per call, 2 ms per factorisation.)
__global__ void __launch_bounds__(256) k_ojw_chol_diag(EdgeDev* edges, int k0, int inject_failure) {
  const EdgeDev E = edges[blockIdx.y];
  if (eig_skip(E) || E.eig->warm != 1) return;
  const int n = E.Lg;
  if (k0 >= n) return;
  const int nb = n - k0 < 64 ? n - k0 : 64;
  __shared__ double sD[64][65];
  const int tid = threadIdx.x, r = tid & 63, cq = tid >> 6;
  double* __restrict__ K = E.G;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int i = e >> 6, j = e & 63;
    sD[i][j] = (i < nb && j <= i) ? K[(size_t)(k0 + i) * n + k0 + j] : ((i == j) ? 1.0 : 0.0);  // (identity beyond the matrix)
  }
  __syncthreads();
  // outer-product form without square roots: A[r][cc] -= A[r][c] A[cc][c] / A[c][c] for c < cc <= r; L[r][c] = A[r][c] / sqrt(A[c][c])
  // at the end.  A non-positive pivot (every thread sees the same one) ends the warm start.
  bool bad = inject_failure != 0;  // (option oj_warm_fail: the tests' way into the fallback)
  for (int c = 0; c < 63 && !bad; ++c) {
    const double piv = sD[c][c];
    if (!(piv > 0.0)) {
      bad = true;
      break;
    }
    const double f = sD[r][c] / piv;  // (r > c only matters)
    // (a fixed trip count: the 16 pairs of LDS reads of a thread are in flight together -- with run-time bounds every
    //  update waited for its own loads, ~150 cycles each)
    double o[16], a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int cc = cq + 4 * m;
      o[m] = sD[cc][c];
      a[m] = sD[r][cc];
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int cc = cq + 4 * m;
      if (cc > c && cc <= r) sD[r][cc] = a[m] - f * o[m];
    }
    __syncthreads();
  }
  if (!bad && !(sD[63][63] > 0.0)) bad = true;
  if (bad) {
    if (tid == 0) E.eig->warm = 2;  // not positive definite to rounding: the pivoted Cholesky takes over
    return;
  }
  for (int e = tid; e < 64 * 64; e += 256) {
    const int i = e >> 6, j = e & 63;
    if (i < nb && j <= i) K[(size_t)(k0 + i) * n + k0 + j] = (i == j) ? sqrt(sD[j][j]) : sD[i][j] / sqrt(sD[j][j]);
  }
}
__global__ void __launch_bounds__(256) k_ojw_chol_trsm(EdgeDev* edges, int k0) {
  const EdgeDev E = edges[blockIdx.y];
  if (eig_skip(E) || E.eig->warm != 1) return;
  const int n = E.Lg;
  const int i0 = k0 + 64 * ((int)blockIdx.x + 1);
  if (i0 >= n) return;
  __shared__ double sL[64][65];
  __shared__ double sX[64][65];
  const int tid = threadIdx.x, r = tid & 63, cq = tid >> 6;
  double* __restrict__ K = E.G;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int i = e >> 6, j = e & 63;
    sL[i][j] = (j <= i) ? K[(size_t)(k0 + i) * n + k0 + j] : 0.0;
    sX[i][j] = (i0 + i < n) ? K[(size_t)(i0 + i) * n + k0 + j] : 0.0;
  }
  __syncthreads();
  // X L_kk^T = A by columns: x[r][c] = a[r][c] / L[c][c] is final once every earlier column has left it; the thread that owns
  // (r, c + 1) divides it right after its update, so the next trip finds it final
  if (cq == 0) sX[r][0] = sX[r][0] / sL[0][0];
  __syncthreads();
  for (int c = 0; c < 63; ++c) {
    const double x = sX[r][c];
    const double dn = sL[c + 1][c + 1];
    double o[16], a[16];  // (fixed trip count: all the LDS reads of a trip in flight together)
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int cc = cq + 4 * m;
      o[m] = sL[cc][c];
      a[m] = sX[r][cc];
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int cc = cq + 4 * m;
      if (cc > c) sX[r][cc] = a[m] - x * o[m];
    }
    // (the owner of column c + 1 finishes it: one division per thread and trip, not one per element)
    if (cq == ((c + 1) & 3)) sX[r][c + 1] = sX[r][c + 1] / dn;
    __syncthreads();
  }
  for (int e = tid; e < 64 * 64; e += 256) {
    const int i = e >> 6, j = e & 63;
    if (i0 + i < n) K[(size_t)(i0 + i) * n + k0 + j] = sX[i][j];
  }
}
__global__ void __launch_bounds__(256) k_ojw_chol_syrk(EdgeDev* edges, int k0) {
  const EdgeDev E = edges[blockIdx.z];
  if (eig_skip(E) || E.eig->warm != 1) return;
  const int n = E.Lg;
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int i0 = k0 + 64 * (bi + 1), j0 = k0 + 64 * (bj + 1);
  if (i0 >= n) return;
  __shared__ double sI[64][65];
  __shared__ double sJ[64][65];
  const int tid = threadIdx.x;
  double* __restrict__ K = E.G;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int i = e >> 6, j = e & 63;
    sI[i][j] = (i0 + i < n) ? K[(size_t)(i0 + i) * n + k0 + j] : 0.0;
    sJ[i][j] = (j0 + i < n) ? K[(size_t)(j0 + i) * n + k0 + j] : 0.0;
  }
  __syncthreads();
  const int lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  v4f64e acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4f64e){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 64; kk += 4) {
    const double a = sI[16 * w + li][kk + lq];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sJ[16 * t + li][kk + lq], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = i0 + 16 * w + lq + 4 * g, j = j0 + 16 * t + li;
      if (i < n && j <= i) K[(size_t)i * n + j] -= acc[t][g];
    }
}
// X (in Gt) becomes the rows G the Jacobi works on; the pivoted Cholesky's launches see "finished"
__global__ void __launch_bounds__(256) k_ojw_commit(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  if (eig_skip(E) || E.eig->warm != 1) return;
  const size_t cnt = (size_t)E.Lg * E.Lg;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < cnt; e += (size_t)gridDim.x * 256) E.G[e] = E.Gt[e];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    EigState* st = E.eig;
    st->rank = E.Lg;
    st->t_slot[0] = st->t_slot[1] = E.Lg;
    st->stop_slot[0] = st->stop_slot[1] = 1;
    st->stopped = 1;
  }
}

// ---- 2. one-sided block Jacobi on the rows of G --------------------------------------------------------------------
__device__ __forceinline__ int oj_row(int bI, int bJ, int t) { return t < OJ_B ? bI * OJ_B + t : bJ * OJ_B + (t - OJ_B); }

// rotation (c, s) that annihilates the coupling apq of a symmetric 2x2 block [app apq; apq aqq]: the smaller angle
// (c >= 1/sqrt 2).  With rho = sqrt(d^2 + 4 apq^2), d = aqq - app:  c^2 = (rho + |d|) / (2 rho),
// s = +-apq / (rho c)  -- two reciprocal square roots (hardware seed + two Newton steps each), no division.  Branchless:
// the "nothing to rotate" test selects at the end (a dependent f64 instruction costs 8.4 cycles and an exec-mask branch as
// much again: tools/ubench/f64_chain.hip, jac_params_chain.hip; this chain is the inner round's critical path).
__device__ __forceinline__ void oj_rotation(double app, double apq, double aqq, double& c, double& s, double& rel2) {
  const double den = fabs(app * aqq), num = apq * apq;
  rel2 = den > 0.0 ? num * __builtin_amdgcn_rcp(den) : 0.0;  // (hardware reciprocal: a convergence measure, not arithmetic)
  const bool on = num > 1e-34 * den && fabs(apq) > 1e-300;
  const double d = aqq - app, hh = 2.0 * apq;
  const double rho2 = d * d + hh * hh;
  double y = __builtin_amdgcn_rsq(rho2);  // 1 / rho
  y = y * (1.5 - 0.5 * rho2 * y * y);
  y = y * (1.5 - 0.5 * rho2 * y * y);
  const double x = 0.5 + 0.5 * fabs(d) * y;  // c^2 in [1/2, 1]
  double z = __builtin_amdgcn_rsq(x);        // 1 / c
  z = z * (1.5 - 0.5 * x * z * z);
  z = z * (1.5 - 0.5 * x * z * z);
  c = on ? x * z : 1.0;
  s = on ? (d >= 0.0 ? hh : -hh) * (0.5 * y) * z : 0.0;
}

// One cyclic sweep (15 rounds of 8 disjoint rotations) on the symmetric 16x16 matrix s_C, executed by ONE wave: lane = 2x2
// block (a_, b_) of the pairing, upper triangle of blocks active, mirrored writes; the accumulated rotation goes to s_R
// (s_R <- s_R J).  Wave-synchronous LDS traffic only -- no workgroup barrier inside.
// SEATED (round 5; the scheme of k_jacobi_seat): the matrix is addressed by seat, not by player -- the players of pair k
// always sit in seats 2 k and 2 k + 1, and after a round everybody on the circle moves one seat on (the circle method of
// oj_rr_pair: the same pairs in the same order).  A lane's block, its two pairs' diagonal blocks and its two rows of s_R
// are then at FIXED addresses (four ds_read2_b64 + four more for the diagonals and s_R instead of fourteen reads at
// addresses recomputed every round), and its results go to fixed places of the next round's layout, computed once per
// kernel (OjSeat).  After the 15 rounds everybody is back in the seat he started from.  The callers keep s_C and s_R in
// SEAT order (oj_seat_of: local row -> seat): 21.5 k -> ~8 k cycles per sweep, which was 42 % of a pair slot.
__device__ __forceinline__ int oj_seat_of(int t) { return t < 8 ? 2 * t : 31 - 2 * t; }  // local row 0..15 -> its seat in round 0
__device__ __forceinline__ int oj_seat_next(int slot) {  // where the player of `slot` sits next round (16 players, 15 rounds)
  const int k = slot >> 1;
  if (slot == 1) return 1;  // the pivot
  int seat = (slot & 1) ? 15 - k : k;
  seat = seat == 0 ? 14 : seat - 1;
  if (seat == 0) return 0;
  return seat < 8 ? 2 * seat : 2 * (15 - seat) + 1;
}
struct OjSeat {  // per-lane element offsets into s_C / s_R (row stride OJ_M + 1)
  int rc, rda, rdb, rr;  // reads: the lane's block, the diagonal blocks of its row / column pair, its two rows of s_R
  int wc[4], wm[4], wr[2];  // writes: the block's four entries, their mirror images, the two columns of s_R
  bool act, offd;
};
__device__ __forceinline__ void oj_seat_init(OjSeat& T, int lane) {
  constexpr int ld = OJ_M + 1;
  const int a_ = lane >> 3, b_ = lane & 7, i0 = lane >> 3;
  T.act = a_ <= b_;
  T.offd = a_ != b_;
  T.rc = 2 * a_ * ld + 2 * b_;
  T.rda = 2 * a_ * ld + 2 * a_;
  T.rdb = 2 * b_ * ld + 2 * b_;
  T.rr = i0 * ld + 2 * b_;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i2 = oj_seat_next(2 * a_ + (q >> 1)), j2 = oj_seat_next(2 * b_ + (q & 1));
    T.wc[q] = i2 * ld + j2;
    T.wm[q] = j2 * ld + i2;
  }
  T.wr[0] = i0 * ld + oj_seat_next(2 * b_);
  T.wr[1] = i0 * ld + oj_seat_next(2 * b_ + 1);
}
__device__ __forceinline__ double oj_inner_sweep(double (*s_C)[OJ_M + 1], double (*s_R)[OJ_M + 1], const OjSeat& T) {
  constexpr int ld = OJ_M + 1;
  double mr = 0.0;  // largest squared relative coupling rotated away (this lane's pairs)
  double* C = &s_C[0][0];
  double* R = &s_R[0][0];
#pragma unroll 1
  for (int rnd = 0; rnd < OJ_M - 1; ++rnd) {
    // every read of the round first; both rotations of this lane's block are computed here (two independent dependency
    // chains that interleave: the same latency as one, and no cross-lane traffic); lanes that share a pair compute
    // identical values
    const double a00 = C[T.rda], a01 = C[T.rda + 1], a11 = C[T.rda + ld + 1];
    const double d00 = C[T.rdb], d01 = C[T.rdb + 1], d11 = C[T.rdb + ld + 1];
    const double b00 = C[T.rc], b01 = C[T.rc + 1], b10 = C[T.rc + ld], b11 = C[T.rc + ld + 1];
    const double r0p = R[T.rr], r0q = R[T.rr + 1], r1p = R[T.rr + 8 * ld], r1q = R[T.rr + 8 * ld + 1];
    double ca, sa, cb, sb, ra2, rb2;
    oj_rotation(a00, a01, a11, ca, sa, ra2);
    oj_rotation(d00, d01, d11, cb, sb, rb2);
    mr = rb2 > mr ? rb2 : mr;  // (every pair is some lane's b_)
    (void)ra2;
    const double t00 = cb * b00 - sb * b01, t01 = sb * b00 + cb * b01;
    const double t10 = cb * b10 - sb * b11, t11 = sb * b10 + cb * b11;
    const double n00 = ca * t00 - sa * t10, n10 = sa * t00 + ca * t10;
    const double n01 = ca * t01 - sa * t11, n11 = sa * t01 + ca * t11;
    __builtin_amdgcn_wave_barrier();  // every read of this round precedes its writes
    if (T.act) {
      C[T.wc[0]] = n00;
      C[T.wc[1]] = n01;
      C[T.wc[2]] = n10;
      C[T.wc[3]] = n11;
      if (T.offd) {
        C[T.wm[0]] = n00;
        C[T.wm[1]] = n01;
        C[T.wm[2]] = n10;
        C[T.wm[3]] = n11;
      }
    }
    R[T.wr[0]] = cb * r0p - sb * r0q;
    R[T.wr[1]] = sb * r0p + cb * r0q;
    R[T.wr[0] + 8 * ld] = cb * r1p - sb * r1q;
    R[T.wr[1] + 8 * ld] = sb * r1p + cb * r1q;
    __builtin_amdgcn_wave_barrier();
  }
  return mr;
}

// the sweep's convergence measure: largest squared relative coupling any rotation of this visit met
__device__ __forceinline__ void oj_report(double mr, int lane, EigState* st) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(mr, o, WAVE);
    mr = ov > mr ? ov : mr;
  }
  if (lane == 0 && mr > 0.0) atomicMax(&st->maxrel_bits, (unsigned long long)__double_as_longlong(mr));
}

// A pair of blocks whose 120 couplings are ALL four orders below the stopping tolerance (OJ_SKIP_REL2) is left alone: no
// inner sweep, no row update (the identity preserves G^T G = Sigma better than any rotation, and rotating a coupling of
// 1e-12 only moves it below f64 resolution).  That is every pair of the LAST sweep -- the one that only confirms what
// the sweep before it achieved --, whose rounds then cost the Gram matrix only.  (Skipping at the tolerance itself,
// 1e-8, was measured too: 12.0 instead of 12.9 ms per factor, but the rows then stay 1e-8 from orthogonal, 7.6e-8 from
// the cold start's, and the comparison with LAPACK fails its 1e-10.)  Returns the largest squared relative
// coupling of the block pair in every lane (the measure oj_rotation reports: hardware reciprocal).
__device__ __forceinline__ double oj_pair_coupling(const double (*s_C)[OJ_M + 1], int lane) {
  double mr = 0.0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = lane + 64 * u, i = e >> 4, j = e & 15;
    if (i < j) {
      const double den = fabs(s_C[i][i] * s_C[j][j]), num = s_C[i][j] * s_C[i][j];
      const double r2 = den > 0.0 ? num * __builtin_amdgcn_rcp(den) : 0.0;
      mr = r2 > mr ? r2 : mr;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(mr, o, WAVE);
    mr = ov > mr ? ov : mr;
  }
  return mr;
}

// K index of matrix instruction jj, lane group lg -> row of the 16-row panel.  Lane groups 0/1 (and 2/3) share an
// LDS cycle: their rows are 8 apart, which with a row stride of 2 (mod 32) doubles puts them on disjoint banks.
__device__ __forceinline__ int oj_krow(int jj, int lg) { return 8 * (lg & 1) + 4 * (lg >> 1) + jj; }

// One round of the block Jacobi for one pair of 8-row blocks.  STAGED: the 16 rows (16 x Lg doubles, <= 131 KB) are
// read from HBM/L2 ONCE into LDS with fully coalesced loads; the Gram matrix and the row update both take their
// matrix-core operands from there.  Not STAGED (Lg > OJ_STAGE_MAX): operands straight from global memory, the update's
// operands requested before the sweep so that they arrive while wave 0 rotates.
// What a round needs of an edge.  For batches of up to OJ_ARGS_MAXB edges the launcher passes these by value in the
// kernel arguments, which removes the dependent round trip through the edge table from every one of the ~1 600 round
// launches of a factorisation (a launch starts with cold caches: every dependent global access costs ~2 us).
struct OjEdge {
  double* G;
  EigState* st;
  const gpet_scalars* sc;
  int Lg, r_cap, injected, pad;
};
struct OjArgs {
  OjEdge e[OJ_ARGS_MAXB];
};

template <bool STAGED, bool ARGS>
__global__ void __launch_bounds__(256) k_oj_round(OjArgs args, EdgeDev* edges, int round, int nblk, double tol2) {
  OjEdge D;
  if (ARGS) {
    D = args.e[blockIdx.y];
  } else {
    const EdgeDev& Et = edges[blockIdx.y];
    D.G = Et.G;
    D.st = Et.eig;
    D.sc = Et.sc;
    D.Lg = Et.Lg;
    D.r_cap = Et.r_cap;
    D.injected = Et.factor_injected;
  }
  struct {
    double* G;
    int Lg;
  } E = {D.G, D.Lg};
  EigState* st = D.st;
  // the edge's state and (STAGED) its 16 rows are requested together: one round trip instead of two.  The rows are
  // read before the rank is known, so the row index is clamped to the buffer and the mask is applied afterwards.
  const int s_done = D.sc->done, s_force = D.sc->force, s_status = D.sc->status;
  const int s_conv = st->converged, rank = st->rank;
  const int Lg = E.Lg;
  int bI, bJ;
  oj_rr_pair(nblk - 1, round, blockIdx.x, bI, bJ);
  extern __shared__ double s_X[];  // STAGED: [16][ldx]
  __shared__ double s_part[4][OJ_M][OJ_M + 1];
  __shared__ double s_C[OJ_M][OJ_M + 1];
  __shared__ double s_R[OJ_M][OJ_M + 1];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  OjSeat seat;  // (the inner sweep's per-lane addresses: once per kernel)
  oj_seat_init(seat, lane);
  const int nch = (Lg + 15) >> 4;  // 16-column chunks = 16-column tiles of the update below
  const int ldx = ((Lg + 31) & ~31) + 2;
  double v[STAGED ? 16 : 1][4];
  if (STAGED) {
    // -- stage: thread t takes columns t, t + 256, ... of every row: 512 contiguous bytes per wave instruction
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int gi = oj_row(bI, bJ, r);
      gi = gi < D.r_cap ? gi : D.r_cap - 1;
      const double* __restrict__ xrow = E.G + (size_t)gi * Lg;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = tid + 256 * i;
        v[r][i] = (k < Lg) ? xrow[k] : 0.0;
      }
    }
  }
  asm volatile("" ::: "memory");  // (the loads above stay above the exits below)
  if ((s_done && !s_force) || s_status != GPET_OK || D.injected || s_conv) return;
  if (bI * OJ_B >= rank) return;  // (bI < bJ: both blocks are empty)
  if (STAGED) {
    const int kmax = nch * 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool live = oj_row(bI, bJ, r) < rank;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = tid + 256 * i;
        if (k < kmax) s_X[r * ldx + k] = live ? v[r][i] : 0.0;
      }
    }
    __syncthreads();
    // -- Gram matrix: A operand == B operand == X[row lr][k]; four independent accumulators per wave
    v4f64e acc[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[jj] = (v4f64e){0.0, 0.0, 0.0, 0.0};
    const double* xr = s_X + lr * ldx + lg;
    for (int ch0 = w; ch0 < nch; ch0 += 16) {  // 4 chunks per batch: 16 LDS reads in flight, then 16 matrix instructions
      double x[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) x[u][jj] = (ch0 + 4 * u < nch) ? xr[(ch0 + 4 * u) * 16 + 4 * jj] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][jj], x[u][jj], acc[jj], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s_part[w][lg + 4 * i][lr] = (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
  } else {
    const int gi = oj_row(bI, bJ, lr);
    const bool valid = gi < rank;
    const double* __restrict__ xrow = E.G + (size_t)gi * Lg;
    v4f64e acc = (v4f64e){0.0, 0.0, 0.0, 0.0};
    for (int ch0 = w; ch0 < nch; ch0 += 4 * OJ_PF) {
      double xv[OJ_PF][4];
#pragma unroll
      for (int u = 0; u < OJ_PF; ++u) {
        const int k0 = (ch0 + 4 * u) * 16 + lg;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int k = k0 + 4 * jj;
          xv[u][jj] = (valid && k < Lg) ? xrow[k] : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < OJ_PF; ++u)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u][jj], xv[u][jj], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s_part[w][lg + 4 * i][lr] = acc[i];
  }
  // not STAGED: operands of the row update requested now (they do not depend on the rotation)
  const double* xb[4];
  bool vb[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int gb = oj_row(bI, bJ, oj_krow(jj, lg));
    vb[jj] = gb < rank;
    xb[jj] = E.G + (size_t)gb * Lg;
  }
  double xp[STAGED ? 1 : OJ_PF][4];
  if (!STAGED) {
#pragma unroll
    for (int u = 0; u < OJ_PF; ++u) {
      const int c = (w + 4 * u) * 16 + lr;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) xp[u][jj] = (vb[jj] && c < Lg) ? xb[jj][c] : 0.0;  // B[K][N = c]
    }
  }
  {
    const int i = tid >> 4, j = tid & 15;
    s_R[i][j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  {
    const int i = tid >> 4, j = tid & 15;
    s_C[oj_seat_of(i)][oj_seat_of(j)] = (s_part[0][i][j] + s_part[1][i][j]) + (s_part[2][i][j] + s_part[3][i][j]);  // (seat order: oj_inner_sweep)
  }
  __syncthreads();
  __shared__ int s_skip;
  if (w == 0) {
    const double mr0 = oj_pair_coupling(s_C, lane);
    const bool skip = mr0 <= tol2 * OJ_SKIP_REL2;  // (uniform over the wave)
    if (lane == 0) s_skip = skip ? 1 : 0;
    if (skip) {
      if (lane == 0 && mr0 > 0.0) atomicMax(&st->maxrel_bits, (unsigned long long)__double_as_longlong(mr0));
    } else {
      oj_report(oj_inner_sweep(s_C, s_R, seat), lane, st);
    }
  }
  __syncthreads();
  if (s_skip) return;  // (the rows stay as they are)
  // -- rows <- R^T rows on the matrix cores: out[a][c] = sum_b R[b][a] X[b][c]; a wave owns whole 16-column tiles
  {
    double ra[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) ra[jj] = s_R[oj_seat_of(oj_krow(jj, lg))][oj_seat_of(lr)];  // A[M = a = lr][K -> row b] (s_R is in seat order)
    double* xo[4];
    bool vo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ga = oj_row(bI, bJ, lg + 4 * i);
      vo[i] = ga < rank;
      xo[i] = E.G + (size_t)ga * Lg;
    }
    if (STAGED) {
      const double* xs[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) xs[jj] = s_X + oj_krow(jj, lg) * ldx + lr;
      for (int ct0 = w; ct0 < nch; ct0 += 16) {  // 4 tiles per batch: 16 LDS reads, 4 independent accumulator chains
        double xv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) xv[u][jj] = (ct0 + 4 * u < nch) ? xs[jj][(ct0 + 4 * u) * 16] : 0.0;
        v4f64e acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = (v4f64e){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[jj], xv[u][jj], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = (ct0 + 4 * u) * 16 + lr;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (vo[i] && ct0 + 4 * u < nch && c < Lg) xo[i][c] = acc[u][i];
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < OJ_PF; ++u) {
        const int c = (w + 4 * u) * 16 + lr;
        v4f64e acc = (v4f64e){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[jj], xp[u][jj], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (vo[i] && c < Lg) xo[i][c] = acc[i];
      }
      for (int ct = w + 4 * OJ_PF; ct < nch; ct += 4) {  // (more than 64 OJ_PF columns: Lg > 1024)
        const int c = ct * 16 + lr;
        const bool cv = c < Lg;
        v4f64e acc = (v4f64e){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const double xv = (vb[jj] && cv) ? xb[jj][c] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[jj], xv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (vo[i] && cv) xo[i][c] = acc[i];
      }
    }
  }
}

// ---- the rounds and sweeps of one factorisation in ONE launch ----------------------------------------------------------
// A round launch costs ~25 us of which ~12 are the round: the rest is start, drain and cold state.
// Here the workgroups of an edge loop over the rounds and sweeps themselves.  Work is handed out by TICKET: a workgroup
// takes the next pair slot (ticket t -> sweep, round, slot; one atomic on the edge's EigState), waits until every slot of
// the earlier rounds has FINISHED (a second monotonic counter), rotates the pair, counts it finished and comes back for
// the next ticket.  Nothing requires the workgroups to be resident together: whoever holds the oldest unfinished ticket
// is running by construction, so fewer resident workgroups than slots (a batch of many edges, other batches' kernels on
// the GPU, another process on the device) only means that a workgroup serves several slots per round.  (Round 3's form
// bound workgroup k to slot k and met at a barrier: it needed all workgroups co-resident, gave up after ~1 s when they
// were not and FAILED the edge -- six traces in flight could starve each other.)  The last slot of a sweep publishes the
// sweep's verdict; tickets of the next sweep wait for it.  The rows a workgroup needs next were written by two other
// workgroups, possibly on another XCD whose L2 is not coherent with this one: rows are read and written with agent-scope
// accesses (sc1: write-through to / read from the level all XCDs share) instead of a cache-wide write-back + invalidate
// per round (a __threadfence() per barrier made a round 29 us: slower than a launch).  Same pairs, same rotations, same
// arithmetic as k_oj_round<STAGED>, whichever workgroup serves a slot: bit-identical rows.  The wait is bounded by WALL
// time (~4 s: a device fault elsewhere, never contention) and then fails the edge instead of hanging the device.
__device__ __forceinline__ bool oj_wait_slot(EigState* st, unsigned int need_done, int need_verdicts, int* s_flag) {
  if (threadIdx.x == 0) {
    int ok = 1;
    const unsigned long long t0 = wall_clock64();  // (constant 100 MHz)
    for (;;) {
      const unsigned int d = __hip_atomic_load(&st->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int v = __hip_atomic_load(&st->verdicts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d >= need_done && v >= need_verdicts) break;
      if (v >= need_verdicts && __hip_atomic_load(&st->converged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // (finished while we waited)
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 400000000ull) {
        ok = 0;
        break;
      }
    }
    *s_flag = ok;
  }
  __syncthreads();
  // (readfirstlane: the verdict is the same in every lane, and the compiler has to KNOW it -- see the loop below)
  return __builtin_amdgcn_readfirstlane(*s_flag) != 0;
}

template <bool ARGS>
__global__ void __launch_bounds__(256) k_oj_persist(OjArgs args, EdgeDev* edges, int nblk, int max_sweeps, double tol2, int half_stage) {
  OjEdge D;
  if (ARGS) {
    D = args.e[blockIdx.y];
  } else {
    const EdgeDev& Et = edges[blockIdx.y];
    D.G = Et.G;
    D.st = Et.eig;
    D.sc = Et.sc;
    D.Lg = Et.Lg;
    D.r_cap = Et.r_cap;
    D.injected = Et.factor_injected;
  }
  EigState* st = D.st;
  // (the same for every workgroup of the edge: they leave together or stay together)
  if ((D.sc->done && !D.sc->force) || D.sc->status != GPET_OK || D.injected || st->converged) return;
  const int rank = st->rank, Lg = D.Lg;
  extern __shared__ double s_X[];  // [16][ldx]
  __shared__ double s_part[4][OJ_M][OJ_M + 1];
  __shared__ double s_C[OJ_M][OJ_M + 1];
  __shared__ double s_R[OJ_M][OJ_M + 1];
  __shared__ int s_flag;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  OjSeat seat;  // (the inner sweep's per-lane addresses: once per kernel)
  oj_seat_init(seat, lane);
  const int nch = (Lg + 15) >> 4;
  // half_stage (round 6; every edge of the launch has an even width, the launcher checks): the panel is staged one 512-column half
  // at a time -- [16][514] doubles = 66 KB instead of 131 KB at 1 024 columns, TWO workgroups per CU.  Both halves arrive in
  // registers anyway (the 32 loads of a thread); the Gram matrix takes them through LDS one after the other, in the order of the
  // one-pass form (the same bits), and the row update does the half that is still in LDS first, then the other half again.
  const bool two_half = half_stage != 0 && Lg > 512;
  const int ldx = half_stage ? 514 : ((Lg + 31) & ~31) + 2;
  const int kmax = nch * 16;
  const unsigned int nslots = (unsigned int)(nblk / 2), per_sweep = nslots * (unsigned int)(nblk - 1);
  unsigned int nbar = 0;  // (slots served by this workgroup)
  __shared__ unsigned int s_ticket;
  __shared__ int s_skip;
#ifdef GPET_OJ_PROF
  long long pt[6] = {0, 0, 0, 0, 0, 0};
#define OJ_T(i) { const long long t_ = clock64(); pt[i] += t_ - tl; tl = t_; }
  long long tl = clock64();
#else
#define OJ_T(i)
#endif
  if (tid == 0) s_ticket = atomicAdd(&st->ticket, 1u);
  for (;;) {
    {
      // Two things the compiler must not get wrong here.  (1) Every exit of this loop has to be a branch it KNOWS to be
      // uniform (hence the readfirstlanes on what comes out of LDS or global memory).  (2) There is exactly ONE
      // `if (tid == 0)` per trip: thread 0 counts the slot finished AND takes the next ticket in the same block at the
      // end of the trip.  With a second one at the top of the loop the compiler threads the two (thread 0: tail -> head;
      // everybody else: neither), which makes two cycles; it nests them, and the lanes that are not thread 0 then run
      // the barriers of the next trip on the old ticket while thread 0 waits for them to leave the loop -- the first
      // form of this kernel hung exactly so.
      __syncthreads();
      const unsigned int t = (unsigned int)__builtin_amdgcn_readfirstlane((int)s_ticket);
      const int sweep = (int)(t / per_sweep);
      const unsigned int rem = t - (unsigned int)sweep * per_sweep;
      const int round = (int)(rem / nslots), slot = (int)(rem - (unsigned int)round * nslots);
      if (sweep >= max_sweeps) break;
      // every slot of the earlier rounds finished, and the verdict of the previous sweep known
      if (!oj_wait_slot(st, t - (unsigned int)slot, sweep, &s_flag)) {
        if (tid == 0) const_cast<gpet_scalars*>(D.sc)->status = GPET_ERR_STATE;
        return;
      }
      if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&st->converged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
      int bI, bJ;
      oj_rr_pair(nblk - 1, round, slot, bI, bJ);
      if (bI * OJ_B < rank) {  // (bI < bJ: otherwise both blocks are empty)
        // -- stage + Gram matrix.  Even widths: thread t takes the column PAIRS 2 t, 2 t + 512 of every row as 16-byte
        //    agent-scope loads -- raw buffer loads with the sc1 policy (__hip_atomic_load has no 16-byte form; the builtin is one
        //    the compiler's wait-count insertion tracks, unlike round 4's inline assembly, and a dead row or a column beyond
        //    the edge is simply an offset beyond the buffer: it reads zero, no branch).  All 32 loads of a thread are issued at
        //    once, columns [0, 512) first: the Gram matrix of that half runs on the matrix cores while the other half is still
        //    in flight (a slot's staging is bound by what ONE CU pulls from beyond its XCD's L2, ~15 k cycles for 128 KB;
        //    the Gram products are 8 k).  Same chunks per wave in the same order as one pass over all columns: the same bits.
        //    Odd widths: thread t takes columns t, t + 256, ... of every row, one pass.
        v4f64e acc[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[jj] = (v4f64e){0.0, 0.0, 0.0, 0.0};
        const double* xr = s_X + lr * ldx + lg;
        auto gram_chunks = [&](int c_lo, int c_hi, int coff) {  // 16-column chunks c_lo + w + 4 u + 16 k < c_hi of this wave; LDS holds chunk c at c - coff
          for (int ch0 = c_lo + w; ch0 < c_hi; ch0 += 16) {
            double x[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) x[u][jj] = (ch0 + 4 * u < c_hi) ? xr[(ch0 + 4 * u - coff) * 16 + 4 * jj] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][jj], x[u][jj], acc[jj], 0, 0, 0);
          }
        };
        typedef double oj_d2 __attribute__((ext_vector_type(2)));
        typedef unsigned int oj_u4 __attribute__((ext_vector_type(4)));
        if ((Lg & 1) == 0) {
          oj_u4 qv[2][16];
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)D.G, 0, (int)((size_t)D.r_cap * (size_t)Lg * sizeof(double)), 0x00020000);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int gi = oj_row(bI, bJ, r), k = 2 * tid + 512 * i;
              const unsigned int off = (gi < rank && k < Lg) ? (unsigned int)(((size_t)gi * Lg + k) * sizeof(double)) : 0xFFFFFFF0u;
              qv[i][r] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);  // (aux 16 = sc1: agent scope)
            }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (h == 1 && Lg <= 512) break;  // (uniform)
            if (h == 1 && two_half) __syncthreads();  // (the Gram products of half 0 have read their columns)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int k = 2 * tid + 512 * h, kl = two_half ? 2 * tid : k;
              if (k < kmax) *reinterpret_cast<oj_d2*>(&s_X[r * ldx + kl]) = __builtin_bit_cast(oj_d2, qv[h][r]);
            }
            __syncthreads();
            OJ_T(0)
            gram_chunks(32 * h, nch < 32 * h + 32 ? nch : 32 * h + 32, two_half ? 32 * h : 0);
            OJ_T(1)
          }
        } else {
          double v[16][4];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int gi = oj_row(bI, bJ, r);
            const bool live = gi < rank;
            gi = gi < D.r_cap ? gi : D.r_cap - 1;
            const double* __restrict__ xrow = D.G + (size_t)gi * Lg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int k = tid + 256 * i;
              v[r][i] = (live && k < Lg) ? __hip_atomic_load(xrow + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int k = tid + 256 * i;
              if (k < kmax) s_X[r * ldx + k] = v[r][i];
            }
          __syncthreads();
          OJ_T(0)
          gram_chunks(0, nch, 0);
          OJ_T(1)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s_part[w][lg + 4 * i][lr] = (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
        {
          const int i = tid >> 4, j = tid & 15;
          s_R[i][j] = (i == j) ? 1.0 : 0.0;
        }
        __syncthreads();
        {
          const int i = tid >> 4, j = tid & 15;
          s_C[oj_seat_of(i)][oj_seat_of(j)] = (s_part[0][i][j] + s_part[1][i][j]) + (s_part[2][i][j] + s_part[3][i][j]);  // (seat order: oj_inner_sweep)
        }
        __syncthreads();
        OJ_T(1)
        if (w == 0) {
          const double mr0 = oj_pair_coupling(s_C, lane);
          const bool skip = mr0 <= tol2 * OJ_SKIP_REL2;  // (uniform over the wave)
          if (lane == 0) s_skip = skip ? 1 : 0;
          if (skip) {
            if (lane == 0 && mr0 > 0.0) atomicMax(&st->maxrel_bits, (unsigned long long)__double_as_longlong(mr0));
          } else {
            oj_report(oj_inner_sweep(s_C, s_R, seat), lane, st);
          }
        }
        __syncthreads();
        OJ_T(2)
        // -- rows <- R^T rows (not for a pair that is within the tolerance already: oj_pair_coupling)
        if (!s_skip) {
          double ra[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) ra[jj] = s_R[oj_seat_of(oj_krow(jj, lg))][oj_seat_of(lr)];  // (s_R is in seat order)
          double* xo[4];
          bool vo[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int ga = oj_row(bI, bJ, lg + 4 * i);
            vo[i] = ga < rank;
            xo[i] = D.G + (size_t)ga * Lg;
          }
          const double* xs[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) xs[jj] = s_X + oj_krow(jj, lg) * ldx + lr;
          auto update_chunks = [&](int c_lo, int c_hi, int coff) {  // chunks c_lo + w + 4 u + 16 k < c_hi; LDS holds chunk c at c - coff
            for (int ct0 = c_lo + w; ct0 < c_hi; ct0 += 16) {
              double xv[4][4];
#pragma unroll
              for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) xv[u][jj] = (ct0 + 4 * u < c_hi) ? xs[jj][(ct0 + 4 * u - coff) * 16] : 0.0;
              v4f64e acc[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) acc[u] = (v4f64e){0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[jj], xv[u][jj], acc[u], 0, 0, 0);
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int c = (ct0 + 4 * u) * 16 + lr;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  if (vo[i] && ct0 + 4 * u < c_hi && c < Lg) __hip_atomic_store(xo[i] + c, acc[u][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
          };
          // (ONE copy of the update loop: two_half runs it twice -- the half the Gram products left in LDS first; then the first half
          //  again from memory (this workgroup owns the 16 rows for the slot: columns [0, 512) are as they were; from its XCD's L2,
          //  where the staging left them).  Kept in registers through the inner sweep instead the kernel needed 320 registers,
          //  requested before the first update 284, with two inlined copies of the loop 260 -- two workgroups per CU allow 256)
          for (int pass = 0; pass < (two_half ? 2 : 1); ++pass) {
            if (pass == 1) {
              const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)D.G, 0, (int)((size_t)D.r_cap * (size_t)Lg * sizeof(double)), 0x00020000);
              oj_u4 q0[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int gi = oj_row(bI, bJ, r), k = 2 * tid;
                const unsigned int off = (gi < rank && k < Lg) ? (unsigned int)(((size_t)gi * Lg + k) * sizeof(double)) : 0xFFFFFFF0u;
                q0[r] = __builtin_amdgcn_raw_buffer_load_b128(rs2, off, 0, 16);
              }
              __syncthreads();
#pragma unroll
              for (int r = 0; r < 16; ++r) *reinterpret_cast<oj_d2*>(&s_X[r * ldx + 2 * tid]) = __builtin_bit_cast(oj_d2, q0[r]);
              __syncthreads();
            }
            const int c_lo = (two_half && pass == 0) ? 32 : 0, c_hi = (two_half && pass == 1) ? 32 : nch;
            update_chunks(c_lo, c_hi, two_half ? c_lo : 0);
          }
        }
      }
      OJ_T(3)
      ++nbar;
      // Every wave's row stores (write-through, sc1) must have COMPLETED before thread 0 counts the slot finished: the
      // workgroup barrier orders nothing in global memory and the compiler puts no vmcnt wait in front of it (the shipped
      // object had none -- tests/test_abi.py greps the ISA for this one), so each wave drains its own stores here.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        s_ticket = atomicAdd(&st->ticket, 1u);  // (the next trip's; read after the barrier at the top)
        const unsigned int d = atomicAdd(&st->bar, 1u) + 1u;
        if (d == (unsigned int)(sweep + 1) * per_sweep) {
          // the sweep's last slot: its verdict (k_oj_check), then the count of published verdicts the next sweep waits for
          const unsigned long long bits = __hip_atomic_load(&st->maxrel_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const double mr2 = __longlong_as_double((long long)bits);
          const int sw = st->sweeps + 1;
          st->sweeps = sw;
          const_cast<gpet_scalars*>(D.sc)->lml = (double)sw;
          __hip_atomic_store(&st->maxrel_bits, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&st->converged, mr2 <= tol2 ? 1 : 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&st->verdicts, sweep + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      OJ_T(4)
    }
  }
#ifdef GPET_OJ_PROF
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 37) && blockIdx.y == 0 && nbar > 0)
    printf("oj persist wg %d: per slot: stage %lld | gram %lld | inner sweep %lld | update + stores %lld | ticket + wait %lld cycles (%u slots)\n", (int)blockIdx.x,
           pt[0] / nbar, pt[1] / nbar, pt[2] / nbar, pt[3] / nbar, pt[4] / nbar, nbar);
#endif
#undef OJ_T
}

// after every sweep: converged when the largest relative coupling met during it was below the tolerance (the
// rotations of that sweep then took it to ~its square)
__global__ void __launch_bounds__(64) k_oj_check(EdgeDev* edges, double tol2) {
  const EdgeDev E = edges[blockIdx.x];
  if (eig_skip(E)) return;
  EigState* st = E.eig;
  if (threadIdx.x != 0 || st->converged) return;
  const double mr2 = __longlong_as_double((long long)st->maxrel_bits);
  st->sweeps += 1;
  if (mr2 <= tol2) st->converged = 1;  // default 1e-16: |g_p.g_q| <= 1e-8 |g_p||g_q| for every pair met in this sweep
  st->maxrel_bits = 0ull;
  E.sc->lml = (double)st->sweeps;  // diagnostics (gpet_scalars.lml: sweeps of the last factorisation)
}

// ---- 3. singular values, order, factor rows ------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_oj_norms(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  if (eig_skip(E)) return;
  const int rank = E.eig->rank, Lg = E.Lg;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= rank) return;
  const double* __restrict__ row = E.G + (size_t)k * Lg;
  double s = 0.0;
  for (int j = lane; j < Lg; j += WAVE) s += row[j] * row[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, WAVE);
  if (lane == 0) E.theta[k] = s;
}

__global__ void __launch_bounds__(256) k_oj_order(EdgeDev* edges) {  // (256 ranks per workgroup: one workgroup for all of them took 0.11 ms at 1 024 rows)
  const EdgeDev E = edges[blockIdx.y];
  if (eig_skip(E)) return;
  const int r = E.eig->rank;
  const int k = blockIdx.x * 256 + threadIdx.x;
  __shared__ double s_th[256];
  const double v = k < r ? E.theta[k] : 0.0;
  int pos = 0;
  for (int j0 = 0; j0 < r; j0 += 256) {
    __syncthreads();
    s_th[threadIdx.x] = (j0 + (int)threadIdx.x < r) ? E.theta[j0 + threadIdx.x] : 0.0;
    __syncthreads();
    const int lim = r - j0 < 256 ? r - j0 : 256;
    for (int jj = 0; jj < lim; ++jj) {
      const double u = s_th[jj];
      pos += (u > v) || (u == v && j0 + jj < k);
    }
  }
  if (k < r) E.order[pos] = k;
}

// A[k, :] = +-G[order[k], :], sign convention sum_j A[k, j] / (j + 1) >= 0 (LAPACK's signs are implementation-defined)
__global__ void __launch_bounds__(256) k_oj_rows(EdgeDev* edges) {
  const EdgeDev E = edges[blockIdx.y];
  if (eig_skip(E)) return;
  const int r = E.eig->rank, Lg = E.Lg, k = blockIdx.x;
  if (k >= r) return;
  __shared__ double s_red[4];
  const double* __restrict__ src = E.G + (size_t)E.order[k] * Lg;
  double part = 0.0;
  for (int j = threadIdx.x; j < Lg; j += 256) part += src[j] / (double)(j + 1);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, WAVE);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = part;
  __syncthreads();
  const double dot = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  const double sg = dot < 0.0 ? -1.0 : 1.0;
  double* __restrict__ dst = E.A + (size_t)k * Lg;
  // (and into this iteration's slot of the ring the next iteration's warm start reads -- rows of full rank only)
  // -- and only rows of a CONVERGED Jacobi: the warm start's identity X^T X = Sigma needs A^T S^-1 A = I, i.e. mutually
  //    orthogonal rows; a factorisation that ran out of sweeps (oj_max_sweeps) leaves rows that are not.
  const bool keep = r == Lg && Lg <= E.r_cap && E.eig->converged != 0;
  double* __restrict__ dst2 = E.Ap + ((size_t)(E.sc->iter & 1) * E.r_cap + k) * Lg;
  for (int j = threadIdx.x; j < Lg; j += 256) {
    const double v = sg * src[j];
    dst[j] = v;
    if (keep) dst2[j] = v;
  }
  if (k == 0 && threadIdx.x == 0) E.ap_tag[E.sc->iter & 1] = keep ? E.sc->iter + 1 : 0;
}

// Enqueues the whole factorisation.  Nothing is read back: a fixed budget of pivot steps and sweeps is launched and
// the kernels turn into no-ops once the device-side tests (tolerance reached / converged) have fired.
// the warm start's launches (after the pivoted Cholesky's init kernel, before its blocks): no-ops for an edge without
// usable previous rows
static void launch_oj_warm(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd) {
  const int warm = gpet_opt_oj_warm();
  if (!warm || bd.Lg > bd.r_cap) return;
  const int n = bd.Lg, nt = cdiv_h(n, 64);
  hipLaunchKernelGGL(k_ojw_begin, dim3(cdiv_h(n, 4), B), dim3(256), 0, st, d_edges, warm);
  hipLaunchKernelGGL(k_ojw_gemm<0>, dim3(nt, nt, B), dim3(256), 0, st, d_edges, warm);
  hipLaunchKernelGGL(k_ojw_gemm<1>, dim3(nt, nt, B), dim3(256), 0, st, d_edges, warm);
  for (int k0 = 0; k0 < n; k0 += 64) {
    hipLaunchKernelGGL(k_ojw_chol_diag, dim3(1, B), dim3(256), 0, st, d_edges, k0, (k0 == 0 && option("oj_warm_fail")) ? 1 : 0);
    const int below = cdiv_h(n - k0 - 64, 64);
    if (below > 0) {
      hipLaunchKernelGGL(k_ojw_chol_trsm, dim3(below, B), dim3(256), 0, st, d_edges, k0);
      hipLaunchKernelGGL(k_ojw_chol_syrk, dim3(below, below, B), dim3(256), 0, st, d_edges, k0);
    }
  }
  hipLaunchKernelGGL(k_ojw_gemm<2>, dim3(nt, nt, B), dim3(256), 0, st, d_edges, warm);
  hipLaunchKernelGGL(k_ojw_commit, dim3(256, B), dim3(256), 0, st, d_edges);
}

// The pivoted Cholesky of a WIDE edge of low rank (BASELINE config 3: 2 048 columns, rank <= 96) over the GPU instead of
// k_pchol's one workgroup (1.2 ms of that factor's 2.2: every pivot streams all previous rows through one CU).  The pivot
// ORDER matters here: these factors end at the rank CAP, and the blocked candidate selection of the any-rank path, which is
// free to take pivots out of order when every pivot will be taken anyway, leaves 1e-7 of the largest entry at rank 96 where
// the greedy order leaves 1e-12 (measured).  Option "pchol_multi": 2 = k_pcb_block with a block ending where the next pivot
// falls under a tenth of the block's first (the global maximum): the order stays within a constant of the greedy one;
// 1 = one pivot per launch in the greedy order itself (k_pcx_step); 0 = k_pchol.  The rows' last bits differ from k_pchol's
// either way, so the prior eigenbasis of the structured loop (whose bits every trace inherits) and edges of up to 1 024
// columns stay on k_pchol.
bool pchol_multi_applies(const BatchDims& bd) {
  const int mode = option("pchol_multi");
  return bd.Lg > 1024 && bd.r_cap <= 96 && mode != 0 && (mode == 1 || cdiv_h(bd.Lg, PCX_COLS) <= 64);
}
hipError_t launch_pchol_multi(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd) {
  (void)hipGetLastError();
  const int nw = cdiv_h(bd.Lg, PCX_COLS);
  const int steps = bd.r_cap < bd.Lg ? bd.r_cap : bd.Lg;
  PcxArgs px_args;
  memset(&px_args, 0, sizeof px_args);
  if (option("pchol_multi") == 1) {
    hipLaunchKernelGGL(k_pcx_init, dim3(nw, B), dim3(256), 0, st, d_edges, nw);
    for (int t = 0; t < steps; ++t) hipLaunchKernelGGL((k_pcx_step<false>), dim3(nw, B), dim3(256), 0, st, px_args, d_edges, t, nw);
    hipLaunchKernelGGL(k_pcx_fin, dim3(B), dim3(64), 0, st, d_edges, steps, nw);
  } else {
    // (every block takes at least its first candidate: `steps` blocks always suffice; one that finds the factorisation
    //  finished returns at once)
    hipLaunchKernelGGL(k_pcb_init, dim3(nw, B), dim3(64), 0, st, d_edges, nw);
    for (int blk = 0; blk < steps; ++blk)
      hipLaunchKernelGGL((k_pcb_block<false>), dim3(nw, B), dim3(256), 0, st, px_args, d_edges, blk, nw, 0.1);
    hipLaunchKernelGGL(k_pcb_fin, dim3(B), dim3(64), 0, st, d_edges, nw);
  }
  return hipGetLastError();
}

hipError_t launch_factor_big(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, const EdgeDev* h_edges) {
  (void)hipGetLastError();
  // small batches: the per-edge pointers travel in the kernel arguments (h_edges = host copy of the edge table)
  OjArgs oj_args;
  memset(&oj_args, 0, sizeof oj_args);
  const bool use_args = h_edges != nullptr && B <= OJ_ARGS_MAXB && option("oj_args");
  if (use_args)
    for (int e = 0; e < B; ++e) {
      OjEdge& d = oj_args.e[e];
      d.G = h_edges[e].G;
      d.st = h_edges[e].eig;
      d.sc = h_edges[e].sc;
      d.Lg = h_edges[e].Lg;
      d.r_cap = h_edges[e].r_cap;
      d.injected = h_edges[e].factor_injected;
    }
  const int nw = cdiv_h(bd.Lg, PCX_COLS);
  const int steps = bd.r_cap < bd.Lg ? bd.r_cap : bd.Lg;
  const int nblk = 2 * cdiv_h(steps, 2 * OJ_B);
  PcxArgs px_args;
  memset(&px_args, 0, sizeof px_args);
  if (use_args)
    for (int e = 0; e < B; ++e) {
      PcxEdge& d = px_args.e[e];
      const EdgeDev& h = h_edges[e];
      d.G = h.G;
      d.Gt = h.Gt;
      d.pcx_d = h.pcx_d;
      d.pcx_cand = h.pcx_cand;
      d.cov = h.cov;
      d.perm = h.perm;
      d.eig = h.eig;
      d.sc = h.sc;
      d.Lg = h.Lg;
      d.r_cap = h.r_cap;
      d.factor_injected = h.factor_injected;
    }
  if (nw <= 64 && !option("pcx_one_pivot")) {
    // blocks of up to PCB_NB pivots; rejected candidates cost extra blocks, so the budget is generous (a block that
    // finds the factorisation finished returns at once)
    hipLaunchKernelGGL(k_pcb_init, dim3(nw, B), dim3(64), 0, st, d_edges, nw);
    launch_oj_warm(st, d_edges, B, bd);
    const int per_block = nw < PCB_NB ? nw : PCB_NB;  // (one candidate per 32-column workgroup: narrow edges offer fewer)
    const int nblocks = 2 * cdiv_h(steps, per_block) + 8;
    for (int blk = 0; blk < nblocks; ++blk) {
      if (use_args) hipLaunchKernelGGL((k_pcb_block<true>), dim3(nw, B), dim3(256), 0, st, px_args, d_edges, blk, nw, 0.0);
      else hipLaunchKernelGGL((k_pcb_block<false>), dim3(nw, B), dim3(256), 0, st, px_args, d_edges, blk, nw, 0.0);
    }
    hipLaunchKernelGGL(k_pcb_fin, dim3(B), dim3(64), 0, st, d_edges, nw);
  } else {
    hipLaunchKernelGGL(k_pcx_init, dim3(nw, B), dim3(256), 0, st, d_edges, nw);
    for (int t = 0; t < steps; ++t) {
      if (use_args) hipLaunchKernelGGL((k_pcx_step<true>), dim3(nw, B), dim3(256), 0, st, px_args, d_edges, t, nw);
      else hipLaunchKernelGGL((k_pcx_step<false>), dim3(nw, B), dim3(256), 0, st, px_args, d_edges, t, nw);
    }
    hipLaunchKernelGGL(k_pcx_fin, dim3(B), dim3(64), 0, st, d_edges, steps, nw);
  }
  const int max_sweeps = gpet_opt_oj_max_sweeps();
  // LDS staging costs a workgroup a whole CU (131 KB): worth it while a round's workgroups (pairs x edges) fit the
  // chip's 256 CUs side by side (one edge: 64 pairs; measured 54 vs 61 ms per factor); a bigger batch runs two
  // register-fed workgroups per CU instead (8 edges: 85 vs 106 ms)
  const bool staged_lds_ok = bd.Lg <= OJ_STAGE_MAX && option("oj_stage");
  const bool staged = staged_lds_ok && (long long)(nblk / 2) * B <= 256;
  const size_t stage_lds = (size_t)OJ_M * (((bd.Lg + 31) & ~31) + 2) * sizeof(double);
  // k_oj_persist with the panel staged one 512-column half at a time (even widths above 512 columns): 66 KB instead of 131 KB,
  // two workgroups per CU -- a batch of eight 1 024-column edges (config 5's chains) holds all its 512 pair slots at once
  // (only where the whole panels cannot all be resident -- one workgroup per CU: a single 1 024-column edge, 64 slots, is 3-4 % slower
  //  with the second read of the first half: 11.6 against 11.2 ms warm; eight edges, 512 slots: 20.9 against 22.6 ms -- the rounds of
  //  a big batch are bound by what the 64 MB of panels pull from beyond the L2s, not by occupancy: profiles/r06_oj_half_stage.txt)
  int n_cus = 256;
  {
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess || n_cus < 1) n_cus = 256;
    (void)hipGetLastError();
  }
  const int half_opt = option("oj_half_stage");
  const int half_stage = (bd.lg_even && bd.Lg > 512 && (half_opt > 0 || (half_opt < 0 && (long long)(nblk / 2) * B > n_cus))) ? 1 : 0;
  const size_t persist_lds = half_stage ? (size_t)OJ_M * 514 * sizeof(double) : stage_lds;
  if (staged) {
    static int attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_done[dev]) {
      (void)hipFuncSetAttribute((const void*)k_oj_round<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
      (void)hipFuncSetAttribute((const void*)k_oj_round<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
      attr_done[dev] = 1;
    }
  }
  const double tol2 = pow(10.0, -2.0 * (double)gpet_opt_oj_tol_exp());
  // staged: the rounds and sweeps in ONE launch (k_oj_persist: pair slots by ticket -- no residency requirement, so the
  // grid is sized to what the device holds at once and a workgroup serves several slots per round when the batch has more
  // slots than that; beyond four slots per workgroup and round the round launches' two register-fed workgroups per CU
  // win); option "oj_persist" = 0: launches
  static int persist_cap[64] = {};  // workgroups of k_oj_persist the device holds at once (occupancy x CUs), per device ...
  static size_t persist_cap_lds[64] = {};  // ... for this much dynamic LDS (the occupancy is asked again when the size changes)
  static bool persist_attr[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (staged_lds_ok && gpet_opt_oj_persist() && dev >= 0 && dev < 64 && (!persist_cap[dev] || persist_cap_lds[dev] != persist_lds)) {
    if (!persist_attr[dev]) {
      (void)hipFuncSetAttribute((const void*)k_oj_persist<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
      (void)hipFuncSetAttribute((const void*)k_oj_persist<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
      persist_attr[dev] = true;
    }
    int occ = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_oj_persist<false>, 256, persist_lds) != hipSuccess || occ < 1) occ = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 64;
    persist_cap[dev] = occ * cus;
    persist_cap_lds[dev] = persist_lds;
    (void)hipGetLastError();
  }
  const int cap = (dev >= 0 && dev < 64 && persist_cap[dev] > 0) ? persist_cap[dev] : 64;
  const long long slots_all = (long long)(nblk / 2) * B;
  if (staged_lds_ok && gpet_opt_oj_persist() && slots_all <= 4LL * cap) {
    int wpe = nblk / 2;  // workgroups per edge
    if (slots_all > cap) wpe = cap / B > 0 ? cap / B : 1;
    if (use_args) hipLaunchKernelGGL((k_oj_persist<true>), dim3(wpe, B), dim3(256), persist_lds, st, oj_args, d_edges, nblk, max_sweeps, tol2, half_stage);
    else hipLaunchKernelGGL((k_oj_persist<false>), dim3(wpe, B), dim3(256), persist_lds, st, oj_args, d_edges, nblk, max_sweeps, tol2, half_stage);
  } else
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    for (int round = 0; round < nblk - 1; ++round) {
      if (staged && use_args)
        hipLaunchKernelGGL((k_oj_round<true, true>), dim3(nblk / 2, B), dim3(256), stage_lds, st, oj_args, d_edges, round, nblk, tol2);
      else if (staged)
        hipLaunchKernelGGL((k_oj_round<true, false>), dim3(nblk / 2, B), dim3(256), stage_lds, st, oj_args, d_edges, round, nblk, tol2);
      else
        hipLaunchKernelGGL((k_oj_round<false, false>), dim3(nblk / 2, B), dim3(256), 0, st, oj_args, d_edges, round, nblk, tol2);
    }
    hipLaunchKernelGGL(k_oj_check, dim3(B), dim3(64), 0, st, d_edges, tol2);
  }
  hipLaunchKernelGGL(k_oj_norms, dim3(cdiv_h(steps, 4), B), dim3(256), 0, st, d_edges);
  hipLaunchKernelGGL(k_oj_order, dim3(cdiv_h(steps, 256), B), dim3(256), 0, st, d_edges);
  hipLaunchKernelGGL(k_oj_rows, dim3(steps, B), dim3(256), 0, st, d_edges);
  return hipGetLastError();
}

int& gpet_opt_oj_tol_exp() {
  static const int i_ = option_index("oj_tol_exp");
  int& v = option_at(i_);
  return v;
}

int& gpet_opt_oj_persist() {
  static const int i_ = option_index("oj_persist");
  int& v = option_at(i_);
  return v;
}

int& gpet_opt_oj_max_sweeps() {
  static const int i_ = option_index("oj_max_sweeps");
  int& v = option_at(i_);
  return v;
}

}  // namespace gpet
