// Device-side view of one edge of a batch (internal; not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpet_hip.h"

namespace gpet {

// Device-side state of the any-rank factorisation (gpet_eig.hip): pivoted Cholesky + one-sided block Jacobi.
struct EigState {
  double tol;                      // 1e-14 * first pivot: where the pivoted Cholesky stops
  unsigned long long maxrel_bits;  // bits of the largest squared relative row coupling met in the current sweep
  int stopped, rank;               // pivoted Cholesky finished; rows of G
  int converged, sweeps;           // Jacobi: every pair orthogonal to the tolerance; sweeps done
  int cand_half;                   // blocked pivoting: which half of pcx_cand holds the latest candidates
  unsigned int bar;                // k_oj_persist: pair slots FINISHED so far (all rounds and sweeps of this factorisation), monotonic
  unsigned int ticket;             // k_oj_persist: pair slots HANDED OUT so far (a workgroup takes the next one when it has finished its last)
  int verdicts;                    // k_oj_persist: sweeps whose convergence verdict has been published
  int warm;                        // warm start (k_ojw_*): 0 = cold (pivoted Cholesky), 1 = the rows come from the previous iteration's, 2 = tried and failed
  int t_slot[2], stop_slot[2];     // blocked pivoting: rows so far / finished, as block (blk & 1) must see them -- a block
                                   // writes the OTHER slot, so workgroups of one launch never read what it writes
};

// One entry per edge, stored in a device array; kernels index it with blockIdx.y.
struct EdgeDev {
  // geometry / clamped ctor parameters (gpet.py:95-158)
  int M, N, x_st, x_en, Lg, S, n_keep, z_cols, r_cap, n_cap, obs_cap, n_init;
  int kernel_type, nu_code;  // nu_code: 0 -> 0.5, 1 -> 1.5, 2 -> 2.5, 3 -> any other nu > 0 (nu_gen; Bessel form by quadrature)
  double nu_gen, inv_gamma_nu;  // general Matern smoothness and 1 / Gamma(nu)
  double* rho_tab;              // [N] general nu: correlation at the integer lags of the pixel grid for this edge's length scale
  int tab_ok, pad_tab;          // 1: coordinates are pixels and the length scale is the constructor's: rho_tab applies
  int fix_endpoints, delta_x, pixel_thresh, algo_thresh, n_bins, a_rows_cap;
  int factor_injected, z_ring;  // z_ring: slots of pre-generated normals (one per upcoming iteration)
  double sigma_f, length_scale, noise_y, jitter;
  // inputs
  const float* grad;      // [M*N] normalised gradient image (values are f32-exact, gpet.py:97)
  const float* grad_kde;  // [M*N] normalised gradient KDE (gpet.py:127)
  const long long* init_xy;  // [n_init*2] sorted by x
  long long* obs_xy;         // [obs_cap*2]
  long long* obs_new;        // [obs_cap*2] scratch for the next observation set
  gpet_scalars* sc;
  // GP workspaces
  double *xt, *yt, *wt;  // [n_cap]
  double* K;             // [n_cap*n_cap] row-major; lower triangle becomes L
  double* alpha;         // [n_cap]
  double* chol_inv;      // [n_cap / 64 + 1][64][64] inverses of L's diagonal blocks (blocked fit, n_cap > 128): the
                         // triangular solves against a panel become matrix products on the matrix cores
  double* solve_z;       // [n_cap] z = L^-1 y of the multi-workgroup triangular solves (blocked fit, n_cap > 128)
  int* solve_flag;       // [2][n_cap / 64 + 1] launch number at which block i of z / of alpha was published (k_chol_solve_mw)
  double* V;             // [n_cap*Lg]  L^-1 K_*^T
  double *mean, *std;    // [Lg]
  double* cov;           // [Lg*Lg]
  double* G;             // [r_cap*Lg] pivoted-Cholesky columns (row t = column t)
  int* perm;             // [r_cap]
  double *C, *W, *theta; // [r_cap*r_cap], [r_cap*r_cap], [r_cap]
  // warm start of the structured path's eigen-decomposition (k_jacobi_prerot): the eigenvectors of iteration k are kept in
  // Wq[k & 1] with the tag (k + 1) + 65536 rank in wq_tag[k & 1] (0: nothing there); Cw = Wq^T C Wq of the previous ones
  double *Wq, *Cw;       // [2 slots][r_cap*r_cap], [r_cap*r_cap]
  int* wq_tag;           // [2]
  int* order;            // [r_cap] eigenvalue order (descending)
  // prior eigenbasis of the unit-amplitude Toeplitz correlation matrix of the grid (structured loop path)
  double* Q0;            // [r_cap*Lg] orthonormal rows q_a^T
  double* lam0;          // [r_cap] eigenvalues of rho
  double* beta;          // [r_cap] c * lam0 * Q0[:, obs] alpha
  double* h0;            // [r_cap] structured path: h0[t] = sum_j Q0[t][j] / (j + 1), the sign convention's weights in the prior eigenbasis
  int r0, structured;    // rank of rho at 1e-14; 1 when the structured path is usable for this edge
  double* jlog;          // [JS_LOG_SWEEPS][m - 1][m / 2][2] rotations (c, s) of the LDS Jacobi, round by round (small batches only, else 1 entry)
  int jlog_cap;          // sweeps the log holds (0: none)
  EigState* eig;         // state of the any-rank factorisation
  double* Gt;            // [Lg][r_cap] transposed copy of G kept by the multi-workgroup pivoted Cholesky (ranks > 96 only)
  double* Ap;            // [2][r_cap*Lg] (ranks > 96 only) the factor rows of the last two iterations: slot k & 1 = iteration k's, the warm start of k + 1
  int* ap_tag;           // [3] [0..1]: iteration + 1 of the rows in the slot when they are of full rank (0: nothing usable);
                         // [2]: 1 + the slot that holds the LAST trace's final rows (set by gpet_batch_set_obs: the first factor of the next trace starts from them)
  double* pcx_d;         // [Lg] remaining diagonal of the multi-workgroup pivoted Cholesky (-1: pivoted)
  double* pcx_cand;      // [2][4 (Lg/32 + 1)][2] per-wave pivot candidates (value, index) of the current / next step
  double* A;             // [a_rows_cap*Lg] factor rows sqrt(s_k) v_k
  double* Z;             // [z_ring][S*z_cols]; slot of iteration k = k % z_ring
  double* Y;             // [S*Yp] samples (+ rows up to the next multiple of 128 and one spare row, never read), row = sample; f32 in the same buffer when y_f32 is set
  int Yp;                // row pitch of Y in elements: Lg rounded up to 16 (128-byte rows of f64: every 128-byte run of the
                         // GEMM's stores is one cache line -- at a pitch of Lg = 500 every other run straddled three)
  int y_f32;             // gpet_batch_set_sample_dtype: 1 = the GEMM stores f32, consumers widen (opt-in; default f64)
  double* costs;         // [S]
  double* cost_part;     // [n_tiles][S][2] per-column-tile partial (arc length, line integral) sums
  double* best_costs;    // [n_keep]
  int* best_idx;         // [n_keep]
  // pixel-selection workspaces (f1)
  double* bins;          // [(N+2)*(M+2)] linear-binning grid, x-major
  double* tmpk;          // [(N+2)*(M+2)] separable-convolution scratch
  float* kde;            // [M*N] normalised curve KDE (stage API); inside gpet_trace_iterate: raw density, band rows only
  int* kde_band;         // [N/16 + 1][2] first / last image row with density, per 16-column tile of the fused KDE
  double* kde_wsum;      // [1] total weight of the gradient KDE's points (sum of colsum in index order; k_kde_wsum)
  double* colsum;        // [N] kept weight per column
  double* colbest;       // [N] best new-pixel score per column
  int* colbest_y;        // [N] row of that pixel
  unsigned int* mm;      // [4] ordered-uint min/max of the raw KDE
  unsigned long long* binbest;  // [n_bins] bits of the best score per bin (scores are >= 0)
  long long* binarg;     // [n_bins] order key of the best candidate per bin
  int bin_lo, fin_n;     // fin_n: training points of the converged fit
  double *fin_x, *fin_y, *fin_w;  // [n_cap] standardised training set of the converged fit (gpet.py:235-238)
  // optimum + transforms of the converged fit: c, l, noise, X_m, X_s, y_m, y_s, m2, s2 (values, not logs);
  // [9], [10]: lattice of fin_x (x_i = x_0 + m_i / hinv, m_i integer): hinv (0: none known) and the largest |m_i - m_j| --
  // the matrix-core objective k_lml16 tabulates the correlation at the lags 0..lagmax
  double* fin_par;       // [12]
  double* fin_out;       // [2 * Lg_max] mean (pixels) then std, in the batch's contiguous output block      // bin index of the first slot (np.round((x - x_st)/delta_x) can be < 0)
};

}  // namespace gpet
