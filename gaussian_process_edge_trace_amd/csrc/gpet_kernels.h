// Kernel launchers (internal).
#pragma once
#include "gpet_dev.h"

namespace gpet {

struct BatchDims {
  int M, N, Lg, S, n_keep, z_cols, r_cap, n_cap, n_bins, obs_cap, z_ring, a_rows_cap, r0_max;
};

hipError_t launch_conv(hipStream_t st, const double* d_img, int M, int N, const double* d_wf, int kh, int kw, int oy,
                       int ox, float* d_tmp, unsigned int* d_minmax);
hipError_t launch_minmax(hipStream_t st, const float* d_in, size_t count, unsigned int* d_minmax);
hipError_t launch_normalise(hipStream_t st, const float* d_in, size_t count, const unsigned int* d_minmax,
                            float* d_out);
hipError_t launch_fit_predict(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, int want_cov,
                              unsigned parts = ~0u);
hipError_t launch_final_predict(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd);
hipError_t launch_final_cov(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd);
hipError_t launch_factor(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, unsigned parts = ~0u,
                         const EdgeDev* h_edges = nullptr);
// dynamic LDS available to k_struct_H (the structured path needs at least U + one row of L + beta in it)
#define STRUCT_H_LDS_MAX (150 * 1024)
hipError_t launch_struct_iteration(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, unsigned parts = ~0u);
hipError_t launch_struct_basis(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd);
hipError_t launch_normals(hipStream_t st, EdgeDev* d_edges, int B, const unsigned int* d_seeds, int add_iter,
                          int iter_abs, int n_ahead, int z_store = 0);
hipError_t launch_kde(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, int mode, unsigned parts = ~0u,
                      int raw_band = 0);
hipError_t launch_pixels(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, int raw_band = 0, unsigned parts = ~0u);
// any-rank factor (gpet_eig.hip): pivoted Cholesky over the whole GPU + one-sided block Jacobi on its rows
hipError_t launch_factor_big(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, const EdgeDev* h_edges = nullptr);
// gpet_set_option "oj_max_sweeps" (default 16; environment GPET_OJ_MAX_SWEEPS): sweep budget of that Jacobi
int& gpet_opt_oj_max_sweeps();
// gpet_set_option "oj_tol_exp" (default 8; environment GPET_OJ_TOL_EXP): the Jacobi stops after a sweep in which every
// pair of rows it met was orthogonal to 10^-x relative (that sweep's rotations then take the couplings to ~their square)
int& gpet_opt_oj_tol_exp();
// process-wide switch (gpet_set_option "scalar_jacobi"; initial value from the environment GPET_SCALAR_JACOBI):
// 1 = factor covariances of rank > 96 with the round-1 whole-GPU scalar Jacobi instead of gpet_eig.hip
int& gpet_opt_scalar_jacobi();
int& gpet_opt_jacobi_variant();
// gpet_set_option "rng_lookahead" (default 1; environment GPET_RNG_LOOKAHEAD): how many iterations the RNG stream of
// the device loop may run ahead of it (gpet_api.hip, gpet_trace_iterate)
int& gpet_opt_rng_lookahead();
// gpet_set_option "lml_two_tiles_from" (default 600; environment GPET_LML_TWO_TILES_FROM): launches of the converged fits' objective with at least this many problems use
// the two-tiles-per-thread kernel also below 129 training points (fewer instructions per problem, longer latency)
int& gpet_opt_lml_two_tiles_from();
hipError_t launch_set_force(hipStream_t st, EdgeDev* d_edges, int B, int v);
hipError_t launch_rho_tab(hipStream_t st, EdgeDev* d_edges, int B, int N);
hipError_t launch_fin_scatter(hipStream_t st, EdgeDev* d_edges, int B, const double* d_stage, const int* d_n, int stride);
hipError_t launch_pixels_reset(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd);
hipError_t launch_sample(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, int rank_max = 0);
hipError_t launch_score(hipStream_t st, EdgeDev* d_edges, int B, const BatchDims& bd, unsigned parts = ~0u);

// converged fit on the device (gpet_lbfgsb.hip): training sets + start points, L-BFGS-B state machines, best restart
size_t lb_prob_bytes();
hipError_t launch_fin_prepare(hipStream_t st, EdgeDev* d_edges, int B, const unsigned int* d_seeds, double* d_starts,
                              double* d_scratch, int scratch_stride, int n_cap);
hipError_t launch_lb_init(hipStream_t st, void* d_probs, int P, const double* d_starts, int* slot_edge, double* slot_theta,
                          int* slot_src);
hipError_t launch_lb_advance(hipStream_t st, void* d_probs, int n_upper, const int* cur_count, const int* slot_src,
                             const double* d_f, const double* d_g, int* next_count, int* next_edge, double* next_theta,
                             int* next_src);
hipError_t launch_lb_pick(hipStream_t st, EdgeDev* d_edges, int B, const void* d_probs, double* d_theta_out);

// objective for more than 250 training points: blocked HBM kernels over a scratch table of virtual edges
size_t lmlbig_scratch_doubles(int ncap_v);
hipError_t launch_lml_big(hipStream_t st, EdgeDev* d_edges, int P, int n_max, const int* d_edge_of, const double* d_theta,
                          double* d_f, double* d_g, void* d_vedges, void* d_vsc, double* d_scratch, double* d_part,
                          int ncap_v);
// d_count (may be null): device word with the true number of problems (workgroups beyond it return at once);
// lag_cap > 0: every training set of the launch sits on a lattice (fin_par[9..10]) with fewer than lag_cap points -- the
// matrix-core kernel k_lml16 and its correlation tables apply (n_max <= 108); 0: the vector kernels
hipError_t launch_lml(hipStream_t st, EdgeDev* d_edges, int P, int n_max, const int* d_edge_of, const double* d_theta,
                      double* d_f, double* d_g, const int* d_count, int lag_cap);
// gpet_set_option "lml_mfma" (default 1; environment GPET_LML_MFMA): 0 = objective of the converged fits on the
// round-2 vector kernels (k_lml / k_lml2) also below 109 training points
int& gpet_opt_lml_mfma();

}  // namespace gpet
