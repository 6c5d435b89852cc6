// Internals shared by the translation units of the C ABI (include/gpet_hip.h): contexts and batches as the library sees them,
// the error / wait helpers, and the few host-side helpers more than one unit needs.  Host-side plumbing only: all arithmetic
// lives in the kernels (gpet_kernels.hip, gpet_eig.hip, gpet_lbfgsb.hip, gpet_rng.hip).
//   gpet_api_ctx.hip     contexts, options, timers, a1 (gradient image), the shared helpers' definitions
//   gpet_api_batch.hip   batches: creation (arena layout), destruction, images, observations, reset, reads / writes
//   gpet_api_stages.hip  the per-stage entry points (a2-a7, f1) and gpet_profile_stage
//   gpet_api_final.hip   the converged fit (f2): objective, device L-BFGS-B, posterior at the optimum
//   gpet_api_loop.hip    the device-resident loop (a8): gpet_trace_iterate
//   gpet_api_comm.hip    multi-GPU helpers on RCCL (8e): communicator, broadcast of the gradient image, gather of the traces
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>
#include <algorithm>
#include <vector>

#include "gpet_kernels.h"
#include "gpet_options.h"

using namespace gpet;

struct gpet_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  char* scratch = nullptr;  // device scratch of the a1 entry points (gpet_grad_image / gpet_normalise_f32), grown on demand
  size_t scratch_bytes = 0;
  std::string err;
};

struct gpet_batch {
  gpet_ctx* ctx = nullptr;
  int B = 0;
  BatchDims bd{};
  std::vector<EdgeDev> h_edges;
  std::vector<gpet_params> params;
  EdgeDev* d_edges = nullptr;
  EdgeDev* d_edges_act = nullptr;      // the edges still running, compacted (gpet_trace_iterate)
  unsigned int* d_seeds_act = nullptr;
  std::vector<EdgeDev> h_edges_act;
  std::vector<unsigned int> h_seeds_act;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  unsigned int* d_seeds = nullptr;
  gpet_scalars* d_scalars = nullptr;   // [B] contiguous: one copy reads every edge's state
  double* d_fin_out = nullptr;         // [B][2][Lg_max] contiguous results of the converged fits
  double* d_fin_par = nullptr;         // [B][12] contiguous hyper-parameters / transforms of the converged fits
  long long* d_obs = nullptr;          // [B][obs_cap_max][2] contiguous observations: one copy reads them all
  long long* d_init = nullptr;         // [B][n_init_max][2] contiguous init points: one copy writes them all
  std::vector<gpet_scalars> h_scalars;
  std::vector<int> h_nobs_prev;        // observations per edge at the last group boundary of the loop (adaptive group sizes)
  int iters_issued = 0;                // iterations enqueued since the last reset (== sc->iter of active edges)
  int rng_mode = 0;                    // 0: MT19937 + polar method = numpy's RandomState stream; 1: Philox4x32-10 + Box-Muller (opt-in)
  hipStream_t side = nullptr;          // RNG stream: normals of upcoming iterations run ahead of the loop
  double* d_fin_stage = nullptr;       // staging of the converged fits' training sets (x | y | w blocks)
  int* d_fin_n = nullptr;
  size_t fin_stage_cap = 0;
  hipStream_t fit = nullptr;           // high-priority stream of the final-fit objective launches: they are tiny and
                                       // latency-bound, and run while OTHER batches' loops keep the GPU busy
  hipEvent_t ev_norm[16] = {};
  hipEvent_t ev_gemm[16] = {};         // sample GEMM of iteration k done: ring slot k % ring may be refilled
  hipEvent_t ev_pix[16] = {};          // pixel selection of iteration k done: the `done` flags of iteration k + 1 are final
  int norm_issued = 0;                 // iterations whose normals have been enqueued on `side`
  hipEvent_t ev_main = nullptr;
  unsigned int* d_minmax = nullptr;    // [2 B]: (min, max) of every gradient image being uploaded
  std::vector<unsigned int> h_mm0;     // their reset values (kept alive for the asynchronous copy)
  float* d_raw = nullptr;  // [M*N] staging of a user gradient image before its re-normalisation (gpet.py:97)
  int share_image = 0;
  bool structured = false;  // every edge can take the prior-eigenbasis loop path
  // converged-fit scratch (grown on demand)
  int lml_cap = 0;
  hipEvent_t ev_l0 = nullptr, ev_l1 = nullptr;  // around every LML kernel launch (gpet_lml_stats)
  double lml_ms = 0.0;
  long long lml_evals = 0;
  int lml_launches = 0;
  int* d_edge_of = nullptr;
  double *d_theta = nullptr, *d_f = nullptr, *d_g = nullptr;
  // device-resident converged fits (gpet_final_fit_all): one allocation, carved
  char* lb_mem = nullptr;
  int lb_cap_P = 0;  // problems the optimiser's workspace holds
  void* lb_probs = nullptr;
  double *lb_starts = nullptr, *lb_scratch = nullptr, *lb_f = nullptr, *lb_g = nullptr, *lb_theta_out = nullptr;
  int* lb_slot_edge[2] = {nullptr, nullptr};
  double* lb_slot_theta[2] = {nullptr, nullptr};
  int* lb_slot_src[2] = {nullptr, nullptr};
  int* lb_count = nullptr;
  unsigned int* lb_seeds = nullptr;
  int lb_scratch_stride = 0;
  std::vector<hipEvent_t> lb_events;
  // objective for more than 250 training points (launch_lml_big): virtual-edge table + per-problem scratch
  char* big_mem = nullptr;
  int big_chunk = 0, big_ncap = 0;
  void *big_vedges = nullptr, *big_vsc = nullptr;
  double *big_scratch = nullptr, *big_part = nullptr;  // pairs around every objective launch of a converged fit (gpet_lml_stats)
  // chunked normal generator (one long MT19937 stream on many workgroups): workspace + the jump tables on the device
  void* mtj_work = nullptr;
  size_t mtj_bytes = 0;
  unsigned int* d_mtj_poly = nullptr;
  // largest lattice lag of every edge's converged-fit training set as the HOST knows it (fin_par[9..10] on the device):
  // -1 = no lattice (caller-supplied x off any grid) -> the vector objective kernels; see fin_lattice()
  std::vector<int> fin_lag;
  bool have_fit = false, have_factor = false, have_normals = false, have_samples = false, have_scores = false;
  OptionSet opts;  // the batch's own copy of the option table (gpet_options.h): taken at creation, gpet_batch_set_option changes it
};
// first statement of every entry point that works on a batch: its option table for the calling thread
#define GPET_BATCH_SCOPE(b) OptionScope gpet_opt_scope_((b) ? &(b)->opts : nullptr)

// ---- helpers defined in gpet_api_ctx.hip ----------------------------------------------------------------------------------
hipError_t gpet_wait(hipStream_t st);  // host wait on a stream: spinning or sleeping (option blocking_sync)
int fail(gpet_ctx* ctx, int code, const char* fmt, ...);
#define HIPCHK(ctx, call)                                                                          \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail((ctx), GPET_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
struct Carver {  // lays buffers out in one arena (256-byte aligned); with base == nullptr it only measures
  size_t off = 0;
  char* base = nullptr;
  template <typename T>
  T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};
int fin_lattice(const double* x, int n, double* hinv);
int& opt_fit_persistent();
hipError_t launch_normals_seq(gpet_batch* b, hipStream_t st, EdgeDev* edges_l, int B_l, const unsigned int* seeds_l, int add_iter,
                              int iter_abs, int n_ahead, int z_store);
int normals_auto(gpet_batch* b, hipStream_t st, EdgeDev* edges_l, int B_l, const unsigned int* seeds_l, int add_iter, int iter_abs,
                 int n_ahead, int z_store, bool allow_chunked = true);
// ---- gpet_api_batch.hip ---------------------------------------------------------------------------------------------------
int fetch_all_scalars(gpet_batch* b);
int check_device_status(gpet_batch* b);
// what the loop's generator stores of a sample row: the r0 (rounded to 4) leading normals a structured batch multiplies
static inline int loop_z_store(const gpet_batch* b) {
  if (!b->structured || b->bd.r0_max < 1 || option("z_store_full")) return 0;
  return (b->bd.r0_max + 3) & ~3;
}
