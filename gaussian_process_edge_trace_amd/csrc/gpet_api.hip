// C ABI of libgpet_hip.so (see include/gpet_hip.h).  Host-side plumbing only: contexts,
// batches, workspace carving, launches, copies.  All arithmetic lives in gpet_kernels.hip.
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
  std::vector<gpet_scalars> h_scalars;
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
  unsigned int* d_minmax = nullptr;
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
};

// Host waits.  hipStreamSynchronize spins on a CPU core; with one process per GPU and a few driver threads per process
// (device loop + converged fits in flight) eight ranks would keep 32 threads spinning on a node's cores.  In blocking
// mode (gpet_set_option("blocking_sync", 1); default: on when WORLD_SIZE > 1, i.e. under torch.distributed.run) a wait
// is an event created with hipEventBlockingSync: the thread sleeps until the GPU signals.
static int opt_blocking_sync() {
  static int& v = option("blocking_sync");
  if (v >= 0) return v;
  static const int by_world = [] {  // (WORLD_SIZE is torch.distributed's variable, not a switch of this library)
    const char* w = getenv("WORLD_SIZE");
    return (w && atoi(w) > 1) ? 1 : 0;
  }();
  return by_world;
}
static hipError_t gpet_wait(hipStream_t st) {
  if (!opt_blocking_sync()) return hipStreamSynchronize(st);
  static thread_local hipEvent_t ev = nullptr;  // (per host thread: waits from different driver threads do not share it)
  static thread_local int ev_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!ev || ev_dev != dev) {
    if (ev) (void)hipEventDestroy(ev);
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming);
    if (e != hipSuccess) {
      ev = nullptr;
      return hipStreamSynchronize(st);
    }
    ev_dev = dev;
  }
  hipError_t e = hipEventRecord(ev, st);
  if (e != hipSuccess) return e;
  return hipEventSynchronize(ev);
}

static int fail(gpet_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

#define HIPCHK(ctx, call)                                                                          \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail((ctx), GPET_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

namespace {
struct Carver {
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

int nu_to_code(double nu) {
  if (nu == 0.5) return 0;
  if (nu == 1.5) return 1;
  if (nu == 2.5) return 2;
  if (nu > 0.0 && nu <= 1e6) return 3;  // any other smoothness: Bessel form by quadrature (gpet.py:134)
  return -1;
}

// lays out one edge's buffers; with base == nullptr only measures
void carve_edge(Carver& cv, EdgeDev& E, bool own_image) {
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
  E.Gt = cv.take<double>(rc > 96 ? Lg * rc : 1);
  E.Ap = cv.take<double>(rc > 96 ? 2 * Lg * rc : 1);
  E.ap_tag = cv.take<int>(3);
  E.pcx_d = cv.take<double>(Lg);
  E.pcx_cand = cv.take<double>(16 * ((size_t)E.N / 32 + 2));  // (indexed with the widest edge of the batch)
  E.jb_cs = cv.take<double>(2 * (rc / 2 + 1) + 2 * 64);  // (+ 64 partial norm pairs of k_jb_norms)
  E.jb_norm = cv.take<double>(2);
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

// Lattice of a caller-supplied training set: h with x_i = x_min + m_i h (the smallest positive gap, refined over the
// whole span), accepted when every point sits on it to 1e-6 of a step.  hinv = 1 / h; returns the largest lag or -1.
static int fin_lattice(const double* x, int n, double* hinv) {
  *hinv = 0.0;
  if (n < 2) return -1;
  double lo = x[0], hi = x[0];
  for (int i = 1; i < n; ++i) {
    lo = x[i] < lo ? x[i] : lo;
    hi = x[i] > hi ? x[i] : hi;
  }
  std::vector<double> srt(x, x + n);
  std::sort(srt.begin(), srt.end());
  double gap = INFINITY;
  for (int i = 1; i < n; ++i) {
    const double d = srt[i] - srt[i - 1];
    if (d > 0.0 && d < gap) gap = d;
  }
  if (!(gap < INFINITY) || !(hi > lo)) return -1;
  const double span = hi - lo, mr = rint(span / gap);
  if (!(mr >= 1.0 && mr < 1048576.0) || fabs(span / gap - mr) > 1e-6) return -1;
  const double hi_ = mr / span;
  for (int i = 0; i < n; ++i) {
    const double t = (x[i] - x[0]) * hi_;
    if (fabs(t - rint(t)) > 1e-6) return -1;
  }
  *hinv = hi_;
  return (int)mr;
}

// converged fits: every (edge, restart) problem from start to optimum in one workgroup (k_lml16_fit) instead of
// lock-step rounds over all running problems: -1 = for problem sets resident at once (<= 1024), 0 = never, 1 = always
// (where the training sets allow it)
static int& opt_fit_persistent() {
  static int& v = option("fit_persistent");
  return v;
}

static int& opt_rng_chunked() {
  static int& v = option("rng_chunked");  // -1: by launch shape
  return v;
}

// One sequential walk per stream: the register-resident generator (four streams per wave, gpet_rng.hip) when the batch is
// homogeneous and the launch has enough streams to fill the GPU with single waves (2 048 = half of its SIMDs; a wave of
// four streams takes ~2.5 ms against 0.6 ms for a three-wave workgroup per stream, so small launches keep the old kernel),
// else one workgroup per stream (k_mt_normals).  The same numbers either way.
static hipError_t launch_normals_seq(gpet_batch* b, hipStream_t st, EdgeDev* edges_l, int B_l, const unsigned int* seeds_l,
                                     int add_iter, int iter_abs, int n_ahead, int z_store) {
  const int opt = gpet_opt_rng4();
  if (b->bd.rng4 && (opt > 0 || (opt < 0 && (long long)B_l * n_ahead >= 2048)))
    return launch_normals4(st, edges_l, B_l, seeds_l, add_iter, iter_abs, n_ahead, z_store, b->bd.Lg, b->bd.S, b->bd.z_cols);
  return launch_normals(st, edges_l, B_l, seeds_l, add_iter, iter_abs, n_ahead, z_store);
}

// The normals of `n_ahead` iterations of B_l edges: one workgroup per stream (k_mt_normals), or -- when that leaves
// most of the GPU idle and the streams are long -- every stream cut into chunks that many workgroups generate at once
// (MT19937 jump-ahead, launch_normals_chunked).  The same numbers either way.
static int normals_auto(gpet_batch* b, hipStream_t st, EdgeDev* edges_l, int B_l, const unsigned int* seeds_l, int add_iter,
                        int iter_abs, int n_ahead, int z_store) {
  gpet_ctx* c = b->ctx;
  if (b->rng_mode == 1) {  // opt-in Philox mode (gpet_batch_set_rng)
    HIPCHK(c, launch_normals_philox(st, edges_l, B_l, b->bd, seeds_l, add_iter, iter_abs, n_ahead, z_store));
    return GPET_OK;
  }
  const int streams = B_l * n_ahead;
  const int nc = mtj_chunks((long long)b->bd.S * b->bd.Lg);
  const int opt = opt_rng_chunked();
  const bool force4 = gpet_opt_rng4() > 0 && b->bd.rng4;  // (tests: the register-resident generator on any launch shape)
  const bool chunked = !force4 && nc >= 2 && (opt > 0 || (opt < 0 && streams <= 32 && nc >= 4));
  if (!chunked) {
    HIPCHK(c, launch_normals_seq(b, st, edges_l, B_l, seeds_l, add_iter, iter_abs, n_ahead, z_store));
    return GPET_OK;
  }
  const size_t need = mtj_work_bytes(streams, nc);
  if (need > b->mtj_bytes) {
    if (b->mtj_work) {
      HIPCHK(c, hipDeviceSynchronize());  // (launches that use the old workspace may still be in flight)
      (void)hipFree(b->mtj_work);
      b->mtj_work = nullptr;
      b->mtj_bytes = 0;
    }
    HIPCHK(c, hipMalloc(&b->mtj_work, need));
    b->mtj_bytes = need;
  }
  if (!b->d_mtj_poly) {
    HIPCHK(c, hipMalloc(&b->d_mtj_poly, mtj_poly_bytes()));
    HIPCHK(c, hipMemcpy(b->d_mtj_poly, mtj_poly_host(), mtj_poly_bytes(), hipMemcpyHostToDevice));
  }
  HIPCHK(c, launch_normals_chunked(st, edges_l, B_l, seeds_l, add_iter, iter_abs, n_ahead, z_store, b->mtj_work, nc, b->d_mtj_poly));
  return GPET_OK;
}

static int fetch_all_scalars(gpet_batch* b);
static int eval_objective(gpet_batch* b, hipStream_t st, int P, int n_max, const int* d_edge_of, const double* d_theta,
                          double* d_f, double* d_g, const int* d_count = nullptr, int lag_cap = 0);

extern "C" {

int gpet_abi_version(void) { return GPET_ABI_VERSION; }

int gpet_set_option(const char* name, int value) {
  int prev = 0, idx = -1;
  for (int i = 0; i < option_count(); ++i)
    if (name && strcmp(option_def(i).name, name) == 0) idx = i;
  if (idx < 0 || option_set(name, value, &prev) != 0) return -1;
  // ("chosen automatically", -1, is reported as the option's largest value + 1: a negative return means "unknown name")
  return prev < 0 ? option_def(idx).hi + 1 : prev;
}

int gpet_get_option(const char* name, int* value) { return option_get(name, value) == 0 ? GPET_OK : GPET_ERR_BAD_ARG; }

int gpet_option_count(void) { return option_count(); }

int gpet_option_info(int index, const char** name, int* value, int* def, int* lo, int* hi, const char** doc) {
  if (index < 0 || index >= option_count()) return GPET_ERR_BAD_ARG;
  const OptionDef& d = option_def(index);
  if (name) *name = d.name;
  if (value) (void)option_get(d.name, value);
  if (def) *def = d.def;
  if (lo) *lo = d.lo;
  if (hi) *hi = d.hi;
  if (doc) *doc = d.doc;
  return GPET_OK;
}

int gpet_ctx_create(int device, void* stream, gpet_ctx** out) {
  if (!out) return GPET_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return GPET_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = new (std::nothrow) gpet_ctx();
  if (!c) return GPET_ERR_HIP;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) {
    delete c;
    return GPET_ERR_HIP;
  }
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return GPET_ERR_HIP;
    }
    c->own_stream = true;
  }
  (void)hipEventCreate(&c->ev0);
  (void)hipEventCreate(&c->ev1);
  *out = c;
  return GPET_OK;
}

void gpet_ctx_destroy(gpet_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* gpet_last_error(const gpet_ctx* c) { return c ? c->err.c_str() : "null context"; }

int gpet_sync(gpet_ctx* c) {
  if (!c) return GPET_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));  // (gpet_wait keys its blocking event on the calling thread's current device)
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

void* gpet_ctx_stream(gpet_ctx* c) { return c ? (void*)c->stream : nullptr; }

int gpet_timer_start(gpet_ctx* c) {
  if (!c) return GPET_ERR_BAD_ARG;
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  return GPET_OK;
}

int gpet_timer_stop_ms(gpet_ctx* c, float* ms) {
  if (!c || !ms) return GPET_ERR_BAD_ARG;
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev1));
  HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
  return GPET_OK;
}

// ---- a1 -------------------------------------------------------------------------------
// The a1 entry points keep their device scratch in the context: one allocation, grown on demand, freed with the
// context -- nothing to leak on an error path and no hipMalloc/hipFree per call.
static int ctx_scratch(gpet_ctx* c, size_t bytes) {
  if (bytes <= c->scratch_bytes) return GPET_OK;
  HIPCHK(c, gpet_wait(c->stream));
  if (c->scratch) (void)hipFree(c->scratch);
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  HIPCHK(c, hipMalloc(&c->scratch, bytes));
  c->scratch_bytes = bytes;
  return GPET_OK;
}

int gpet_grad_image(gpet_ctx* c, const double* img, int M, int N, const double* kern, int kh, int kw, float* out) {
  if (!c || !img || !kern || !out || M <= 0 || N <= 0 || kh <= 0 || kw <= 0) return fail(c, GPET_ERR_BAD_ARG, "gpet_grad_image: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t px = (size_t)M * N;
  // scipy.ndimage.convolve == correlate with the flipped kernel; even extents shift the origin
  std::vector<double> wf((size_t)kh * kw);
  for (int a = 0; a < kh; ++a)
    for (int b = 0; b < kw; ++b) wf[(size_t)a * kw + b] = kern[(size_t)(kh - 1 - a) * kw + (kw - 1 - b)];
  const int oy = kh / 2 - ((kh % 2 == 0) ? 1 : 0), ox = kw / 2 - ((kw % 2 == 0) ? 1 : 0);
  Carver meas;
  (void)meas.take<double>(px);
  (void)meas.take<double>(wf.size());
  (void)meas.take<float>(px);
  (void)meas.take<float>(px);
  (void)meas.take<unsigned int>(2);
  int rc = ctx_scratch(c, meas.off + 256);
  if (rc) return rc;
  Carver cv;
  cv.base = c->scratch;
  double* d_img = cv.take<double>(px);
  double* d_wf = cv.take<double>(wf.size());
  float* d_tmp = cv.take<float>(px);
  float* d_out = cv.take<float>(px);
  unsigned int* d_mm = cv.take<unsigned int>(2);
  const unsigned int mm0[2] = {0xFFFFFFFFu, 0u};
  HIPCHK(c, hipMemcpyAsync(d_img, img, px * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_wf, wf.data(), wf.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_mm, mm0, sizeof mm0, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, launch_conv(c->stream, d_img, M, N, d_wf, kh, kw, oy, ox, d_tmp, d_mm));
  HIPCHK(c, launch_normalise(c->stream, d_tmp, px, d_mm, d_out));
  HIPCHK(c, hipMemcpyAsync(out, d_out, px * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

int gpet_normalise_f32(gpet_ctx* c, const float* img, size_t count, float* out) {
  if (!c || !img || !out || count == 0) return fail(c, GPET_ERR_BAD_ARG, "gpet_normalise_f32: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  Carver meas;
  (void)meas.take<float>(count);
  (void)meas.take<float>(count);
  (void)meas.take<unsigned int>(2);
  int rc = ctx_scratch(c, meas.off + 256);
  if (rc) return rc;
  Carver cv;
  cv.base = c->scratch;
  float* d_in = cv.take<float>(count);
  float* d_out = cv.take<float>(count);
  unsigned int* d_mm = cv.take<unsigned int>(2);
  const unsigned int mm0[2] = {0xFFFFFFFFu, 0u};
  HIPCHK(c, hipMemcpyAsync(d_in, img, count * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_mm, mm0, sizeof mm0, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, launch_minmax(c->stream, d_in, count, d_mm));
  HIPCHK(c, launch_normalise(c->stream, d_in, count, d_mm, d_out));
  HIPCHK(c, hipMemcpyAsync(out, d_out, count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

// ---- batch ----------------------------------------------------------------------------

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
  const hipMemcpyKind up = (flags & GPET_GRAD_ON_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  for (int g = 0; g < n_img; ++g) {
    if (!grad[g]) return fail(c, GPET_ERR_BAD_ARG, "gradient image %d is a null pointer", g);
    const unsigned int mm0[2] = {0xFFFFFFFFu, 0u};
    HIPCHK(c, hipMemcpyAsync(b->d_raw, grad[g], px * sizeof(float), up, c->stream));
    HIPCHK(c, hipMemcpyAsync(b->d_minmax, mm0, sizeof mm0, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_minmax(c->stream, b->d_raw, px, b->d_minmax));
    HIPCHK(c, launch_normalise(c->stream, b->d_raw, px, b->d_minmax, (float*)b->h_edges[g].grad));
    HIPCHK(c, gpet_wait(c->stream));  // (the host copy of mm0 / a pageable source must stay valid)
  }
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
  for (int e = 0; e < B; ++e) carve_edge(meas, tmp[e], !b->share_image);
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
  b->h_scalars.resize(B);
  for (int e = 0; e < B; ++e) {
    EdgeDev& E = b->h_edges[e];
    E.sc = b->d_scalars + e;
    E.fin_out = b->d_fin_out + (size_t)e * 2 * bd.Lg;
    carve_edge(cv, E, !b->share_image);
    E.fin_par = b->d_fin_par + (size_t)e * 12;                 // (batch-contiguous; the per-edge carve is unused)
    E.obs_xy = b->d_obs + (size_t)e * 2 * bd.obs_cap;
    if (b->share_image) {
      E.grad = shared_grad;
      E.grad_kde = shared_kde;
    }
  }
  HIPCHK(c, hipMalloc(&b->d_edges, sizeof(EdgeDev) * B));
  HIPCHK(c, hipMalloc(&b->d_seeds, sizeof(unsigned int) * B));
  HIPCHK(c, hipMalloc(&b->d_minmax, sizeof(unsigned int) * 2));
  {
    // (the stream the normals run ahead of the loop on: default priority -- lowest / highest were measured, +-0)
    HIPCHK(c, hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
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
  for (int e = 0; e < B; ++e) {
    EdgeDev& E = b->h_edges[e];
    HIPCHK(c, hipMemcpyAsync((void*)E.init_xy, init_xy[e], sizeof(long long) * 2 * E.n_init, hipMemcpyHostToDevice,
                             c->stream));
    gpet_scalars s0;
    memset(&s0, 0, sizeof s0);
    s0.score_thresh = params[e].score_thresh;
    s0.done = (0 >= E.algo_thresh) ? 1 : 0;  // gpet.py:829 with no observations yet
    HIPCHK(c, hipMemcpyAsync(E.sc, &s0, sizeof s0, hipMemcpyHostToDevice, c->stream));
  }
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
      HIPCHK(c, launch_struct_basis(c->stream, b->d_edges, B, b->bd));
      int rc2 = fetch_all_scalars(b);
      if (rc2) return rc2;
      int r0_max = 0;
      for (int e = 0; e < B; ++e) {
        EdgeDev& E = b->h_edges[e];
        const gpet_scalars& s = b->h_scalars[e];
        if (s.status != GPET_OK || s.rank < 1 || s.rank >= E.r_cap) ok = false;  // rank capacity reached
        E.r0 = s.rank;
        if (s.rank > r0_max) r0_max = s.rank;
      }
      b->bd.r0_max = r0_max;
      // (n_cap <= 128: k_struct_H keeps U in LDS -- it fits with L streamed row by row; larger: U in HBM, blocked)
      if (b->bd.n_cap <= 128 &&
          ((size_t)b->bd.n_cap * (r0_max | 1) + b->bd.n_cap + b->bd.r_cap) * sizeof(double) > (size_t)STRUCT_H_LDS_MAX)
        ok = false;
      // edges of the same grid length, first column, kernel and length scale have the same prior eigenbasis bit for bit
      // (k_rho_fill forms the lags as fl((x_st+i)/l) - fl((x_st+j)/l), which depends on x_st in the last bits unless l is
      // a power of two -- so x_st is part of the match; the amplitude is not: the matrix has unit amplitude): they all
      // read the first such edge's copy, which then stays in L2 for the whole batch (k_struct_H gathers its rows,
      // k_struct_rows streams it: 288 KB per edge at rank 72, Lg 500) -- option "shared_basis" = 0: every edge its own
      if (ok && option("shared_basis")) {
        for (int e = 1; e < B; ++e) {
          EdgeDev& E = b->h_edges[e];
          for (int j = 0; j < e; ++j) {
            const EdgeDev& F = b->h_edges[j];
            if (F.Lg == E.Lg && F.x_st == E.x_st && F.kernel_type == E.kernel_type && F.nu_code == E.nu_code && F.nu_gen == E.nu_gen &&
                F.length_scale == E.length_scale && F.r0 == E.r0 && F.r_cap == E.r_cap) {
              E.Q0 = F.Q0;
              E.lam0 = F.lam0;
              break;
            }
            if (j >= 8) break;  // (batches are homogeneous or nearly so: a short search)
          }
        }
      }
      // back to the pristine scalar state
      for (int e = 0; e < B; ++e) {
        EdgeDev& E = b->h_edges[e];
        E.structured = ok ? 1 : 0;
        gpet_scalars s0;
        memset(&s0, 0, sizeof s0);
        s0.score_thresh = params[e].score_thresh;
        s0.done = (0 >= E.algo_thresh) ? 1 : 0;
        HIPCHK(c, hipMemcpyAsync(E.sc, &s0, sizeof s0, hipMemcpyHostToDevice, c->stream));
      }
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

static int read_scalars(gpet_batch* b, int e, gpet_scalars* s) {
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipMemcpyAsync(s, b->h_edges[e].sc, sizeof *s, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

static int fetch_all_scalars(gpet_batch* b) {
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipMemcpyAsync(b->h_scalars.data(), b->d_scalars, sizeof(gpet_scalars) * b->B, hipMemcpyDeviceToHost,
                           c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

static int check_device_status(gpet_batch* b) {
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

int gpet_batch_set_rng(gpet_batch* b, int mode) {
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
  if (!b || e < 0 || e >= b->B) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  EdgeDev& E = b->h_edges[e];
  E.factor_injected = 0;
  HIPCHK(c, hipMemcpyAsync(b->d_edges + e, &E, sizeof E, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, gpet_wait(c->stream));
  return GPET_OK;
}

// Normals the loop stores per sample row: a structured batch's factors have at most r0_max rows (the posterior lives in
// the prior's r0 eigen-directions), so only that many columns of the z_cols-wide block are ever multiplied.  The stage
// API (gpet_gp_normals) always stores the whole block: its factor may come from the generic path or from the caller.
static inline int loop_z_store(const gpet_batch* b) {
  if (!b->structured || b->bd.r0_max < 1 || option("z_store_full")) return 0;
  return (b->bd.r0_max + 3) & ~3;
}

// ---- stages ---------------------------------------------------------------------------
int gpet_gp_fit_predict(gpet_batch* b, int want_cov) {
  if (!b) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 1));
  HIPCHK(c, launch_fit_predict(c->stream, b->d_edges, b->B, b->bd, want_cov));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 0));
  b->have_fit = true;
  return check_device_status(b);
}

int gpet_gp_factor(gpet_batch* b) {
  if (!b) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  if (!b->have_fit) return fail(c, GPET_ERR_STATE, "gpet_gp_factor before gpet_gp_fit_predict");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 1));
  HIPCHK(c, launch_factor(c->stream, b->d_edges, b->B, b->bd, ~0u, b->h_edges.data()));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 0));
  b->have_factor = true;
  return check_device_status(b);
}

int gpet_gp_normals(gpet_batch* b, const uint32_t* seeds) {
  if (!b || !seeds) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(b->d_seeds, seeds, sizeof(uint32_t) * b->B, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 1));
  {
    int rcn = normals_auto(b, c->stream, b->d_edges, b->B, b->d_seeds, 0, -1, 1, 0);
    if (rcn) return rcn;
  }
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 0));
  HIPCHK(c, gpet_wait(c->stream));
  b->have_normals = true;
  return GPET_OK;
}

int gpet_gp_sample(gpet_batch* b) {
  if (!b) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  if (!b->have_fit || !b->have_factor || !b->have_normals)
    return fail(c, GPET_ERR_STATE, "gpet_gp_sample needs fit, factor and normals first");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 1));
  HIPCHK(c, launch_sample(c->stream, b->d_edges, b->B, b->bd));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 0));
  b->have_samples = true;
  return GPET_OK;
}

int gpet_score_curves(gpet_batch* b) {
  if (!b) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  if (!b->have_samples) return fail(c, GPET_ERR_STATE, "gpet_score_curves before samples exist");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 1));
  HIPCHK(c, launch_score(c->stream, b->d_edges, b->B, b->bd));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 0));
  b->have_scores = true;
  return GPET_OK;
}

int gpet_curve_kde(gpet_batch* b) {
  if (!b) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  if (!b->have_scores) return fail(c, GPET_ERR_STATE, "gpet_curve_kde before gpet_score_curves");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 1));
  HIPCHK(c, launch_kde(c->stream, b->d_edges, b->B, b->bd, 0));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 0));
  return check_device_status(b);
}

int gpet_final_cov(gpet_batch* b) {
  if (!b) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  for (int e = 0; e < b->B; ++e)
    if (b->h_edges[e].fin_n < 1) return fail(c, GPET_ERR_STATE, "gpet_final_cov before gpet_final_predict_all");
  HIPCHK(c, launch_final_cov(c->stream, b->d_edges, b->B, b->bd));
  HIPCHK(c, gpet_wait(c->stream));
  b->have_fit = true;  // (mean in the caller's hands, covariance in GPET_BUF_COV: gpet_gp_factor may follow)
  return GPET_OK;
}

int gpet_select_pixels(gpet_batch* b) {
  if (!b) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  if (!b->have_scores) return fail(c, GPET_ERR_STATE, "gpet_select_pixels before gpet_score_curves");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 1));
  HIPCHK(c, launch_kde(c->stream, b->d_edges, b->B, b->bd, 0));
  HIPCHK(c, launch_pixels(c->stream, b->d_edges, b->B, b->bd));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 0));
  b->iters_issued += 1;  // k_pix_select advanced every active edge's iteration counter
  return check_device_status(b);
}

static int batch_reset(gpet_batch* b, bool next_frame) {
  gpet_ctx* c = b->ctx;
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
  if (!b) return GPET_ERR_BAD_ARG;
  return batch_reset(b, false);
}

int gpet_batch_set_images(gpet_batch* b, const float* const* grad, unsigned int flags) {
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

int gpet_profile_stage(gpet_batch* b, int stage, int reps, float* ms_per_rep) {
  if (!b || !ms_per_rep || reps < 1) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  for (int r = 0; r < reps; ++r) {
    switch (stage) {
      case 0:
        if (b->structured) HIPCHK(c, launch_struct_iteration(c->stream, b->d_edges, b->B, b->bd, 1u | 2u));
        else HIPCHK(c, launch_fit_predict(c->stream, b->d_edges, b->B, b->bd, 1));
        break;
      case 1:
        if (b->structured) HIPCHK(c, launch_struct_iteration(c->stream, b->d_edges, b->B, b->bd, 4u | 8u));
        else HIPCHK(c, launch_factor(c->stream, b->d_edges, b->B, b->bd, ~0u, b->h_edges.data()));
        break;
      case 120: case 121: case 122: case 123:  // structured path: fit, (U, H, mean), Jacobi, factor rows
        HIPCHK(c, launch_struct_iteration(c->stream, b->d_edges, b->B, b->bd, 1u << (stage - 120))); break;
      case 2:
        if (b->rng_mode == 1) HIPCHK(c, launch_normals_philox(c->stream, b->d_edges, b->B, b->bd, b->d_seeds, 1, -1, b->bd.z_ring, loop_z_store(b)));
        else HIPCHK(c, launch_normals_seq(b, c->stream, b->d_edges, b->B, b->d_seeds, 1, -1, b->bd.z_ring, loop_z_store(b)));
        break;
      case 3: HIPCHK(c, launch_sample(c->stream, b->d_edges, b->B, b->bd, b->structured ? b->bd.r0_max : 0)); break;
      case 4: HIPCHK(c, launch_score(c->stream, b->d_edges, b->B, b->bd)); break;
      case 5: HIPCHK(c, launch_kde(c->stream, b->d_edges, b->B, b->bd, 0, ~0u, 1)); break;  // (the loop form: raw, band only)
      case 6: HIPCHK(c, launch_pixels_reset(c->stream, b->d_edges, b->B, b->bd)); break;  // (reset only: selection mutates the loop state)
      // single kernels: 100+ fit/predict/cov, 110+ pchol/gram/jacobi/rows, 130 gemm, 140+ score/topk, 150+ kde prep/fused/normalise
      case 100: case 101: case 102:
        HIPCHK(c, launch_fit_predict(c->stream, b->d_edges, b->B, b->bd, 1, 1u << (stage - 100))); break;
      case 110: case 111: case 112: case 113:
        HIPCHK(c, launch_factor(c->stream, b->d_edges, b->B, b->bd, 1u << (stage - 110))); break;
      case 130: HIPCHK(c, launch_sample(c->stream, b->d_edges, b->B, b->bd, b->structured ? b->bd.r0_max : 0)); break;
      case 140: case 141: HIPCHK(c, launch_score(c->stream, b->d_edges, b->B, b->bd, 1u << (stage - 140))); break;
      case 150: case 151: HIPCHK(c, launch_kde(c->stream, b->d_edges, b->B, b->bd, 0, 1u << (stage - 150), 1)); break;
      case 152: HIPCHK(c, launch_kde(c->stream, b->d_edges, b->B, b->bd, 0, 4u, 0)); break;  // (stage-API form only)
      // 160: the column scan of the pixel selection, loop form (reads the raw KDE band of stage 151; it only raises
      // per-bin maxima to values they already hold, so repeating it leaves the loop state as it was)
      case 160: HIPCHK(c, launch_pixels(c->stream, b->d_edges, b->B, b->bd, 1, 1u)); break;
      default: return fail(c, GPET_ERR_BAD_ARG, "gpet_profile_stage: unknown stage %d", stage);
    }
  }
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev1));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *ms_per_rep = ms / (float)reps;
  return check_device_status(b);
}

int gpet_final_set_training(gpet_batch* b, int e, const double* xs, const double* ys, const double* w, int n) {
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
  if (!b || !dst) return GPET_ERR_BAD_ARG;
  HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
  int rc = fetch_all_scalars(b);
  if (rc) return rc;
  memcpy(dst, b->h_scalars.data(), sizeof(gpet_scalars) * b->B);
  return GPET_OK;
}

int gpet_batch_read_obs_all(gpet_batch* b, int64_t* dst, int32_t* counts, int stride_obs) {
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

int gpet_select_pixels_only(gpet_batch* b) {
  if (!b) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 1));
  HIPCHK(c, launch_pixels_reset(c->stream, b->d_edges, b->B, b->bd));
  HIPCHK(c, launch_pixels(c->stream, b->d_edges, b->B, b->bd));
  HIPCHK(c, launch_set_force(c->stream, b->d_edges, b->B, 0));
  b->iters_issued += 1;
  return check_device_status(b);
}

int gpet_trace_iterate(gpet_batch* b, const uint32_t* base_seeds, int max_iters, int* n_active) {
  if (!b || !base_seeds || !n_active || max_iters < 0) return GPET_ERR_BAD_ARG;
  gpet_ctx* c = b->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(b->d_seeds, base_seeds, sizeof(uint32_t) * b->B, hipMemcpyHostToDevice, c->stream));
  // Normals: the seeds of upcoming iterations are known (gpet.py:839), so the RNG stream runs ahead of the loop on
  // its own HIP stream, one launch per iteration, `look` iterations ahead (gpet_set_option("rng_lookahead", n), default
  // 1: the draws of iteration k+1 are enqueued when iteration k starts and run next to it), never past the horizon of
  // the iterations enqueued together.  An edge that finishes still gets the draws already enqueued for it, so a deeper
  // look-ahead only wastes generator work (n = 4: 19 % of it; measured loop time of a batch alone: n = 1, 2, 4 within
  // 1 %).  n = 0 orders the draws of iteration k after the pixel selection of iteration k-1 -- nothing is drawn for
  // finished edges, but the generator then competes with the eigen-solver for the start of every iteration: 187 instead
  // of 179 ms per loop of 1024 edges, 70 instead of 56 ms at 256.  A ring slot is refilled only after the sample GEMM
  // that read it (one ring earlier) has completed.
  const int ring = b->bd.z_ring;
  int look = gpet_opt_rng_lookahead();
  // Automatic (default): a batch that fills the GPU is throughput-bound in the generator, so one iteration ahead wastes
  // the least; a small batch is LATENCY-bound in it -- a stream is sequential, one workgroup per (edge, iteration),
  // 2.1 ms for the 500 k normals of a 500-column edge against 0.9 ms for the rest of an iteration -- so the streams of
  // the next 8 iterations are generated side by side, by one launch, across group boundaries.
  const bool deep = look < 0 ? (b->B <= 64) : (look > 4);
  if (look < 0) look = deep ? 8 : 1;
  if (look > ring - 1) look = ring - 1;
  // The iterations are enqueued in groups of 8, then 4 and -- once the first edges have finished -- 2: after every
  // group the host reads the `done` flags, stops if no edge is left and otherwise launches the next group on a
  // COMPACTED copy of the edge table (only the edges still running).  Every kernel skips finished edges by itself, but
  // it still starts one workgroup per edge and tile to find that out: 2.2 ms per iteration for 1024 finished edges, and
  // the last iterations of a batch run with a handful of edges left.
  int remaining = max_iters, group = 8, active = b->B;
  bool flags_known = false;
  *n_active = b->B;
  if (max_iters == 0) {
    int rc0 = check_device_status(b);
    if (rc0) return rc0;
    active = 0;
    for (int e = 0; e < b->B; ++e) active += b->h_scalars[e].done ? 0 : 1;
    *n_active = active;
    return GPET_OK;
  }
  while (remaining > 0) {
    const int n_it = group < remaining ? group : remaining;
    EdgeDev* edges_l = b->d_edges;
    unsigned int* seeds_l = b->d_seeds;
    int B_l = b->B;
    if (flags_known && active < b->B) {
      b->h_edges_act.clear();
      b->h_seeds_act.clear();
      for (int e = 0; e < b->B; ++e)
        if (!b->h_scalars[e].done) {
          b->h_edges_act.push_back(b->h_edges[e]);
          b->h_seeds_act.push_back(base_seeds[e]);
        }
      B_l = (int)b->h_edges_act.size();
      if (!b->d_edges_act) {
        HIPCHK(c, hipMalloc(&b->d_edges_act, sizeof(EdgeDev) * b->B));
        HIPCHK(c, hipMalloc(&b->d_seeds_act, sizeof(unsigned int) * b->B));
      }
      // (the previous group has completed -- check_device_status synchronised -- so the tables may be overwritten)
      HIPCHK(c, hipMemcpyAsync(b->d_edges_act, b->h_edges_act.data(), sizeof(EdgeDev) * B_l, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(b->d_seeds_act, b->h_seeds_act.data(), sizeof(unsigned int) * B_l, hipMemcpyHostToDevice, c->stream));
      edges_l = b->d_edges_act;
      seeds_l = b->d_seeds_act;
    }
    const int first = b->iters_issued, horizon = first + n_it;
    if (b->norm_issued < first) b->norm_issued = first;
    HIPCHK(c, hipEventRecord(b->ev_main, c->stream));  // the seeds and the edge table are on the device
    HIPCHK(c, hipStreamWaitEvent(b->side, b->ev_main, 0));
    for (int it = 0; it < n_it; ++it) {
      // every kernel skips edges whose `done` flag is set, so edges that finish inside a group cost little.
      const int cur = first + it;
      // GPET_RNG_INLINE: 0 = the generator runs ahead of the loop on its own stream (small batches: always); 2 = the streams
      // of ALL the iterations of a group in one launch on the loop's own stream (batches above 64 edges: the default -- the
      // launch fills the GPU and runs beside nothing, 157-159 instead of 161-162 ms per step of 1 024 traces); 1 = one
      // iteration per launch on the loop's stream (an experiment: 179 ms)
      const int rng_inline_opt = option("rng_inline");
      const int rng_inline = rng_inline_opt >= 0 ? rng_inline_opt : (deep ? 0 : 2);
      if (rng_inline == 1) {  // experiment: the normals of this iteration on the loop's own stream, overlapping nothing
        int rcn = normals_auto(b, c->stream, edges_l, B_l, seeds_l, 1, cur, 1, loop_z_store(b));
        if (rcn) return rcn;
        HIPCHK(c, hipEventRecord(b->ev_norm[cur % 16], c->stream));
        b->norm_issued = cur + 1;
      } else if (rng_inline == 2 && b->norm_issued <= cur) {
        // experiment: the streams of ALL the iterations of this group (up to ring - 1) in one launch on the loop's own stream:
        // nothing beside it, and enough workgroups to fill the GPU
        int n = horizon - cur;
        if (n > ring - 1) n = ring - 1;
        int rcn = normals_auto(b, c->stream, edges_l, B_l, seeds_l, 1, cur, n, loop_z_store(b));
        if (rcn) return rcn;
        for (int q = cur; q < cur + n; ++q) HIPCHK(c, hipEventRecord(b->ev_norm[q % 16], c->stream));
        b->norm_issued = cur + n;
      }
      if (!rng_inline && deep && b->norm_issued - cur <= look / 2) {
        // small batch: the streams of the next `n` iterations in ONE launch (blockIdx.x = iteration), side by side.
        // Their ring slots were last read by the sample GEMMs of iterations <= cur - 1 (outstanding + n <= ring).
        const int j = b->norm_issued;
        int n = ring - (j - cur);
        if (n > look) n = look;
        if (cur - 1 >= first) HIPCHK(c, hipStreamWaitEvent(b->side, b->ev_gemm[(cur - 1) % 16], 0));
        {
          int rcn = normals_auto(b, b->side, edges_l, B_l, seeds_l, 1, j, n, loop_z_store(b));
          if (rcn) return rcn;
        }
        for (int q = j; q < j + n; ++q) HIPCHK(c, hipEventRecord(b->ev_norm[q % 16], b->side));
        b->norm_issued = j + n;
      }
      const int look_now = look;  // (after the GEMM instead of beside the eigen-solver was measured: +-0)
      while (!rng_inline && !deep && b->norm_issued <= cur + look_now && b->norm_issued < horizon) {
        const int j = b->norm_issued;
        if (look == 0) {
          if (j - 1 >= first) HIPCHK(c, hipStreamWaitEvent(b->side, b->ev_pix[(j - 1) % 16], 0));
        } else if (j - ring >= first) {
          HIPCHK(c, hipStreamWaitEvent(b->side, b->ev_gemm[(j - ring) % 16], 0));
        }
        {
          int rcn = normals_auto(b, b->side, edges_l, B_l, seeds_l, 1, j, 1, loop_z_store(b));
          if (rcn) return rcn;
        }
        HIPCHK(c, hipEventRecord(b->ev_norm[j % 16], b->side));
        b->norm_issued = j + 1;
      }
      if (b->structured) {
        HIPCHK(c, launch_struct_iteration(c->stream, edges_l, B_l, b->bd));
      } else {
        HIPCHK(c, launch_fit_predict(c->stream, edges_l, B_l, b->bd, 1));
        HIPCHK(c, launch_factor(c->stream, edges_l, B_l, b->bd, ~0u, edges_l == b->d_edges ? b->h_edges.data() : b->h_edges_act.data()));
      }
      HIPCHK(c, hipStreamWaitEvent(c->stream, b->ev_norm[cur % 16], 0));
      // samples + scores: the GEMM writes all S rows and the scorer reads them back (two fused forms that never wrote the sample
      // matrix were built in rounds 3 and 4, bit-identical, and measured slower: an f64 matrix instruction and f64 vector
      // work do not overlap, DESIGN.md history)
      const int rank_max = b->structured ? b->bd.r0_max : 0;
      HIPCHK(c, launch_sample(c->stream, edges_l, B_l, b->bd, rank_max));
      HIPCHK(c, hipEventRecord(b->ev_gemm[cur % 16], c->stream));
      HIPCHK(c, launch_score(c->stream, edges_l, B_l, b->bd));
      // loop form: the density stays raw and band-limited in HBM; the pixel kernels normalise on the fly
      HIPCHK(c, launch_kde(c->stream, edges_l, B_l, b->bd, 0, ~0u, 1));
      HIPCHK(c, launch_pixels(c->stream, edges_l, B_l, b->bd, 1));
      HIPCHK(c, hipEventRecord(b->ev_pix[cur % 16], c->stream));
      b->iters_issued += 1;
    }
    b->have_fit = b->have_factor = b->have_normals = b->have_samples = b->have_scores = true;
    HIPCHK(c, gpet_wait(b->side));  // (its launches read the compacted tables too)
    int rc = check_device_status(b);
    if (rc) return rc;
    active = 0;
    for (int e = 0; e < b->B; ++e) active += b->h_scalars[e].done ? 0 : 1;
    flags_known = true;
    *n_active = active;
    remaining -= n_it;
    if (active == 0) break;
    group = active == b->B ? (group < 4 ? group : 4) : 2;
  }
  return GPET_OK;
}

}  // extern "C"
