// C ABI of libgpet_hip.so (include/gpet_hip.h), part 1: contexts, options, timers, the gradient image (a1), and the helpers the
// other parts share.
#include "gpet_api_internal.h"

// Host waits.  hipStreamSynchronize spins on a CPU core; with one process per GPU and a few driver threads per process
// (device loop + converged fits in flight) eight ranks would keep 32 threads spinning on a node's cores.  In blocking
// mode (gpet_set_option("blocking_sync", 1); default: on when WORLD_SIZE > 1, i.e. under torch.distributed.run) a wait
// is an event created with hipEventBlockingSync: the thread sleeps until the GPU signals.
static int opt_blocking_sync() {
  static const int i_ = option_index("blocking_sync");
  int& v = option_at(i_);
  if (v >= 0) return v;
  static const int by_world = [] {  // (WORLD_SIZE is torch.distributed's variable, not a switch of this library)
    const char* w = getenv("WORLD_SIZE");
    return (w && atoi(w) > 1) ? 1 : 0;
  }();
  return by_world;
}
hipError_t gpet_wait(hipStream_t st) {
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

int fail(gpet_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}


// Lattice of a caller-supplied training set: h with x_i = x_min + m_i h (the smallest positive gap, refined over the
// whole span), accepted when every point sits on it to 1e-6 of a step.  hinv = 1 / h; returns the largest lag or -1.
int fin_lattice(const double* x, int n, double* hinv) {
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
int& opt_fit_persistent() {
  static const int i_ = option_index("fit_persistent");
  int& v = option_at(i_);
  return v;
}

static int& opt_rng_chunked() {
  static const int i_ = option_index("rng_chunked");
  int& v = option_at(i_);  // -1: by launch shape
  return v;
}

// One sequential walk per stream: the register-resident generator (four streams per wave, gpet_rng.hip) when the batch is
// homogeneous and the launch has enough streams to fill the GPU with single waves (2 048 = half of its SIMDs; a wave of
// four streams takes ~2.5 ms against 0.6 ms for a three-wave workgroup per stream, so small launches keep the old kernel),
// else one workgroup per stream (k_mt_normals).  The same numbers either way.
hipError_t launch_normals_seq(gpet_batch* b, hipStream_t st, EdgeDev* edges_l, int B_l, const unsigned int* seeds_l,
                                     int add_iter, int iter_abs, int n_ahead, int z_store) {
  const int opt = gpet_opt_rng4();
  if (b->bd.rng4 && (opt > 0 || (opt < 0 && (long long)B_l * n_ahead >= 2048)))
    return launch_normals4(st, edges_l, B_l, seeds_l, add_iter, iter_abs, n_ahead, z_store, b->bd.Lg, b->bd.S, b->bd.z_cols);
  return launch_normals(st, edges_l, B_l, seeds_l, add_iter, iter_abs, n_ahead, z_store);
}

// The normals of `n_ahead` iterations of B_l edges: one workgroup per stream (k_mt_normals), or -- when that leaves
// most of the GPU idle and the streams are long -- every stream cut into chunks that many workgroups generate at once
// (MT19937 jump-ahead, launch_normals_chunked).  The same numbers either way.
int normals_auto(gpet_batch* b, hipStream_t st, EdgeDev* edges_l, int B_l, const unsigned int* seeds_l, int add_iter,
                        int iter_abs, int n_ahead, int z_store, bool allow_chunked) {
  gpet_ctx* c = b->ctx;
  if (b->rng_mode == 1) {  // opt-in Philox mode (gpet_batch_set_rng)
    HIPCHK(c, launch_normals_philox(st, edges_l, B_l, b->bd, seeds_l, add_iter, iter_abs, n_ahead, z_store));
    return GPET_OK;
  }
  const int streams = B_l * n_ahead;
  const int nc = mtj_chunks((long long)b->bd.S * b->bd.Lg);
  const int opt = opt_rng_chunked();
  const bool force4 = gpet_opt_rng4() > 0 && b->bd.rng4;  // (tests: the register-resident generator on any launch shape)
  // (the chunked form works in the batch's ONE jump workspace: a launch that runs beside another chunked launch of the same batch
  //  -- the tail of a small batch's first normals on the fit stream, gpet_trace_iterate -- must take the sequential kernel)
  const bool chunked = allow_chunked && !force4 && nc >= 2 && (opt > 0 || (opt < 0 && streams <= 32 && nc >= 4));
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

}  // extern "C"
