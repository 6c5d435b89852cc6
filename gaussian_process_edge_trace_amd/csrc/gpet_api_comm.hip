// The C ABI's multi-GPU helpers (include/gpet_hip.h, "collectives"): one process per GPU, independent edges in
// contiguous blocks per rank, ONE broadcast of the shared gradient image(s) and ONE gather of the finished traces -- there is
// no collective on the data path of a trace.  RCCL is bound at run time (dlopen of librccl.so.1: a process that has already
// loaded a copy, e.g. torch's, gets that one), so the library itself links nothing but the HIP runtime and a single-GPU
// host never needs RCCL installed.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "gpet_api_internal.h"


namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};
Rccl& rccl() {
  static Rccl r = [] {
    Rccl q;
    // GPET_RCCL_LIB: the one library to bind instead of the default names (a deployment's own build of RCCL; the tests point
    // it at a file that does not exist to see GPET_ERR_UNSUPPORTED come back)
    const char* forced = getenv("GPET_RCCL_LIB");
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (forced && *forced) q.lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    else
      for (const char* n : names)
        if ((q.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!q.lib) {
      const char* m = dlerror();  // (ONE call: dlerror clears the message it returns)
      q.why = std::string("RCCL not found (dlopen ") + (forced && *forced ? forced : "librccl.so.1") + "): " + (m ? m : "?");
      return q;
    }
#define GPET_RCCL_SYM(field, name)                                      \
  q.field = reinterpret_cast<decltype(q.field)>(dlsym(q.lib, name));   \
  if (!q.field && q.why.empty()) q.why = std::string("RCCL symbol missing: ") + name;
    GPET_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    GPET_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    GPET_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    GPET_RCCL_SYM(Broadcast, "ncclBroadcast")
    GPET_RCCL_SYM(AllGather, "ncclAllGather")
    GPET_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef GPET_RCCL_SYM
    return q;
  }();
  return r;
}
}  // namespace

struct gpet_comm {
  gpet_ctx* ctx = nullptr;
  int device = 0;           // (its own copy: gpet_comm_destroy must not need the context, which the caller may have destroyed first)
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
  char* scratch = nullptr;  // device staging of the gather (send block | world receive blocks)
  size_t scratch_bytes = 0;
};

#define NCCLCHK(ctx, call)                                                                                       \
  do {                                                                                                           \
    ncclResult_t r_ = (call);                                                                                    \
    if (r_ != ncclSuccess)                                                                                       \
      return fail((ctx), GPET_ERR_HIP, "%s failed: %s (%s:%d)", #call, rccl().GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

static void block_of(size_t n, int world, int rank, size_t& lo, size_t& hi) {  // sharding.edge_slice
  const size_t base = n / (size_t)world, rem = n % (size_t)world;
  lo = (size_t)rank * base + std::min((size_t)rank, rem);
  hi = lo + base + ((size_t)rank < rem ? 1 : 0);
}

extern "C" {

int gpet_comm_unique_id(void* id128) {
  if (!id128) return GPET_ERR_BAD_ARG;
  static_assert(sizeof(ncclUniqueId) <= GPET_COMM_ID_BYTES, "unique id fits the ABI's buffer");
  Rccl& r = rccl();
  if (!r.why.empty()) return GPET_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (r.GetUniqueId(&id) != ncclSuccess) return GPET_ERR_HIP;
  memset(id128, 0, GPET_COMM_ID_BYTES);
  memcpy(id128, &id, sizeof id);
  return GPET_OK;
}

int gpet_comm_create(gpet_ctx* ctx, const void* id128, int world, int rank, gpet_comm** out) {
  if (!ctx || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) return GPET_ERR_BAD_ARG;
  *out = nullptr;
  gpet_comm* c = new (std::nothrow) gpet_comm();
  if (!c) return GPET_ERR_HIP;
  c->ctx = ctx;
  c->device = ctx->device;
  c->world = world;
  c->rank = rank;
  if (world > 1 || option("comm_force_rccl")) {  // (a single rank needs no communicator: its collectives are copies)
    Rccl& r = rccl();
    if (!r.why.empty()) {
      delete c;
      return fail(ctx, GPET_ERR_UNSUPPORTED, "%s", r.why.c_str());
    }
    if (hipSetDevice(ctx->device) != hipSuccess) {
      delete c;
      return fail(ctx, GPET_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    }
    ncclUniqueId id;
    if (id128) memcpy(&id, id128, sizeof id);
    else if (r.GetUniqueId(&id) != ncclSuccess) {
      delete c;
      return fail(ctx, GPET_ERR_HIP, "ncclGetUniqueId failed");
    }
    const ncclResult_t e = r.CommInitRank(&c->comm, world, id, rank);
    if (e != ncclSuccess) {
      delete c;
      return fail(ctx, GPET_ERR_HIP, "ncclCommInitRank(world %d, rank %d) failed: %s", world, rank, r.GetErrorString(e));
    }
  }
  *out = c;
  return GPET_OK;
}

void gpet_comm_destroy(gpet_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  delete c;
}

int gpet_comm_rank(const gpet_comm* c) { return c ? c->rank : -1; }
int gpet_comm_world(const gpet_comm* c) { return c ? c->world : -1; }

int gpet_comm_block(const gpet_comm* c, int64_t n_units, int64_t* lo, int64_t* hi) {
  if (!c || n_units < 0 || !lo || !hi) return GPET_ERR_BAD_ARG;
  size_t a, b;
  block_of((size_t)n_units, c->world, c->rank, a, b);
  *lo = (int64_t)a;
  *hi = (int64_t)b;
  return GPET_OK;
}

// device memory for the broadcast buffer, for hosts that do not talk to the HIP runtime themselves
int gpet_dev_alloc(gpet_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return GPET_ERR_BAD_ARG;
  *out = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMalloc(out, bytes ? bytes : 1));
  return GPET_OK;
}
int gpet_dev_free(gpet_ctx* ctx, void* p) {
  if (!ctx) return GPET_ERR_BAD_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (p) HIPCHK(ctx, hipFree(p));
  return GPET_OK;
}
int gpet_dev_copy(gpet_ctx* ctx, void* dst, const void* src, size_t bytes, int to_host) {
  if (!ctx || (bytes && (!dst || !src))) return GPET_ERR_BAD_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, gpet_wait(ctx->stream));
  return GPET_OK;
}

int gpet_bcast_grad(gpet_comm* c, float* d_grad, size_t count, int root) {
  if (!c || !d_grad || root < 0 || root >= c->world) return GPET_ERR_BAD_ARG;
  gpet_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (c->comm) NCCLCHK(ctx, rccl().Broadcast(d_grad, d_grad, count, ncclFloat, root, c->comm, ctx->stream));
  // (the library's batches consume the buffer on the same stream: no wait needed for them; the caller may read it after gpet_sync)
  return GPET_OK;
}

int gpet_allgather_i64(gpet_comm* c, const int64_t* h_local, const int64_t* counts, int64_t* h_all) {
  if (!c || !counts || !h_all) return GPET_ERR_BAD_ARG;
  gpet_ctx* ctx = c->ctx;
  size_t cap = 0, total = 0;
  for (int r = 0; r < c->world; ++r) {
    if (counts[r] < 0) return GPET_ERR_BAD_ARG;
    cap = std::max(cap, (size_t)counts[r]);
    total += (size_t)counts[r];
  }
  const size_t mine = (size_t)counts[c->rank];
  if (mine > 0 && !h_local) return GPET_ERR_BAD_ARG;
  if (total == 0) return GPET_OK;
  if (!c->comm) {
    memcpy(h_all, h_local, mine * sizeof(int64_t));
    return GPET_OK;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // blocks of different lengths: padded to the longest (the traces of a rank's edges: blocks differ by one edge at most)
  const size_t need = (size_t)(c->world + 1) * cap * sizeof(int64_t);
  if (need > c->scratch_bytes) {
    if (c->scratch) (void)hipFree(c->scratch);
    c->scratch = nullptr;
    c->scratch_bytes = 0;
    HIPCHK(ctx, hipMalloc(&c->scratch, need));
    c->scratch_bytes = need;
  }
  int64_t* d_send = reinterpret_cast<int64_t*>(c->scratch);
  int64_t* d_recv = d_send + cap;
  HIPCHK(ctx, hipMemsetAsync(d_send, 0, cap * sizeof(int64_t), ctx->stream));
  if (mine) HIPCHK(ctx, hipMemcpyAsync(d_send, h_local, mine * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  NCCLCHK(ctx, rccl().AllGather(d_send, d_recv, cap, ncclInt64, c->comm, ctx->stream));
  size_t off = 0;
  for (int r = 0; r < c->world; ++r) {
    if (counts[r])
      HIPCHK(ctx, hipMemcpyAsync(h_all + off, d_recv + (size_t)r * cap, (size_t)counts[r] * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    off += (size_t)counts[r];
  }
  HIPCHK(ctx, gpet_wait(ctx->stream));
  return GPET_OK;
}

int gpet_gather_traces(gpet_comm* c, const int64_t* h_local, int64_t n_edges, int64_t edge_len, int64_t* h_all) {
  if (!c || n_edges < 0 || edge_len < 0 || !h_all) return GPET_ERR_BAD_ARG;
  std::vector<int64_t> counts((size_t)c->world);
  for (int r = 0; r < c->world; ++r) {
    size_t lo, hi;
    block_of((size_t)n_edges, c->world, r, lo, hi);
    counts[(size_t)r] = (int64_t)(hi - lo) * edge_len * 2;
  }
  return gpet_allgather_i64(c, h_local, counts.data(), h_all);
}

}  // extern "C"
