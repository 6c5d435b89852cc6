// The table of process-wide options (gpet_options.h).  Host code only.
#include "gpet_options.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace gpet {

namespace {

// name, default, lo, hi, meaning.  (-1 = automatic where lo == -1.)
const OptionDef kDefs[] = {
    {"blocking_sync", -1, -1, 1, "host waits sleep on a hipEventBlockingSync event instead of spinning in hipStreamSynchronize; -1: on when WORLD_SIZE > 1 (several ranks per host share its cores)"},
    {"rng4", -1, -1, 1, "normals by the register-resident generator k_mt_normals4 (four MT19937 streams per wave): -1 = launches of >= 2048 streams, 0 = never, 1 = whenever the batch is homogeneous"},
    {"rng_chunked", -1, -1, 1, "one MT19937 stream on many workgroups by jump-ahead: -1 = launches of <= 32 streams of >= 4 chunks, 0 = never, 1 = always"},
    {"rng_lookahead", -1, -1, 15, "iterations the normals may run ahead of the device loop on the side stream; -1: 8 up to 64 edges, else 1"},
    {"rng_head", -1, -1, 8, "batches of 2..32 edges: leading iterations of a trace whose normals are generated chunked (jump-ahead, one launch per iteration) beside the sequential launch of the following ones; -1: 4, 0: off"},
    {"loop_adaptive_groups", 1, 0, 1, "device loop of batches up to 64 edges: after the first group of 8 iterations the next group is what the slowest running edge still needs at its last rate of new observations; 0: the fixed 8 / 4 / 2 / 2 ladder"},
    {"rng_refill_at", -1, -1, 15, "small batches: the side stream refills the normals ring when at most this many generated iterations are left ahead of the loop; -1: look-ahead - 2 (round 5: look-ahead / 2)"},
    {"side_own_queue", 0, 0, 1, "1: a batch's RNG look-ahead stream is created with an all-CUs mask (hipExtStreamCreateWithCUMask), which makes the runtime give it a hardware queue of its own instead of one from the process's shared pool (read when a batch is created).  Off: in a process that creates contexts repeatedly it removes the occasional 32-edge loop whose look-ahead stream shares its own loop's queue (16 instead of 12 ms), but every such queue is one more for the hardware scheduler -- with the bench's eight objects in flight a 32-edge batch then takes 32 instead of 16 ms and the headline loses 3.5 %"},
    {"loop_fused_tail", -1, -1, 1, "device loop: k_score_combine + k_topk_sort + k_kde_prep as one launch per iteration (k_score_tail, the same bits); -1: batches up to 64 edges (latency chains), 1: always where the shape allows, 0: never"},
    {"rng_inline", -1, -1, 2, "where the loop's normals are generated: 0 = side stream, 1 = one iteration per launch on the loop's stream, 2 = all iterations of a group in one launch on the loop's stream; -1: 2 above 64 edges, else 0"},
    {"z_store_full", 0, 0, 1, "1: the structured loop stores all z_cols normals of a sample row instead of the r0 (rounded to 4) its factors multiply"},
    {"fit_persistent", -1, -1, 1, "converged fits as one workgroup per (edge, restart) problem: -1 = problem sets resident at once (<= 1024), 0 = lock-step rounds, 1 = always"},
    {"fin_prepare_serial", 0, 0, 1, "1: training sets of the converged fits by one thread per edge (cross-check of the one-wave-per-edge kernel)"},
    {"lml_mfma", 1, 0, 1, "objective of the converged fits on the f64 matrix cores (k_lml16) where the training set allows it; 0: register-tile kernels k_lml / k_lml2"},
    {"lml_two_tiles_from", 600, 1, 0x3fffffff, "problems per launch from which k_lml2 (two 4x4 tiles per thread) replaces k_lml"},
    {"jacobi_variant", 1, 0, 1, "LDS Jacobi of ranks <= 96: 1 = seated, rotation parameters one round ahead, one barrier per round (k_jacobi_ahead); 0 = seated, three barriers per round (k_jacobi_seat: the cross-check)"},
    {"jacobi_warm", 1, 0, 1, "structured loop: the eigen-decomposition of an iteration starts from the previous iteration's eigenvectors (k_jacobi_prerot) instead of the identity"},
    {"jacobi_wreg", 6, 0, 8, "k_jacobi_seat: 7 x this many rows of the eigenvector matrix in the worker waves' registers instead of LDS (0, 4, 6, 8)"},
    {"jacobi_logw", 1, 0, 1, "batches that have a rotation log (<= jlog_max_b edges): eigenvectors by a separate pass over the logged rotations"},
    {"wpass_lds", 0, 0, 1, "rotation-log pass (k_jacobi_wpass): 1 = eight rows of W per workgroup share every tile of the log through LDS (k_jacobi_wpass_lds: 68 against 80 us per launch of 32 edges, the same bits -- but the 32-edge loop then takes 13.6 instead of 11.8 ms: its 512-thread, 49 KB workgroups wait for CUs the look-ahead generator occupies); 0 = every wave streams the log itself"},
    {"jlog_max_b", 32, 0, 4096, "largest batch that gets a rotation log (read when a batch is created): measured per step of a batch alone -- 32 edges 21.5 against 23.1 ms, 64 edges 27.1 either way, 1 024 edges slower (3.15 against 1.73 ms per launch)"},
    {"oj_warm", 1, 0, 1, "any-rank factor: rows of full rank start from the previous iteration's rows (k_ojw_*: A Sigma A^T, its Cholesky factor, one product) instead of the pivoted Cholesky (across the frames of a sequence only where the caller asks: gpet_batch_set_images with GPET_IMAGES_NEXT_FRAME)"},
    {"oj_warm_fail", 0, 0, 1, "testing: the warm start's Cholesky reports a non-positive pivot, so that the factor falls back to the pivoted Cholesky"},
    {"oj_persist", 1, 0, 1, "any-rank Jacobi: rounds and sweeps in one launch, pair slots handed out by ticket (k_oj_persist); 0: one launch per round"},
    {"oj_stage", 1, 0, 1, "any-rank Jacobi: a pair's 16 rows staged in LDS; 0: operands from global memory"},
    {"oj_half_stage", -1, -1, 1, "any-rank Jacobi (k_oj_persist), even widths above 512 columns: a pair's 16-row panel staged in LDS one 512-column half at a time (66 KB, two workgroups per CU, the first half read a second time for the row update) instead of whole (131 KB at 1 024 columns, one per CU); the same bits; -1: where a round has more pair slots than the GPU has CUs"},
    {"oj_args", 1, 0, 1, "any-rank factor: per-edge pointers of small batches in the kernel arguments; 0: through the edge table"},
    {"oj_tol_exp", 8, 4, 15, "any-rank Jacobi stops after a sweep whose pairs were all orthogonal to 10^-x relative"},
    {"oj_max_sweeps", 16, 1, 64, "sweep budget of the any-rank Jacobi"},
    {"comm_force_rccl", 0, 0, 1, "testing: gpet_comm_create builds an RCCL communicator also for a world of one (whose collectives are otherwise plain copies)"},
    {"pchol_multi", 2, 0, 2, "edges wider than 1 024 columns of rank <= 96: pivoted Cholesky over the GPU instead of one workgroup (k_pchol): 2 = blocks of pivots within a tenth of the block's first (k_pcb_block), 1 = one pivot per launch in the greedy order (k_pcx_step)"},
    {"pcx_one_pivot", 0, 0, 1, "1: multi-workgroup pivoted Cholesky one pivot per launch (cross-check of the blocked candidate selection)"},
    {"solve_mw", 1, 0, 1, "blocked fit: alpha by one workgroup per 64-row block and direction (k_chol_solve_mw); 0: one workgroup per edge"},
    {"diag_in_syrk", 1, 0, 1, "blocked fit: the trailing update's first workgroup factors the next diagonal block; 0: a launch of its own"},
    {"topk_rank", 0, 0, 1, "1: argsort of the costs by rank counting (k_topk) also where the bitonic sort applies"},
    {"struct_path", 1, 0, 1, "structured loop path (prior eigenbasis of the pixel grid) where it applies; 0: the generic kernels"},
    {"shared_basis", 1, 0, 1, "edges of one geometry share one prior eigenbasis; 0: every edge its own copy"},
};
constexpr int kCount = (int)(sizeof(kDefs) / sizeof(kDefs[0]));
static_assert(kCount <= kMaxOptions, "OptionSet holds every option");
int g_val[kMaxOptions];
std::once_flag g_once;
std::mutex g_mu;  // writers of the process-wide table and the snapshots of it
thread_local OptionSet* tl_set = nullptr;

int clampv(const OptionDef& d, int v) { return v < d.lo ? d.lo : (v > d.hi ? d.hi : v); }

void init_all() {
  for (int i = 0; i < kCount; ++i) {
    const OptionDef& d = kDefs[i];
    char env[64] = "GPET_";
    size_t n = strlen(env);
    for (const char* p = d.name; *p && n + 1 < sizeof env; ++p) env[n++] = (*p >= 'a' && *p <= 'z') ? (char)(*p - 32) : *p;
    env[n] = 0;
    const char* e = getenv(env);
    g_val[i] = e ? clampv(d, atoi(e)) : d.def;
  }
}

int find(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < kCount; ++i)
    if (strcmp(kDefs[i].name, name) == 0) return i;
  return -1;
}

}  // namespace

int option_index(const char* name) {
  const int i = find(name);
  if (i < 0) {
    fprintf(stderr, "libgpet_hip: unknown option '%s'\n", name ? name : "(null)");
    abort();
  }
  return i;
}

int& option_at(int i) {
  std::call_once(g_once, init_all);
  return tl_set ? tl_set->v[i] : g_val[i];
}

int& option(const char* name) { return option_at(option_index(name)); }

int option_set(const char* name, int value, int* previous) {
  std::call_once(g_once, init_all);
  const int i = find(name);
  if (i < 0) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  if (previous) *previous = g_val[i];
  g_val[i] = clampv(kDefs[i], value);
  return 0;
}

int option_get(const char* name, int* value) {
  std::call_once(g_once, init_all);
  const int i = find(name);
  if (i < 0) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  if (value) *value = g_val[i];
  return 0;
}

void option_snapshot(OptionSet* out) {
  std::call_once(g_once, init_all);
  std::lock_guard<std::mutex> lk(g_mu);
  memcpy(out->v, g_val, sizeof out->v);
}

int option_set_in(OptionSet* s, const char* name, int value, int* previous) {
  const int i = find(name);
  if (i < 0 || !s) return -1;
  if (previous) *previous = s->v[i];
  s->v[i] = clampv(kDefs[i], value);
  return 0;
}

int option_get_in(const OptionSet* s, const char* name, int* value) {
  const int i = find(name);
  if (i < 0 || !s) return -1;
  if (value) *value = s->v[i];
  return 0;
}

OptionScope::OptionScope(OptionSet* s) : prev(tl_set), active(s != nullptr) {
  if (active) tl_set = s;
}
OptionScope::~OptionScope() {
  if (active) tl_set = prev;
}

int option_count() { return kCount; }
const OptionDef& option_def(int i) { return kDefs[i]; }

}  // namespace gpet
