// C ABI, part 3: the per-stage entry points (a2-a7, f1) and gpet_profile_stage.
#include "gpet_api_internal.h"

extern "C" {


// ---- stages ---------------------------------------------------------------------------
int gpet_gp_fit_predict(gpet_batch* b, int want_cov) {
  GPET_BATCH_SCOPE(b);
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
  GPET_BATCH_SCOPE(b);
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
  GPET_BATCH_SCOPE(b);
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
  GPET_BATCH_SCOPE(b);
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
  GPET_BATCH_SCOPE(b);
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
  GPET_BATCH_SCOPE(b);
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
  GPET_BATCH_SCOPE(b);
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
  GPET_BATCH_SCOPE(b);
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

int gpet_profile_stage(gpet_batch* b, int stage, int reps, float* ms_per_rep) {
  GPET_BATCH_SCOPE(b);
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

int gpet_select_pixels_only(gpet_batch* b) {
  GPET_BATCH_SCOPE(b);
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

}  // extern "C"
