// C ABI, part 5: the device-resident loop (a8).
#include "gpet_api_internal.h"

extern "C" {

int gpet_trace_iterate(gpet_batch* b, const uint32_t* base_seeds, int max_iters, int* n_active) {
  GPET_BATCH_SCOPE(b);
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
  bool fit_used = false;  // the head of a small batch's normals put a launch on the fit stream (this call)
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
      // (refill when fewer than `rng_refill_at` iterations are left in the ring.  Round 5 refilled at look / 2 = 4: four iterations of
      //  a 32-edge batch take 2.4 ms, the sequential launch that refills the ring 3 ms -- every refill stalled the loop, and by how
      //  much depended on when the launch got going: 13.5 or 16.7 ms per loop from one run to the next.  At 6 the launch has a
      //  3.6 ms lead.)
      const int refill_at = option("rng_refill_at") < 0 ? (look > 2 ? look - 2 : look / 2) : option("rng_refill_at");
      if (!rng_inline && deep && b->norm_issued - cur <= refill_at) {
        // small batch: the streams of the next `n` iterations in ONE launch (blockIdx.x = iteration), side by side.
        // Their ring slots were last read by the sample GEMMs of iterations <= cur - 1 (outstanding + n <= ring).
        const int j = b->norm_issued;
        int n = ring - (j - cur);
        if (n > look) n = look;
        if (cur - 1 >= first) HIPCHK(c, hipStreamWaitEvent(b->side, b->ev_gemm[(cur - 1) % 16], 0));
        // The HEAD of a trace of 2..32 edges (round 6).  A stream is sequential -- one workgroup walks 1.27 M MT19937 words in
        // ~3 ms -- and the first iteration of a trace has nothing to hide that behind: the loop of a 32-edge batch stood still
        // for ~2.5 ms before its first sample GEMM.  So the first `head` iterations are generated CHUNKED (jump-ahead: a launch
        // of <= 32 streams is cut into chunks on many workgroups, ~0.7 ms), one launch per iteration on the side stream, while
        // the sequential launch of the iterations after them runs beside them on the (idle until the converged fits) fit stream.
        int head = 0;
        if (j == 0 && cur == 0 && B_l >= 2 && B_l <= 32 && b->rng_mode == 0) {
          head = option("rng_head");
          if (head < 0) head = 4;
          if (head > n - 1) head = n - 1;
          if (head < 0) head = 0;
        }
        for (int q = 0; q < head; ++q) {
          int rcn = normals_auto(b, b->side, edges_l, B_l, seeds_l, 1, j + q, 1, loop_z_store(b));
          if (rcn) return rcn;
          HIPCHK(c, hipEventRecord(b->ev_norm[(j + q) % 16], b->side));
        }
        hipStream_t rest = head > 0 ? b->fit : b->side;
        if (head > 0) HIPCHK(c, hipStreamWaitEvent(b->fit, b->ev_main, 0));  // (the seeds and the edge table are on the device)
        if (n > head) {
          int rcn = normals_auto(b, rest, edges_l, B_l, seeds_l, 1, j + head, n - head, loop_z_store(b), head == 0);
          if (rcn) return rcn;
        }
        for (int q = j + head; q < j + n; ++q) HIPCHK(c, hipEventRecord(b->ev_norm[q % 16], rest));
        // (later refills run on the side stream BESIDE this tail -- different ring slots, and the tail is the sequential kernel,
        //  which has no workspace; the host waits for the fit stream too at the end of the group)
        if (head > 0) fit_used = true;
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
      // loop form: the density stays raw and band-limited in HBM; the pixel kernels normalise on the fly
      const int tail_opt = option("loop_fused_tail");
      if ((tail_opt > 0 || (tail_opt < 0 && b->B <= 64)) && score_tail_applies(b->bd)) {
        HIPCHK(c, launch_score_kde_fused_tail(c->stream, edges_l, B_l, b->bd));  // (small batches: three launches fewer per iteration)
      } else {
        HIPCHK(c, launch_score(c->stream, edges_l, B_l, b->bd));
        HIPCHK(c, launch_kde(c->stream, edges_l, B_l, b->bd, 0, ~0u, 1));
      }
      HIPCHK(c, launch_pixels(c->stream, edges_l, B_l, b->bd, 1));
      HIPCHK(c, hipEventRecord(b->ev_pix[cur % 16], c->stream));
      b->iters_issued += 1;
    }
    b->have_fit = b->have_factor = b->have_normals = b->have_samples = b->have_scores = true;
    HIPCHK(c, gpet_wait(b->side));  // (its launches read the compacted tables too)
    if (fit_used) {
      HIPCHK(c, gpet_wait(b->fit));
      fit_used = false;
    }
    int rc = check_device_status(b);
    if (rc) return rc;
    active = 0;
    for (int e = 0; e < b->B; ++e) active += b->h_scalars[e].done ? 0 : 1;
    flags_known = true;
    *n_active = active;
    remaining -= n_it;
    if (active == 0) break;
    group = active == b->B ? (group < 4 ? group : 4) : 2;
    // Small batches (round 6): a latency chain, where an iteration enqueued for edges that have finished costs its ~16 empty
    // launches (~90 us) and a group boundary a host round trip of about the same -- the fixed 8 / 4 / 2 / 2 ladder spent 0.5 ms
    // of a 6.6 ms single-edge loop on the two.  The observation sets grow at a steady rate (pixel_thresh or a few more per
    // iteration), so the next group is what the slowest running edge still needs at the rate of the group
    // just finished: usually ONE more group that ends on the last iteration.
    if (b->B <= 64 && option("loop_adaptive_groups")) {
      if ((int)b->h_nobs_prev.size() != b->B) b->h_nobs_prev.assign(b->B, 0);
      int need = 1;
      for (int e = 0; e < b->B; ++e) {
        const gpet_scalars& s_ = b->h_scalars[e];
        if (!s_.done) {
          const int got = s_.n_obs - b->h_nobs_prev[e];
          const double rate = got > 0 ? (double)got / (double)n_it : 1.0;
          const int left = b->h_edges[e].algo_thresh - s_.n_obs;
          int est = (int)ceil((double)(left > 0 ? left : 1) / rate);
          if (est > need) need = est;
        }
        b->h_nobs_prev[e] = s_.n_obs;
      }
      group = need < 1 ? 1 : (need > 8 ? 8 : need);
    }
  }
  return GPET_OK;
}

}  // extern "C"
