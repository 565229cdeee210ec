// gjx_smc.hip -- one smc_step (smc.py:731-767) of the localization family in two host calls.
//
// The step needs ONE host decision (lax.cond on ESS < N // 2, smc.py:747-753).  Everything
// before it (extend + two-pass normaliser) and everything after it (normalised weights, CDF,
// bisection, gather, weight reset, K fused moves) is enqueued by one C call each, so that the
// stream is not starved by per-launch host overhead: at N = 2^21 per GPU the Python-driven
// sequence of the same kernels left the GPU idle for ~30 % of the step.
#include "gjx_common.h"

// Key schedule of the K rejuvenation moves (smc.py:756-765 scan; pjax.py:1360-1389): move m runs a
// fresh interpreter on fold_in(scan_key, m).  mh (mcmc.py:267-309): x', y', theta' in model order
// (selected sites only), then u.  hmc (mcmc.py:434-578): momenta in sorted-key order theta, x, y
// (selected only), one split for the leapfrog scan, then u.  out: n_moves x 4 x 2.
GJX_API int32_t gjx_loc_move_keys(const uint32_t scan_key[2], int32_t n_moves, int32_t kind,
                                  int32_t sel_mask, uint32_t* out) {
  GJX_REQUIRE(scan_key && out, GJX_EINVAL, "gjx_loc_move_keys: null argument");
  GJX_REQUIRE(n_moves >= 0 && (kind == 1 || kind == 2) && sel_mask >= 1 && sel_mask <= 7, GJX_EINVAL,
              "gjx_loc_move_keys: bad argument");
  static const int mh_bits[3] = {1, 2, 4}, hmc_bits[3] = {4, 1, 2};
  for (int m = 0; m < n_moves; ++m) {
    uint32_t st[2], nx[2];
    gjx_tf2x32(scan_key[0], scan_key[1], 0u, (uint32_t)m, st[0], st[1]);  // fold_in(scan_key, m)
    uint32_t* o = out + (int64_t)m * 8;
    memset(o, 0, sizeof(uint32_t) * 8);
    auto next = [&](uint32_t* sub) {
      gjx_tf2x32(st[0], st[1], 0u, 1u, sub[0], sub[1]);
      gjx_tf2x32(st[0], st[1], 0u, 0u, nx[0], nx[1]);
      st[0] = nx[0];
      st[1] = nx[1];
    };
    uint32_t dummy[2];
    for (int slot = 0; slot < 3; ++slot)
      if (sel_mask & (kind == 1 ? mh_bits[slot] : hmc_bits[slot])) next(o + 2 * slot);
    if (kind == 2) next(dummy);  // lax.scan over the leapfrog steps: one split, no draw
    next(o + 6);
  }
  return GJX_OK;
}

GJX_API int32_t gjx_loc_smc_extend(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                                   int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                                   const float* d_px, const float* d_py, const float* d_pth,
                                   const float* d_logw_in, float* d_x, float* d_y, float* d_th,
                                   float* d_mx, float* d_my, float* d_mth, float* d_logw,
                                   float* d_stats5, float* d_ray_scores, int64_t scores_stride,
                                   void* stream) {
  GJX_REQUIRE(ctx && d_stats5, GJX_EINVAL, "gjx_loc_smc_extend: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_smc_extend: n must be positive");
  // pass 1 of the normaliser (block maxima) is the extend kernel's epilogue; then the max fold, the
  // tile sums and their fold
  hipStream_t s = (hipStream_t)stream;
  float* cell = ctx->d_scalars + GJX_MAX_CELL;   // -inf between steps; raised by extend, reset by the normaliser
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 0, s));
  int32_t rc = gjx_loc_extend_max(ctx, world, keys, is_initial, offset, n, obs, d_px, d_py, d_pth, d_logw_in,
                                  d_x, d_y, d_th, d_mx, d_my, d_mth, d_logw, d_ray_scores, scores_stride,
                                  nullptr, nullptr, cell, stream);
  if (rc != GJX_OK) return rc;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 1, s));
  rc = gjx_normalise_from_max_partials(ctx, d_logw, n, cell, 1, d_stats5, s);
  if (rc != GJX_OK) return rc;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 2, s));
  return GJX_OK;
}

// The scan and the search of the resampling a step MAY take, enqueued before the host has read the ESS (they need the
// lse, which is on the device): the host's round trip for the lax.cond then runs beside them instead of between the
// normaliser and the scan.  The ancestors stay in the context's workspace; gjx_loc_smc_resample_moves with
// do_resample = 2 uses them, with do_resample = 0 ignores them.
GJX_API int32_t gjx_loc_smc_presearch(gjx_ctx* ctx, const float* d_logw, const float* d_stats5, int64_t n, float u,
                                      void* stream) {
  GJX_REQUIRE(ctx && d_logw && d_stats5, GJX_EINVAL, "gjx_loc_smc_presearch: null argument");
  GJX_REQUIRE(n > 0 && n <= ctx->n_max, GJX_ELIMIT, "gjx_loc_smc_presearch: n=%lld not in (0, n_max=%lld]", (long long)n,
              (long long)ctx->n_max);
  hipStream_t s = (hipStream_t)stream;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 3, s));
  int32_t rc = gjx_cumsum_normalized_f32(ctx, d_logw, d_stats5 + 1, n, ctx->d_tmp_cdf, stream);
  if (rc != GJX_OK) return rc;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 4, s));
  rc = gjx_searchsorted_systematic(ctx->d_tmp_cdf, n, u, 0, n, n, ctx->d_tmp_anc, stream);
  if (rc != GJX_OK) return rc;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 5, s));
  return GJX_OK;
}

GJX_API int32_t gjx_loc_smc_resample_moves(gjx_ctx* ctx, const gjx_loc_world* world, int32_t do_resample,
                                           float u, const float* d_stats5, int32_t kind,
                                           const uint32_t (*move_keys)[4][2], int32_t n_moves,
                                           int32_t sel_mask, float step_size, int32_t n_steps,
                                           int32_t is_initial, uint64_t offset, int64_t n,
                                           const float* obs, float* const* d_cols, float* const* d_cols_alt,
                                           float* d_logw, const float* d_ray_scores,
                                           int64_t scores_stride, int64_t* d_accept_counts,
                                           void* ev_moves_begin,
                                           void* ev_moves_end, void* stream) {
  GJX_REQUIRE(ctx && d_cols && d_logw, GJX_EINVAL, "gjx_loc_smc_resample_moves: null argument");
  GJX_REQUIRE(n > 0 && n <= ctx->n_max, GJX_ELIMIT,
              "gjx_loc_smc_resample_moves: n=%lld not in (0, n_max=%lld]", (long long)n,
              (long long)ctx->n_max);
  GJX_REQUIRE(kind >= 0 && kind <= 2, GJX_EINVAL, "gjx_loc_smc_resample_moves: kind must be 0, 1 or 2");
  hipStream_t s = (hipStream_t)stream;
  float* const* cols = d_cols;
  if (do_resample) {
    GJX_REQUIRE(d_stats5 && d_cols_alt, GJX_EINVAL,
                "gjx_loc_smc_resample_moves: resampling needs the statistics and the alternate columns");
    int32_t rc = GJX_OK;
    if (do_resample != 2) {   // 2: gjx_loc_smc_presearch has left the ancestors (and the stage marks 3..5)
      rc = gjx_loc_smc_presearch(ctx, d_logw, d_stats5, n, u, stream);
      if (rc != GJX_OK) return rc;
    }
    cols = d_cols_alt;
    const float* src[6];
    for (int c = 0; c < 6; ++c) src[c] = d_cols[c];
    if (kind == 1 && n_moves > 0) {
      // mh: the gather rides on the move kernel's loads (VALU-bound kernel, the extra traffic is free)
      GJX_REQUIRE(move_keys != nullptr, GJX_EINVAL, "gjx_loc_smc_resample_moves: move keys are null");
      if (ev_moves_begin) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_moves_begin, s));
      rc = gjx_loc_mh_resampled(ctx, world, move_keys, n_moves, sel_mask, is_initial, offset, n, obs,
                                ctx->d_tmp_anc, 0, src, d_cols_alt, d_ray_scores, scores_stride, d_logw,
                                d_accept_counts, stream);  // also sets log_weights := 0 (smc.py:646)
      if (rc != GJX_OK) return rc;
      if (ev_moves_end) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_moves_end, s));
      GJX_CHECK_HIP(gjx_stage_mark(ctx, 6, s));
      return GJX_OK;
    }
    rc = gjx_gather_columns_f32(ctx->d_tmp_anc, 0, n, src, d_cols_alt, 6, stream);
    if (rc != GJX_OK) return rc;
    GJX_CHECK_HIP(hipMemsetAsync(d_logw, 0, sizeof(float) * n, s));  // log_weights := 0 (smc.py:646)
  }
  if (!do_resample)
    for (int k = 3; k <= 5; ++k) GJX_CHECK_HIP(gjx_stage_mark(ctx, k, s));   // empty stages: zero-length intervals
  if (kind == 0 || n_moves == 0) {
    GJX_CHECK_HIP(gjx_stage_mark(ctx, 6, s));
    return GJX_OK;
  }
  GJX_REQUIRE(move_keys != nullptr, GJX_EINVAL, "gjx_loc_smc_resample_moves: move keys are null");
  // (with d_accept_counts the timed interval also holds the 8-byte-per-move memset of the counters)
  if (ev_moves_begin) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_moves_begin, s));
  int32_t rc;
  if (kind == 1 && d_ray_scores && !do_resample) {
    const float* src[6];
    for (int c = 0; c < 6; ++c) src[c] = cols[c];
    rc = gjx_loc_mh_resampled(ctx, world, move_keys, n_moves, sel_mask, is_initial, offset, n, obs, nullptr, 0,
                              src, cols, d_ray_scores, scores_stride, nullptr, d_accept_counts, stream);
  } else if (kind == 1)
    rc = gjx_loc_mh(ctx, world, move_keys, n_moves, sel_mask, is_initial, offset, n, obs, cols[0], cols[1],
                    cols[2], cols[3], cols[4], cols[5], cols[0], cols[1], cols[2], d_accept_counts, stream);
  else
    rc = gjx_loc_hmc(ctx, world, move_keys, n_moves, sel_mask, step_size, n_steps, is_initial, offset, n, obs,
                     cols[0], cols[1], cols[2], cols[3], cols[4], cols[5], cols[0], cols[1], cols[2],
                     d_accept_counts, stream);
  if (rc != GJX_OK) return rc;
  if (ev_moves_end) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_moves_end, s));
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 6, s));
  return GJX_OK;
}

// The whole step in ONE call, with the lax.cond taken on the device (kernel = mh, one GPU): extend, normaliser whose
// fold kernel decides ESS < n / 2 (d_decision[2]; d_decision = {1, n, take, flags}, the caller sets [0], [1] once and
// zeroes [3]; d_range = {0, take ? n : 0}; *d_lml += lse - log n when taken;
// d_trace2 = {ess, take}), the scan and the search that return at once when the decision is "no", and the move
// launch that reads the decision (resampled from ctx->d_tmp_anc with weights := 0, or every particle in place with
// weights kept).  Output always in d_cols_alt (the caller flips).  ev_stats (optional) is recorded after the
// normaliser, so that the caller can fetch d_stats5 on another stream while the moves run: the host never stands
// between two kernels of a step.  d_decision[3] |= 2 if a position disagrees with the decision's range.
GJX_API int32_t gjx_loc_smc_step_auto(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                                      int32_t is_initial, int64_t n, const float* obs, float u,
                                      const uint32_t (*move_keys)[4][2], int32_t n_moves, int32_t sel_mask,
                                      float* const* d_cols, float* const* d_cols_alt, float* d_logw, float* d_stats5,
                                      float* d_ray_scores, int64_t scores_stride, int32_t* d_decision,
                                      int64_t* d_range, float* d_lml, float* d_trace2, int64_t* d_accept_counts,
                                      void* ev_stats, void* ev_moves_begin, void* ev_moves_end, void* stream) {
  GJX_REQUIRE(ctx && world && d_cols && d_cols_alt && d_logw && d_stats5 && d_decision && d_range && d_lml &&
                  d_trace2 && move_keys,
              GJX_EINVAL, "gjx_loc_smc_step_auto: null argument");
  GJX_REQUIRE(n > 0 && n <= ctx->n_max && n_moves > 0, GJX_ELIMIT,
              "gjx_loc_smc_step_auto: n=%lld not in (0, n_max=%lld], or no moves", (long long)n, (long long)ctx->n_max);
  hipStream_t s = (hipStream_t)stream;
  float* cell = ctx->d_scalars + GJX_MAX_CELL;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 0, s));
  int32_t rc = gjx_loc_extend_max(ctx, world, keys, is_initial, 0, n, obs, is_initial ? nullptr : d_cols[0],
                                  is_initial ? nullptr : d_cols[1], is_initial ? nullptr : d_cols[2], d_logw, d_cols[0],
                                  d_cols[1], d_cols[2], d_cols[3], d_cols[4], d_cols[5], d_logw, d_ray_scores,
                                  scores_stride, nullptr, nullptr, cell, stream);
  if (rc != GJX_OK) return rc;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 1, s));
  GjxDecide dec;
  dec.go = d_decision + 2;   // d_decision = {1, n, take, flags}: the move launch's view of a one-shard "split"
  dec.range = d_range;
  dec.n = n;
  dec.half_n = n / 2;
  dec.log_n = (float)log((double)n);
  dec.lml = d_lml;
  dec.trace2 = d_trace2;
  rc = gjx_normalise_decide(ctx, d_logw, n, cell, 1, d_stats5, dec, s);
  if (rc != GJX_OK) return rc;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 2, s));
  if (ev_stats) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_stats, s));
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 3, s));   // no host decision: a zero-length stage
  rc = gjx_resample_systematic_logw_if(ctx, d_logw, d_stats5 + 1, n, u, d_decision + 2, d_range, ctx->d_tmp_anc, s);
  if (rc != GJX_OK) return rc;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 5, s));
  if (d_accept_counts) GJX_CHECK_HIP(hipMemsetAsync(d_accept_counts, 0, sizeof(int64_t) * n_moves, s));
  const float* src[6];
  for (int c = 0; c < 6; ++c) src[c] = d_cols[c];
  if (ev_moves_begin) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_moves_begin, s));
  rc = gjx_loc_mh_overlap(ctx, world, move_keys, n_moves, sel_mask, is_initial, 0, n, obs, nullptr, src, d_cols_alt,
                          d_ray_scores, scores_stride, d_logw, d_accept_counts, d_decision, ctx->d_tmp_anc, d_range, 0, -1,
                          stream);
  if (rc != GJX_OK) return rc;
  if (ev_moves_end) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_moves_end, s));
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 6, s));
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// The same step with the particle axis sharded over ranks (genjax_amd/parallel.py runs the
// collectives between the calls; kernels of the resampling in gjx_resample.hip).
// ---------------------------------------------------------------------------
// extend + the maximum of THIS shard's new log-weights (-> all_gather, gjx_shard_tile_sums)
GJX_API int32_t gjx_loc_shard_extend(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                                     int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                                     const float* d_px, const float* d_py, const float* d_pth,
                                     const float* d_logw_in, float* d_x, float* d_y, float* d_th, float* d_mx,
                                     float* d_my, float* d_mth, float* d_logw, float* d_max1,
                                     float* d_ray_scores, int64_t scores_stride, void* stream) {
  GJX_REQUIRE(ctx && d_max1, GJX_EINVAL, "gjx_loc_shard_extend: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_shard_extend: n must be positive");
  hipStream_t s = (hipStream_t)stream;
  int n_partials = 0;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 0, s));
  int32_t rc = gjx_loc_extend_max(ctx, world, keys, is_initial, offset, n, obs, d_px, d_py, d_pth, d_logw_in,
                                  d_x, d_y, d_th, d_mx, d_my, d_mth, d_logw, d_ray_scores, scores_stride,
                                  ctx->d_partials, &n_partials, nullptr, stream);
  if (rc != GJX_OK) return rc;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 1, s));
  return gjx_fold_max_partials(ctx->d_partials, n_partials, d_max1, s);
}

int32_t gjx_loc_shard_extend_cell(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2], int32_t is_initial,
                                  uint64_t offset, int64_t n, const float* obs, const float* d_px, const float* d_py,
                                  const float* d_pth, const float* d_logw_in, float* d_x, float* d_y, float* d_th,
                                  float* d_mx, float* d_my, float* d_mth, float* d_logw, float* d_max1,
                                  float* d_ray_scores, int64_t scores_stride, void* stream) {
  GJX_REQUIRE(ctx && d_max1, GJX_EINVAL, "gjx_loc_shard_extend: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_shard_extend: n must be positive");
  hipStream_t s = (hipStream_t)stream;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 0, s));
  int32_t rc = gjx_loc_extend_max(ctx, world, keys, is_initial, offset, n, obs, d_px, d_py, d_pth, d_logw_in, d_x, d_y,
                                  d_th, d_mx, d_my, d_mth, d_logw, d_ray_scores, scores_stride, nullptr, nullptr, d_max1,
                                  stream);
  if (rc != GJX_OK) return rc;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 1, s));
  return GJX_OK;
}

// After the all_to_all of the rows: received rows into the halo [n, n + n_recv) of the current columns
// (and their per-ray scores recomputed there: same device function, same bits as the owner's stored ones),
// local ancestor indices, then exactly what gjx_loc_smc_resample_moves does after its search.
GJX_API int32_t gjx_shard_resample_moves(gjx_ctx* ctx, const gjx_loc_world* world, const int32_t* d_cand_idx,
                                         const int64_t* d_range, const float* d_recv_rows, int64_t n_recv,
                                         int32_t rank, int64_t n, int64_t halo_cap, int32_t* d_anc,
                                         int32_t* d_err, int32_t kind, const uint32_t (*move_keys)[4][2],
                                         int32_t n_moves, int32_t sel_mask, float step_size, int32_t n_steps,
                                         int32_t is_initial, const float* obs, float* const* d_cols,
                                         float* const* d_cols_alt, float* d_logw, float* d_ray_scores,
                                         int64_t scores_stride, int64_t* d_accept_counts, void* ev_moves_begin,
                                         void* ev_moves_end, void* stream) {
  GJX_REQUIRE(ctx && world && d_cols && d_cols_alt && d_logw && d_anc, GJX_EINVAL,
              "gjx_shard_resample_moves: null argument");
  GJX_REQUIRE(n > 0 && n <= ctx->n_max, GJX_ELIMIT, "gjx_shard_resample_moves: n=%lld not in (0, n_max=%lld]",
              (long long)n, (long long)ctx->n_max);
  GJX_REQUIRE(kind >= 0 && kind <= 2 && rank >= 0, GJX_EINVAL, "gjx_shard_resample_moves: bad kind or rank");
  GJX_REQUIRE(d_ray_scores == nullptr || scores_stride >= n + halo_cap, GJX_EINVAL,
              "gjx_shard_resample_moves: the score rows must hold the halo too (stride %lld < %lld)",
              (long long)scores_stride, (long long)(n + halo_cap));
  hipStream_t s = (hipStream_t)stream;
  const uint64_t offset = (uint64_t)rank * (uint64_t)n;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 5, s));
  int32_t rc = gjx_shard_expand(d_cand_idx, d_range, d_recv_rows, n_recv, rank, n, d_cols, halo_cap, d_anc, d_err, s);
  if (rc != GJX_OK) return rc;
  const float* src[6];
  for (int c = 0; c < 6; ++c) src[c] = d_cols[c];
  if (kind == 1 && n_moves > 0) {
    GJX_REQUIRE(move_keys != nullptr, GJX_EINVAL, "gjx_shard_resample_moves: move keys are null");
    if (d_ray_scores && n_recv > 0) {
      rc = gjx_loc_ray_scores_strided(world, obs, d_cols[0] + n, d_cols[1] + n, d_cols[2] + n, n_recv,
                                      d_ray_scores + n, scores_stride, stream);
      if (rc != GJX_OK) return rc;
    }
    if (ev_moves_begin) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_moves_begin, s));
    rc = gjx_loc_mh_resampled(ctx, world, move_keys, n_moves, sel_mask, is_initial, offset, n, obs, d_anc, 0, src,
                              d_cols_alt, d_ray_scores, scores_stride, d_logw, d_accept_counts, stream);
    if (rc != GJX_OK) return rc;
    if (ev_moves_end) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_moves_end, s));
    GJX_CHECK_HIP(gjx_stage_mark(ctx, 6, s));
    return GJX_OK;
  }
  rc = gjx_gather_columns_f32(d_anc, 0, n, src, d_cols_alt, 6, stream);
  if (rc != GJX_OK) return rc;
  GJX_CHECK_HIP(hipMemsetAsync(d_logw, 0, sizeof(float) * n, s));   // log_weights := 0 (smc.py:646)
  if (kind == 0 || n_moves == 0) {
    GJX_CHECK_HIP(gjx_stage_mark(ctx, 6, s));
    return GJX_OK;
  }
  GJX_REQUIRE(move_keys != nullptr, GJX_EINVAL, "gjx_shard_resample_moves: move keys are null");
  float* const* c = d_cols_alt;
  if (ev_moves_begin) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_moves_begin, s));
  if (kind == 1)
    rc = gjx_loc_mh(ctx, world, move_keys, n_moves, sel_mask, is_initial, offset, n, obs, c[0], c[1], c[2], c[3], c[4],
                    c[5], c[0], c[1], c[2], d_accept_counts, stream);
  else
    rc = gjx_loc_hmc(ctx, world, move_keys, n_moves, sel_mask, step_size, n_steps, is_initial, offset, n, obs, c[0],
                     c[1], c[2], c[3], c[4], c[5], c[0], c[1], c[2], d_accept_counts, stream);
  if (rc != GJX_OK) return rc;
  if (ev_moves_end) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_moves_end, s));
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 6, s));
  return GJX_OK;
}

// The resampling + mh moves of the sharded step without the host between the search and the moves.
// gjx_shard_split (gjx_resample.hip) has left the decision and the own-sourced run in d_split; this launch moves
// every position whose ancestor (read from the search results) is an own particle -- all positions, as their own
// ancestors and with their weights kept, when the decision is "no" -- into the alternate columns.  The caller
// flips the column sets either way.  d_split[3] |= 2 if a position disagrees with the run the host is told.
GJX_API int32_t gjx_shard_moves_main(gjx_ctx* ctx, const gjx_loc_world* world, const int32_t* d_cand_idx,
                                     const int64_t* d_range, int32_t rank, int64_t n, int32_t* d_split,
                                     const uint32_t (*move_keys)[4][2], int32_t n_moves,
                                     int32_t sel_mask, int32_t is_initial, const float* obs, float* const* d_cols,
                                     float* const* d_cols_alt, float* d_logw, float* d_ray_scores,
                                     int64_t scores_stride, int64_t* d_accept_counts, void* ev_moves_begin,
                                     void* ev_moves_end, void* stream) {
  GJX_REQUIRE(ctx && world && d_cols && d_cols_alt && d_logw && d_cand_idx && d_range && d_split && move_keys,
              GJX_EINVAL, "gjx_shard_moves_main: null argument");
  GJX_REQUIRE(n > 0 && n <= ctx->n_max && rank >= 0 && n_moves > 0, GJX_ELIMIT,
              "gjx_shard_moves_main: n=%lld not in (0, n_max=%lld], or no moves", (long long)n, (long long)ctx->n_max);
  hipStream_t s = (hipStream_t)stream;
  const float* src[6];
  for (int c = 0; c < 6; ++c) src[c] = d_cols[c];
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 5, s));
  if (ev_moves_begin) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_moves_begin, s));
  int32_t rc = gjx_loc_mh_overlap(ctx, world, move_keys, n_moves, sel_mask, is_initial, (uint64_t)rank * (uint64_t)n, n,
                                  obs, nullptr, src, d_cols_alt, d_ray_scores, scores_stride, d_logw, d_accept_counts,
                                  d_split, d_cand_idx, d_range, 0, -1, stream);
  if (rc != GJX_OK) return rc;
  if (ev_moves_end) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_moves_end, s));
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 6, s));
  return GJX_OK;
}

// The rest of the same step once the rows have arrived (any stream: it touches the halo, the positions
// [0, p0) and [p1, n) of the alternate columns and of the weights, and d_anc_fix -- nothing the main launch
// writes or reads besides the read-only current columns): rows -> halo, their per-ray scores, ancestors, moves.
GJX_API int32_t gjx_shard_moves_fixup(gjx_ctx* ctx, const gjx_loc_world* world, const float* d_recv_rows,
                                      int64_t n_recv, int32_t rank, int64_t n, int64_t halo_cap, int64_t p0,
                                      int64_t p1, int32_t* d_anc_fix, int32_t* d_err,
                                      const uint32_t (*move_keys)[4][2], int32_t n_moves, int32_t sel_mask,
                                      int32_t is_initial, const float* obs, float* const* d_cols,
                                      float* const* d_cols_alt, float* d_logw, float* d_ray_scores,
                                      int64_t scores_stride, int64_t* d_accept_counts, void* stream) {
  GJX_REQUIRE(ctx && world && d_cols && d_cols_alt && d_logw && d_anc_fix && d_err && move_keys, GJX_EINVAL,
              "gjx_shard_moves_fixup: null argument");
  GJX_REQUIRE(n > 0 && n <= ctx->n_max && rank >= 0 && n_moves > 0, GJX_ELIMIT,
              "gjx_shard_moves_fixup: n=%lld not in (0, n_max=%lld], or no moves", (long long)n, (long long)ctx->n_max);
  GJX_REQUIRE(d_ray_scores == nullptr || scores_stride >= n + halo_cap, GJX_EINVAL,
              "gjx_shard_moves_fixup: the score rows must hold the halo too (stride %lld < %lld)",
              (long long)scores_stride, (long long)(n + halo_cap));
  hipStream_t s = (hipStream_t)stream;
  int32_t rc = gjx_shard_expand_fix(d_recv_rows, n_recv, rank, n, p0, p1, d_cols, halo_cap, d_anc_fix, d_err, s);
  if (rc != GJX_OK) return rc;
  const int64_t n_fix = p0 + (n - p1);
  if (n_fix == 0) return GJX_OK;
  GJX_REQUIRE(n_recv > 0, GJX_EINVAL, "gjx_shard_moves_fixup: %lld positions wait for rows, none arrived",
              (long long)n_fix);
  if (d_ray_scores) {
    rc = gjx_loc_ray_scores_strided(world, obs, d_cols[0] + n, d_cols[1] + n, d_cols[2] + n, n_recv, d_ray_scores + n,
                                    scores_stride, stream);
    if (rc != GJX_OK) return rc;
  }
  const float* src[6];
  for (int c = 0; c < 6; ++c) src[c] = d_cols[c];
  return gjx_loc_mh_overlap(ctx, world, move_keys, n_moves, sel_mask, is_initial, (uint64_t)rank * (uint64_t)n, n_fix,
                            obs, d_anc_fix, src, d_cols_alt, d_ray_scores, scores_stride, d_logw, d_accept_counts,
                            nullptr, nullptr, nullptr, p0, p1, stream);
}

// Stage timing of the two entry points above: n = 7 events (gjx_event_create) recorded at the stage
// boundaries of every following call on this context, n = 0 switches it off.
GJX_API int32_t gjx_ctx_set_stage_events(gjx_ctx* ctx, void* const* events, int32_t n) {
  GJX_REQUIRE(ctx != nullptr, GJX_EINVAL, "gjx_ctx_set_stage_events: ctx is null");
  GJX_REQUIRE(n == 0 || (n == GJX_N_STAGE_EVENTS && events), GJX_EINVAL,
              "gjx_ctx_set_stage_events: pass %d events or none", GJX_N_STAGE_EVENTS);
  for (int k = 0; k < n; ++k) {
    GJX_REQUIRE(events[k] != nullptr, GJX_EINVAL, "gjx_ctx_set_stage_events: event %d is null", k);
    ctx->stage_ev[k] = (hipEvent_t)events[k];
  }
  ctx->n_stage_ev = n;
  return GJX_OK;
}

// HIP events for callers that time a launch enqueued inside one of the fused entry points
// (bench.py's roofline: the events are recorded on the launch stream around the move kernel).
GJX_API int32_t gjx_event_create(void** out) {
  GJX_REQUIRE(out != nullptr, GJX_EINVAL, "gjx_event_create: out is null");
  hipEvent_t e;
  GJX_CHECK_HIP(hipEventCreate(&e));
  *out = (void*)e;
  return GJX_OK;
}

GJX_API int32_t gjx_event_record(void* ev, void* stream) {
  GJX_REQUIRE(ev != nullptr, GJX_EINVAL, "gjx_event_record: event is null");
  GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return GJX_OK;
}

GJX_API int32_t gjx_event_destroy(void* ev) {
  if (ev) GJX_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
  return GJX_OK;
}

// One wave that keeps its stream busy for about `microseconds` (wall clock, the 100 MHz constant counter) and
// touches no memory: the set-up probe of the sharded driver asks whether a second stream runs BESIDE the stream
// this spins on (genjax_amd/parallel.py).  The loop ends on the clock or after a fixed number of iterations,
// whichever comes first, so the wave always retires.
__global__ void k_spin(long long ticks, int max_iter) {
  const long long t0 = wall_clock64();
  for (int it = 0; it < max_iter; ++it) {
    if (wall_clock64() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(32);
  }
}

GJX_API int32_t gjx_spin_us(int32_t microseconds, void* stream) {
  GJX_REQUIRE(microseconds >= 0 && microseconds <= 100000, GJX_ELIMIT, "gjx_spin_us: 0..100000 microseconds, got %d",
              microseconds);
  if (microseconds == 0) return GJX_OK;
  // s_sleep 32 = 2048 cycles ~ 1 us at the shader clock: 4 iterations per microsecond is a generous cap
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100LL,
                     microseconds * 4 + 64);
  GJX_CHECK_HIP(hipGetLastError());
  return GJX_OK;
}

GJX_API int32_t gjx_event_elapsed_ms(void* begin, void* end, float* ms) {
  GJX_REQUIRE(begin && end && ms, GJX_EINVAL, "gjx_event_elapsed_ms: null argument");
  GJX_CHECK_HIP(hipEventSynchronize((hipEvent_t)end));
  GJX_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)begin, (hipEvent_t)end));
  return GJX_OK;
}
