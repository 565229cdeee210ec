// gjx_models.hip -- further fused model families.
//
// gjx_poly_is: init(npoint_curve, (xs,), n, {"ys": {"obs": ys}}) of examples/curvefit
// (core.py:57-100,114-136; smc.py:272-317): a, b, c ~ N(0, 1) drawn in program order from three
// sub-keys, logw = sum_j log N(ys_j; a + b x_j + c x_j^2, sigma).  16 B written per particle.
#include "gjx_common.h"

struct PolyK {
  int P;
  float inv_sigma, lp_const;
  float xs[GJX_MAX_POINTS], ys_s[GJX_MAX_POINTS];
  uint32_t k[3][2];
};

__global__ void __launch_bounds__(GJX_BLOCK)
    k_poly_is(PolyK p, uint64_t offset, int64_t n, float* __restrict__ a, float* __restrict__ b,
              float* __restrict__ c, float* __restrict__ logw) {
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * GJX_BLOCK) {
    const uint64_t gi = offset + (uint64_t)i;
    // normal(0, 1): sample * 1 + 0 is the identity
    const float va = gjx_normal_at(p.k[0][0], p.k[0][1], gi);
    const float vb = gjx_normal_at(p.k[1][0], p.k[1][1], gi);
    const float vc = gjx_normal_at(p.k[2][0], p.k[2][1], gi);
    float s = 0.0f;
    for (int j = 0; j < p.P; ++j) {
      const float xj = p.xs[j];
      // polyfn: a + b*x + c*x**2  (core.py:72-75)
      const float mean = gjx_add_rn(gjx_add_rn(va, gjx_mul_rn(vb, xj)), gjx_mul_rn(vc, gjx_mul_rn(xj, xj)));
      const float z = p.ys_s[j] - mean * p.inv_sigma;
      s += -0.5f * (z * z) - p.lp_const;
    }
    a[i] = va;
    b[i] = vb;
    c[i] = vc;
    logw[i] = s;
  }
}

GJX_API int32_t gjx_poly_is(const uint32_t keys[3][2], uint64_t offset, int64_t n, const float* xs,
                            const float* ys, int32_t n_points, float sigma, float* d_a, float* d_b,
                            float* d_c, float* d_logw, void* stream) {
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(keys && xs && ys && d_a && d_b && d_c && d_logw, GJX_EINVAL,
              "gjx_poly_is: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_poly_is: n must not be negative");
  GJX_REQUIRE(n_points >= 1 && n_points <= GJX_MAX_POINTS, GJX_ELIMIT,
              "gjx_poly_is: n_points=%d not in [1, %d]", n_points, GJX_MAX_POINTS);
  GJX_REQUIRE(sigma > 0.0f, GJX_EINVAL, "gjx_poly_is: sigma must be positive");
  PolyK p;
  memset(&p, 0, sizeof(p));
  p.P = n_points;
  p.inv_sigma = 1.0f / sigma;
  p.lp_const = HALF_LOG_2PI_F + logf(sigma);
  for (int j = 0; j < n_points; ++j) {
    p.xs[j] = xs[j];
    p.ys_s[j] = ys[j] / sigma;
  }
  memcpy(p.k, keys, sizeof(p.k));
  k_poly_is<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(p, offset, n, d_a, d_b, d_c,
                                                                     d_logw);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// linear-Gaussian state-space model (reference src/genjax/extras/state_space.py:377-431)
//   state ~ MVN(is_initial ? mu0 : A prev, is_initial ? Sigma0 : Q)   @ "state"
//   obs   ~ MVN(C state, R)                                           @ "obs" (constrained)
// bootstrap-PF extend: draw the state column from ONE key (MultivariateNormalFullCovariance =
// mean + chol(cov) @ normal(key, (n, d)), flat index i*d + r) and add the observation log-density
// to the log-weights.  Cholesky factors are uniform and computed once on the host.
// State layout: SoA, d_state rows of n floats.
// ---------------------------------------------------------------------------
#define GJX_LG_MAXD 4

struct LgK {
  int ds, dobs, is_initial;
  float mu0[GJX_LG_MAXD];
  float Lx[GJX_LG_MAXD * GJX_LG_MAXD];   // chol(Sigma0) or chol(Q) for THIS step
  float A[GJX_LG_MAXD * GJX_LG_MAXD];
  float C[GJX_LG_MAXD * GJX_LG_MAXD];
  float LR[GJX_LG_MAXD * GJX_LG_MAXD];   // chol(R)
  float obs[GJX_LG_MAXD];
  float lognorm;                          // sum log LR_ii + dobs/2 log 2pi
  uint32_t key[4][2];
};

// DS, DO: the state / observation dimensions as template parameters, so that the small per-particle vectors
// live in registers (with run-time bounds they are indexed dynamically and end up in scratch memory:
// k_lgssm_extend took 13 us for 2^20 particles that way).
template <int DS, int DO>
__device__ __forceinline__ float lg_obs_logpdf(const LgK& p, const float* x) {
  float z[DO], q = 0.0f;
#pragma unroll
  for (int r = 0; r < DO; ++r) {
    float m = 0.0f;
#pragma unroll
    for (int c = 0; c < DS; ++c) m += p.C[r * DS + c] * x[c];
    float s = p.obs[r] - m;
#pragma unroll
    for (int k = 0; k < r; ++k) s -= p.LR[r * DO + k] * z[k];
    z[r] = s / p.LR[r * DO + r];
    q += z[r] * z[r];
  }
  return -0.5f * q - p.lognorm;
}

template <int DS>
__device__ __forceinline__ void lg_draw_state(const LgK& p, const uint32_t (&key)[2], uint64_t gi,
                                              const float* mean, float* x) {
  float z[DS];
#pragma unroll
  for (int r = 0; r < DS; ++r) z[r] = gjx_normal_at(key[0], key[1], gi * (uint64_t)DS + (uint64_t)r);
#pragma unroll
  for (int r = 0; r < DS; ++r) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c <= r; ++c) s += p.Lx[r * DS + c] * z[c];
    x[r] = mean[r] + s;
  }
}

template <int DS>
__device__ __forceinline__ void lg_mean(const LgK& p, const float* prev, int64_t i, int64_t n,
                                        float* mean) {
  if (p.is_initial) {
#pragma unroll
    for (int r = 0; r < DS; ++r) mean[r] = p.mu0[r];
  } else {
    float pv[DS];
#pragma unroll
    for (int c = 0; c < DS; ++c) pv[c] = prev[(int64_t)c * n + i];
#pragma unroll
    for (int r = 0; r < DS; ++r) {
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < DS; ++c) s += p.A[r * DS + c] * pv[c];
      mean[r] = s;
    }
  }
}

// extend.  prev may be `state` itself (in place: particle i reads only its own previous state) unless go_prev is
// given: then the PREVIOUS step's resampling rides on this kernel's loads -- particle i continues from ancestor
// anc_prev[i] of the other column set with log-weight 0 when *go_prev, from itself otherwise
// (resample_vectorized_trace + log_weights := 0, smc.py:139-175, 646, without a gather pass of their own).
// MAXP: the block maxima of the new log-weights, pass 1 of the normaliser.  (One atomic max per block into a single
// word was measured instead: 2048 same-address atomics serialise in L2 and took the kernel from 11 to 29 us.)
template <int DS, int DO, bool MAXP>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_lgssm_extend(LgK p, uint64_t offset, int64_t n, const float* prev, const float* logw_in,
                   float* state, float* mean_out, float* logw, float* __restrict__ max_partials,
                   const int32_t* __restrict__ go_prev, const int32_t* __restrict__ anc_prev) {
  const bool via = go_prev != nullptr && go_prev[0] != 0;
  float m = -__builtin_inff();
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * GJX_BLOCK) {
    float mean[DS], x[DS];
    lg_mean<DS>(p, prev, via ? (int64_t)anc_prev[i] : i, n, mean);
    const float lw = via ? 0.0f : logw_in[i];
    lg_draw_state<DS>(p, p.key[0], offset + (uint64_t)i, mean, x);
#pragma unroll
    for (int r = 0; r < DS; ++r) {
      state[(int64_t)r * n + i] = x[r];
      mean_out[(int64_t)r * n + i] = mean[r];
    }
    const float out = lw + lg_obs_logpdf<DS, DO>(p, x);
    logw[i] = out;
    m = fmaxf(m, out);
  }
  if (MAXP) {
    __shared__ float sm[GJX_BLOCK / 64];
    m = gjx_wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) max_partials[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  }
}

#define LG_DISPATCH_DO(DSV, dobs, ...)                              \
  switch (dobs) {                                                   \
    case 1: { constexpr int DS = DSV, DO = 1; __VA_ARGS__; } break; \
    case 2: { constexpr int DS = DSV, DO = 2; __VA_ARGS__; } break; \
    case 3: { constexpr int DS = DSV, DO = 3; __VA_ARGS__; } break; \
    default: { constexpr int DS = DSV, DO = 4; __VA_ARGS__; } break; \
  }
#define LG_DISPATCH(ds, dobs, ...)                    \
  switch (ds) {                                       \
    case 1: LG_DISPATCH_DO(1, dobs, __VA_ARGS__) break; \
    case 2: LG_DISPATCH_DO(2, dobs, __VA_ARGS__) break; \
    case 3: LG_DISPATCH_DO(3, dobs, __VA_ARGS__) break; \
    default: LG_DISPATCH_DO(4, dobs, __VA_ARGS__) break; \
  }

// n_moves fused mh(trace, sel("state")) moves: fresh prior draw at the stored mean, weight =
// logpdf_obs(new) - logpdf_obs(old), accept iff log u < min(0, w).  keys[m] = {state', u}.
struct LgMoveKeys {
  uint32_t k[GJX_MAX_MOVES][2][2];
};

template <int DS, int DO>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_lgssm_mh(LgK p, LgMoveKeys keys, int n_moves, uint64_t offset, int64_t n, const float* state_in,
               const float* mean_in, float* state, int64_t* __restrict__ accept_counts) {
  __shared__ uint32_t counts[GJX_MAX_MOVES];
  if (threadIdx.x < GJX_MAX_MOVES) counts[threadIdx.x] = 0u;
  __syncthreads();
  const int64_t nthreads = (int64_t)gridDim.x * GJX_BLOCK;
  const int64_t n_round = ((n + nthreads - 1) / nthreads) * nthreads;
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n_round; i += nthreads) {
    const bool live = i < n;
    const int64_t ii = live ? i : 0;
    float cur[DS], mean[DS], prop[DS];
#pragma unroll
    for (int r = 0; r < DS; ++r) {
      cur[r] = state_in[(int64_t)r * n + ii];
      mean[r] = mean_in[(int64_t)r * n + ii];
    }
    float lp_old = lg_obs_logpdf<DS, DO>(p, cur);
    const uint64_t gi = offset + (uint64_t)ii;
    for (int m = 0; m < n_moves; ++m) {
      lg_draw_state<DS>(p, keys.k[m][0], gi, mean, prop);
      const float u = gjx_uniform_at(keys.k[m][1][0], keys.k[m][1][1], gi, 0.0f, 1.0f);
      const float lp_new = lg_obs_logpdf<DS, DO>(p, prop);
      const float wt = lp_new - lp_old;
      const bool accept = live && (logf(u) < gjx_min0(wt));
#pragma unroll
      for (int r = 0; r < DS; ++r) cur[r] = accept ? prop[r] : cur[r];
      lp_old = accept ? lp_new : lp_old;
      gjx_count_accepts(counts, m, accept);
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < DS; ++r) state[(int64_t)r * n + i] = cur[r];
    }
  }
  gjx_flush_accepts(counts, n_moves, accept_counts);
}


static int host_cholesky(const float* cov, int d, float* L) {
  for (int i = 0; i < d * d; ++i) L[i] = 0.0f;
  for (int r = 0; r < d; ++r)
    for (int c = 0; c <= r; ++c) {
      float s = cov[r * d + c];
      for (int k = 0; k < c; ++k) s -= L[r * d + k] * L[c * d + k];
      if (r == c) {
        if (!(s > 0.0f)) return -1;
        L[r * d + c] = sqrtf(s);
      } else {
        L[r * d + c] = s / L[c * d + c];
      }
    }
  return 0;
}

static int32_t build_lgk(const gjx_lgssm_params* m, int is_initial, const float* obs, LgK* p,
                         const char* who) {
  GJX_REQUIRE(m && obs, GJX_EINVAL, "%s: null parameter block", who);
  GJX_REQUIRE(m->d_state >= 1 && m->d_state <= GJX_LG_MAXD && m->d_obs >= 1 && m->d_obs <= GJX_LG_MAXD,
              GJX_ELIMIT, "%s: d_state=%d / d_obs=%d not in [1, %d]", who, m->d_state, m->d_obs,
              GJX_LG_MAXD);
  memset(p, 0, sizeof(*p));
  p->ds = m->d_state;
  p->dobs = m->d_obs;
  p->is_initial = is_initial;
  const int d = m->d_state, o = m->d_obs;
  for (int i = 0; i < d; ++i) p->mu0[i] = m->initial_mean[i];
  GJX_REQUIRE(host_cholesky(is_initial ? m->initial_cov : m->Q, d, p->Lx) == 0, GJX_EINVAL,
              "%s: state covariance is not positive definite", who);
  GJX_REQUIRE(host_cholesky(m->R, o, p->LR) == 0, GJX_EINVAL,
              "%s: observation covariance is not positive definite", who);
  for (int i = 0; i < d * d; ++i) p->A[i] = m->A[i];
  for (int i = 0; i < o * d; ++i) p->C[i] = m->C[i];
  float ln = 0.0f;
  for (int i = 0; i < o; ++i) {
    p->obs[i] = obs[i];
    ln += logf(p->LR[i * o + i]);
  }
  p->lognorm = ln + (float)o * HALF_LOG_2PI_F;
  return GJX_OK;
}

GJX_API int32_t gjx_lgssm_extend(gjx_ctx* ctx, const gjx_lgssm_params* model, const uint32_t key[2],
                                 int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                                 const float* d_prev, const float* d_logw_in, float* d_state,
                                 float* d_mean, float* d_logw, void* stream) {
  (void)ctx;
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(key && d_logw_in && d_state && d_mean && d_logw, GJX_EINVAL, "gjx_lgssm_extend: null argument");
  GJX_REQUIRE(is_initial || d_prev, GJX_EINVAL, "gjx_lgssm_extend: previous state required unless is_initial");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_lgssm_extend: n must not be negative");
  LgK p;
  int32_t rc = build_lgk(model, is_initial != 0, obs, &p, "gjx_lgssm_extend");
  if (rc != GJX_OK) return rc;
  p.key[0][0] = key[0];
  p.key[0][1] = key[1];
  LG_DISPATCH(p.ds, p.dobs, (k_lgssm_extend<DS, DO, false><<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(
                                p, offset, n, d_prev, d_logw_in, d_state, d_mean, d_logw, nullptr, nullptr, nullptr)));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_lgssm_mh(gjx_ctx* ctx, const gjx_lgssm_params* model, const uint32_t (*keys)[2][2],
                             int32_t n_moves, int32_t is_initial, uint64_t offset, int64_t n,
                             const float* obs, const float* d_state_in, const float* d_mean,
                             float* d_state, int64_t* d_accept_counts, void* stream) {
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(ctx && keys && d_state_in && d_mean && d_state, GJX_EINVAL, "gjx_lgssm_mh: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_lgssm_mh: n must not be negative");
  GJX_REQUIRE(n_moves >= 1 && n_moves <= GJX_MAX_MOVES, GJX_ELIMIT,
              "gjx_lgssm_mh: n_moves=%d not in [1, %d]", n_moves, GJX_MAX_MOVES);
  LgK p;
  int32_t rc = build_lgk(model, is_initial != 0, obs, &p, "gjx_lgssm_mh");
  if (rc != GJX_OK) return rc;
  LgMoveKeys mk;
  memset(&mk, 0, sizeof(mk));
  memcpy(mk.k, keys, sizeof(uint32_t) * 4 * n_moves);
  hipStream_t s = (hipStream_t)stream;
  if (d_accept_counts) GJX_CHECK_HIP(hipMemsetAsync(d_accept_counts, 0, sizeof(int64_t) * n_moves, s));
  LG_DISPATCH(p.ds, p.dobs, (k_lgssm_mh<DS, DO><<<gjx_grid_for(n), GJX_BLOCK, 0, s>>>(p, mk, n_moves, offset, n, d_state_in,
                                                                                    d_mean, d_state, d_accept_counts)));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// One smc_step of the bootstrap particle filter (smc.py:731-767) in ONE host call, with the lax.cond on
// ESS < N // 2 decided on the device: the host enqueues step after step and never waits.  At N = 2^20 a
// step is ~10 short launches; a host read-back per step (the form of the localization step) costs more
// than all of them together.  What the host would have done between the calls happens in the normaliser's fold kernel:
// the decision, log_marginal_estimate += lse - log N when it resamples (smc.py:628-632), and the step's
// diagnostics {ess, resampled} into a device array that is read once after the last step.
// ---------------------------------------------------------------------------
struct LgCols {
  const float* src[2 * GJX_LG_MAXD];
  float* dst[2 * GJX_LG_MAXD];
  int n;
};

// resample_vectorized_trace (smc.py:139-175) under the device decision: dst = src[go ? anc : identity];
// log-weights := 0 when resampled (smc.py:646).  The host flips the column sets either way.
__global__ void __launch_bounds__(GJX_BLOCK)
    k_lg_gather(LgCols c, const int32_t* __restrict__ go, const int32_t* __restrict__ anc, int64_t n,
                float* __restrict__ logw) {
  const bool take = go[0] != 0;
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * GJX_BLOCK) {
    const int64_t a = take ? (int64_t)anc[i] : i;
    for (int k = 0; k < c.n; ++k) c.dst[k][i] = c.src[k][a];
    if (take) logw[i] = 0.0f;
  }
}

// One step.  fuse_prev: the previous step left its resampling to this one (d_go / ctx->d_tmp_anc still hold its
// decision and ancestors): the particles are read from (d_state, d_mean) through them and the new ones written to
// the *_alt columns.  Otherwise the extend runs in place on (d_state, d_mean).  gather_now: this step's own
// resampling is materialised into the *_alt columns before returning (needed when moves follow).
static int32_t lg_smc_step(gjx_ctx* ctx, const gjx_lgssm_params* model, const uint32_t key[2], int32_t is_initial,
                           int64_t n, const float* obs, float u, bool fuse_prev, bool gather_now, float* d_state,
                           float* d_mean, float* d_state_alt, float* d_mean_alt, float* d_logw, float* d_stats5,
                           float* d_lml, float* d_trace2, int32_t* d_go, int64_t* d_range,
                           const uint32_t (*move_keys)[2][2], int32_t n_moves,
                           int64_t* d_accept_counts, hipStream_t s) {
  LgK p;
  int32_t rc = build_lgk(model, is_initial != 0, obs, &p, "gjx_lgssm_smc_run");
  if (rc != GJX_OK) return rc;
  p.key[0][0] = key[0];
  p.key[0][1] = key[1];
  const int grid = gjx_grid_for(n);
  const int ds = model->d_state;
  float* out_state = fuse_prev ? d_state_alt : d_state;
  float* out_mean = fuse_prev ? d_mean_alt : d_mean;
  LG_DISPATCH(p.ds, p.dobs, (k_lgssm_extend<DS, DO, true><<<grid, GJX_BLOCK, 0, s>>>(
                                p, 0, n, is_initial ? nullptr : d_state, d_logw, out_state, out_mean, d_logw,
                                ctx->d_partials, fuse_prev ? d_go : nullptr, ctx->d_tmp_anc)));
  GJX_LAUNCH_CHECK();
  GjxDecide dec;   // the lax.cond, log_marginal_estimate += lse - log N, and the step's diagnostics: in the fold kernel
  dec.go = d_go;
  dec.range = d_range;
  dec.n = n;
  dec.half_n = n / 2;
  dec.log_n = (float)log((double)n);
  dec.lml = d_lml;
  dec.trace2 = d_trace2;
  rc = gjx_normalise_decide(ctx, d_logw, n, ctx->d_partials, grid, d_stats5, dec, s);
  if (rc != GJX_OK) return rc;
  rc = gjx_resample_systematic_logw_if(ctx, d_logw, d_stats5 + 1, n, u, d_go, d_range, ctx->d_tmp_anc, s);
  if (rc != GJX_OK) return rc;
  if (!gather_now) return GJX_OK;
  float* from_state = out_state;
  float* from_mean = out_mean;
  float* to_state = fuse_prev ? d_state : d_state_alt;
  float* to_mean = fuse_prev ? d_mean : d_mean_alt;
  LgCols c;
  c.n = 2 * ds;
  for (int r = 0; r < ds; ++r) {
    c.src[r] = from_state + (int64_t)r * n;
    c.dst[r] = to_state + (int64_t)r * n;
    c.src[ds + r] = from_mean + (int64_t)r * n;
    c.dst[ds + r] = to_mean + (int64_t)r * n;
  }
  for (int k = 2 * ds; k < 2 * GJX_LG_MAXD; ++k) c.src[k] = nullptr, c.dst[k] = nullptr;
  k_lg_gather<<<grid, GJX_BLOCK, 0, s>>>(c, d_go, ctx->d_tmp_anc, n, d_logw);
  GJX_LAUNCH_CHECK();
  if (n_moves > 0)
    return gjx_lgssm_mh(ctx, model, move_keys, n_moves, is_initial, 0, n, obs, to_state, to_mean, to_state,
                        d_accept_counts, (void*)s);
  return GJX_OK;
}

// The whole filter: rejuvenation_smc(linear_gaussian, observations = T rows, mcmc_kernel = mh(sel("state")) | None,
// n_rejuvenation_moves = n_moves, systematic resampling) under seed(...) in ONE host call -- the Seed key
// schedule (pjax.py:1318-1401) walked here on the host: the init stage runs on the caller's interpreter
// (interp_key, advanced in place), then one split for the scan over observations[1:], step t on a fresh
// interpreter on fold_in(scan, t - 1).  A stage takes: the "state" site key, one split for the cond (whose
// sub-interpreter yields the resampling uniform), and with a kernel one split for the scan over the moves
// (move m on fold_in(., m): the regenerated site, then the accept uniform).
struct HostSeq {
  uint32_t k[2];
  void next(uint32_t sub[2]) {
    uint32_t nx[2];
    gjx_tf2x32(k[0], k[1], 0u, 1u, sub[0], sub[1]);
    gjx_tf2x32(k[0], k[1], 0u, 0u, nx[0], nx[1]);
    k[0] = nx[0];
    k[1] = nx[1];
  }
};

static float host_uniform01(const uint32_t key[2]) {   // uniform.sample(0, 1) of a scalar site
  uint32_t a, b;
  gjx_tf2x32(key[0], key[1], 0u, 0u, a, b);
  const uint32_t bits = ((a ^ b) >> 9) | 0x3F800000u;
  float f;
  memcpy(&f, &bits, sizeof(f));
  return f - 1.0f;
}

GJX_API int32_t gjx_lgssm_smc_run(gjx_ctx* ctx, const gjx_lgssm_params* model, uint32_t interp_key[2], int32_t T,
                                  const float* obs_rows, int64_t n, int32_t n_moves, float* d_state, float* d_mean,
                                  float* d_state_alt, float* d_mean_alt, float* d_logw, float* d_stats5, float* d_lml,
                                  float* d_trace, int32_t* d_go, int64_t* d_range, int64_t* d_accept_counts,
                                  int32_t* final_set, void* stream) {
  GJX_REQUIRE(ctx && model && interp_key && obs_rows && final_set, GJX_EINVAL, "gjx_lgssm_smc_run: null argument");
  GJX_REQUIRE(T >= 1 && d_trace, GJX_EINVAL, "gjx_lgssm_smc_run: T must be positive and the trace array given");
  GJX_REQUIRE(n_moves >= 0 && n_moves <= GJX_MAX_MOVES, GJX_ELIMIT, "gjx_lgssm_smc_run: n_moves=%d not in [0, %d]",
              n_moves, GJX_MAX_MOVES);
  GJX_REQUIRE(ctx && d_state && d_mean && d_state_alt && d_mean_alt && d_logw && d_stats5 && d_lml && d_go && d_range,
              GJX_EINVAL, "gjx_lgssm_smc_run: null device argument");
  GJX_REQUIRE(n > 0 && n <= ctx->n_max, GJX_ELIMIT, "gjx_lgssm_smc_run: n=%lld not in (0, n_max=%lld]", (long long)n,
              (long long)ctx->n_max);
  hipStream_t s = (hipStream_t)stream;
  GJX_CHECK_HIP(hipMemsetAsync(d_logw, 0, sizeof(float) * n, s));
  GJX_CHECK_HIP(hipMemsetAsync(d_lml, 0, sizeof(float), s));
  HostSeq top{{interp_key[0], interp_key[1]}}, scan{{0u, 0u}};
  float* st[2] = {d_state, d_state_alt};
  float* mn[2] = {d_mean, d_mean_alt};
  // Without moves a step leaves its resampling to the next step's extend (no gather pass); the last one is
  // materialised after the loop.  With moves every step gathers (the moves need the resampled particles).
  const bool defer = n_moves == 0;
  int cur = 0;
  for (int t = 0; t < T; ++t) {
    HostSeq step;
    if (t > 0) gjx_tf2x32(scan.k[0], scan.k[1], 0u, (uint32_t)(t - 1), step.k[0], step.k[1]);   // fold_in
    HostSeq& seq = t == 0 ? top : step;
    uint32_t k_ext[2], kc[2], k_u[2], mk[GJX_MAX_MOVES][2][2];
    seq.next(k_ext);
    seq.next(kc);
    HostSeq cond{{kc[0], kc[1]}};
    cond.next(k_u);
    if (n_moves > 0) {
      uint32_t sr[2];
      seq.next(sr);
      for (int m = 0; m < n_moves; ++m) {
        HostSeq mv;
        gjx_tf2x32(sr[0], sr[1], 0u, (uint32_t)m, mv.k[0], mv.k[1]);
        mv.next(mk[m][0]);
        mv.next(mk[m][1]);
      }
    }
    const bool fuse_prev = defer && t > 0;
    const bool gather_now = !defer || t == T - 1;
    int32_t rc = lg_smc_step(ctx, model, k_ext, t == 0, n, obs_rows + (int64_t)t * model->d_obs, host_uniform01(k_u),
                             fuse_prev, gather_now, st[cur], mn[cur], st[1 - cur], mn[1 - cur], d_logw, d_stats5, d_lml,
                             d_trace + 2 * (int64_t)t, d_go, d_range, n_moves > 0 ? mk : nullptr,
                             n_moves, d_accept_counts ? d_accept_counts + (int64_t)t * n_moves : nullptr, s);
    if (rc != GJX_OK) return rc;
    // where the particles are now: extend wrote to the other set iff fuse_prev; the gather moved them once more
    if (fuse_prev != gather_now) cur = 1 - cur;
    if (t == 0) {
      uint32_t sk[2];
      top.next(sk);   // scan over observations[1:]: one split
      scan.k[0] = sk[0];
      scan.k[1] = sk[1];
    }
  }
  interp_key[0] = top.k[0];
  interp_key[1] = top.k[1];
  *final_set = cur;
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// Outlier-mixture curve fitting: enumerative Gibbs over the outlier indicators + HMC over
// (curve.a, curve.b, curve.c, outlier_rate), one thread per chain / particle, n_iters kernel
// applications fused (examples/curvefit/core.py:468-552 model, :934-1014 kernel; harness
// examples/curvefit/figs.py:907-950).  Each chain carries its own Threefry key:
//   chain key = split(key, n)[i];  iteration: key, sub = split(key);  Seed(sub): S -> Gibbs
//   categorical noise (P, 2); S x 4 -> momenta a, b, c, rate; S -> leapfrog scan (unused); S -> u.
// State per chain: 4 floats + a 64-bit outlier mask + the carried key (32 B); the data set
// (<= 64 points) travels in the kernel arguments.  VALU-bound: per iteration ~(2 n_steps + 4)
// passes over the points plus ~(2 P + 21) Threefry blocks.
// ---------------------------------------------------------------------------
struct OutK {
  int P;
  float xs[GJX_MAX_POINTS], ys[GJX_MAX_POINTS];
  float alpha, beta, sigma, lo, hi;
  float lbeta, lp_out, inv_var, log_sigma;
};

__device__ __forceinline__ void key_split2(const uint32_t (&k)[2], uint32_t (&k0)[2], uint32_t (&k1)[2]) {
  gjx_tf2x32(k[0], k[1], 0u, 0u, k0[0], k0[1]);
  gjx_tf2x32(k[0], k[1], 0u, 1u, k1[0], k1[1]);
}

// self.key, sub = split(self.key)  (pjax.py:1337)
__device__ __forceinline__ void seed_next(uint32_t (&st)[2], uint32_t (&sub)[2]) {
  uint32_t nx[2];
  key_split2(st, nx, sub);
  st[0] = nx[0];
  st[1] = nx[1];
}

__device__ __forceinline__ float nlogpdf(float x, float mu, float inv_sigma, float log_sigma) {
  const float z = (x - mu) * inv_sigma;
  return -0.5f * (z * z) - (HALF_LOG_2PI_F + log_sigma);
}

__device__ __forceinline__ float bern_lp(float x, float p) {
  const float a = (x == 0.0f) ? 0.0f : x * logf(p);
  const float b = (1.0f - x == 0.0f) ? 0.0f : (1.0f - x) * log1pf(-p);
  return a + b;
}

__device__ float mt_gamma(const uint32_t (&key)[2], uint64_t stream, float shape) {
  const float a = shape < 1.0f ? shape + 1.0f : shape;
  const float d = a - 1.0f / 3.0f, c = 1.0f / sqrtf(9.0f * d);
  float g = d;
  for (uint64_t t = 0; t < 64; ++t) {
    const float x = gjx_normal_at(key[0], key[1], (stream << 16) + 2 * t);
    const float u = gjx_uniform_at(key[0], key[1], (stream << 16) + 2 * t + 1, 0.0f, 1.0f);
    float v = 1.0f + c * x;
    if (v <= 0.0f) continue;
    v = v * v * v;
    g = d * v;
    if (logf(u) < 0.5f * x * x + d - d * v + d * logf(v)) break;
  }
  if (shape < 1.0f) {
    const float u = gjx_uniform_at(key[0], key[1], (stream << 16) + 200, 0.0f, 1.0f);
    g = g * powf(u, 1.0f / shape);
  }
  return g;
}

// assess() of the model and its gradient w.r.t. (a, b, c, rate)  (mcmc.py:59-88, 508)
__device__ __forceinline__ float outlier_logp_grad(const OutK& m, const float (&q)[4], uint64_t mask,
                                                   float (&g)[4]) {
  const float a = q[0], b = q[1], c = q[2], rate = q[3];
  const float t1 = (m.alpha - 1.0f == 0.0f) ? 0.0f : (m.alpha - 1.0f) * logf(rate);
  const float t2 = (m.beta - 1.0f == 0.0f) ? 0.0f : (m.beta - 1.0f) * log1pf(-rate);
  float lp = (rate < 0.0f || rate > 1.0f) ? -__builtin_inff() : t1 + t2 - m.lbeta;
  lp += -0.5f * (a * a) - HALF_LOG_2PI_F;
  lp += -0.5f * (b * b) - HALF_LOG_2PI_F;
  lp += -0.5f * (c * c) - HALF_LOG_2PI_F;
  float ga = -a, gb = -b, gc = -c;
  const float inv_rate = 1.0f / rate, inv_1mr = 1.0f / (1.0f - rate);
  float grate = (m.alpha - 1.0f == 0.0f ? 0.0f : (m.alpha - 1.0f) * inv_rate) -
                (m.beta - 1.0f == 0.0f ? 0.0f : (m.beta - 1.0f) * inv_1mr);
  const float lp_o1 = bern_lp(1.0f, rate), lp_o0 = bern_lp(0.0f, rate);
  const float inv_sigma = 1.0f / m.sigma;
  float ys_lp = 0.0f;
  for (int j = 0; j < m.P; ++j) {
    const float x = m.xs[j], y = m.ys[j];
    const bool o = (mask >> j) & 1ull;
    const float x2 = x * x;
    const float mean = a + b * x + c * x2;
    const float lo_pt = (y < m.lo || y > m.hi) ? -__builtin_inff() : m.lp_out;
    const float pt = o ? (lp_o1 + lo_pt) : (lp_o0 + nlogpdf(y, mean, inv_sigma, m.log_sigma));
    const float r = o ? 0.0f : (y - mean) * m.inv_var;
    ga += r;
    gb += r * x;
    gc += r * x2;
    grate += o ? inv_rate : -inv_1mr;
    ys_lp += pt;
  }
  lp += ys_lp;
  g[0] = ga;
  g[1] = gb;
  g[2] = gc;
  g[3] = grate;
  return lp;
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_outlier_init(OutK m, uint32_t bk0, uint32_t bk1, uint64_t offset, int64_t n,
                   uint32_t* __restrict__ chain_keys, float* __restrict__ a, float* __restrict__ b,
                   float* __restrict__ c, float* __restrict__ rate,
                   unsigned long long* __restrict__ outliers) {
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * GJX_BLOCK) {
    const uint64_t gi = offset + (uint64_t)i;
    uint32_t st[2], sub[2];
    gjx_tf2x32(bk0, bk1, (uint32_t)(gi >> 32), (uint32_t)gi, st[0], st[1]);  // split(key, n)[i]
    chain_keys[2 * i] = st[0];
    chain_keys[2 * i + 1] = st[1];
    seed_next(st, sub);  // outlier_rate ~ beta (see the header comment on the sampler)
    const float ga = mt_gamma(sub, 0, m.alpha), gb = mt_gamma(sub, 1, m.beta);
    const float rt = ga / (ga + gb);
    rate[i] = rt;
    seed_next(st, sub);
    a[i] = gjx_add_rn(gjx_mul_rn(gjx_normal_at(sub[0], sub[1], 0), 1.0f), 0.0f);
    seed_next(st, sub);
    b[i] = gjx_add_rn(gjx_mul_rn(gjx_normal_at(sub[0], sub[1], 0), 1.0f), 0.0f);
    seed_next(st, sub);
    c[i] = gjx_add_rn(gjx_mul_rn(gjx_normal_at(sub[0], sub[1], 0), 1.0f), 0.0f);
    seed_next(st, sub);  // is_outlier ~ flip(rate), sample_shape (P,)
    unsigned long long mask = 0ull;
    for (int j = 0; j < m.P; ++j)
      if (gjx_uniform_at(sub[0], sub[1], (uint64_t)j, 0.0f, 1.0f) < rt) mask |= 1ull << j;
    outliers[i] = mask;
  }
}

#ifndef GJX_OUT_WAVES_N
#define GJX_OUT_WAVES_N 0
#endif
#if GJX_OUT_WAVES_N > 0
#define GJX_OUT_WAVES __attribute__((amdgpu_waves_per_eu(GJX_OUT_WAVES_N, GJX_OUT_WAVES_N)))
#else
#define GJX_OUT_WAVES
#endif
// DIAG: the same iteration (ONE per launch) that also stores the quantities of its decisions in (10 + P) columns
// of n floats: 0..3 the HMC proposal, 4 log alpha before min(0, .), 5 log u, 6 / 7 assess() at the start / end
// point, 8 / 9 the momentum log-densities there, 10 + j the Gibbs margin v1 - v0 of point j (gjx_outlier_gibbs_hmc_diag).
template <bool DIAG>
__global__ void __launch_bounds__(GJX_BLOCK) GJX_OUT_WAVES
    k_outlier_gibbs_hmc(OutK m, int n_iters, float eps, float half_eps, int n_steps, int64_t n,
                        uint32_t* __restrict__ chain_keys, float* __restrict__ a,
                        float* __restrict__ b, float* __restrict__ c, float* __restrict__ rate,
                        unsigned long long* __restrict__ outliers,
                        int64_t* __restrict__ accept_count, float* __restrict__ diag) {
  __shared__ uint32_t counts[1];
  if (threadIdx.x == 0) counts[0] = 0u;
  __syncthreads();
  const float tiny = 1.17549435e-38f;
  const float inv_sigma = 1.0f / m.sigma;
  const int64_t nthreads = (int64_t)gridDim.x * GJX_BLOCK;
  const int64_t n_round = ((n + nthreads - 1) / nthreads) * nthreads;
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n_round; i += nthreads) {
    const bool live = i < n;
    const int64_t ii = live ? i : 0;
    uint32_t key[2] = {chain_keys[2 * ii], chain_keys[2 * ii + 1]};
    float q[4] = {a[ii], b[ii], c[ii], rate[ii]};
    unsigned long long mask = outliers[ii];
    uint32_t n_acc = 0;
    for (int it = 0; it < n_iters; ++it) {
      uint32_t st[2], sub[2];
      seed_next(key, st);  // key, subkey = jrand.split(key)   (figs.py:928)
      // ---- enumerative Gibbs (core.py:934-985): per point logits {not outlier, outlier}
      seed_next(st, sub);
      const float lp0 = bern_lp(0.0f, q[3]), lp1 = bern_lp(1.0f, q[3]);
      unsigned long long nm = 0ull;
      for (int j = 0; j < m.P; ++j) {
        const float x = m.xs[j], y = m.ys[j];
        const float mean = q[0] + q[1] * x + q[2] * (x * x);
        const float l0 = lp0 + nlogpdf(y, mean, inv_sigma, m.log_sigma);
        const float l1 = lp1 + ((y < m.lo || y > m.hi) ? -__builtin_inff() : m.lp_out);
        // categorical.sample(logits (P, 2)) = argmax(logits + gumbel(key, (P, 2)))  [3P]
        const float u0 = gjx_uniform_at(sub[0], sub[1], 2ull * j, tiny, 1.0f);
        const float u1 = gjx_uniform_at(sub[0], sub[1], 2ull * j + 1, tiny, 1.0f);
        const float v0 = l0 + (-logf(-logf(u0))), v1 = l1 + (-logf(-logf(u1)));
        if (v1 > v0) nm |= 1ull << j;
        if constexpr (DIAG)
          if (live) diag[(int64_t)(10 + j) * n + i] = v1 - v0;
      }
      mask = nm;
      // ---- hmc on (a, b, c, rate) (mcmc.py:434-578)
      float p[4], g[4], gd[4];
      const float prev_model = outlier_logp_grad(m, q, mask, g);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        seed_next(st, sub);
        p[k] = gjx_add_rn(gjx_mul_rn(gjx_normal_at(sub[0], sub[1], 0), 1.0f), 0.0f);
      }
      seed_next(st, sub);  // leapfrog scan_p: one split, unused (pjax.py:1381)
      seed_next(st, sub);
      const float u = gjx_uniform_at(sub[0], sub[1], 0, 0.0f, 1.0f);
      const float prev_mom = (((-0.5f * (p[0] * p[0]) - HALF_LOG_2PI_F) +
                               (-0.5f * (p[1] * p[1]) - HALF_LOG_2PI_F)) +
                              (-0.5f * (p[2] * p[2]) - HALF_LOG_2PI_F)) +
                             (-0.5f * (p[3] * p[3]) - HALF_LOG_2PI_F);
      float qq[4] = {q[0], q[1], q[2], q[3]};
      for (int l = 0; l < n_steps; ++l) {
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = p[k] + half_eps * g[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) qq[k] = qq[k] + eps * p[k];
        (void)outlier_logp_grad(m, qq, mask, g);
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = p[k] + half_eps * g[k];
      }
      const float new_model = outlier_logp_grad(m, qq, mask, gd);
      const float new_mom = (((-0.5f * (p[0] * p[0]) - HALF_LOG_2PI_F) +
                              (-0.5f * (p[1] * p[1]) - HALF_LOG_2PI_F)) +
                             (-0.5f * (p[2] * p[2]) - HALF_LOG_2PI_F)) +
                            (-0.5f * (p[3] * p[3]) - HALF_LOG_2PI_F);
      float log_alpha = (new_model + new_mom) - (prev_model + prev_mom);
      if constexpr (DIAG) {
        if (live) {
#pragma unroll
          for (int k = 0; k < 4; ++k) diag[(int64_t)k * n + i] = qq[k];
          diag[4 * n + i] = log_alpha;
          diag[5 * n + i] = logf(u);
          diag[6 * n + i] = prev_model;
          diag[7 * n + i] = new_model;
          diag[8 * n + i] = prev_mom;
          diag[9 * n + i] = new_mom;
        }
      }
      log_alpha = gjx_min0(log_alpha);
      const bool accept = live && (logf(u) < log_alpha);
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = accept ? qq[k] : q[k];
      n_acc += accept ? 1u : 0u;
    }
    if (live) {
      chain_keys[2 * i] = key[0];
      chain_keys[2 * i + 1] = key[1];
      a[i] = q[0];
      b[i] = q[1];
      c[i] = q[2];
      rate[i] = q[3];
      outliers[i] = mask;
    }
    // wave-level sum of the accept counts -> one LDS add per wave
    uint32_t s = n_acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&counts[0], s);
  }
  __syncthreads();
  if (threadIdx.x == 0 && accept_count != nullptr)
    atomicAdd(reinterpret_cast<unsigned long long*>(accept_count), (unsigned long long)counts[0]);
}

static int32_t build_outk(const gjx_outlier_model* m, OutK* k, const char* who) {
  GJX_REQUIRE(m != nullptr, GJX_EINVAL, "%s: model is null", who);
  GJX_REQUIRE(m->n_points >= 1 && m->n_points <= GJX_MAX_POINTS, GJX_ELIMIT,
              "%s: n_points=%d not in [1, %d]", who, m->n_points, GJX_MAX_POINTS);
  GJX_REQUIRE(m->alpha > 0.0f && m->beta > 0.0f && m->inlier_sigma > 0.0f && m->out_hi > m->out_lo,
              GJX_EINVAL, "%s: alpha, beta, inlier_sigma must be positive and out_hi > out_lo", who);
  memset(k, 0, sizeof(*k));
  k->P = m->n_points;
  for (int j = 0; j < m->n_points; ++j) {
    k->xs[j] = m->xs[j];
    k->ys[j] = m->ys[j];
  }
  k->alpha = m->alpha;
  k->beta = m->beta;
  k->sigma = m->inlier_sigma;
  k->lo = m->out_lo;
  k->hi = m->out_hi;
  k->lbeta = lgammaf(m->alpha) + lgammaf(m->beta) - lgammaf(m->alpha + m->beta);
  k->lp_out = -logf(m->out_hi - m->out_lo);
  k->inv_var = 1.0f / (m->inlier_sigma * m->inlier_sigma);
  k->log_sigma = logf(m->inlier_sigma);
  return GJX_OK;
}

GJX_API int32_t gjx_outlier_init(const gjx_outlier_model* model, const uint32_t base_key[2],
                                 uint64_t offset, int64_t n, uint32_t* d_chain_keys, float* d_a,
                                 float* d_b, float* d_c, float* d_rate, uint64_t* d_outliers,
                                 void* stream) {
  if (n == 0) return GJX_OK;
  GJX_REQUIRE(base_key && d_chain_keys && d_a && d_b && d_c && d_rate && d_outliers, GJX_EINVAL,
              "gjx_outlier_init: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_outlier_init: n must not be negative");
  OutK k;
  int32_t rc = build_outk(model, &k, "gjx_outlier_init");
  if (rc != GJX_OK) return rc;
  k_outlier_init<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      k, base_key[0], base_key[1], offset, n, d_chain_keys, d_a, d_b, d_c, d_rate,
      reinterpret_cast<unsigned long long*>(d_outliers));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_outlier_gibbs_hmc(const gjx_outlier_model* model, int32_t n_iters,
                                      float step_size, int32_t n_steps, int64_t n,
                                      uint32_t* d_chain_keys, float* d_a, float* d_b, float* d_c,
                                      float* d_rate, uint64_t* d_outliers, int64_t* d_accept_count,
                                      void* stream) {
  if (n == 0) return GJX_OK;
  GJX_REQUIRE(d_chain_keys && d_a && d_b && d_c && d_rate && d_outliers, GJX_EINVAL,
              "gjx_outlier_gibbs_hmc: null argument");
  GJX_REQUIRE(n > 0 && n_iters >= 0, GJX_EINVAL, "gjx_outlier_gibbs_hmc: bad sizes");
  GJX_REQUIRE(n_steps >= 1 && n_steps <= 4096, GJX_ELIMIT,
              "gjx_outlier_gibbs_hmc: n_steps=%d out of range", n_steps);
  OutK k;
  int32_t rc = build_outk(model, &k, "gjx_outlier_gibbs_hmc");
  if (rc != GJX_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (d_accept_count) GJX_CHECK_HIP(hipMemsetAsync(d_accept_count, 0, sizeof(int64_t), s));
  const float half_eps = (float)((double)step_size / 2.0);  // (step_size / 2.0) * g, mcmc.py:516
  k_outlier_gibbs_hmc<false><<<gjx_grid_heavy(n), GJX_BLOCK, 0, s>>>(
      k, n_iters, step_size, half_eps, n_steps, n, d_chain_keys, d_a, d_b, d_c, d_rate,
      reinterpret_cast<unsigned long long*>(d_outliers), d_accept_count, nullptr);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ONE iteration of the same kernel code that also reports what it decided on (d_diag: (10 + n_points) columns of
// n floats, layout at k_outlier_gibbs_hmc): the single-move entry point the fused launch is tied to bit for bit by
// the parity tests, as gjx_loc_hmc_propose is for gjx_loc_hmc.
GJX_API int32_t gjx_outlier_gibbs_hmc_diag(const gjx_outlier_model* model, float step_size, int32_t n_steps,
                                           int64_t n, uint32_t* d_chain_keys, float* d_a, float* d_b, float* d_c,
                                           float* d_rate, uint64_t* d_outliers, float* d_diag,
                                           int64_t* d_accept_count, void* stream) {
  if (n == 0) return GJX_OK;
  GJX_REQUIRE(d_chain_keys && d_a && d_b && d_c && d_rate && d_outliers && d_diag, GJX_EINVAL,
              "gjx_outlier_gibbs_hmc_diag: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_outlier_gibbs_hmc_diag: bad sizes");
  GJX_REQUIRE(n_steps >= 1 && n_steps <= 4096, GJX_ELIMIT, "gjx_outlier_gibbs_hmc_diag: n_steps=%d out of range",
              n_steps);
  OutK k;
  int32_t rc = build_outk(model, &k, "gjx_outlier_gibbs_hmc_diag");
  if (rc != GJX_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (d_accept_count) GJX_CHECK_HIP(hipMemsetAsync(d_accept_count, 0, sizeof(int64_t), s));
  const float half_eps = (float)((double)step_size / 2.0);
  k_outlier_gibbs_hmc<true><<<gjx_grid_heavy(n), GJX_BLOCK, 0, s>>>(
      k, 1, step_size, half_eps, n_steps, n, d_chain_keys, d_a, d_b, d_c, d_rate,
      reinterpret_cast<unsigned long long*>(d_outliers), d_accept_count, d_diag);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}
