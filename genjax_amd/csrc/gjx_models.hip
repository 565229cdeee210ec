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

__device__ __forceinline__ float lg_obs_logpdf(const LgK& p, const float* x) {
  float z[GJX_LG_MAXD], q = 0.0f;
  for (int r = 0; r < p.dobs; ++r) {
    float m = 0.0f;
    for (int c = 0; c < p.ds; ++c) m += p.C[r * p.ds + c] * x[c];
    float s = p.obs[r] - m;
    for (int k = 0; k < r; ++k) s -= p.LR[r * p.dobs + k] * z[k];
    z[r] = s / p.LR[r * p.dobs + r];
    q += z[r] * z[r];
  }
  return -0.5f * q - p.lognorm;
}

__device__ __forceinline__ void lg_draw_state(const LgK& p, const uint32_t (&key)[2], uint64_t gi,
                                              const float* mean, float* x) {
  float z[GJX_LG_MAXD];
  for (int r = 0; r < p.ds; ++r) z[r] = gjx_normal_at(key[0], key[1], gi * (uint64_t)p.ds + (uint64_t)r);
  for (int r = 0; r < p.ds; ++r) {
    float s = 0.0f;
    for (int c = 0; c <= r; ++c) s += p.Lx[r * p.ds + c] * z[c];
    x[r] = mean[r] + s;
  }
}

__device__ __forceinline__ void lg_mean(const LgK& p, const float* prev, int64_t i, int64_t n,
                                        float* mean) {
  if (p.is_initial) {
    for (int r = 0; r < p.ds; ++r) mean[r] = p.mu0[r];
  } else {
    float pv[GJX_LG_MAXD];
    for (int c = 0; c < p.ds; ++c) pv[c] = prev[(int64_t)c * n + i];
    for (int r = 0; r < p.ds; ++r) {
      float s = 0.0f;
      for (int c = 0; c < p.ds; ++c) s += p.A[r * p.ds + c] * pv[c];
      mean[r] = s;
    }
  }
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_lgssm_extend(LgK p, uint64_t offset, int64_t n, const float* prev, const float* logw_in,
                   float* state, float* mean_out, float* logw) {
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * GJX_BLOCK) {
    float mean[GJX_LG_MAXD], x[GJX_LG_MAXD];
    lg_mean(p, prev, i, n, mean);
    const float lw = logw_in[i];
    lg_draw_state(p, p.key[0], offset + (uint64_t)i, mean, x);
    for (int r = 0; r < p.ds; ++r) {
      state[(int64_t)r * n + i] = x[r];
      mean_out[(int64_t)r * n + i] = mean[r];
    }
    logw[i] = lw + lg_obs_logpdf(p, x);
  }
}

// n_moves fused mh(trace, sel("state")) moves: fresh prior draw at the stored mean, weight =
// logpdf_obs(new) - logpdf_obs(old), accept iff log u < min(0, w).  keys[m] = {state', u}.
struct LgMoveKeys {
  uint32_t k[GJX_MAX_MOVES][2][2];
};

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
    float cur[GJX_LG_MAXD], mean[GJX_LG_MAXD], prop[GJX_LG_MAXD];
    for (int r = 0; r < p.ds; ++r) {
      cur[r] = state_in[(int64_t)r * n + ii];
      mean[r] = mean_in[(int64_t)r * n + ii];
    }
    float lp_old = lg_obs_logpdf(p, cur);
    const uint64_t gi = offset + (uint64_t)ii;
    for (int m = 0; m < n_moves; ++m) {
      lg_draw_state(p, keys.k[m][0], gi, mean, prop);
      const float u = gjx_uniform_at(keys.k[m][1][0], keys.k[m][1][1], gi, 0.0f, 1.0f);
      const float lp_new = lg_obs_logpdf(p, prop);
      const float wt = lp_new - lp_old;
      const bool accept = live && (logf(u) < fminf(0.0f, wt));
      for (int r = 0; r < p.ds; ++r) cur[r] = accept ? prop[r] : cur[r];
      lp_old = accept ? lp_new : lp_old;
      gjx_count_accepts(counts, m, accept);
    }
    if (live)
      for (int r = 0; r < p.ds; ++r) state[(int64_t)r * n + i] = cur[r];
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
  k_lgssm_extend<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(p, offset, n, d_prev, d_logw_in,
                                                                          d_state, d_mean, d_logw);
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
  k_lgssm_mh<<<gjx_grid_for(n), GJX_BLOCK, 0, s>>>(p, mk, n_moves, offset, n, d_state_in, d_mean,
                                                    d_state, d_accept_counts);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}
