// gjx_core.hip -- context, errors, host key arithmetic, stand-alone draw and log-density kernels.
#include <stdarg.h>
#include <string.h>

#include "gjx_common.h"

// ---------------------------------------------------------------------------
// errors / context
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void gjx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

GJX_API const char* gjx_last_error(void) { return g_err; }
GJX_API int32_t gjx_version(void) { return 100; }

static void ctx_free(gjx_ctx* c) {
  (void)hipFree(c->d_partials);
  (void)hipFree(c->d_tile_sums);
  (void)hipFree(c->d_scalars);
  for (int l = 0; l < 4; ++l) {
    (void)hipFree(c->d_scan_lvl[l]);
    (void)hipFree(c->d_scan_out[l]);
  }
  (void)hipFree(c->d_tmp_cdf);
  (void)hipFree(c->d_tmp_anc);
  delete c;
}

static int32_t ctx_alloc(gjx_ctx* c) {
  GJX_CHECK_HIP(hipMalloc(&c->d_partials, sizeof(float) * GJX_MAX_PARTIALS));
  GJX_CHECK_HIP(hipMalloc(&c->d_tile_sums, sizeof(float) * 2 * ((c->n_max + GJX_SCAN_TILE - 1) / GJX_SCAN_TILE + 1)));
  GJX_CHECK_HIP(hipMalloc(&c->d_scalars, sizeof(float) * 16));
  {
    float init[16];
    for (int k = 0; k < 16; ++k) init[k] = 0.0f;
    init[GJX_MAX_CELL] = -INFINITY;
    GJX_CHECK_HIP(hipMemcpy(c->d_scalars, init, sizeof(init), hipMemcpyHostToDevice));
  }
  int64_t m = c->n_max;
  for (int l = 0; l < 4; ++l) {
    m = (m + GJX_SCAN_TILE - 1) / GJX_SCAN_TILE;
    c->scan_cap[l] = m;
    GJX_CHECK_HIP(hipMalloc(&c->d_scan_lvl[l], sizeof(float) * (m + 1)));
    GJX_CHECK_HIP(hipMalloc(&c->d_scan_out[l], sizeof(float) * (m + 1)));
    if (m <= 1) break;
  }
  GJX_CHECK_HIP(hipMalloc(&c->d_tmp_cdf, sizeof(float) * c->n_max));
  GJX_CHECK_HIP(hipMalloc(&c->d_tmp_anc, sizeof(int32_t) * c->n_max));
  // the library's static scratch array (gjx_resample_systematic_logw_static): resolve this
  // device's address now, so that no later call has to do it under a graph capture
  float* unused = nullptr;
  return gjx_static_scratch(&unused);
}

GJX_API int32_t gjx_ctx_create(int32_t device, int64_t n_max, gjx_ctx** out) {
  GJX_REQUIRE(out != nullptr, GJX_EINVAL, "gjx_ctx_create: out is null");
  GJX_REQUIRE(n_max > 0 && n_max < ((int64_t)1 << 31), GJX_EINVAL,
              "gjx_ctx_create: n_max %lld out of range (0, 2^31)", (long long)n_max);
  int prev = -1;
  GJX_CHECK_HIP(hipGetDevice(&prev));
  GJX_CHECK_HIP(hipSetDevice(device));
  gjx_ctx* c = new gjx_ctx();
  memset(c, 0, sizeof(*c));
  c->device = device;
  c->n_max = n_max;
  const int32_t rc = ctx_alloc(c);   // on failure: free what was allocated, keep the error text
  if (rc != GJX_OK) ctx_free(c);
  (void)hipSetDevice(prev);          // the caller's current device is left as it was
  if (rc != GJX_OK) return rc;
  *out = c;
  return GJX_OK;
}

GJX_API int32_t gjx_ctx_destroy(gjx_ctx* c) {
  if (!c) return GJX_OK;
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(c->device);
  ctx_free(c);
  if (prev >= 0) (void)hipSetDevice(prev);
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// host key arithmetic
// ---------------------------------------------------------------------------
GJX_API void gjx_key(uint64_t seed, uint32_t key[2]) {
  key[0] = (uint32_t)(seed >> 32);
  key[1] = (uint32_t)(seed & 0xFFFFFFFFu);
}
GJX_API void gjx_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]) {
  gjx_tf2x32(key[0], key[1], ctr[0], ctr[1], out[0], out[1]);
}
GJX_API void gjx_split(const uint32_t key[2], uint32_t n, uint32_t* out) {
  for (uint32_t j = 0; j < n; ++j) gjx_tf2x32(key[0], key[1], 0u, j, out[2 * j], out[2 * j + 1]);
}
GJX_API void gjx_fold_in(const uint32_t key[2], uint32_t data, uint32_t out[2]) {
  gjx_tf2x32(key[0], key[1], 0u, data, out[0], out[1]);
}

// ---------------------------------------------------------------------------
// stand-alone draw kernels (parity tests, generic GFI path)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(GJX_BLOCK) k_bits(uint32_t k0, uint32_t k1, uint64_t offset,
                                                    int64_t n, uint32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * GJX_BLOCK)
    out[i] = gjx_bits_at(k0, k1, offset + (uint64_t)i);
}

__global__ void __launch_bounds__(GJX_BLOCK) k_uniform(uint32_t k0, uint32_t k1, uint64_t offset,
                                                       int64_t n, float minval, float maxval,
                                                       float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * GJX_BLOCK)
    out[i] = gjx_uniform_at(k0, k1, offset + (uint64_t)i, minval, maxval);
}

__global__ void __launch_bounds__(GJX_BLOCK) k_normal(uint32_t k0, uint32_t k1, uint64_t offset,
                                                      int64_t n, float loc, float scale,
                                                      float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * GJX_BLOCK)
    out[i] = gjx_add_rn(gjx_mul_rn(gjx_normal_at(k0, k1, offset + (uint64_t)i), scale), loc);
}

#define GJX_ARGS_1D(name)                                                      \
  GJX_REQUIRE(n >= 0, GJX_EINVAL, name ": n < 0");                             \
  GJX_REQUIRE(d_out != nullptr || n == 0, GJX_EINVAL, name ": d_out is null"); \
  if (n == 0) return GJX_OK;

GJX_API int32_t gjx_random_bits_u32(const uint32_t key[2], uint64_t offset, int64_t n,
                                    uint32_t* d_out, void* stream) {
  GJX_ARGS_1D("gjx_random_bits_u32");
  k_bits<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(key[0], key[1], offset, n, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_random_uniform_f32(const uint32_t key[2], uint64_t offset, int64_t n,
                                       float minval, float maxval, float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_random_uniform_f32");
  k_uniform<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(key[0], key[1], offset, n,
                                                                     minval, maxval, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_random_normal_f32(const uint32_t key[2], uint64_t offset, int64_t n, float loc,
                                      float scale, float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_random_normal_f32");
  k_normal<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(key[0], key[1], offset, n, loc,
                                                                    scale, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// element-wise log densities (TFP formulas)
// ---------------------------------------------------------------------------
#define GJX_GRID_STRIDE(i, n)                                                \
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < (n);   \
       i += (int64_t)gridDim.x * GJX_BLOCK)

__global__ void __launch_bounds__(GJX_BLOCK)
    k_normal_logpdf(const float* __restrict__ x, const float* __restrict__ mu, int ms,
                    const float* __restrict__ sg, int ss, int64_t n, float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    float s = sg[i * ss], m = mu[i * ms];
    float z = x[i] / s - m / s;
    out[i] = -0.5f * (z * z) - (HALF_LOG_2PI_F + logf(s));
  }
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_uniform_logpdf(const float* __restrict__ x, const float* __restrict__ lo, int ls,
                     const float* __restrict__ hi, int hs, int64_t n, float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    float l = lo[i * ls], h = hi[i * hs], v = x[i];
    out[i] = (v < l || v > h) ? -__builtin_inff() : -logf(h - l);
  }
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_bernoulli_logpdf(const float* __restrict__ x, const float* __restrict__ p, int ps, int64_t n,
                       float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    float v = x[i], q = p[i * ps];
    float a = (v == 0.0f) ? 0.0f : v * logf(q);
    float b = (1.0f - v == 0.0f) ? 0.0f : (1.0f - v) * log1pf(-q);
    out[i] = a + b;
  }
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_beta_logpdf(const float* __restrict__ x, const float* __restrict__ a, int as,
                  const float* __restrict__ b, int bs, int64_t n, float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    float v = x[i], al = a[i * as], be = b[i * bs];
    float lbeta = lgammaf(al) + lgammaf(be) - lgammaf(al + be);
    float t1 = (al - 1.0f == 0.0f) ? 0.0f : (al - 1.0f) * logf(v);
    float t2 = (be - 1.0f == 0.0f) ? 0.0f : (be - 1.0f) * log1pf(-v);
    out[i] = (v < 0.0f || v > 1.0f) ? -__builtin_inff() : t1 + t2 - lbeta;
  }
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_categorical_logpdf(const int32_t* __restrict__ x, const float* __restrict__ logits, int ls,
                         int64_t n, int k, float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    const float* l = logits + (int64_t)i * ls * k;
    float m = l[0];
    for (int j = 1; j < k; ++j) m = fmaxf(m, l[j]);
    float s = 0.0f;
    for (int j = 0; j < k; ++j) s += expf(l[j] - m);
    int xi = x[i];
    xi = xi < 0 ? 0 : (xi >= k ? k - 1 : xi);
    out[i] = (l[xi] - m) - logf(s);
  }
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_mvnormal_logpdf(const float* __restrict__ x, const float* __restrict__ mu, int ms,
                      const float* __restrict__ cov, int cs, int64_t n, int d,
                      float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    float L[64], z[8];
    const float* C = cov + (int64_t)i * cs * d * d;
    const float* M = mu + (int64_t)i * ms * d;
    for (int r = 0; r < d; ++r)
      for (int c = 0; c <= r; ++c) {
        float s = C[r * d + c];
        for (int k = 0; k < c; ++k) s -= L[r * d + k] * L[c * d + k];
        L[r * d + c] = (r == c) ? sqrtf(s) : s / L[c * d + c];
      }
    float q = 0.0f, logdet = 0.0f;
    for (int r = 0; r < d; ++r) {
      float s = x[i * d + r] - M[r];
      for (int k = 0; k < r; ++k) s -= L[r * d + k] * z[k];
      z[r] = s / L[r * d + r];
      q += z[r] * z[r];
      logdet += logf(L[r * d + r]);
    }
    out[i] = -0.5f * q - logdet - (float)d * HALF_LOG_2PI_F;
  }
}

#define GJX_STRIDE_OK(s) ((s) == 0 || (s) == 1)

GJX_API int32_t gjx_normal_logpdf_f32(const float* d_x, const float* d_mu, int32_t mu_stride,
                                      const float* d_sigma, int32_t sigma_stride, int64_t n,
                                      float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_normal_logpdf_f32");
  GJX_REQUIRE(d_x && d_mu && d_sigma, GJX_EINVAL, "gjx_normal_logpdf_f32: null input");
  GJX_REQUIRE(GJX_STRIDE_OK(mu_stride) && GJX_STRIDE_OK(sigma_stride), GJX_EINVAL,
              "gjx_normal_logpdf_f32: strides must be 0 or 1");
  k_normal_logpdf<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      d_x, d_mu, mu_stride, d_sigma, sigma_stride, n, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_uniform_logpdf_f32(const float* d_x, const float* d_lo, int32_t lo_stride,
                                       const float* d_hi, int32_t hi_stride, int64_t n,
                                       float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_uniform_logpdf_f32");
  GJX_REQUIRE(d_x && d_lo && d_hi, GJX_EINVAL, "gjx_uniform_logpdf_f32: null input");
  GJX_REQUIRE(GJX_STRIDE_OK(lo_stride) && GJX_STRIDE_OK(hi_stride), GJX_EINVAL,
              "gjx_uniform_logpdf_f32: strides must be 0 or 1");
  k_uniform_logpdf<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      d_x, d_lo, lo_stride, d_hi, hi_stride, n, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_bernoulli_logpdf_f32(const float* d_x, const float* d_p, int32_t p_stride,
                                         int64_t n, float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_bernoulli_logpdf_f32");
  GJX_REQUIRE(d_x && d_p, GJX_EINVAL, "gjx_bernoulli_logpdf_f32: null input");
  GJX_REQUIRE(GJX_STRIDE_OK(p_stride), GJX_EINVAL, "gjx_bernoulli_logpdf_f32: bad stride");
  k_bernoulli_logpdf<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(d_x, d_p, p_stride,
                                                                              n, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_beta_logpdf_f32(const float* d_x, const float* d_a, int32_t a_stride,
                                    const float* d_b, int32_t b_stride, int64_t n, float* d_out,
                                    void* stream) {
  GJX_ARGS_1D("gjx_beta_logpdf_f32");
  GJX_REQUIRE(d_x && d_a && d_b, GJX_EINVAL, "gjx_beta_logpdf_f32: null input");
  GJX_REQUIRE(GJX_STRIDE_OK(a_stride) && GJX_STRIDE_OK(b_stride), GJX_EINVAL,
              "gjx_beta_logpdf_f32: strides must be 0 or 1");
  k_beta_logpdf<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(d_x, d_a, a_stride, d_b,
                                                                         b_stride, n, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_categorical_logpdf_f32(const int32_t* d_x, const float* d_logits,
                                           int32_t logits_stride, int64_t n, int32_t k,
                                           float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_categorical_logpdf_f32");
  GJX_REQUIRE(d_x && d_logits && k > 0, GJX_EINVAL, "gjx_categorical_logpdf_f32: bad input");
  GJX_REQUIRE(GJX_STRIDE_OK(logits_stride), GJX_EINVAL, "gjx_categorical_logpdf_f32: bad stride");
  k_categorical_logpdf<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      d_x, d_logits, logits_stride, n, k, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_mvnormal_logpdf_f32(const float* d_x, const float* d_mu, int32_t mu_stride,
                                        const float* d_cov, int32_t cov_stride, int64_t n,
                                        int32_t d, float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_mvnormal_logpdf_f32");
  GJX_REQUIRE(d_x && d_mu && d_cov, GJX_EINVAL, "gjx_mvnormal_logpdf_f32: null input");
  GJX_REQUIRE(d >= 1 && d <= 8, GJX_ELIMIT, "gjx_mvnormal_logpdf_f32: d=%d not in [1, 8]", d);
  GJX_REQUIRE(GJX_STRIDE_OK(mu_stride) && GJX_STRIDE_OK(cov_stride), GJX_EINVAL,
              "gjx_mvnormal_logpdf_f32: strides must be 0 or 1");
  k_mvnormal_logpdf<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      d_x, d_mu, mu_stride, d_cov, cov_stride, n, d, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// samplers with column parameters (generic GFI path)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(GJX_BLOCK)
    k_normal_cols(uint32_t k0, uint32_t k1, uint64_t offset, int64_t n, const float* __restrict__ loc,
                  int ls, const float* __restrict__ scale, int ss, float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n)
    out[i] = gjx_add_rn(gjx_mul_rn(gjx_normal_at(k0, k1, offset + (uint64_t)i), scale[i * ss]), loc[i * ls]);
}

// tfd.Uniform.sample: low + (high - low) * uniform(key)
__global__ void __launch_bounds__(GJX_BLOCK)
    k_uniform_cols(uint32_t k0, uint32_t k1, uint64_t offset, int64_t n, const float* __restrict__ lo,
                   int ls, const float* __restrict__ hi, int hs, float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    const float u = gjx_uniform_at(k0, k1, offset + (uint64_t)i, 0.0f, 1.0f);
    const float l = lo[i * ls], h = hi[i * hs];
    out[i] = gjx_add_rn(l, gjx_mul_rn(h - l, u));
  }
}

// tfd.Bernoulli.sample: uniform(key) < probs
__global__ void __launch_bounds__(GJX_BLOCK)
    k_bernoulli(uint32_t k0, uint32_t k1, uint64_t offset, int64_t n, const float* __restrict__ p,
                int ps, float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    const float u = gjx_uniform_at(k0, k1, offset + (uint64_t)i, 0.0f, 1.0f);
    out[i] = (u < p[i * ps]) ? 1.0f : 0.0f;
  }
}

// tfd.Categorical.sample = argmax(logits[..., None] + gumbel(key, logits.shape + (m,)), axis=-2):
//   per-row logits (stride 1): n rows, one draw each, noise index row*k + c
//   shared logits  (stride 0): n draws, noise index c*n + j
__global__ void __launch_bounds__(GJX_BLOCK)
    k_categorical(uint32_t k0, uint32_t k1, uint64_t offset, int64_t n, const float* __restrict__ logits,
                  int ls, int k, int32_t* __restrict__ out) {
  const float tiny = 1.17549435e-38f;
  GJX_GRID_STRIDE(i, n) {
    float best = 0.0f;
    int arg = 0;
    for (int c = 0; c < k; ++c) {
      const uint64_t idx = ls ? ((uint64_t)(offset + i) * (uint64_t)k + (uint64_t)c)
                              : ((uint64_t)c * (uint64_t)n + (uint64_t)i + offset);
      const float u = gjx_uniform_at(k0, k1, idx, tiny, 1.0f);
      const float v = logits[(int64_t)i * ls * k + c] + (-logf(-logf(u)));
      if (c == 0 || v > best) {
        best = v;
        arg = c;
      }
    }
    out[i] = arg;
  }
}

// MultivariateNormalFullCovariance.sample: mu + chol(cov) @ normal(key, shape + [d])
__global__ void __launch_bounds__(GJX_BLOCK)
    k_mvnormal_sample(uint32_t k0, uint32_t k1, uint64_t offset, int64_t n, int d,
                      const float* __restrict__ mu, int ms, const float* __restrict__ cov, int cs,
                      float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    float L[64], z[8];
    const float* C = cov + (int64_t)i * cs * d * d;
    const float* M = mu + (int64_t)i * ms * d;
    for (int r = 0; r < d; ++r)
      for (int c = 0; c <= r; ++c) {
        float s = C[r * d + c];
        for (int q = 0; q < c; ++q) s -= L[r * d + q] * L[c * d + q];
        L[r * d + c] = (r == c) ? sqrtf(s) : s / L[c * d + c];
      }
    for (int r = 0; r < d; ++r) z[r] = gjx_normal_at(k0, k1, (offset + (uint64_t)i) * (uint64_t)d + (uint64_t)r);
    for (int r = 0; r < d; ++r) {
      float s = 0.0f;
      for (int q = 0; q <= r; ++q) s += L[r * d + q] * z[q];
      out[i * d + r] = M[r] + s;
    }
  }
}

GJX_API int32_t gjx_random_normal_cols_f32(const uint32_t key[2], uint64_t offset, int64_t n,
                                           const float* d_loc, int32_t loc_stride,
                                           const float* d_scale, int32_t scale_stride, float* d_out,
                                           void* stream) {
  GJX_ARGS_1D("gjx_random_normal_cols_f32");
  GJX_REQUIRE(d_loc && d_scale, GJX_EINVAL, "gjx_random_normal_cols_f32: null parameter");
  GJX_REQUIRE(GJX_STRIDE_OK(loc_stride) && GJX_STRIDE_OK(scale_stride), GJX_EINVAL,
              "gjx_random_normal_cols_f32: strides must be 0 or 1");
  k_normal_cols<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      key[0], key[1], offset, n, d_loc, loc_stride, d_scale, scale_stride, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_random_uniform_cols_f32(const uint32_t key[2], uint64_t offset, int64_t n,
                                            const float* d_lo, int32_t lo_stride, const float* d_hi,
                                            int32_t hi_stride, float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_random_uniform_cols_f32");
  GJX_REQUIRE(d_lo && d_hi, GJX_EINVAL, "gjx_random_uniform_cols_f32: null parameter");
  GJX_REQUIRE(GJX_STRIDE_OK(lo_stride) && GJX_STRIDE_OK(hi_stride), GJX_EINVAL,
              "gjx_random_uniform_cols_f32: strides must be 0 or 1");
  k_uniform_cols<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      key[0], key[1], offset, n, d_lo, lo_stride, d_hi, hi_stride, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_random_bernoulli_f32(const uint32_t key[2], uint64_t offset, int64_t n,
                                         const float* d_p, int32_t p_stride, float* d_out,
                                         void* stream) {
  GJX_ARGS_1D("gjx_random_bernoulli_f32");
  GJX_REQUIRE(d_p && GJX_STRIDE_OK(p_stride), GJX_EINVAL, "gjx_random_bernoulli_f32: bad parameter");
  k_bernoulli<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(key[0], key[1], offset, n, d_p,
                                                                       p_stride, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_random_categorical_i32(const uint32_t key[2], uint64_t offset, int64_t n,
                                           const float* d_logits, int32_t logits_stride, int32_t k,
                                           int32_t* d_out, void* stream) {
  GJX_ARGS_1D("gjx_random_categorical_i32");
  GJX_REQUIRE(d_logits && k > 0 && GJX_STRIDE_OK(logits_stride), GJX_EINVAL,
              "gjx_random_categorical_i32: bad parameter");
  k_categorical<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      key[0], key[1], offset, n, d_logits, logits_stride, k, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_random_mvnormal_f32(const uint32_t key[2], uint64_t offset, int64_t n, int32_t d,
                                        const float* d_mu, int32_t mu_stride, const float* d_cov,
                                        int32_t cov_stride, float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_random_mvnormal_f32");
  GJX_REQUIRE(d_mu && d_cov, GJX_EINVAL, "gjx_random_mvnormal_f32: null parameter");
  GJX_REQUIRE(d >= 1 && d <= 8, GJX_ELIMIT, "gjx_random_mvnormal_f32: d=%d not in [1, 8]", d);
  GJX_REQUIRE(GJX_STRIDE_OK(mu_stride) && GJX_STRIDE_OK(cov_stride), GJX_EINVAL,
              "gjx_random_mvnormal_f32: strides must be 0 or 1");
  k_mvnormal_sample<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      key[0], key[1], offset, n, d, d_mu, mu_stride, d_cov, cov_stride, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// Metropolis-Hastings accept / select for the generic column path (mcmc.py:293-305):
//   log_alpha = jnp.minimum(0, w);  accept = log(uniform(key)[i]) < log_alpha;
//   leaf = select(accept, new_leaf, old_leaf)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(GJX_BLOCK)
    k_mh_accept(uint32_t k0, uint32_t k1, uint64_t offset, int64_t n, const float* __restrict__ logw,
                int ws, uint8_t* __restrict__ accept) {
  GJX_GRID_STRIDE(i, n) {
    const float u = gjx_uniform_at(k0, k1, offset + (uint64_t)i, 0.0f, 1.0f);
    accept[i] = (logf(u) < gjx_min0(logw[i * ws])) ? (uint8_t)1 : (uint8_t)0;
  }
}

// out[i * inner + j] = accept[i] ? a[...] : b[...]   (rows of `inner` 4-byte words)
__global__ void __launch_bounds__(GJX_BLOCK)
    k_select_rows(const uint8_t* __restrict__ accept, const uint32_t* __restrict__ a,
                  const uint32_t* __restrict__ b, int64_t n, int64_t inner,
                  uint32_t* __restrict__ out) {
  const int64_t total = n * inner;
  GJX_GRID_STRIDE(e, total) {
    const int64_t i = e / inner;
    out[e] = accept[i] ? a[e] : b[e];
  }
}

GJX_API int32_t gjx_mh_accept(const uint32_t key[2], uint64_t offset, int64_t n,
                              const float* d_log_weight, int32_t weight_stride, uint8_t* d_out,
                              void* stream) {
  GJX_ARGS_1D("gjx_mh_accept");
  GJX_REQUIRE(key && d_log_weight && GJX_STRIDE_OK(weight_stride), GJX_EINVAL,
              "gjx_mh_accept: bad parameter");
  k_mh_accept<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(key[0], key[1], offset, n,
                                                                       d_log_weight, weight_stride, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_select_rows_b32(const uint8_t* d_accept, const void* d_new, const void* d_old,
                                    int64_t n, int64_t inner, void* d_out, void* stream) {
  GJX_ARGS_1D("gjx_select_rows_b32");
  GJX_REQUIRE(d_accept && d_new && d_old && inner >= 1, GJX_EINVAL, "gjx_select_rows_b32: bad parameter");
  k_select_rows<<<gjx_grid_for(n * inner), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      d_accept, (const uint32_t*)d_new, (const uint32_t*)d_old, n, inner, (uint32_t*)d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}
