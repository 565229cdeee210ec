// gjx_generic.hip -- kernels of the generic (column-at-a-time) GFI path that go beyond plain draws
// and log-densities: Beta draws, Bernoulli(logits), the log-density GRADIENTS that generic hmc
// differentiates through (mcmc.py:434-578 with jax.grad of assess, mcmc.py:59-88, 508, 523), and the
// few per-particle column operations the inference surface needs (so that no per-particle
// arithmetic runs outside this library).
#include "gjx_common.h"

#define GJX_GRID_STRIDE(i, n)                                              \
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < (n); \
       i += (int64_t)gridDim.x * GJX_BLOCK)
#define GJX_STRIDE_OK(s) ((s) == 0 || (s) == 1)
#define GJX_ARGS_1D(name)                                                      \
  GJX_REQUIRE(n >= 0, GJX_EINVAL, name ": n < 0");                             \
  GJX_REQUIRE(d_out != nullptr || n == 0, GJX_EINVAL, name ": d_out is null"); \
  if (n == 0) return GJX_OK;

// ---------------------------------------------------------------------------
// beta(a, b).sample  (distributions.py:62-65).  [3P] tfd.Beta draws two gammas with a
// data-dependent rejection sampler whose stream is not reproduced; this is the same law on
// counter-based draws: Ga / (Ga + Gb) with Marsaglia-Tsang gammas, element i on the counters
// (i << 20) + (stream << 16) + ... of the site key (as gjx_outlier_init draws its rate).
// ---------------------------------------------------------------------------
__device__ float mt_gamma_at(uint32_t k0, uint32_t k1, uint64_t base, uint64_t stream, float shape) {
  const float a = shape < 1.0f ? shape + 1.0f : shape;
  const float d = a - 1.0f / 3.0f, c = 1.0f / sqrtf(9.0f * d);
  float g = d;
  for (uint64_t t = 0; t < 64; ++t) {
    const float x = gjx_normal_at(k0, k1, base + (stream << 16) + 2 * t);
    const float u = gjx_uniform_at(k0, k1, base + (stream << 16) + 2 * t + 1, 0.0f, 1.0f);
    float v = 1.0f + c * x;
    if (v <= 0.0f) continue;
    v = v * v * v;
    g = d * v;
    if (logf(u) < 0.5f * x * x + d - d * v + d * logf(v)) break;
  }
  if (shape < 1.0f) {
    const float u = gjx_uniform_at(k0, k1, base + (stream << 16) + 200, 0.0f, 1.0f);
    g = g * powf(u, 1.0f / shape);
  }
  return g;
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_random_beta(uint32_t k0, uint32_t k1, uint64_t offset, int64_t n, const float* __restrict__ a, int as,
                  const float* __restrict__ b, int bs, float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    const uint64_t base = (offset + (uint64_t)i) << 20;
    const float ga = mt_gamma_at(k0, k1, base, 0, a[i * as]), gb = mt_gamma_at(k0, k1, base, 1, b[i * bs]);
    out[i] = ga / (ga + gb);
  }
}

GJX_API int32_t gjx_random_beta_f32(const uint32_t key[2], uint64_t offset, int64_t n, const float* d_a,
                                    int32_t a_stride, const float* d_b, int32_t b_stride, float* d_out,
                                    void* stream) {
  GJX_ARGS_1D("gjx_random_beta_f32");
  GJX_REQUIRE(key && d_a && d_b && GJX_STRIDE_OK(a_stride) && GJX_STRIDE_OK(b_stride), GJX_EINVAL,
              "gjx_random_beta_f32: bad parameter");
  GJX_REQUIRE(offset + (uint64_t)n < ((uint64_t)1 << 44), GJX_ELIMIT, "gjx_random_beta_f32: index beyond 2^44");
  k_random_beta<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(key[0], key[1], offset, n, d_a, a_stride,
                                                                        d_b, b_stride, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// bernoulli(logits)  (distributions.py:22-25; [3P] tfd.Bernoulli(logits=...)):
//   log_prob(x) = -x softplus(-l) - (1 - x) softplus(l)   (multiply_no_nan), finite for any |l|
//   sample      = uniform(key) < sigmoid(l)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float softplusf(float l) { return log1pf(expf(-fabsf(l))) + fmaxf(l, 0.0f); }

__global__ void __launch_bounds__(GJX_BLOCK)
    k_bernoulli_logits_logpdf(const float* __restrict__ x, const float* __restrict__ logits, int ls, int64_t n,
                              float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    const float v = x[i], l = logits[i * ls];
    const float a = (v == 0.0f) ? 0.0f : v * softplusf(-l);
    const float b = (1.0f - v == 0.0f) ? 0.0f : (1.0f - v) * softplusf(l);
    out[i] = -a - b;
  }
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_random_bernoulli_logits(uint32_t k0, uint32_t k1, uint64_t offset, int64_t n,
                              const float* __restrict__ logits, int ls, float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    const float l = logits[i * ls];
    const float p = 1.0f / (1.0f + expf(-l));
    out[i] = gjx_uniform_at(k0, k1, offset + (uint64_t)i, 0.0f, 1.0f) < p ? 1.0f : 0.0f;
  }
}

GJX_API int32_t gjx_bernoulli_logits_logpdf_f32(const float* d_x, const float* d_logits, int32_t logits_stride,
                                                int64_t n, float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_bernoulli_logits_logpdf_f32");
  GJX_REQUIRE(d_x && d_logits && GJX_STRIDE_OK(logits_stride), GJX_EINVAL,
              "gjx_bernoulli_logits_logpdf_f32: bad parameter");
  k_bernoulli_logits_logpdf<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(d_x, d_logits, logits_stride, n,
                                                                                    d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_random_bernoulli_logits_f32(const uint32_t key[2], uint64_t offset, int64_t n,
                                                const float* d_logits, int32_t logits_stride, float* d_out,
                                                void* stream) {
  GJX_ARGS_1D("gjx_random_bernoulli_logits_f32");
  GJX_REQUIRE(key && d_logits && GJX_STRIDE_OK(logits_stride), GJX_EINVAL,
              "gjx_random_bernoulli_logits_f32: bad parameter");
  k_random_bernoulli_logits<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(key[0], key[1], offset, n,
                                                                                    d_logits, logits_stride, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// log-density gradients (what jax.grad produces for the log_prob of each distribution): every
// output is the upstream gradient g[i] times the partial derivative; outputs are nullable.
//   kind 0 normal(x; mu, sigma): d/dx = -(x - mu) / sigma^2, d/dmu = -d/dx, d/dsigma = (z^2 - 1) / sigma
//   kind 1 uniform(x; lo, hi):   d/dx = 0, d/dlo = 1 / (hi - lo), d/dhi = -1 / (hi - lo)  (inside the support)
//   kind 2 beta(x; a, b):        d/dx = (a - 1) / x - (b - 1) / (1 - x);  d/da = log x - psi(a) + psi(a + b),
//                                d/db = log1p(-x) - psi(b) + psi(a + b)
//   kind 3 bernoulli(x; p):      d/dp = x / p - (1 - x) / (1 - p)   (p is parameter 1; no d/dx)
// ---------------------------------------------------------------------------
__device__ float digammaf_dev(float x) {
  float r = 0.0f;
  while (x < 6.0f) {   // recurrence up, then the asymptotic series
    r -= 1.0f / x;
    x += 1.0f;
  }
  const float f = 1.0f / (x * x);
  return r + logf(x) - 0.5f / x - f * (1.0f / 12.0f - f * (1.0f / 120.0f - f * (1.0f / 252.0f)));
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_logpdf_grad(int kind, const float* __restrict__ x, const float* __restrict__ p1, int s1,
                  const float* __restrict__ p2, int s2, const float* __restrict__ g, int gs, int64_t n,
                  float* __restrict__ gx, float* __restrict__ g1, float* __restrict__ g2) {
  GJX_GRID_STRIDE(i, n) {
    const float v = x[i], a = p1[i * s1], b = p2 ? p2[i * s2] : 0.0f, up = g[i * gs];
    float dx = 0.0f, d1 = 0.0f, d2 = 0.0f;
    if (kind == 0) {
      const float inv = 1.0f / b, z = (v - a) * inv;
      dx = -z * inv;
      d1 = z * inv;
      d2 = (z * z - 1.0f) * inv;
    } else if (kind == 1) {
      const float w = 1.0f / (b - a);
      const bool in = !(v < a || v > b);
      d1 = in ? w : 0.0f;
      d2 = in ? -w : 0.0f;
    } else if (kind == 2) {
      dx = ((a - 1.0f == 0.0f) ? 0.0f : (a - 1.0f) / v) - ((b - 1.0f == 0.0f) ? 0.0f : (b - 1.0f) / (1.0f - v));
      if (g1 || g2) {
        const float pab = digammaf_dev(a + b);
        d1 = logf(v) - digammaf_dev(a) + pab;
        d2 = log1pf(-v) - digammaf_dev(b) + pab;
      }
    } else {
      d1 = ((v == 0.0f) ? 0.0f : v / a) - ((1.0f - v == 0.0f) ? 0.0f : (1.0f - v) / (1.0f - a));
    }
    if (gx) gx[i] = up * dx;
    if (g1) g1[i] = up * d1;
    if (g2) g2[i] = up * d2;
  }
}

GJX_API int32_t gjx_logpdf_grad_f32(int32_t kind, const float* d_x, const float* d_p1, int32_t p1_stride,
                                    const float* d_p2, int32_t p2_stride, const float* d_upstream,
                                    int32_t upstream_stride, int64_t n, float* d_gx, float* d_g1, float* d_g2,
                                    void* stream) {
  GJX_REQUIRE(n >= 0, GJX_EINVAL, "gjx_logpdf_grad_f32: n < 0");
  if (n == 0) return GJX_OK;
  GJX_REQUIRE(kind >= 0 && kind <= 3, GJX_EINVAL, "gjx_logpdf_grad_f32: kind %d not in [0, 3]", kind);
  GJX_REQUIRE(d_x && d_p1 && (d_p2 || kind == 3) && d_upstream, GJX_EINVAL, "gjx_logpdf_grad_f32: null input");
  GJX_REQUIRE(GJX_STRIDE_OK(p1_stride) && GJX_STRIDE_OK(p2_stride) && GJX_STRIDE_OK(upstream_stride), GJX_EINVAL,
              "gjx_logpdf_grad_f32: strides must be 0 or 1");
  k_logpdf_grad<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(kind, d_x, d_p1, p1_stride, d_p2, p2_stride,
                                                                        d_upstream, upstream_stride, n, d_gx, d_g1,
                                                                        d_g2);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// per-particle column operations of the inference surface
// ---------------------------------------------------------------------------
// out = y + a * x with separately rounded multiply and add (the leapfrog updates p + (eps / 2) g and
// x + eps p of hmc, mcmc.py:515-528); x, y have n elements (stride 1) or are scalars (stride 0)
__global__ void __launch_bounds__(GJX_BLOCK)
    k_axpy(float a, const float* __restrict__ x, int xs, const float* __restrict__ y, int ys, int64_t n,
           float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) out[i] = gjx_add_rn(y[i * ys], gjx_mul_rn(a, x[i * xs]));
}

GJX_API int32_t gjx_axpy_f32(float a, const float* d_x, int32_t x_stride, const float* d_y, int32_t y_stride,
                             int64_t n, float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_axpy_f32");
  GJX_REQUIRE(d_x && d_y && GJX_STRIDE_OK(x_stride) && GJX_STRIDE_OK(y_stride), GJX_EINVAL, "gjx_axpy_f32: bad parameter");
  k_axpy<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(a, d_x, x_stride, d_y, y_stride, n, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// out = (a + b) - (c + d): hmc's log_alpha = (new_model + new_momentum) - (prev_model + prev_momentum)
// (mcmc.py:559-561); with d = null it is (a + b) - c (mala's model_weight + backward - forward, mcmc.py:414);
// with c = d = null it is a + b (tree_reduce(add) of the momentum scores)
__global__ void __launch_bounds__(GJX_BLOCK)
    k_sum_diff(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
               const float* __restrict__ d, int64_t n, float* __restrict__ out) {
  GJX_GRID_STRIDE(i, n) {
    const float s = gjx_add_rn(a[i], b[i]);
    out[i] = c ? gjx_sub_rn(s, d ? gjx_add_rn(c[i], d[i]) : c[i]) : s;
  }
}

GJX_API int32_t gjx_sum_diff_f32(const float* d_a, const float* d_b, const float* d_c, const float* d_d, int64_t n,
                                 float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_sum_diff_f32");
  GJX_REQUIRE(d_a && d_b && (d_c || !d_d), GJX_EINVAL, "gjx_sum_diff_f32: bad parameter (d without c)");
  k_sum_diff<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(d_a, d_b, d_c, d_d, n, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// out = a - *scalar: the diagnostic weights logw - logsumexp(logw) (smc.py:259-261)
__global__ void __launch_bounds__(GJX_BLOCK)
    k_sub_scalar(const float* __restrict__ a, const float* __restrict__ s, int64_t n, float* __restrict__ out) {
  const float v = s[0];
  GJX_GRID_STRIDE(i, n) out[i] = a[i] - v;
}

GJX_API int32_t gjx_sub_scalar_f32(const float* d_a, const float* d_scalar, int64_t n, float* d_out, void* stream) {
  GJX_ARGS_1D("gjx_sub_scalar_f32");
  GJX_REQUIRE(d_a && d_scalar, GJX_EINVAL, "gjx_sub_scalar_f32: null input");
  k_sub_scalar<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(d_a, d_scalar, n, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ParticleCollection.estimate (smc.py:206-245): out[k] = sum_i exp(logw[i] - *lse) v[i * inner + k] for
// k < inner: block partials in double (fixed grid, fixed order), folded by the last launch.
__global__ void __launch_bounds__(GJX_BLOCK)
    k_wsum_partial(const float* __restrict__ logw, const float* __restrict__ lse, const float* __restrict__ v,
                   int64_t n, int inner, int k, double* __restrict__ partials) {
  const float l = lse[0];
  double s = 0.0;
  GJX_GRID_STRIDE(i, n) s += (double)(expf(logw[i] - l) * v[i * inner + k]);
  __shared__ double sm[GJX_BLOCK];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = GJX_BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = sm[0];
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_wsum_final(const double* __restrict__ partials, int np, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += GJX_BLOCK) s += partials[i];
  __shared__ double sm[GJX_BLOCK];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = GJX_BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)sm[0];
}

GJX_API int32_t gjx_weighted_sum_f32(gjx_ctx* ctx, const float* d_logw, const float* d_lse, const float* d_values,
                                     int64_t n, int32_t inner, float* d_out, void* stream) {
  GJX_REQUIRE(ctx && d_logw && d_lse && d_values && d_out, GJX_EINVAL, "gjx_weighted_sum_f32: null argument");
  GJX_REQUIRE(n > 0 && inner >= 1 && inner <= 4096, GJX_EINVAL, "gjx_weighted_sum_f32: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const int grid = gjx_grid_for(n);           // <= 2048 blocks: the partials fit ctx->d_tile_sums' first 16 KiB
  double* partials = reinterpret_cast<double*>(ctx->d_partials);   // GJX_MAX_PARTIALS floats = 32768 doubles
  for (int k = 0; k < inner; ++k) {
    k_wsum_partial<<<grid, GJX_BLOCK, 0, s>>>(d_logw, d_lse, d_values, n, inner, k, partials);
    GJX_LAUNCH_CHECK();
    k_wsum_final<<<1, GJX_BLOCK, 0, s>>>(partials, grid, d_out + k);
    GJX_LAUNCH_CHECK();
  }
  return GJX_OK;
}
