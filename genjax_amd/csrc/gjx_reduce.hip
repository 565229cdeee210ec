// gjx_reduce.hip -- two-pass log-weight normaliser: max, then sum exp / sum exp^2.
//
// Replaces jax.scipy.special.logsumexp + effective_sample_size (smc.py:24-76 and the
// logsumexp call sites smc.py:74,127,201,232,261,623,628).  HBM-bound: 4 B/particle per pass.
// Pass structure (deterministic, no atomics):
//   k_max_partial   : grid-stride float4 loads -> wave max (shuffles) -> LDS -> block partial
//   k_max_final     : one block folds <= 2048 partials
//   k_sumexp_partial: reads the max from a device scalar, accumulates exp(a-m), exp(a-m)^2
//   k_sum_final     : one block folds the partials in double, writes the sums
#include "gjx_common.h"

template <bool VEC>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_max_partial(const float* __restrict__ a, int64_t n, float* __restrict__ partials) {
  float m = -__builtin_inff();
  const int64_t tid = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * GJX_BLOCK;
  if (VEC) {
    const int64_t n4 = n >> 2;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    for (int64_t i = tid; i < n4; i += nthreads) {
      float4 v = a4[i];
      m = fmaxf(fmaxf(m, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads) m = fmaxf(m, a[i]);
  } else {
    for (int64_t i = tid; i < n; i += nthreads) m = fmaxf(m, a[i]);
  }
  __shared__ float sm[GJX_BLOCK / 64];
  m = gjx_wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = sm[0];
    for (int w = 1; w < GJX_BLOCK / 64; ++w) r = fmaxf(r, sm[w]);
    partials[blockIdx.x] = r;
  }
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_max_final(const float* __restrict__ partials, int np, float* __restrict__ out) {
  float m = -__builtin_inff();
  for (int i = threadIdx.x; i < np; i += GJX_BLOCK) m = fmaxf(m, partials[i]);
  __shared__ float sm[GJX_BLOCK / 64];
  m = gjx_wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = sm[0];
    for (int w = 1; w < GJX_BLOCK / 64; ++w) r = fmaxf(r, sm[w]);
    out[0] = r;
  }
}

template <bool VEC>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_sumexp_partial(const float* __restrict__ a, const float* __restrict__ d_max, int n_max,
                     float* __restrict__ max_out, int64_t n, float* __restrict__ partials) {
  // d_max: the maximum itself (n_max = 1) or the n_max block partials of k_max_partial, folded
  // here by every block (max is exact in any order) so that no separate final launch is needed;
  // block 0 publishes the folded maximum for k_sum_final.
  float m = d_max[0];
  if (n_max > 1) {
    __shared__ float smx[GJX_BLOCK / 64];
    for (int i = threadIdx.x + GJX_BLOCK; i < n_max; i += GJX_BLOCK) m = fmaxf(m, d_max[i]);
    if ((int)threadIdx.x < n_max) m = fmaxf(m, d_max[threadIdx.x]);
    m = gjx_wave_max(m);
    if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = m;
    __syncthreads();
    m = smx[0];
    for (int w = 1; w < GJX_BLOCK / 64; ++w) m = fmaxf(m, smx[w]);
  }
  if (max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) max_out[0] = m;
  // logsumexp: amax = where(isfinite(amax), amax, 0)
  if (!(fabsf(m) < __builtin_inff())) m = 0.0f;
  float s1 = 0.0f, s2 = 0.0f;
  const int64_t tid = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * GJX_BLOCK;
  if (VEC) {
    const int64_t n4 = n >> 2;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    for (int64_t i = tid; i < n4; i += nthreads) {
      float4 v = a4[i];
      float e0 = expf(v.x - m), e1 = expf(v.y - m), e2 = expf(v.z - m), e3 = expf(v.w - m);
      s1 += (e0 + e1) + (e2 + e3);
      s2 += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads) {
      float e = expf(a[i] - m);
      s1 += e;
      s2 += e * e;
    }
  } else {
    for (int64_t i = tid; i < n; i += nthreads) {
      float e = expf(a[i] - m);
      s1 += e;
      s2 += e * e;
    }
  }
  __shared__ float sm[2][GJX_BLOCK / 64];
  s1 = gjx_wave_sum(s1);
  s2 = gjx_wave_sum(s2);
  if ((threadIdx.x & 63) == 0) {
    sm[0][threadIdx.x >> 6] = s1;
    sm[1][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float r1 = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
    float r2 = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
    partials[2 * blockIdx.x] = r1;
    partials[2 * blockIdx.x + 1] = r2;
  }
}

// Folds the block partials in double (fixed order), writes {sum exp, sum exp^2}; when
// d_out5 is given also {max, lse, ess, s1, s2}.
__global__ void __launch_bounds__(GJX_BLOCK)
    k_sum_final(const float* __restrict__ partials, int np, const float* __restrict__ d_max,
                float* __restrict__ out2, float* __restrict__ out5) {
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < np; i += GJX_BLOCK) {
    s1 += (double)partials[2 * i];
    s2 += (double)partials[2 * i + 1];
  }
  __shared__ double sm[2][GJX_BLOCK];
  sm[0][threadIdx.x] = s1;
  sm[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = GJX_BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sm[0][threadIdx.x] += sm[0][threadIdx.x + o];
      sm[1][threadIdx.x] += sm[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float f1 = (float)sm[0][0], f2 = (float)sm[1][0];
    if (out2) {
      out2[0] = f1;
      out2[1] = f2;
    }
    if (out5) {
      float m = d_max[0];
      float mm = (fabsf(m) < __builtin_inff()) ? m : 0.0f;
      out5[0] = m;
      out5[1] = logf(f1) + mm;                 // logsumexp
      out5[2] = (f1 * f1) / f2;                // 1 / sum (exp(a-m)/S)^2
      out5[3] = f1;
      out5[4] = f2;
    }
  }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

GJX_API int32_t gjx_max_f32(gjx_ctx* ctx, const float* d_a, int64_t n, float* d_out1,
                            void* stream) {
  GJX_REQUIRE(ctx && d_a && d_out1, GJX_EINVAL, "gjx_max_f32: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_max_f32: n must be positive (got %lld)", (long long)n);
  hipStream_t s = (hipStream_t)stream;
  int grid = gjx_grid_for((n + 3) / 4);
  if (aligned16(d_a))
    k_max_partial<true><<<grid, GJX_BLOCK, 0, s>>>(d_a, n, ctx->d_partials);
  else
    k_max_partial<false><<<grid, GJX_BLOCK, 0, s>>>(d_a, n, ctx->d_partials);
  GJX_LAUNCH_CHECK();
  k_max_final<<<1, GJX_BLOCK, 0, s>>>(ctx->d_partials, grid, d_out1);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// d_max: the maximum (n_max = 1) or n_max block partials to fold (then d_max_out receives it).
// The sum partials live in the upper half of ctx->d_partials so that they never overwrite
// maxima other blocks are still folding.
static int32_t sumexp_impl(gjx_ctx* ctx, const float* d_a, const float* d_max, int n_max,
                           float* d_max_out, int64_t n, float* d_out2, float* d_out5,
                           hipStream_t s) {
  int grid = gjx_grid_for((n + 3) / 4);
  float* sums = ctx->d_partials + 2 * GJX_MAX_GRID;
  if (aligned16(d_a))
    k_sumexp_partial<true><<<grid, GJX_BLOCK, 0, s>>>(d_a, d_max, n_max, d_max_out, n, sums);
  else
    k_sumexp_partial<false><<<grid, GJX_BLOCK, 0, s>>>(d_a, d_max, n_max, d_max_out, n, sums);
  GJX_LAUNCH_CHECK();
  k_sum_final<<<1, GJX_BLOCK, 0, s>>>(sums, grid, n_max > 1 ? d_max_out : d_max, d_out2, d_out5);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_sumexp_f32(gjx_ctx* ctx, const float* d_a, const float* d_max1, int64_t n,
                               float* d_out2, void* stream) {
  GJX_REQUIRE(ctx && d_a && d_max1 && d_out2, GJX_EINVAL, "gjx_sumexp_f32: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_sumexp_f32: n must be positive (got %lld)", (long long)n);
  return sumexp_impl(ctx, d_a, d_max1, 1, nullptr, n, d_out2, nullptr, (hipStream_t)stream);
}

GJX_API int32_t gjx_logsumexp_ess(gjx_ctx* ctx, const float* d_logw, int64_t n, float* d_out5,
                                  void* stream) {
  GJX_REQUIRE(ctx && d_logw && d_out5, GJX_EINVAL, "gjx_logsumexp_ess: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_logsumexp_ess: n must be positive (got %lld)", (long long)n);
  // pass 1: block maxima; pass 2 folds them itself (no separate final launch), sums exp and exp^2
  hipStream_t s = (hipStream_t)stream;
  const int grid = gjx_grid_for((n + 3) / 4);
  if (aligned16(d_logw))
    k_max_partial<true><<<grid, GJX_BLOCK, 0, s>>>(d_logw, n, ctx->d_partials);
  else
    k_max_partial<false><<<grid, GJX_BLOCK, 0, s>>>(d_logw, n, ctx->d_partials);
  GJX_LAUNCH_CHECK();
  return sumexp_impl(ctx, d_logw, ctx->d_partials, grid, ctx->d_scalars, n, nullptr, d_out5, s);
}
