// gjx_reduce.hip -- two-pass log-weight normaliser: max, then sum exp / sum exp^2.
//
// Replaces jax.scipy.special.logsumexp + effective_sample_size (smc.py:24-76 and the
// logsumexp call sites smc.py:74,127,201,232,261,623,628).  HBM-bound: 4 B/particle per pass.
// Pass structure (deterministic, no atomics):
//   k_max_partial : grid-stride float4 loads -> wave max (shuffles) -> LDS -> block partial
//                   (in the fused smc step the extend kernel writes these partials itself)
//   k_max_final   : one block folds the partials
//   k_tile_sums   : per 2048-tile sums of exp(a - M), exp(a - M)^2, fixed association, one wave per tile
//   k_fold_tiles  : one block folds the tile sums in double with one fixed association, writes lse / ESS
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

constexpr int GJX_FOLD_BLOCK = GJX_FOLD_THREADS;

__global__ void __launch_bounds__(GJX_FOLD_BLOCK)
    k_max_final(const float* __restrict__ partials, int np, float* __restrict__ out) {
  float m = -__builtin_inff();
  const int np4 = np >> 2;   // partials come from hipMalloc: 16-byte aligned
  const float4* p4 = reinterpret_cast<const float4*>(partials);
  for (int i = threadIdx.x; i < np4; i += GJX_FOLD_BLOCK) {
    const float4 v = p4[i];
    m = fmaxf(fmaxf(m, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
  }
  for (int i = (np4 << 2) + threadIdx.x; i < np; i += GJX_FOLD_BLOCK) m = fmaxf(m, partials[i]);
  __shared__ float sm[GJX_FOLD_BLOCK / 64];
  m = gjx_wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = sm[0];
    for (int w = 1; w < GJX_FOLD_BLOCK / 64; ++w) r = fmaxf(r, sm[w]);
    out[0] = r;
  }
}

// ---------------------------------------------------------------------------
// pass 2: sum exp(a - M) and sum exp(a - M)^2 with ONE association for any launch shape and
// any sharding of the particle axis.  The unit is the 2048-element tile of the GLOBAL index
// (shards start at multiples of 2048): a tile's two sums are a fixed tree over its elements
// (8 per lane pairwise, gjx_wave_total63 over each 512-element quarter, ((q0 + q1) + (q2 + q3)) over the quarters),
// relative to the GLOBAL maximum M; the tile sums of the whole particle set are then folded in
// double in global tile order by one block (k_fold_tiles).  A rank of a sharded run computes
// the tiles it owns and all-gathers them, so lse and ESS have the same bits for 1, 2, 4, 8 GPUs.
// ---------------------------------------------------------------------------
// max over d_max[0 .. n_max): the maxima of the ranks of a sharded run, or block maxima straight from the
// kernel that wrote the log-weights (a few thousand at most).  Block-cooperative; max is exact in any order.
template <int THREADS>
__device__ __forceinline__ float fold_maxima(const float* __restrict__ d_max, int n_max, float* sm /* THREADS/64 + 1 */) {
  if (n_max == 1) return d_max[0];
  float m = -__builtin_inff();
  for (int i = threadIdx.x; i < n_max; i += THREADS) m = fmaxf(m, d_max[i]);
  m = gjx_wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = sm[0];
    for (int w = 1; w < THREADS / 64; ++w) r = fmaxf(r, sm[w]);
    sm[THREADS / 64] = r;
  }
  __syncthreads();
  return sm[THREADS / 64];
}

// {lse, ESS, ...} from the two folded sums: shared by the stand-alone fold kernel and the in-kernel fold.
// ESS follows the reference's formulation, 1 / sum_i exp(a_i - lse)^2 (smc.py:74-76) =
// exp(2 (lse - M)) / sum_i exp(a_i - M)^2 with lse the ROUNDED f32 value: at |lse| ~ 1e3 the rounding of lse
// moves the reference's ESS by ~1e-4 relative, and the resampling decision ESS < N // 2 has to fall the same way.
__device__ __forceinline__ void finish_stats(double d1, double d2, float m, float* __restrict__ out2,
                                             float* __restrict__ out5, const GjxDecide& dec) {
  const float f1 = (float)d1, f2 = (float)d2;
  if (out2) {
    out2[0] = f1;
    out2[1] = f2;
  }
  if (out5) {
    const float mm = (fabsf(m) < __builtin_inff()) ? m : 0.0f;
    const float lse = logf(f1) + mm;               // logsumexp
    const float ess = (float)(exp(2.0 * ((double)lse - (double)mm)) / (double)f2);
    out5[0] = m;
    out5[1] = lse;
    out5[2] = ess;
    out5[3] = f1;
    out5[4] = f2;
    if (dec.go) {   // the lax.cond of the smc step decided here (gjx_lgssm_smc_step)
      const bool take = (double)ess < (double)dec.half_n;   // python compares the f32 ESS with an int: exact in double
      dec.go[0] = take ? 1 : 0;
      dec.range[0] = 0;
      dec.range[1] = take ? dec.n : 0;
      if (take) dec.lml[0] = gjx_add_rn(dec.lml[0], gjx_sub_rn(lse, dec.log_n));
      dec.trace2[0] = ess;
      dec.trace2[1] = take ? 1.0f : 0.0f;
    }
  }
}

#ifndef GJX_FOLD_INFLIGHT
#define GJX_FOLD_INFLIGHT 8    // 8-byte loads a fold thread keeps in flight: one round for 2^24 particles; 32 spills at 1024 threads
#endif
// The fold of ALL tile sums in double, ONE association for any block size that divides 1024: virtual thread v of
// 1024 takes tiles v, v + 1024, ... in order; the 64 virtual threads of virtual wave w = v / 64 are summed by
// gjx_wave_total63_f64; the 16 wave sums are added in order, W0 + W1 + ... + W15, by one thread.  The in-kernel
// fold of the one-device normaliser (256 threads, four virtual threads each) and the stand-alone fold of the
// sharded one (1024 threads) give the same bits.  One barrier; all loads of a thread in flight together.
// Returns the two sums in thread 0.  sm: 2 x 16 doubles.
template <int THREADS>
__device__ __forceinline__ void fold_tiles_1024(const float* __restrict__ tiles, int64_t ntiles, double (*sm)[16],
                                                double& out1, double& out2) {
  const float2* t2 = reinterpret_cast<const float2*>(tiles);
  constexpr int J = GJX_FOLD_THREADS / THREADS, U = GJX_FOLD_INFLIGHT / J;   // J virtual threads per thread, U loads in flight for each
  double s1[J], s2[J];
#pragma unroll
  for (int j = 0; j < J; ++j) s1[j] = s2[j] = 0.0;
  for (int64_t base = 0; base < ntiles; base += (int64_t)U * GJX_FOLD_THREADS) {
    float2 t[J][U];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t i = base + (int64_t)k * GJX_FOLD_THREADS + threadIdx.x + j * THREADS;
        t[j][k] = (i < ntiles) ? t2[i] : make_float2(0.0f, 0.0f);
      }
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int k = 0; k < U; ++k)
        if (base + (int64_t)k * GJX_FOLD_THREADS + threadIdx.x + j * THREADS < ntiles) {   // in tile order per virtual thread
          s1[j] += (double)t[j][k].x;
          s2[j] += (double)t[j][k].y;
        }
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    s1[j] = gjx_wave_total63_f64(s1[j]);
    s2[j] = gjx_wave_total63_f64(s2[j]);
    if ((threadIdx.x & 63) == 63) {
      sm[0][(threadIdx.x >> 6) + j * (THREADS / 64)] = s1[j];
      sm[1][(threadIdx.x >> 6) + j * (THREADS / 64)] = s2[j];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a1 = sm[0][0], a2 = sm[1][0];
    for (int w = 1; w < 16; ++w) {
      a1 += sm[0][w];
      a2 += sm[1][w];
    }
    out1 = a1;
    out2 = a2;
  }
}

// One WAVE per tile: the lane loads its 8 elements of each of the tile's four 512-element quarters up front (eight
// 16-byte loads in flight per lane, 8 KB per wave), sums them pairwise, the quarters are summed over the wave by
// DPP (gjx_wave_total63: no LDS, no barrier) and combined as (q0 + q1) + (q2 + q3); lane 63 stores the tile's
// two sums as one 8-byte word.
// (Folding the tile sums inside this launch -- the last block to arrive on a counter does it behind an agent-scope
// acquire -- was built and measured: 27.1 us against 20.1 us for this kernel plus the fold launch at n = 2^24,
// benchmarks/tools/ubench/hbm_stages.hip: the arrival atomic, the acquire and the write-through tile stores cost more
// than the launch boundary they replace.  Not kept.)
__global__ void __launch_bounds__(GJX_BLOCK)
    k_tile_sums(const float* __restrict__ a, int64_t n, const float* __restrict__ d_max, int n_max,
                float* __restrict__ tiles /* 2 per tile */) {
  __shared__ float smx[GJX_BLOCK / 64 + 1];
  float m = fold_maxima<GJX_BLOCK>(d_max, n_max, smx);
  if (!(fabsf(m) < __builtin_inff())) m = 0.0f;   // logsumexp: amax = where(isfinite(amax), amax, 0)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ntiles = (n + GJX_SCAN_TILE - 1) / GJX_SCAN_TILE;
  const bool vec = (((uintptr_t)a) & 15u) == 0;
  for (int64_t tile = (int64_t)blockIdx.x * (GJX_BLOCK / 64) + wave; tile < ntiles;
       tile += (int64_t)gridDim.x * (GJX_BLOCK / 64)) {
    const int64_t t0 = tile * GJX_SCAN_TILE;
    float e[4][GJX_SCAN_ITEMS];
    if (vec && t0 + GJX_SCAN_TILE <= n) {
      float4 u[4], v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* p = a + t0 + q * (GJX_SCAN_TILE / 4) + lane * GJX_SCAN_ITEMS;
        u[q] = *reinterpret_cast<const float4*>(p);
        v[q] = *reinterpret_cast<const float4*>(p + 4);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        e[q][0] = u[q].x; e[q][1] = u[q].y; e[q][2] = u[q].z; e[q][3] = u[q].w;
        e[q][4] = v[q].x; e[q][5] = v[q].y; e[q][6] = v[q].z; e[q][7] = v[q].w;
#pragma unroll
        for (int k = 0; k < GJX_SCAN_ITEMS; ++k) e[q][k] = expf(e[q][k] - m);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t i0 = t0 + q * (GJX_SCAN_TILE / 4) + lane * GJX_SCAN_ITEMS;
#pragma unroll
        for (int k = 0; k < GJX_SCAN_ITEMS; ++k) e[q][k] = (i0 + k < n) ? expf(a[i0 + k] - m) : 0.0f;
      }
    }
    float s1[4], s2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* x = e[q];
      s1[q] = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
      s2[q] = ((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3])) +
              ((x[4] * x[4] + x[5] * x[5]) + (x[6] * x[6] + x[7] * x[7]));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s1[q] = gjx_wave_total63(s1[q]);
      s2[q] = gjx_wave_total63(s2[q]);
    }
    if (lane == 63)
      reinterpret_cast<float2*>(tiles)[tile] =
          make_float2((s1[0] + s1[1]) + (s1[2] + s1[3]), (s2[0] + s2[1]) + (s2[2] + s2[3]));
  }
}

// Folds the tile sums of the WHOLE particle set (fold_tiles_1024) and writes {sum exp, sum exp^2}; when out5 is
// given also {max, lse, ess, s1, s2}.  The sharded normaliser's last step, and the one-device normaliser's when
// there are more block maxima than a sum kernel's blocks should fold.
__global__ void __launch_bounds__(GJX_FOLD_BLOCK)
    k_fold_tiles(const float* __restrict__ tiles, int64_t ntiles, const float* __restrict__ d_max, int n_max,
                 float* __restrict__ out2, float* __restrict__ out5, GjxDecide dec, float* __restrict__ reset_cell) {
  __shared__ float smx[GJX_FOLD_BLOCK / 64 + 1];
  const float m = out5 ? fold_maxima<GJX_FOLD_BLOCK>(d_max, n_max, smx) : 0.0f;
  __shared__ double sm[2][16];
  double f1 = 0.0, f2 = 0.0;
  fold_tiles_1024<GJX_FOLD_BLOCK>(tiles, ntiles, sm, f1, f2);
  if (threadIdx.x == 0) {
    finish_stats(f1, f2, m, out2, out5, dec);
    // the running-maximum cell of the fused step goes back to -inf for the next step's extend: every block of the
    // sum launch has read it (that launch is complete), this block has (fold_maxima above, behind a barrier)
    if (reset_cell) *reset_cell = -__builtin_inff();
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
  k_max_final<<<1, GJX_FOLD_BLOCK, 0, s>>>(ctx->d_partials, grid, d_out1);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// Passes 2 and 3 over one device's whole particle set: tile sums relative to the maximum in
// d_max[0], then their fold.
#ifndef GJX_SUMS_GRID
#define GJX_SUMS_GRID GJX_MAX_GRID
#endif
static int sums_grid(int64_t ntiles) {
  const int64_t blocks = (ntiles + GJX_BLOCK / 64 - 1) / (GJX_BLOCK / 64);
  static int cap = 0;
  if (!cap) {
    const char* e = getenv("GJX_SUMS_GRID");   // tuning only
    cap = e ? atoi(e) : GJX_SUMS_GRID;
    if (cap < 1 || cap > 65535) cap = GJX_SUMS_GRID;
  }
  return (int)(blocks > cap ? cap : (blocks < 1 ? 1 : blocks));
}

static int32_t sumexp_impl(gjx_ctx* ctx, const float* d_a, const float* d_max1, int n_max, int64_t n, float* d_out2,
                           float* d_out5, hipStream_t s, const GjxDecide& dec = GjxDecide{},
                           float* d_reset_cell = nullptr) {
  GJX_REQUIRE(n <= ctx->n_max, GJX_ELIMIT, "normaliser: n=%lld exceeds the context's n_max=%lld", (long long)n,
              (long long)ctx->n_max);
  const int64_t ntiles = (n + GJX_SCAN_TILE - 1) / GJX_SCAN_TILE;
  k_tile_sums<<<sums_grid(ntiles), GJX_BLOCK, 0, s>>>(d_a, n, d_max1, n_max, ctx->d_tile_sums);
  GJX_LAUNCH_CHECK();
  k_fold_tiles<<<1, GJX_FOLD_BLOCK, 0, s>>>(ctx->d_tile_sums, ntiles, d_max1, n_max, d_out2, d_out5, dec, d_reset_cell);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_sumexp_f32(gjx_ctx* ctx, const float* d_a, const float* d_max1, int64_t n,
                               float* d_out2, void* stream) {
  GJX_REQUIRE(ctx && d_a && d_max1 && d_out2, GJX_EINVAL, "gjx_sumexp_f32: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_sumexp_f32: n must be positive (got %lld)", (long long)n);
  return sumexp_impl(ctx, d_a, d_max1, 1, n, d_out2, nullptr, (hipStream_t)stream);
}

int32_t gjx_fold_max_partials(const float* d_partials, int n_partials, float* d_out1, hipStream_t s) {
  k_max_final<<<1, GJX_FOLD_BLOCK, 0, s>>>(d_partials, n_partials, d_out1);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// Sharded normaliser (genjax_amd/parallel.py): every rank all-gathers its maximum, computes the sums of
// ITS 2048-tiles relative to the global maximum (the same tiles, the same tree, the same bits as one
// device computes for them), all-gathers the tile sums, and folds all of them in global tile order:
// lse and ESS have the bits of the single-device normaliser for any number of ranks.
GJX_API int32_t gjx_shard_tile_sums(const float* d_logw, int64_t n, const float* d_max_all, int32_t n_ranks,
                                    float* d_tiles2, void* stream) {
  GJX_REQUIRE(d_logw && d_max_all && d_tiles2, GJX_EINVAL, "gjx_shard_tile_sums: null argument");
  GJX_REQUIRE(n > 0 && n % GJX_SCAN_TILE == 0, GJX_EINVAL,
              "gjx_shard_tile_sums: shard size %lld is not a positive multiple of %d", (long long)n, GJX_SCAN_TILE);
  GJX_REQUIRE(n_ranks >= 1 && n_ranks <= GJX_MAX_RANKS, GJX_EINVAL, "gjx_shard_tile_sums: bad rank count %d", n_ranks);
  const int64_t ntiles = n / GJX_SCAN_TILE;
  k_tile_sums<<<sums_grid(ntiles), GJX_BLOCK, 0, (hipStream_t)stream>>>(d_logw, n, d_max_all, n_ranks, d_tiles2);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_shard_stats_fold(const float* d_tiles_all, int64_t n_tiles_all, const float* d_max_all,
                                     int32_t n_ranks, float* d_stats5, void* stream) {
  GJX_REQUIRE(d_tiles_all && d_max_all && d_stats5, GJX_EINVAL, "gjx_shard_stats_fold: null argument");
  GJX_REQUIRE(n_tiles_all >= 1 && n_ranks >= 1 && n_ranks <= GJX_MAX_RANKS, GJX_EINVAL,
              "gjx_shard_stats_fold: bad sizes (%lld tiles, %d ranks)", (long long)n_tiles_all, n_ranks);
  k_fold_tiles<<<1, GJX_FOLD_BLOCK, 0, (hipStream_t)stream>>>(d_tiles_all, n_tiles_all, d_max_all, n_ranks, nullptr,
                                                              d_stats5, GjxDecide{}, nullptr);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

int32_t gjx_shard_stats_fold_reset(const float* d_tiles_all, int64_t n_tiles_all, const float* d_max_all, int32_t n_ranks,
                                   float* d_stats5, float* d_reset_cell, void* stream) {
  GJX_REQUIRE(d_tiles_all && d_max_all && d_stats5, GJX_EINVAL, "gjx_shard_stats_fold: null argument");
  GJX_REQUIRE(n_tiles_all >= 1 && n_ranks >= 1 && n_ranks <= GJX_MAX_RANKS, GJX_EINVAL,
              "gjx_shard_stats_fold: bad sizes (%lld tiles, %d ranks)", (long long)n_tiles_all, n_ranks);
  k_fold_tiles<<<1, GJX_FOLD_BLOCK, 0, (hipStream_t)stream>>>(d_tiles_all, n_tiles_all, d_max_all, n_ranks, nullptr,
                                                              d_stats5, GjxDecide{}, d_reset_cell);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// Folds n_partials block maxima (as written by k_max_partial or by the extend kernel's epilogue)
// into ctx->d_scalars[0], then runs passes 2 and 3.
int32_t gjx_normalise_from_max_partials(gjx_ctx* ctx, const float* d_logw, int64_t n, const float* d_partials,
                                        int n_partials, float* d_out5, hipStream_t s) {
  return gjx_normalise_decide(ctx, d_logw, n, d_partials, n_partials, d_out5, GjxDecide{}, s);
}

// The same, with the smc step's lax.cond decided in the fold kernel when dec.go is set.  A few thousand block
// maxima are folded by the sum kernels themselves (every block reads them from L2: one launch less, which is
// what a step of ~10 short launches at N = 2^20 is made of); more go through k_max_final first.
int32_t gjx_normalise_decide(gjx_ctx* ctx, const float* d_logw, int64_t n, const float* d_partials, int n_partials,
                             float* d_out5, const GjxDecide& dec, hipStream_t s) {
  // n_partials == 1 with the context's running-maximum cell: the fused step's extend raised it by atomics; the
  // fold kernel puts it back to -inf for the next step
  if (n_partials == 1 && d_partials == ctx->d_scalars + GJX_MAX_CELL)
    return sumexp_impl(ctx, d_logw, d_partials, 1, n, nullptr, d_out5, s, dec, ctx->d_scalars + GJX_MAX_CELL);
  if (n_partials <= 4096) return sumexp_impl(ctx, d_logw, d_partials, n_partials, n, nullptr, d_out5, s, dec);
  k_max_final<<<1, GJX_FOLD_BLOCK, 0, s>>>(d_partials, n_partials, ctx->d_scalars);
  GJX_LAUNCH_CHECK();
  return sumexp_impl(ctx, d_logw, ctx->d_scalars, 1, n, nullptr, d_out5, s, dec);
}

GJX_API int32_t gjx_logsumexp_ess(gjx_ctx* ctx, const float* d_logw, int64_t n, float* d_out5,
                                  void* stream) {
  GJX_REQUIRE(ctx && d_logw && d_out5, GJX_EINVAL, "gjx_logsumexp_ess: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_logsumexp_ess: n must be positive (got %lld)", (long long)n);
  hipStream_t s = (hipStream_t)stream;
  const int grid = gjx_grid_for((n + 3) / 4);
  if (aligned16(d_logw))
    k_max_partial<true><<<grid, GJX_BLOCK, 0, s>>>(d_logw, n, ctx->d_partials);
  else
    k_max_partial<false><<<grid, GJX_BLOCK, 0, s>>>(d_logw, n, ctx->d_partials);
  GJX_LAUNCH_CHECK();
  return gjx_normalise_from_max_partials(ctx, d_logw, n, ctx->d_partials, grid, d_out5, s);
}
