// gjx_common.h -- shared host/device helpers for libgjx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "../../include/gjx.h"

#define GJX_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------
void gjx_set_error(const char* fmt, ...);

#define GJX_CHECK_HIP(expr)                                                            \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      gjx_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,                 \
                    hipGetErrorString(_e));                                            \
      return GJX_EHIP;                                                                 \
    }                                                                                  \
  } while (0)

#define GJX_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) {                   \
      gjx_set_error(__VA_ARGS__);    \
      return (code);                 \
    }                                \
  } while (0)

#define GJX_LAUNCH_CHECK() GJX_CHECK_HIP(hipGetLastError())

// ---------------------------------------------------------------------------
// context: per-GPU workspace
// ---------------------------------------------------------------------------
constexpr int GJX_BLOCK = 256;          // 4 waves of 64
constexpr int GJX_MAX_GRID = 256 * 8;   // 256 CUs x 8 blocks: grid-stride beyond this
constexpr int GJX_HEAVY_GRID = 256 * 128;  // VALU-bound kernels: fine-grained blocks balance better (measured)
constexpr int GJX_MAX_PARTIALS = 65536;    // >= the largest grid of a kernel that writes block maxima
constexpr int GJX_SCAN_ITEMS = 8;       // elements per thread in scan tiles
constexpr int GJX_SCAN_TILE = GJX_BLOCK * GJX_SCAN_ITEMS;  // 2048 = 2^11
constexpr int GJX_FOLD_THREADS = 1024;  // the one-block folds / scans: 16 waves hide the load latency
constexpr int GJX_MAX_RANKS = 64;       // ranks of a sharded run (= GJX_MAX_PIECES of include/gjx.h)

struct gjx_ctx {
  int device;
  int64_t n_max;
  float* d_partials;     // GJX_MAX_PARTIALS floats: block maxima (k_max_partial, extend / move epilogues)
  float* d_tile_sums;    // 2 floats per 2048-tile of n_max: {sum exp, sum exp^2} of the normaliser
  float* d_scalars;      // 16 floats: max / sums hand-over between passes; [8] = the running-maximum cell of the
                         // fused step (-inf between steps: extend raises it, the normaliser's fold kernel resets it)
  float* d_scan_lvl[4];  // tile totals (level l has ceil(n / TILE^(l+1)) entries), then scans
  float* d_scan_out[4];
  int64_t scan_cap[4];
  float* d_tmp_cdf;      // n_max floats (cdf when the caller does not ask for it)
  int32_t* d_tmp_anc;    // n_max int32: ancestors of the fused smc-step entry points
  // optional stage timing of the fused smc-step entry points (gjx_ctx_set_stage_events): events
  // recorded on the launch stream at the stage boundaries, GJX_N_STAGE_EVENTS of them or none
  hipEvent_t stage_ev[8];
  int n_stage_ev;
};

constexpr int GJX_MAX_CELL = 8;         // index of the running-maximum cell in gjx_ctx::d_scalars
constexpr int GJX_N_STAGE_EVENTS = 7;
// boundaries: 0 before extend | 1 after extend (block maxima included) | 2 after the normaliser |
// 3 before the scan | 4 after the scan | 5 after the search = before the moves | 6 after the moves
static inline hipError_t gjx_stage_mark(gjx_ctx* ctx, int k, hipStream_t s) {
  return (ctx->n_stage_ev == GJX_N_STAGE_EVENTS) ? hipEventRecord(ctx->stage_ev[k], s) : hipSuccess;
}

// Device address of the CURRENT device's copy of the library's static scratch array
// (gjx_resample.hip; the workspace of gjx_resample_systematic_logw_static), resolved with
// hipGetSymbolAddress once per device.
int32_t gjx_static_scratch(float** out);

// The smc step's lax.cond (ESS < N // 2, smc.py:747-753) taken on the device by the normaliser's fold kernel:
// go[0] = take, range = {0, take ? n : 0}, lml += lse - log N when taken, trace2 = {ess, take}.  go == nullptr: off.
struct GjxDecide {
  int32_t* go = nullptr;
  int64_t* range = nullptr;
  int64_t n = 0, half_n = 0;
  float log_n = 0.0f;
  float* lml = nullptr;
  float* trace2 = nullptr;
};
int32_t gjx_normalise_decide(gjx_ctx* ctx, const float* d_logw, int64_t n, const float* d_partials, int n_partials,
                             float* d_out5, const GjxDecide& dec, hipStream_t s);

// k_max_final over n_partials block maxima -> ctx->d_scalars[0], then the sum passes of the
// normaliser over d_logw -> d_out5 = {max, lse, ess, sum exp, sum exp^2} (gjx_reduce.hip).
int32_t gjx_normalise_from_max_partials(gjx_ctx* ctx, const float* d_logw, int64_t n, const float* d_partials,
                                        int n_partials, float* d_out5, hipStream_t s);

// gjx_loc_extend_scores plus the block maxima of the new log-weights (gjx_loc.hip)
int32_t gjx_loc_extend_max(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                           int32_t is_initial, uint64_t offset, int64_t n, const float* obs, const float* d_px,
                           const float* d_py, const float* d_pth, const float* d_logw_in, float* d_x,
                           float* d_y, float* d_th, float* d_mx, float* d_my, float* d_mth, float* d_logw,
                           float* d_ray_scores, int64_t scores_stride, float* d_max_partials,
                           int* n_partials, float* d_max_cell, void* stream);

// gjx_loc_shard_extend whose shard maximum is raised in *d_max1 by the extend kernel's own atomics (no fold launch):
// *d_max1 must hold -inf on entry (gjx_smc.hip)
int32_t gjx_loc_shard_extend_cell(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2], int32_t is_initial,
                                  uint64_t offset, int64_t n, const float* obs, const float* d_px, const float* d_py,
                                  const float* d_pth, const float* d_logw_in, float* d_x, float* d_y, float* d_th,
                                  float* d_mx, float* d_my, float* d_mth, float* d_logw, float* d_max1,
                                  float* d_ray_scores, int64_t scores_stride, void* stream);

// gjx_shard_stats_fold that also puts *d_reset_cell back to -inf (gjx_reduce.hip)
int32_t gjx_shard_search_only(const float* d_coarse, int32_t lc, const float* d_cdf_ext, int32_t n_ranks, int32_t rank,
                              int64_t n, float u, int64_t cand_cap, int64_t* d_range, int32_t* d_cand_idx,
                              int32_t* d_counts, void* stream);
int32_t gjx_shard_stats_fold_reset(const float* d_tiles_all, int64_t n_tiles_all, const float* d_max_all, int32_t n_ranks,
                                   float* d_stats5, float* d_reset_cell, void* stream);

// k_max_final over n_partials block maxima -> d_out1 (gjx_reduce.hip)
int32_t gjx_fold_max_partials(const float* d_partials, int n_partials, float* d_out1, hipStream_t s);

// Sharded step, receiving side (gjx_resample.hip): received rows -> the halo [n, n + n_recv) of the six
// columns; d_anc[jl] = local index (own particle, or n + row) of the ancestor of position rank n + jl.
int32_t gjx_shard_expand(const int32_t* d_cand_idx, const int64_t* d_range, const float* d_rows, int64_t n_recv,
                         int32_t rank, int64_t n, float* const* d_cols6, int64_t halo_cap, int32_t* d_anc,
                         int32_t* d_err, hipStream_t s);

// The same for the overlapped step: rows -> halo, ancestors of the positions [0, p0) and [p1, n) only, into a
// buffer of their own (gjx_resample.hip)
int32_t gjx_shard_expand_fix(const float* d_rows, int64_t n_recv, int32_t rank, int64_t n, int64_t p0, int64_t p1,
                             float* const* d_cols6, int64_t halo_cap, int32_t* d_anc_fix, int32_t* d_err,
                             hipStream_t s);

// The mh launches of the overlapped step (gjx_loc.hip).  Main: d_split = {p0 + 1, p1, go, flags} on the device (one
// GPU: {1, n, go, flags}, every position own-sourced; flags |= 2 when a position disagrees with the run), ancestors
// from the search (d_idx, d_range), offset = the shard's first global index.  Fix-up: d_split null, ancestors from
// d_anc for the positions [0, p0) and [p1, ...).  Neither zeroes the acceptance counters.
int32_t gjx_loc_mh_overlap(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t (*keys)[4][2], int32_t n_moves,
                           int32_t sel_mask, int32_t is_initial, uint64_t offset, int64_t n_work, const float* obs,
                           const int32_t* d_anc, const float* const* d_src6, float* const* d_dst6,
                           const float* d_ray_scores, int64_t scores_stride, float* d_logw_reset,
                           int64_t* d_accept_counts, int32_t* d_split, const int32_t* d_idx, const int64_t* d_range,
                           int64_t p0, int64_t p1, void* stream);

// gjx_loc_ray_scores with the rows `stride` floats apart (gjx_loc.hip)
int32_t gjx_loc_ray_scores_strided(const gjx_loc_world* world, const float* obs, const float* d_x, const float* d_y,
                                   const float* d_th, int64_t n, float* d_ray_scores, int64_t stride, void* stream);

// gjx_resample_systematic_logw under a decision taken on the device (gjx_resample.hip)
int32_t gjx_resample_systematic_logw_if(gjx_ctx* ctx, const float* d_logw, const float* d_lse, int64_t n, float u,
                                        const int32_t* d_go, const int64_t* d_range, int32_t* d_ancestors,
                                        hipStream_t s);

static inline int gjx_grid_for(int64_t n) {
  int64_t b = (n + GJX_BLOCK - 1) / GJX_BLOCK;
  if (b < 1) b = 1;
  return (int)(b > GJX_MAX_GRID ? GJX_MAX_GRID : b);
}

// Grid of the VALU-bound per-particle kernels (extend / mh / hmc).  Measured on MI355X at
// N = 2^24 (k_loc_mh, 10 moves): 1280 blocks 4.99 ms, 2048 4.64, 4096 4.43, 8192 4.32,
// 16384 4.26, 65536 4.27 -- many short blocks let the dispatcher even out the tail.  Re-measured with
// the gather and the stored scores in the launch: 8192 2.89, 16384 2.85, 32768 2.82, 65536 2.82 ms.
// The environment variable GJX_HEAVY_GRID overrides the cap (tuning only).
static inline int gjx_grid_heavy(int64_t n) {
  static int env_cap = -1;
  if (env_cap < 0) {
    const char* e = getenv("GJX_HEAVY_GRID");
    env_cap = e ? atoi(e) : 0;
  }
  int cap = env_cap > 0 ? env_cap : GJX_HEAVY_GRID;
  if (cap > 65536) cap = 65536;
  int64_t b = (n + GJX_BLOCK - 1) / GJX_BLOCK;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}

// Acceptance counts: wave ballot -> LDS counter per move -> ONE 64-bit atomic add per
// (block, move) into the caller's counters (zeroed by the entry point before the launch).
// Integer adds commute, so the totals are deterministic.
__device__ __forceinline__ void gjx_count_accepts(uint32_t* lds_counts, int move, bool accept) {
  const unsigned long long b = __ballot(accept);
  if ((threadIdx.x & 63) == 0) atomicAdd(&lds_counts[move], (uint32_t)__popcll(b));
}
__device__ __forceinline__ void gjx_flush_accepts(const uint32_t* lds_counts, int n_moves,
                                                  int64_t* accept_counts) {
  __syncthreads();
  if (accept_counts != nullptr && (int)threadIdx.x < n_moves)
    atomicAdd(reinterpret_cast<unsigned long long*>(accept_counts) + threadIdx.x,
              (unsigned long long)lds_counts[threadIdx.x]);
}

// ---------------------------------------------------------------------------
// Threefry-2x32 (20 rounds) and jax.random layouts -- device + host
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t gjx_rotl(uint32_t x, int r) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_rotateleft32(x, r);  // v_alignbit_b32
#else
  return (x << r) | (x >> (32 - r));
#endif
}

#define GJX_TF_ROUND(r)   \
  x0 += x1;               \
  x1 = gjx_rotl(x1, (r)); \
  x1 ^= x0;

__host__ __device__ __forceinline__ void gjx_tf2x32(uint32_t k0, uint32_t k1, uint32_t c0,
                                                    uint32_t c1, uint32_t& o0, uint32_t& o1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  uint32_t x0 = c0 + k0, x1 = c1 + k1;
  GJX_TF_ROUND(13) GJX_TF_ROUND(15) GJX_TF_ROUND(26) GJX_TF_ROUND(6)
  x0 += k1; x1 += k2 + 1u;
  GJX_TF_ROUND(17) GJX_TF_ROUND(29) GJX_TF_ROUND(16) GJX_TF_ROUND(24)
  x0 += k2; x1 += k0 + 2u;
  GJX_TF_ROUND(13) GJX_TF_ROUND(15) GJX_TF_ROUND(26) GJX_TF_ROUND(6)
  x0 += k0; x1 += k1 + 3u;
  GJX_TF_ROUND(17) GJX_TF_ROUND(29) GJX_TF_ROUND(16) GJX_TF_ROUND(24)
  x0 += k1; x1 += k2 + 4u;
  GJX_TF_ROUND(13) GJX_TF_ROUND(15) GJX_TF_ROUND(26) GJX_TF_ROUND(6)
  x0 += k2; x1 += k0 + 5u;
  o0 = x0;
  o1 = x1;
}

// Unfused IEEE f32 multiply / add / subtract.  HIP's __fmul_rn/__fadd_rn are plain operators
// and hipcc's default -ffp-contract=fast-honor-pragmas would fuse them into an fma; these
// helpers carry no `contract` flag, so `a * b + c` written with them rounds twice, as the
// reference's separate XLA HLO multiply and add do.
__device__ __forceinline__ float gjx_mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float gjx_add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float gjx_sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

// random_bits (32-bit) under jax_threefry_partitionable: counter = 64-bit global index.
__device__ __forceinline__ uint32_t gjx_bits_at(uint32_t k0, uint32_t k1, uint64_t i) {
  uint32_t a, b;
  gjx_tf2x32(k0, k1, (uint32_t)(i >> 32), (uint32_t)i, a, b);
  return a ^ b;
}

__device__ __forceinline__ float gjx_u01(uint32_t bits) {
  return __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
}

// jax.random.uniform: max(minval, u01 * (maxval - minval) + minval), unfused like XLA's HLO.
__device__ __forceinline__ float gjx_uniform_at(uint32_t k0, uint32_t k1, uint64_t i, float minval,
                                                float maxval) {
  float f = gjx_u01(gjx_bits_at(k0, k1, i));
  float v = gjx_add_rn(gjx_mul_rn(f, maxval - minval), minval);
  return fmaxf(minval, v);
}

// XLA ErfInv32 (Giles' single-precision polynomial) on w = -log1p(-x*x).  As in the reference
// x*x is ROUNDED to f32 before the logarithm (that rounding is amplified by 1/(1-x^2) in the
// tails, so a fused 1 - x*x would drift from the reference there); 1 - x2 is then exact for
// x2 >= 0.5 and within 2^-25 below, and the logarithm is v_log_f32.
__device__ __forceinline__ float gjx_erfinv(float x) {
  const float x2 = gjx_mul_rn(x, x);
  float w = -0.69314718055994530942f * __builtin_amdgcn_logf(gjx_sub_rn(1.0f, x2));
  float p;
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f;
    p = fmaf(p, w, 3.43273939e-07f);
    p = fmaf(p, w, -3.5233877e-06f);
    p = fmaf(p, w, -4.39150654e-06f);
    p = fmaf(p, w, 0.00021858087f);
    p = fmaf(p, w, -0.00125372503f);
    p = fmaf(p, w, -0.00417768164f);
    p = fmaf(p, w, 0.246640727f);
    p = fmaf(p, w, 1.50140941f);
  } else {
    w = sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = fmaf(p, w, 0.000100950558f);
    p = fmaf(p, w, 0.00134934322f);
    p = fmaf(p, w, -0.00367342844f);
    p = fmaf(p, w, 0.00573950773f);
    p = fmaf(p, w, -0.0076224613f);
    p = fmaf(p, w, 0.00943887047f);
    p = fmaf(p, w, 1.00167406f);
    p = fmaf(p, w, 2.83297682f);
  }
  float r = p * x;
  return (fabsf(x) == 1.0f) ? x * __builtin_inff() : r;
}

// jax.random.normal: sqrt(2) * erf_inv(uniform(minval = nextafter(-1, 0), maxval = 1)).
__device__ __forceinline__ float gjx_normal_at(uint32_t k0, uint32_t k1, uint64_t i) {
  const float lo = -0.99999994f;  // nextafter(-1, 0)
  float f = gjx_u01(gjx_bits_at(k0, k1, i));
  float u = fmaxf(lo, gjx_add_rn(gjx_mul_rn(f, 2.0f), lo));  // (1 - lo) rounds to 2.0f
  return 1.41421356237309504880f * gjx_erfinv(u);
}

// ---------------------------------------------------------------------------
// wave64 / block reductions (fixed tree => deterministic)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float gjx_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int gjx_wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float gjx_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Sum over the wave by DPP (no LDS traffic), valid in LANE 63 only.  Association: xor-1, xor-2 butterflies inside
// the quads, then the two quads of an 8-group, the two 8-groups of a 16-lane row (mirrors: every lane of a group
// already holds the group's sum, so the mirrored partner's value IS the other group's sum), then rows (1 + 0) and
// (3 + 2), then (3 + 2) + (1 + 0).
template <int CTRL>
__device__ __forceinline__ float gjx_dpp_add_all(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float gjx_dpp_add_rows(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v),
                                                                   CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float gjx_wave_total63(float v) {
  v = gjx_dpp_add_all<0xB1>(v);         // quad_perm [1,0,3,2]
  v = gjx_dpp_add_all<0x4E>(v);         // quad_perm [2,3,0,1]
  v = gjx_dpp_add_all<0x141>(v);        // row_half_mirror
  v = gjx_dpp_add_all<0x140>(v);        // row_mirror
  v = gjx_dpp_add_rows<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = gjx_dpp_add_rows<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

// The same association in double precision (two 32-bit DPP moves per step).
template <int CTRL, int ROWMASK, bool ALL>
__device__ __forceinline__ double gjx_dpp_add_f64(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const int lo = (int)(uint32_t)b, hi = (int)(uint32_t)(b >> 32);
  const int plo = ALL ? __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xf, false);
  const int phi = ALL ? __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xf, false);
  return v + __builtin_bit_cast(double, (unsigned long long)(uint32_t)plo | ((unsigned long long)(uint32_t)phi << 32));
}
__device__ __forceinline__ double gjx_wave_total63_f64(double v) {
  v = gjx_dpp_add_f64<0xB1, 0xf, true>(v);
  v = gjx_dpp_add_f64<0x4E, 0xf, true>(v);
  v = gjx_dpp_add_f64<0x141, 0xf, true>(v);
  v = gjx_dpp_add_f64<0x140, 0xf, true>(v);
  v = gjx_dpp_add_f64<0x142, 0xa, false>(v);
  v = gjx_dpp_add_f64<0x143, 0xc, false>(v);
  return v;
}

// max(*cell, v) -> *cell for a cell that starts at -inf and only grows within a launch: floats order like their
// bits as signed integers when the sign bit is clear and like their bits as unsigned integers, reversed, when it
// is set, so one integer atomic does it; max is exact in any order, the result does not depend on the order of
// arrival.  The plain pre-read may be stale, i.e. too small: the only cost is an atomic that changes nothing.
// NaN never enters (fmaxf drops it too).
__device__ __forceinline__ void gjx_atomic_max_f32(float* cell, float v) {
  if (!(v == v)) return;
  const float cur = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v < cur || (v == cur && v != 0.0f)) return;
  const uint32_t b = __float_as_uint(v);
  if (b >> 31)
    atomicMin(reinterpret_cast<unsigned int*>(cell), b);
  else
    atomicMax(reinterpret_cast<int*>(cell), (int)b);
}

#define HALF_LOG_2PI_F 0.9189385332046727f

// jnp.minimum(0.0, x) (mcmc.py:293-296, 563-566): unlike fminf it PROPAGATES NaN, so a NaN
// acceptance ratio rejects (log_u < nan is False).
__device__ __forceinline__ float gjx_min0(float x) { return (x > 0.0f) ? 0.0f : x; }
