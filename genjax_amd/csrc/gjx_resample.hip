// gjx_resample.hip -- systematic / multinomial resampling and the ancestor gather.
//
// Reference: systematic_resample smc.py:79-136, resample_vectorized_trace smc.py:139-175.
//
// Bit-exactness contract (BASELINE north_star: "ancestor indices bit-exact given the
// reference's weights and uniforms"):
//   * jnp.cumsum lowers to lax.associative_scan(add).  Its f32 association is
//       prefix(k) = fold-left over the dyadic blocks of (k+1), MSB -> LSB,
//                   each block a balanced pairwise tree over an aligned range.
//     It depends on k only, so it is reproduced hierarchically with 2048-element tiles:
//     balanced-tree tile totals, the same scan applied recursively to the totals, and an
//     in-tile fold that starts from the scanned total of the previous tile.
//   * the resulting f32 CDF is NOT guaranteed monotone, so jnp.searchsorted's answer depends
//     on its exact procedure (fixed-depth bisection, method='scan'); k_searchsorted emulates
//     that procedure step for step instead of using a merge-path search.
// All kernels are HBM/L2-bound: totals 4 B read, apply 4 B read + 4 B write, search 4 B
// write + cached CDF probes, gather (4 + 4) B per column.
#include <pthread.h>

#include "gjx_common.h"

// ---------------------------------------------------------------------------
// w = exp(logw - lse)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(GJX_BLOCK)
    k_norm_weights(const float* __restrict__ logw, const float* __restrict__ d_lse, int64_t n,
                   float* __restrict__ w) {
  const float lse = d_lse[0];
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * GJX_BLOCK)
    w[i] = expf(logw[i] - lse);
}

GJX_API int32_t gjx_normalized_weights(const float* d_logw, const float* d_lse, int64_t n,
                                       float* d_w, void* stream) {
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(d_logw && d_lse && d_w, GJX_EINVAL, "gjx_normalized_weights: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_normalized_weights: n must not be negative");
  k_norm_weights<<<gjx_grid_for(n), GJX_BLOCK, 0, (hipStream_t)stream>>>(d_logw, d_lse, n, d_w);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// association-exact cumsum
// ---------------------------------------------------------------------------
// LDS layout of the thread-level tree: level l (256 >> l nodes) starts at 512 - (512 >> l).
__device__ __forceinline__ int lvl_off(int l) { return 512 - (512 >> l); }

struct TileRegs {
  float e[GJX_SCAN_ITEMS];
  float p01, p23, p45, p67, q0, q1, tt;
};

// the thread-level partial sums of the eight loaded elements
__device__ __forceinline__ void tile_sums(TileRegs& r) {
  r.p01 = r.e[0] + r.e[1];
  r.p23 = r.e[2] + r.e[3];
  r.p45 = r.e[4] + r.e[5];
  r.p67 = r.e[6] + r.e[7];
  r.q0 = r.p01 + r.p23;
  r.q1 = r.p45 + r.p67;
  r.tt = r.q0 + r.q1;
}

// LOGW: the input holds log-weights and the scanned values are w = exp(logw - lse)
// (k_norm_weights fused into the loads: the same expf of the same difference, so the same bits,
// without the write and two reads of a weight vector).
template <bool LOGW>
__device__ __forceinline__ void tile_load(const float* __restrict__ in, float lse, int64_t n,
                                          int64_t base, TileRegs& r) {
  const int64_t i0 = base + (int64_t)threadIdx.x * GJX_SCAN_ITEMS;
  if (i0 + GJX_SCAN_ITEMS <= n && ((((uintptr_t)(in + i0)) & 15u) == 0)) {
    const float4 a = *reinterpret_cast<const float4*>(in + i0);
    const float4 b = *reinterpret_cast<const float4*>(in + i0 + 4);
    r.e[0] = a.x; r.e[1] = a.y; r.e[2] = a.z; r.e[3] = a.w;
    r.e[4] = b.x; r.e[5] = b.y; r.e[6] = b.z; r.e[7] = b.w;
    if (LOGW) {
#pragma unroll
      for (int k = 0; k < GJX_SCAN_ITEMS; ++k) r.e[k] = expf(r.e[k] - lse);
    }
  } else {
#pragma unroll
    for (int k = 0; k < GJX_SCAN_ITEMS; ++k)
      r.e[k] = (i0 + k < n) ? (LOGW ? expf(in[i0 + k] - lse) : in[i0 + k]) : 0.0f;
  }
  tile_sums(r);
}

// Builds levels 0..8 of the balanced tree over the 256 thread totals.
__device__ __forceinline__ void tile_tree(float* tree, float tt) {
  tree[threadIdx.x] = tt;
  __syncthreads();
#pragma unroll
  for (int l = 1; l <= 8; ++l) {
    const int cnt = GJX_BLOCK >> l;
    if ((int)threadIdx.x < cnt) {
      const int src = lvl_off(l - 1) + 2 * threadIdx.x;
      tree[lvl_off(l) + threadIdx.x] = tree[src] + tree[src + 1];
    }
    __syncthreads();
  }
}

template <bool LOGW>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_scan_totals(const float* __restrict__ in, const float* __restrict__ d_lse, int64_t n,
                  float* __restrict__ totals, const int32_t* __restrict__ d_go = nullptr) {
  if (d_go && !*d_go) return;   // a resampling decided on the device (gjx_lgssm_smc_step) said no
  __shared__ float tree[512];
  const float lse = LOGW ? d_lse[0] : 0.0f;
  for (int64_t tile = blockIdx.x; tile * GJX_SCAN_TILE < n; tile += gridDim.x) {
    TileRegs r;
    tile_load<LOGW>(in, lse, n, tile * GJX_SCAN_TILE, r);
    tile_tree(tree, r.tt);
    if (threadIdx.x == 0) totals[tile] = tree[lvl_off(8)];
    __syncthreads();
  }
}

// fold of `carry` with the thread-level dyadic blocks of c (0 <= c < 256), MSB -> LSB
__device__ __forceinline__ float fold_hi(const float* tree, float carry, int c) {
  float acc = carry;
#pragma unroll
  for (int b = 7; b >= 0; --b)
    if (c & (1 << b)) acc = acc + tree[lvl_off(b) + ((c >> (b + 1)) << 1)];
  return acc;
}

// The scan of one loaded tile (base = index of its first element, gt = its GLOBAL tile id) against `scanned`, the
// inclusive scan (same association) of the tile totals indexed by global tile id, or null when the whole input is
// one tile.  Ends with a barrier: `tree` can be reused.
__device__ __forceinline__ void apply_tile(const TileRegs& r, float* tree, const float* __restrict__ scanned,
                                           int64_t gt, int64_t n, int64_t base, float* __restrict__ out) {
  tile_tree(tree, r.tt);
  // 0.0f + x == x exactly (weights carry no -0), so 0 stands for "no carry".
  const float carry = (scanned != nullptr && gt > 0) ? scanned[gt - 1] : 0.0f;
  const int i = threadIdx.x;
  const float h0 = fold_hi(tree, carry, i);
  float h1;
  if (i + 1 == GJX_BLOCK)
    h1 = (scanned != nullptr) ? scanned[gt] : tree[lvl_off(8)];
  else
    h1 = fold_hi(tree, carry, i + 1);
  float o[GJX_SCAN_ITEMS];
  o[0] = h0 + r.e[0];
  o[1] = h0 + r.p01;
  o[2] = (h0 + r.p01) + r.e[2];
  o[3] = h0 + r.q0;
  o[4] = (h0 + r.q0) + r.e[4];
  o[5] = (h0 + r.q0) + r.p45;
  o[6] = ((h0 + r.q0) + r.p45) + r.e[6];
  o[7] = h1;
  const int64_t i0 = base + (int64_t)i * GJX_SCAN_ITEMS;
  if (i0 + GJX_SCAN_ITEMS <= n && ((((uintptr_t)(out + i0)) & 15u) == 0)) {
    *reinterpret_cast<float4*>(out + i0) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(out + i0 + 4) = make_float4(o[4], o[5], o[6], o[7]);
  } else {
#pragma unroll
    for (int k = 0; k < GJX_SCAN_ITEMS; ++k)
      if (i0 + k < n) out[i0 + k] = o[k];
  }
  __syncthreads();
}

// tile_offset: global id of this launch's tile 0.
template <bool LOGW>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_scan_apply(const float* __restrict__ in, const float* __restrict__ d_lse, int64_t n,
                 const float* __restrict__ scanned, int64_t tile_offset, float* __restrict__ out,
                 const int32_t* __restrict__ d_go = nullptr) {
  if (d_go && !*d_go) return;
  __shared__ float tree[512];
  const float lse = LOGW ? d_lse[0] : 0.0f;
  for (int64_t tile = blockIdx.x; tile * GJX_SCAN_TILE < n; tile += gridDim.x) {
    TileRegs r;
    const int64_t base = tile * GJX_SCAN_TILE;
    tile_load<LOGW>(in, lse, n, base, r);
    apply_tile(r, tree, scanned, tile + tile_offset, n, base, out);
  }
}

static inline int scan_grid(int64_t ntiles) {
  return (int)(ntiles > GJX_MAX_GRID ? GJX_MAX_GRID : ntiles);
}

// Workspace of the hierarchical scan: per level the tile totals and their scan.  It lives in a
// gjx_ctx (hipMalloc'ed by gjx_ctx_create) or in the library's static scratch array
// (gjx_resample_systematic_logw_static).
struct ScanWs {
  float* lvl[4];
  float* out[4];
  int64_t cap[4];
  int64_t n_max;
};

static inline ScanWs scan_ws_of(const gjx_ctx* ctx) {
  ScanWs w;
  for (int l = 0; l < 4; ++l) {
    w.lvl[l] = ctx->d_scan_lvl[l];
    w.out[l] = ctx->d_scan_out[l];
    w.cap[l] = ctx->scan_cap[l];
  }
  w.n_max = ctx->n_max;
  return w;
}

// d_lse != nullptr (top level only): `in` holds log-weights, the scan runs over exp(in - *d_lse)
static int32_t cumsum_rec(const ScanWs& ws, const float* in, const float* d_lse, int64_t n, float* out,
                          int level, hipStream_t s, const int32_t* d_go = nullptr) {
  const int64_t ntiles = (n + GJX_SCAN_TILE - 1) / GJX_SCAN_TILE;
  if (ntiles == 1) {
    if (d_lse)
      k_scan_apply<true><<<1, GJX_BLOCK, 0, s>>>(in, d_lse, n, nullptr, 0, out, d_go);
    else
      k_scan_apply<false><<<1, GJX_BLOCK, 0, s>>>(in, nullptr, n, nullptr, 0, out, d_go);
    GJX_LAUNCH_CHECK();
    return GJX_OK;
  }
  GJX_REQUIRE(level < 4 && ws.lvl[level] && ntiles <= ws.cap[level], GJX_ELIMIT,
              "gjx_cumsum_f32: n=%lld exceeds the workspace's n_max=%lld", (long long)n,
              (long long)ws.n_max);
  float* T = ws.lvl[level];
  float* S = ws.out[level];
  if (d_lse)
    k_scan_totals<true><<<scan_grid(ntiles), GJX_BLOCK, 0, s>>>(in, d_lse, n, T, d_go);
  else
    k_scan_totals<false><<<scan_grid(ntiles), GJX_BLOCK, 0, s>>>(in, nullptr, n, T, d_go);
  GJX_LAUNCH_CHECK();
  int32_t rc = cumsum_rec(ws, T, nullptr, ntiles, S, level + 1, s, d_go);
  if (rc != GJX_OK) return rc;
  if (d_lse)
    k_scan_apply<true><<<scan_grid(ntiles), GJX_BLOCK, 0, s>>>(in, d_lse, n, S, 0, out, d_go);
  else
    k_scan_apply<false><<<scan_grid(ntiles), GJX_BLOCK, 0, s>>>(in, nullptr, n, S, 0, out, d_go);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_cumsum_f32(gjx_ctx* ctx, const float* d_w, int64_t n, float* d_cdf,
                               void* stream) {
  GJX_REQUIRE(ctx && d_w && d_cdf, GJX_EINVAL, "gjx_cumsum_f32: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_cumsum_f32: n must be positive");
  return cumsum_rec(scan_ws_of(ctx), d_w, nullptr, n, d_cdf, 0, (hipStream_t)stream);
}

GJX_API int32_t gjx_cumsum_normalized_f32(gjx_ctx* ctx, const float* d_logw, const float* d_lse,
                                          int64_t n, float* d_cdf, void* stream) {
  GJX_REQUIRE(ctx && d_logw && d_lse && d_cdf, GJX_EINVAL, "gjx_cumsum_normalized_f32: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_cumsum_normalized_f32: n must be positive");
  return cumsum_rec(scan_ws_of(ctx), d_logw, d_lse, n, d_cdf, 0, (hipStream_t)stream);
}

// Sharded form: local tile totals, then (after the caller has all-gathered the totals of all
// shards and scanned them with gjx_cumsum_f32) the in-tile fold against the global scan.
GJX_API int32_t gjx_cumsum_tile_totals(const float* d_w, int64_t n, float* d_totals,
                                       void* stream) {
  GJX_REQUIRE(d_w && d_totals, GJX_EINVAL, "gjx_cumsum_tile_totals: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_cumsum_tile_totals: n must be positive");
  const int64_t ntiles = (n + GJX_SCAN_TILE - 1) / GJX_SCAN_TILE;
  k_scan_totals<false><<<scan_grid(ntiles), GJX_BLOCK, 0, (hipStream_t)stream>>>(d_w, nullptr, n, d_totals);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_cumsum_apply(const float* d_w, int64_t n, const float* d_scanned_totals,
                                 int64_t tile_offset, float* d_cdf, void* stream) {
  GJX_REQUIRE(d_w && d_cdf && d_scanned_totals, GJX_EINVAL, "gjx_cumsum_apply: null argument");
  GJX_REQUIRE(n > 0 && tile_offset >= 0, GJX_EINVAL, "gjx_cumsum_apply: bad size");
  const int64_t ntiles = (n + GJX_SCAN_TILE - 1) / GJX_SCAN_TILE;
  k_scan_apply<false><<<scan_grid(ntiles), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      d_w, nullptr, n, d_scanned_totals, tile_offset, d_cdf);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// searchsorted (side='left', method='scan') over systematic positions
// ---------------------------------------------------------------------------
// jnp.searchsorted's fixed-depth bisection: low = 0, high = n; `levels` times
//   mid = (low + high) >> 1;  go_left = pos <= cdf[mid];  (low, high) = go_left ? (low, mid) : (mid, high)
// The chain of `levels` dependent probes is latency-bound (L2), so every thread runs
// GJX_SEARCH_ILP independent bisections side by side.
//
// Two-level form (sharded runs): the first `lc` levels probe a coarse table that holds
// cdf[mid] for every node of the bisection tree's first lc levels in heap order (node 1 = root,
// children 2i (left) and 2i + 1 (right); gjx_bisect_plan lists the mids), the remaining levels
// probe a slab cdf[slab_base .. slab_base + slab_len) that only has to cover the intervals the
// launch's positions reach after lc levels (gjx_searchsorted_window reports that range).
// Measured alternatives at N = 2^24 (0.147 ms with every position walked from the root, 4 per thread):
// probing both children speculatively (two levels per round) 0.142 ms; wave-uniform upper levels through
// scalar loads 0.35 ms (the SMEM round trip serialises the chain).
#ifndef GJX_SEARCH_ILP_N
#define GJX_SEARCH_ILP_N 8
#endif
constexpr int GJX_SEARCH_ILP = GJX_SEARCH_ILP_N;

__device__ __forceinline__ float systematic_pos(int64_t j, float u, float n_total_f) {
  // positions = (arange(n) + u) / n   (smc.py:132): int32 -> f32, add, IEEE divide
  return __fdiv_rn(gjx_add_rn((float)(int32_t)j, u), n_total_f);
}

// Where cdf[g] of a global index g lives: one contiguous slab starting at global index slab_base.
constexpr int GJX_SHARD_ROW = 7;   // floats of an exchanged row: first position + six columns

struct SlabSrc {
  const float* cdf;
  int64_t slab_base;
  __device__ __forceinline__ float at(uint32_t g) const { return cdf[(int64_t)g - slab_base]; }
};

// A wave owns GJX_SEARCH_ILP * 64 CONSECUTIVE positions.  Positions increase with j and every
// decision `pos <= cdf[mid]` is monotone in the position, so all positions between two positions
// that took the same branch took it too: the wave first walks the levels on which its FIRST and LAST
// position agree (one probe per level for the whole wave, lanes alternate between the two), and
// only the remaining levels -- log2 of the spread of the chunk's ancestors, ~9 of 25 at N = 2^24 with
// flat weights, fewer when the weights have collapsed -- are walked per position.  Same probes, same
// decisions, same result as walking every position from the root.
template <class Src>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_searchsorted(Src src, const float* __restrict__ coarse,
                   int lc, uint32_t n_cdf, float u, int64_t j0, int64_t m, float n_total_f,
                   int levels, int32_t* __restrict__ idx, const int64_t* __restrict__ d_range = nullptr) {
  if (d_range) {   // positions [d_range[0], d_range[1]) known only on the device (gjx_shard_search)
    j0 = d_range[0];
    m = d_range[1] - d_range[0];
  }
  constexpr int64_t per_wave = 64 * GJX_SEARCH_ILP;
  const int64_t per_block = (int64_t)(GJX_BLOCK / 64) * per_wave;
  const int lane = threadIdx.x & 63;
  for (int64_t b = (int64_t)blockIdx.x * per_block; b < m; b += (int64_t)gridDim.x * per_block) {
    const int64_t w0 = b + (int64_t)(threadIdx.x >> 6) * per_wave;
    // shared prefix of the chunk's first and last position
    uint32_t low0 = 0u, high0 = n_cdf, node0 = 1u;
    int l0 = 0;
    {
      int64_t je = (lane & 1) ? w0 + per_wave - 1 : w0;
      if (je >= m) je = m - 1;
      const float pe = systematic_pos(j0 + je, u, n_total_f);
      for (; l0 < levels; ++l0) {
        const uint32_t mid = (low0 + high0) >> 1;
        const float c = (l0 < lc) ? coarse[node0] : src.at(mid);
        const bool go_left = pe <= c;
        const unsigned long long bal = __ballot(go_left);
        if (bal != 0ull && bal != ~0ull) break;   // the two ends part ways here (wave-uniform)
        high0 = go_left ? mid : high0;
        low0 = go_left ? low0 : mid;
        node0 = 2u * node0 + (go_left ? 0u : 1u);
      }
    }
    float pos[GJX_SEARCH_ILP];
    uint32_t low[GJX_SEARCH_ILP], high[GJX_SEARCH_ILP], node[GJX_SEARCH_ILP];
#pragma unroll
    for (int k = 0; k < GJX_SEARCH_ILP; ++k) {
      int64_t j = w0 + (int64_t)k * 64 + lane;
      if (j >= m) j = m - 1;  // duplicate work keeps the lanes uniform; not stored
      pos[k] = systematic_pos(j0 + j, u, n_total_f);
      low[k] = low0;
      high[k] = high0;
      node[k] = node0;
    }
    for (int l = l0; l < levels; ++l) {
      float c[GJX_SEARCH_ILP];
      uint32_t mid[GJX_SEARCH_ILP];
#pragma unroll
      for (int k = 0; k < GJX_SEARCH_ILP; ++k) {
        mid[k] = (low[k] + high[k]) >> 1;
        c[k] = (l < lc) ? coarse[node[k]] : src.at(mid[k]);
      }
#pragma unroll
      for (int k = 0; k < GJX_SEARCH_ILP; ++k) {
        const bool go_left = pos[k] <= c[k];
        high[k] = go_left ? mid[k] : high[k];
        low[k] = go_left ? low[k] : mid[k];
        node[k] = 2u * node[k] + (go_left ? 0u : 1u);
      }
    }
#pragma unroll
    for (int k = 0; k < GJX_SEARCH_ILP; ++k) {
      const int64_t j = w0 + (int64_t)k * 64 + lane;
      if (j < m) idx[j] = (int32_t)(high[k] > n_cdf - 1u ? n_cdf - 1u : high[k]);  // gather clamps
    }
  }
}

// ---------------------------------------------------------------------------
// Windowed form (the whole CDF resident, single device): the same bisection, probes served from LDS.
// A WAVE owns 64 * GJX_SEARCH_ILP consecutive positions [A, Z] and works on its own (no block barrier: a wave that
// waits for memory leaves its SIMD to the others).  Phase 1: the wave walks the COMPLETE paths of A and Z, five levels
// per round (a half-wave loads the 31 nodes of the next five levels at once and resolves them with one ballot: 5
// dependent memory round trips instead of 25) -> their final intervals and go-right bit strings.
// Every decision is monotone in the position, so for A <= j <= Z:
//   * a node with mid < high_A lies on A's path where A went right (had A gone left there, high_A <= mid; had j left
//     A's path earlier -- A left, j right at a node m >= high_A -- all its later mids are >= m): j goes right;
//   * a node with mid > low_Z lies on Z's path where Z went left (mirror argument): j goes left;
//   * every other probe index lies in [high_A, low_Z]: that window is staged in the wave's slice of LDS once
//     (coalesced), with a -inf sentinel below and a +inf sentinel above, and a clamped index turns the two rules into
//     ordinary probes.
// j starts at the node where A's and Z's paths part (their common prefix is j's too) and walks the remaining
// levels -- log2(window) + 1, ~10 of 25 for flat weights, 0 when the wave's positions share one ancestor -- in LDS.
// Same probes where a probe decides anything, same decisions, same result as k_searchsorted for ANY cdf, monotone or
// not.  A window larger than GJX_WIN_CAP entries is probed in global memory under the same rules.
// Instruction count matters here (integer min / max / compare / select issue at half rate on gfx950): when n_cdf is
// a power of two every interval of level l has n_cdf >> l entries, so a position carries its low end alone, as a
// byte offset into the window -- add, clamp (one v_med3), LDS read, compare, select per level.
// ---------------------------------------------------------------------------
#ifndef GJX_WIN_CAP
#define GJX_WIN_CAP 1024
#endif

#ifndef GJX_WIN_WAVES
#define GJX_WIN_WAVES 6
#endif
template <bool POW2>
__global__ void __launch_bounds__(GJX_BLOCK, GJX_WIN_WAVES)
    k_searchsorted_win(const float* __restrict__ cdf, uint32_t n_cdf, float u, int64_t j0, int64_t m,
                       float n_total_f, int levels, int32_t* __restrict__ idx,
                       const int64_t* __restrict__ d_range) {
  if (d_range) {
    j0 = d_range[0];
    m = d_range[1] - d_range[0];
  }
  constexpr int64_t per_wave = 64 * GJX_SEARCH_ILP;
  // positions = (j + u) / n (smc.py:132).  When n is a power of two the IEEE quotient is the product with 1 / n
  // (both are the correctly rounded value of the same real number): two instructions instead of the division's twelve
  const bool n_total_pow2 = POW2 && n_total_f == (float)n_cdf;
  const float inv_n = 1.0f / n_total_f;
  auto position = [&](int64_t j) -> float {
    return n_total_pow2 ? gjx_mul_rn(gjx_add_rn((float)(int32_t)j, u), inv_n) : systematic_pos(j, u, n_total_f);
  };
  __shared__ float win_all[(GJX_BLOCK / 64) * (GJX_WIN_CAP + 2)];
  float* win = win_all + (threadIdx.x >> 6) * (GJX_WIN_CAP + 2);
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (GJX_BLOCK / 64) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (GJX_BLOCK / 64);
  for (int64_t w0 = wave * per_wave; w0 < m; w0 += n_waves * per_wave) {
    // ---- phase 1: lanes 0..31 walk A = w0, lanes 32..63 walk Z = the chunk's last position
    // (the walks stay on the vector unit: moved to scalar registers -- their states are wave-uniform per half -- the
    // kernel was 17 % slower, the CU's one scalar unit being shared by its four SIMDs)
    uint32_t high_a, low_z, path_a, path_z;
    {
      const int t = lane & 31, half = lane >> 5;
      int64_t je = half ? w0 + per_wave - 1 : w0;
      if (je >= m) je = m - 1;
      const float pe = position(j0 + je);
      uint32_t low = 0u, high = n_cdf, path = 0u;
      const int d = 31 - __clz(t | 1);   // depth of heap node t inside a five-level round (t = 0 idles)
      if (POW2) {
        // n_cdf = 2^K: the interval of level l is [low, low + 2^(K - l)), so node t of a round (depth d, i-th of its
        // depth) probes low + (i << (s - d)) + (1 << (s - d) >> 1) with s = K - l, and the five decisions move low
        // by their bits times the halved sizes -- shifts instead of the interval walk
        const int i = t - (1 << d);
        const int K = levels - 1;
        bool last_right = false;
        for (int l = 0; l < levels; l += 5) {
          const int depth = levels - l < 5 ? levels - l : 5;
          const int s_ = K - l;
          const bool live = (t >= 1) & (d < depth);
          const int sh = live ? s_ - d : 0;
          const uint32_t mid_t = low + ((uint32_t)i << sh) + ((1u << sh) >> 1);
          const float c = live ? cdf[mid_t < n_cdf ? mid_t : n_cdf - 1u] : 0.0f;
          const unsigned long long bal = __ballot(pe <= c);
          const uint32_t mask = (uint32_t)(bal >> (32 * half));
          uint32_t node = 1u;
          for (int q = 0; q < depth; ++q) node = 2u * node + (((mask >> node) & 1u) ^ 1u);
          const uint32_t bits = node - (1u << depth);          // the round's go-right bits, first level first
          path = (path << depth) | bits;
          low += s_ >= depth ? bits << (s_ - depth) : bits >> (depth - s_);   // the last level (size 1) moves nothing
          last_right = bits & 1u;
        }
        high = low + (last_right ? 1u : 0u);
      } else {
        for (int l = 0; l < levels; l += 5) {
          const int depth = levels - l < 5 ? levels - l : 5;
          uint32_t lo = low, hi = high;
          for (int k = d - 1; k >= 0; --k) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((t >> k) & 1) lo = mid; else hi = mid;
          }
          const uint32_t mid_t = (lo + hi) >> 1;
          const bool live = (t >= 1) & (d < depth);
          const float c = live ? cdf[mid_t < n_cdf ? mid_t : n_cdf - 1u] : 0.0f;
          const unsigned long long bal = __ballot(pe <= c);
          const uint32_t mask = (uint32_t)(bal >> (32 * half));
          uint32_t node = 1u;
          for (int q = 0; q < depth; ++q) {
            const bool gl = (mask >> node) & 1u;
            const uint32_t mid = (low + high) >> 1;
            high = gl ? mid : high;
            low = gl ? low : mid;
            path = (path << 1) | (gl ? 0u : 1u);
            node = 2u * node + (gl ? 0u : 1u);
          }
        }
      }
      // lane 0 holds A's result, lane 32 Z's: wave-uniform from here on (scalar registers)
      high_a = __builtin_amdgcn_readlane(high, 0);
      low_z = __builtin_amdgcn_readlane(low, 32);
      path_a = __builtin_amdgcn_readlane(path, 0);
      path_z = __builtin_amdgcn_readlane(path, 32);
    }
    const uint32_t diff = path_a ^ path_z;
    const int l0 = diff ? levels - (32 - __builtin_clz(diff)) : levels;   // leading levels on which A and Z agree
    uint32_t low0 = 0u, high0 = n_cdf;
    for (int q = 0; q < l0; ++q) {
      const uint32_t mid = (low0 + high0) >> 1;
      if ((path_a >> (levels - 1 - q)) & 1u) low0 = mid; else high0 = mid;
    }
    const int32_t base = (int32_t)high_a - 1;   // window index 0 (the -inf sentinel) in global terms
    const int32_t wd = (int32_t)low_z - (int32_t)high_a + 1;
    const int32_t w = wd > 0 ? wd : 0;          // window entries
    const bool staged = w <= GJX_WIN_CAP;
    if (staged && l0 < levels) {
      for (int i = lane; i < w; i += 64) win[1 + i] = cdf[(int64_t)base + 1 + i];
      if (lane == 0) {
        win[0] = -__builtin_inff();
        win[w + 1] = __builtin_inff();
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave's own LDS writes before its reads
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    float pos[GJX_SEARCH_ILP];
    int32_t res[GJX_SEARCH_ILP];   // relative to base
#pragma unroll
    for (int k = 0; k < GJX_SEARCH_ILP; ++k) {
      int64_t j = w0 + (int64_t)k * 64 + lane;
      if (j >= m) j = m - 1;
      pos[k] = position(j0 + j);
      res[k] = (int32_t)high0 - base;
    }
    if (POW2 && staged && l0 < levels) {
      int32_t lo4[GJX_SEARCH_ILP];
      bool gl[GJX_SEARCH_ILP];
#pragma unroll
      for (int k = 0; k < GJX_SEARCH_ILP; ++k) lo4[k] = ((int32_t)low0 - base) * 4;
      const int32_t top4 = (w + 1) * 4;
      for (int l = l0; l < levels; ++l) {
        const int32_t half4 = (int32_t)((n_cdf >> l) >> 1) * 4;   // 0 on the last level: the interval of one entry
#pragma unroll
        for (int k = 0; k < GJX_SEARCH_ILP; ++k) {
          const int32_t mid4 = lo4[k] + half4;
          int32_t r4;
          asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r4) : "v"(mid4), "v"(top4));
          const float c = *(const float*)((const char*)win + r4);
          gl[k] = pos[k] <= c;
          lo4[k] = gl[k] ? lo4[k] : mid4;
        }
      }
#pragma unroll
      for (int k = 0; k < GJX_SEARCH_ILP; ++k) res[k] = (lo4[k] >> 2) + (gl[k] ? 0 : 1);   // high after the last level
    } else if (l0 < levels) {
      int32_t lo[GJX_SEARCH_ILP];
#pragma unroll
      for (int k = 0; k < GJX_SEARCH_ILP; ++k) lo[k] = (int32_t)low0 - base;
      for (int l = l0; l < levels; ++l) {
#pragma unroll
        for (int k = 0; k < GJX_SEARCH_ILP; ++k) {
          const int32_t mid = (lo[k] + res[k]) >> 1;
          float c;
          if (staged)
            c = win[min(max(mid, 0), w + 1)];
          else
            c = mid <= 0 ? -__builtin_inff() : (mid > w ? __builtin_inff() : cdf[(int64_t)base + mid]);
          const bool gl = pos[k] <= c;
          res[k] = gl ? mid : res[k];
          lo[k] = gl ? lo[k] : mid;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < GJX_SEARCH_ILP; ++k) {
      const int64_t j = w0 + (int64_t)k * 64 + lane;
      const uint32_t h = (uint32_t)(res[k] + base);
      if (j < m) idx[j] = (int32_t)(h > n_cdf - 1u ? n_cdf - 1u : h);  // gather clamps
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // this round's LDS reads before the next round's writes
    __builtin_amdgcn_wave_barrier();
  }
}

static int bisect_levels(int64_t n_cdf) {
  int levels = 0;
  while (((int64_t)1 << levels) < n_cdf + 1) ++levels;
  return levels;
}

// GJX_SEARCH_WINDOWED=0 keeps the plain per-wave bisection for the single-device searches (A/B measurements and the
// tests that compare the two kernels); read once.
static bool search_windowed(int64_t n_cdf) {
  static const bool on = [] {
    const char* e = getenv("GJX_SEARCH_WINDOWED");
    return !(e && e[0] == '0');
  }();
  return on && n_cdf <= ((int64_t)1 << 30);   // the windowed kernel keeps its intervals as signed offsets from the window
}

static inline int search_grid(int64_t m) {
  const int64_t per_block = (int64_t)GJX_BLOCK * GJX_SEARCH_ILP;   // 4 waves x 64 x ILP positions
  int64_t b = (m + per_block - 1) / per_block;
  return (int)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

// m = the most positions the launch can have (the grid; d_range may shrink it on the device)
static void launch_search_win(const float* d_cdf, int64_t n_cdf, float u, int64_t j0, int64_t m, float n_total_f,
                              int32_t* d_idx, const int64_t* d_range, hipStream_t s) {
  const int64_t mm = d_range ? 0 : m;
  if ((n_cdf & (n_cdf - 1)) == 0)
    k_searchsorted_win<true><<<search_grid(m), GJX_BLOCK, 0, s>>>(d_cdf, (uint32_t)n_cdf, u, j0, mm, n_total_f,
                                                                  bisect_levels(n_cdf), d_idx, d_range);
  else
    k_searchsorted_win<false><<<search_grid(m), GJX_BLOCK, 0, s>>>(d_cdf, (uint32_t)n_cdf, u, j0, mm, n_total_f,
                                                                   bisect_levels(n_cdf), d_idx, d_range);
}

GJX_API int32_t gjx_searchsorted_systematic(const float* d_cdf, int64_t n_cdf, float u, int64_t j0,
                                            int64_t m, int64_t n_total, int32_t* d_idx,
                                            void* stream) {
  GJX_REQUIRE(d_cdf && d_idx, GJX_EINVAL, "gjx_searchsorted_systematic: null argument");
  GJX_REQUIRE(n_cdf > 0 && n_cdf < ((int64_t)1 << 31) && m > 0 && n_total > 0 && j0 >= 0,
              GJX_EINVAL, "gjx_searchsorted_systematic: bad sizes");
  if (search_windowed(n_cdf))
    launch_search_win(d_cdf, n_cdf, u, j0, m, (float)n_total, d_idx, nullptr, (hipStream_t)stream);
  else
    k_searchsorted<SlabSrc><<<search_grid(m), GJX_BLOCK, 0, (hipStream_t)stream>>>(
        SlabSrc{d_cdf, 0}, nullptr, 0, (uint32_t)n_cdf, u, j0, m, (float)n_total, bisect_levels(n_cdf),
        d_idx);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_bisect_levels(int64_t n_cdf) { return n_cdf > 0 ? bisect_levels(n_cdf) : 0; }

GJX_API int32_t gjx_bisect_plan(int64_t n_cdf, int32_t lc, int32_t* probe_idx) {
  GJX_REQUIRE(probe_idx && n_cdf > 0 && n_cdf < ((int64_t)1 << 31), GJX_EINVAL,
              "gjx_bisect_plan: bad argument");
  GJX_REQUIRE(lc >= 0 && lc <= 20 && lc <= bisect_levels(n_cdf), GJX_ELIMIT,
              "gjx_bisect_plan: lc=%d not in [0, min(20, levels)]", lc);
  // heap-ordered tree of (low, high) states; node i probes mid = (low + high) >> 1
  const int64_t nodes = (int64_t)1 << lc;
  uint32_t* lo = (uint32_t*)malloc(sizeof(uint32_t) * 2 * nodes);
  GJX_REQUIRE(lo != nullptr, GJX_ENOMEM, "gjx_bisect_plan: out of host memory");
  uint32_t* hi = lo + nodes;
  probe_idx[0] = 0;
  if (nodes > 1) {
    lo[1] = 0u;
    hi[1] = (uint32_t)n_cdf;
  }
  for (int64_t i = 1; i < nodes; ++i) {
    const uint32_t mid = (lo[i] + hi[i]) >> 1;
    probe_idx[i] = (int32_t)mid;
    if (2 * i + 1 < nodes) {
      lo[2 * i] = lo[i];
      hi[2 * i] = mid;
      lo[2 * i + 1] = mid;
      hi[2 * i + 1] = hi[i];
    }
  }
  free(lo);
  return GJX_OK;
}

// Range r (r < n_ranges) covers the positions [j0 + r m, j0 + (r + 1) m); out[2r], out[2r + 1] =
// {low of its first position, high of its last position} after the lc coarse levels.
__global__ void k_search_window(const float* __restrict__ coarse, int lc, uint32_t n_cdf, float u,
                                int64_t j0, int64_t m, int n_ranges, float n_total_f,
                                int64_t* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * n_ranges) return;
  const int r = t >> 1, last = t & 1;
  const int64_t j = j0 + (int64_t)r * m + (last ? m - 1 : 0);
  const float pos = systematic_pos(j, u, n_total_f);
  uint32_t low = 0u, high = n_cdf, node = 1u;
  for (int l = 0; l < lc; ++l) {
    const uint32_t mid = (low + high) >> 1;
    const bool go_left = pos <= coarse[node];
    high = go_left ? mid : high;
    low = go_left ? low : mid;
    node = 2u * node + (go_left ? 0u : 1u);
  }
  out[t] = last ? (int64_t)high : (int64_t)low;
}

GJX_API int32_t gjx_searchsorted_window(const float* d_coarse, int32_t lc, int64_t n_cdf, float u,
                                        int64_t j0, int64_t m, int32_t n_ranges, int64_t n_total,
                                        int64_t* d_out, void* stream) {
  GJX_REQUIRE(d_coarse && d_out, GJX_EINVAL, "gjx_searchsorted_window: null argument");
  GJX_REQUIRE(n_cdf > 0 && n_cdf < ((int64_t)1 << 31) && m > 0 && n_total > 0 && j0 >= 0 &&
                  lc >= 0 && lc <= bisect_levels(n_cdf) && n_ranges >= 1 && n_ranges <= 4096,
              GJX_EINVAL, "gjx_searchsorted_window: bad sizes");
  k_search_window<<<(2 * n_ranges + 63) / 64, 64, 0, (hipStream_t)stream>>>(
      d_coarse, lc, (uint32_t)n_cdf, u, j0, m, n_ranges, (float)n_total, d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_searchsorted_systematic_2l(const float* d_coarse, int32_t lc,
                                               const float* d_cdf_slab, int64_t slab_base,
                                               int64_t slab_len, int64_t n_cdf, float u,
                                               int64_t j0, int64_t m, int64_t n_total,
                                               int32_t* d_idx, void* stream) {
  GJX_REQUIRE(d_coarse && d_cdf_slab && d_idx, GJX_EINVAL,
              "gjx_searchsorted_systematic_2l: null argument");
  GJX_REQUIRE(n_cdf > 0 && n_cdf < ((int64_t)1 << 31) && m > 0 && n_total > 0 && j0 >= 0 &&
                  slab_base >= 0 && slab_len > 0 && slab_base + slab_len <= n_cdf && lc >= 0 &&
                  lc <= bisect_levels(n_cdf),
              GJX_EINVAL, "gjx_searchsorted_systematic_2l: bad sizes");
  // The caller guarantees (gjx_searchsorted_window) that after lc levels every interval of this
  // launch lies inside the slab; all later probes stay inside their interval.
  k_searchsorted<SlabSrc><<<search_grid(m), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      SlabSrc{d_cdf_slab, slab_base}, d_coarse, lc, (uint32_t)n_cdf, u, j0, m, (float)n_total,
      bisect_levels(n_cdf), d_idx);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_resample_systematic_logw(gjx_ctx* ctx, const float* d_logw, const float* d_lse,
                                             int64_t n, float u, int32_t* d_ancestors,
                                             float* d_cdf_out, void* stream) {
  GJX_REQUIRE(ctx && d_logw && d_lse && d_ancestors, GJX_EINVAL,
              "gjx_resample_systematic_logw: null argument");
  GJX_REQUIRE(n > 0 && n <= ctx->n_max, GJX_ELIMIT,
              "gjx_resample_systematic_logw: n=%lld not in (0, n_max=%lld]", (long long)n,
              (long long)ctx->n_max);
  float* cdf = d_cdf_out ? d_cdf_out : ctx->d_tmp_cdf;
  int32_t rc = gjx_cumsum_normalized_f32(ctx, d_logw, d_lse, n, cdf, stream);
  if (rc != GJX_OK) return rc;
  return gjx_searchsorted_systematic(cdf, n, u, 0, n, n, d_ancestors, stream);
}

// systematic_resample whose lax.cond (smc.py:747-753) was decided ON THE DEVICE: d_range = {0, take ? n : 0}
// and *d_go = take, both written by an earlier kernel of the stream.  When take = 0 every kernel returns at
// once and d_ancestors is left alone; the host never waits for the decision.
int32_t gjx_resample_systematic_logw_if(gjx_ctx* ctx, const float* d_logw, const float* d_lse, int64_t n, float u,
                                        const int32_t* d_go, const int64_t* d_range, int32_t* d_ancestors,
                                        hipStream_t s) {
  GJX_REQUIRE(n > 0 && n <= ctx->n_max, GJX_ELIMIT, "resampling: n=%lld not in (0, n_max=%lld]", (long long)n,
              (long long)ctx->n_max);
  int32_t rc = cumsum_rec(scan_ws_of(ctx), d_logw, d_lse, n, ctx->d_tmp_cdf, 0, s, d_go);
  if (rc != GJX_OK) return rc;
  GJX_CHECK_HIP(gjx_stage_mark(ctx, 4, s));   // scan | search boundary of the step's stage timing (off unless set)
  if (search_windowed(n))
    launch_search_win(ctx->d_tmp_cdf, n, u, 0, n, (float)n, d_ancestors, d_range, s);
  else
    k_searchsorted<SlabSrc><<<search_grid(n), GJX_BLOCK, 0, s>>>(SlabSrc{ctx->d_tmp_cdf, 0}, nullptr, 0, (uint32_t)n,
                                                                 u, 0, 0, (float)n, bisect_levels(n), d_ancestors,
                                                                 d_range);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_resample_systematic(gjx_ctx* ctx, const float* d_weights, int64_t n, float u,
                                        int32_t* d_ancestors, float* d_cdf_out, void* stream) {
  GJX_REQUIRE(ctx && d_weights && d_ancestors, GJX_EINVAL, "gjx_resample_systematic: null argument");
  GJX_REQUIRE(n > 0 && n <= ctx->n_max, GJX_ELIMIT,
              "gjx_resample_systematic: n=%lld not in (0, n_max=%lld]", (long long)n,
              (long long)ctx->n_max);
  float* cdf = d_cdf_out ? d_cdf_out : ctx->d_tmp_cdf;
  int32_t rc = gjx_cumsum_f32(ctx, d_weights, n, cdf, stream);
  if (rc != GJX_OK) return rc;
  return gjx_searchsorted_systematic(cdf, n, u, 0, n, n, d_ancestors, stream);
}

// ---------------------------------------------------------------------------
// gjx_resample_systematic_logw for callers that cannot allocate (code under HIP graph capture,
// tight serving loops, harnesses that forbid hipMalloc): the same kernels, with the CDF and the
// scan's tile totals in a static device array instead of a gjx_ctx.  Every GPU the library runs
// on holds its own copy of the array (16.0 MiB) until the process ends.
// ---------------------------------------------------------------------------
constexpr int64_t GJX_STATIC_TILES = GJX_STATIC_MAX_N / GJX_SCAN_TILE;   // 2048: one tile of totals
constexpr int64_t GJX_STATIC_FLOATS = GJX_STATIC_MAX_N + 2 * (GJX_STATIC_TILES + 64);
static __device__ __align__(256) float scratch[GJX_STATIC_FLOATS];

// `scratch` is a host-side stand-in in host code, not a device address: the array's memory is
// reached through the per-device pointer hipGetSymbolAddress returns (each GPU has its own copy).
constexpr int GJX_MAX_DEVICES = 64;
static float* g_scratch_ptr[GJX_MAX_DEVICES];   // written once per device, under g_scratch_mu
static pthread_mutex_t g_scratch_mu = PTHREAD_MUTEX_INITIALIZER;

// Address of the CURRENT device's copy of the array (resolved on the first call per device).
int32_t gjx_static_scratch(float** out) {
  int dev = -1;
  GJX_CHECK_HIP(hipGetDevice(&dev));
  GJX_REQUIRE(dev >= 0 && dev < GJX_MAX_DEVICES, GJX_ELIMIT, "gjx static scratch: device %d not in [0, %d)",
              dev, GJX_MAX_DEVICES);
  pthread_mutex_lock(&g_scratch_mu);
  float* p = g_scratch_ptr[dev];
  hipError_t e = hipSuccess;
  if (p == nullptr) {
    e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e == hipSuccess) g_scratch_ptr[dev] = p;
  }
  pthread_mutex_unlock(&g_scratch_mu);
  if (e != hipSuccess || p == nullptr) {
    gjx_set_error("gjx static scratch: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
    return GJX_EHIP;
  }
  *out = p;
  return GJX_OK;
}

GJX_API int32_t gjx_resample_systematic_logw_static(const float* d_logw, const float* d_lse, int64_t n,
                                                    float u, int32_t* d_ancestors, void* stream) {
  GJX_REQUIRE(d_logw && d_lse && d_ancestors, GJX_EINVAL,
              "gjx_resample_systematic_logw_static: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_resample_systematic_logw_static: n must be positive");
  GJX_REQUIRE(n <= GJX_STATIC_MAX_N, GJX_ELIMIT,
              "gjx_resample_systematic_logw_static: n=%lld exceeds the static workspace (GJX_STATIC_MAX_N=%lld); "
              "use gjx_resample_systematic_logw with a gjx_ctx",
              (long long)n, (long long)GJX_STATIC_MAX_N);
  float* ws_base = nullptr;
  int32_t rc = gjx_static_scratch(&ws_base);
  if (rc != GJX_OK) return rc;
  // layout: cdf[GJX_STATIC_MAX_N] | tile totals | scanned totals.  n <= 2^22 has at most 2048 tiles, whose
  // totals are ONE tile of the next level: no deeper levels.  Every element that is read is
  // written first by the same call, so nothing has to start at zero.
  float* cdf = ws_base;
  ScanWs ws;
  for (int l = 0; l < 4; ++l) ws.lvl[l] = ws.out[l] = nullptr, ws.cap[l] = 0;
  ws.lvl[0] = ws_base + GJX_STATIC_MAX_N;
  ws.out[0] = ws.lvl[0] + GJX_STATIC_TILES + 64;
  ws.cap[0] = GJX_STATIC_TILES;
  ws.n_max = GJX_STATIC_MAX_N;
  rc = cumsum_rec(ws, d_logw, d_lse, n, cdf, 0, (hipStream_t)stream);
  if (rc != GJX_OK) return rc;
  return gjx_searchsorted_systematic(cdf, n, u, 0, n, n, d_ancestors, stream);
}

// ---------------------------------------------------------------------------
// multinomial resampling: Gumbel-max over an n x m noise matrix (O(n m))
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(GJX_BLOCK)
    k_multinomial(uint32_t k0, uint32_t k1, const float* __restrict__ logw, int64_t n, int64_t m,
                  int32_t* __restrict__ idx) {
  const float tiny = 1.17549435e-38f;
  for (int64_t j = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; j < m;
       j += (int64_t)gridDim.x * GJX_BLOCK) {
    float best = 0.0f;
    int32_t arg = 0;
    for (int64_t i = 0; i < n; ++i) {
      // noise element (category i, sample j) sits at flat index i*m + j  ([3P] TFP layout)
      const float uu = gjx_uniform_at(k0, k1, (uint64_t)(i * m + j), tiny, 1.0f);
      const float g = -logf(-logf(uu));
      const float v = logw[i] + g;
      if (i == 0 || v > best) {
        best = v;
        arg = (int32_t)i;
      }
    }
    idx[j] = arg;
  }
}

GJX_API int32_t gjx_resample_multinomial(const uint32_t key[2], const float* d_logw, int64_t n,
                                         int64_t m, int32_t* d_ancestors, void* stream) {
  GJX_REQUIRE(key && d_logw && d_ancestors, GJX_EINVAL, "gjx_resample_multinomial: null argument");
  GJX_REQUIRE(n > 0 && m > 0, GJX_EINVAL, "gjx_resample_multinomial: bad sizes");
  GJX_REQUIRE(n <= (1 << 16) && m <= (1 << 16), GJX_ELIMIT,
              "gjx_resample_multinomial: the n x m Gumbel matrix is O(n*m); n, m <= 65536 "
              "(got n=%lld m=%lld) -- use systematic resampling for large n",
              (long long)n, (long long)m);
  k_multinomial<<<gjx_grid_for(m), GJX_BLOCK, 0, (hipStream_t)stream>>>(key[0], key[1], d_logw, n,
                                                                         m, d_ancestors);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// ancestor gather over SoA columns
// ---------------------------------------------------------------------------
struct GatherCols {
  const float* src[GJX_MAX_COLS];
  float* dst[GJX_MAX_COLS];
};

template <int NC>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_gather(GatherCols c, const int32_t* __restrict__ idx, int64_t idx_base, int64_t m) {
  for (int64_t j = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; j < m;
       j += (int64_t)gridDim.x * GJX_BLOCK) {
    const int64_t a = (int64_t)idx[j] - idx_base;
    float v[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) v[k] = c.src[k][a];
#pragma unroll
    for (int k = 0; k < NC; ++k) c.dst[k][j] = v[k];
  }
}

GJX_API int32_t gjx_gather_columns_f32(const int32_t* d_idx, int64_t idx_base, int64_t m,
                                       const float* const* src_cols, float* const* dst_cols,
                                       int32_t ncols, void* stream) {
  GJX_REQUIRE(d_idx && src_cols && dst_cols, GJX_EINVAL, "gjx_gather_columns_f32: null argument");
  GJX_REQUIRE(m > 0, GJX_EINVAL, "gjx_gather_columns_f32: m must be positive");
  GJX_REQUIRE(ncols >= 1 && ncols <= GJX_MAX_COLS, GJX_ELIMIT,
              "gjx_gather_columns_f32: ncols=%d not in [1, %d]", ncols, GJX_MAX_COLS);
  hipStream_t s = (hipStream_t)stream;
  int done = 0;
  while (done < ncols) {
    const int left = ncols - done;
    const int chunk = left >= 6 ? 6 : (left >= 3 ? 3 : (left == 2 ? 2 : 1));
    GatherCols c;
    for (int k = 0; k < chunk; ++k) {
      GJX_REQUIRE(src_cols[done + k] && dst_cols[done + k], GJX_EINVAL,
                  "gjx_gather_columns_f32: column %d is null", done + k);
      GJX_REQUIRE(src_cols[done + k] != dst_cols[done + k], GJX_EINVAL,
                  "gjx_gather_columns_f32: column %d gathers in place", done + k);
      c.src[k] = src_cols[done + k];
      c.dst[k] = dst_cols[done + k];
    }
    const int grid = gjx_grid_for(m);
    if (chunk == 6)
      k_gather<6><<<grid, GJX_BLOCK, 0, s>>>(c, d_idx, idx_base, m);
    else if (chunk == 3)
      k_gather<3><<<grid, GJX_BLOCK, 0, s>>>(c, d_idx, idx_base, m);
    else if (chunk == 2)
      k_gather<2><<<grid, GJX_BLOCK, 0, s>>>(c, d_idx, idx_base, m);
    else
      k_gather<1><<<grid, GJX_BLOCK, 0, s>>>(c, d_idx, idx_base, m);
    GJX_LAUNCH_CHECK();
    done += chunk;
  }
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// Sharded smc step (no counterpart in the single-device reference).
// One process per GPU holds the contiguous global range [rank n, (rank + 1) n), n a multiple of
// 2048; the host runs the collectives between these calls (genjax_amd/parallel.py).
//
// Resampling is OWNER-PUSH: positions j are increasing and every bisection decision is monotone
// in the position, so ancestors are non-decreasing in j and the positions whose ancestor lies in
// one shard form one contiguous range.  The OWNER of the ancestors finds that range, runs the
// remaining bisection levels on its own CDF slice, and sends every DISTINCT selected particle
// once per destination rank together with the first position that selected it:
//   gjx_shard_cdf_begin   tile totals of exp(logw - lse) + the raw first / last tile  -> all_gather
//   gjx_shard_cdf_finish  global-association CDF slice (+ one tile of either neighbour),
//                         this rank's share of the coarse bisection table              -> all_reduce
//   gjx_shard_search      candidate positions, their ancestors, rows per destination   -> all_gather
//   gjx_shard_pack        rows {first position, x, y, th, mx, my, mth} by destination   -> all_to_all
//   gjx_shard_expand      received rows -> halo of the columns, local ancestor indices
// A rank receives at most one 28-byte row per distinct remote ancestor of its n positions.
// ---------------------------------------------------------------------------
constexpr int GJX_STRIP = GJX_SCAN_TILE;   // CDF entries of either neighbour a rank holds

// The message of gjx_shard_cdf_begin in one launch: all blocks but the last form the tile totals (k_scan_totals'
// body: same loads, same tree), the last one copies the raw first and last tile behind them.
__device__ __forceinline__ void shard_msg_block(float* tree, int block, int n_blocks, const float* __restrict__ logw,
                                                const float* __restrict__ d_lse, int64_t n, float* __restrict__ msg) {
  const int64_t ntiles = n / GJX_SCAN_TILE;
  if (block + 1 == n_blocks) {
    float* strips = msg + ntiles;
    for (int k = threadIdx.x; k < GJX_SCAN_TILE; k += GJX_BLOCK) {
      strips[k] = logw[k];
      strips[GJX_SCAN_TILE + k] = logw[n - GJX_SCAN_TILE + k];
    }
    return;
  }
  const float lse = d_lse[0];
  for (int64_t tile = block; tile < ntiles; tile += n_blocks - 1) {
    TileRegs r;
    tile_load<true>(logw, lse, n, tile * GJX_SCAN_TILE, r);
    tile_tree(tree, r.tt);
    if (threadIdx.x == 0) msg[tile] = tree[lvl_off(8)];
    __syncthreads();
  }
}

// One record of GJX_PROBE_REC floats per coarse-table probe this rank owns (probe = node of the first lc bisection
// levels whose mid lies in the shard): what apply_tile ADDS, in order, to the carry of the probe's tile to form
// cdf[mid] -- the tree nodes of fold_hi, then the in-thread partial sums.  The carry (scan of ALL shards' tile
// totals) is not known yet; every rank redoes the few additions once it is (k_table_from_records): the same
// operands in the same order, so the same bits as the owner's k_scan_apply -- and no all_reduce of the table.
// rec[0] = number of addends (int bits; -1: the element is its tile's last, cdf[mid] = scanned[tile]).
constexpr int GJX_PROBE_REC = 12;

__device__ __forceinline__ void probe_record_block(float* tree, int j, const float* __restrict__ logw,
                                                   const float* __restrict__ d_lse, int64_t n, int64_t shard_lo,
                                                   const int32_t* __restrict__ nodes, const int32_t* __restrict__ mids,
                                                   float* __restrict__ recs) {
  if (nodes[j] == 0) return;   // padding (block-uniform)
  const int64_t local = (int64_t)mids[j] - shard_lo;
  const int64_t base = (local / GJX_SCAN_TILE) * GJX_SCAN_TILE;
  const int within = (int)(local - base), c = within / GJX_SCAN_ITEMS, k = within % GJX_SCAN_ITEMS;
  TileRegs r;
  tile_load<true>(logw, d_lse[0], n, base, r);
  tile_tree(tree, r.tt);
  if ((int)threadIdx.x != c) return;
  float* rec = recs + (int64_t)j * GJX_PROBE_REC;
  int cnt = 0;
  const int cf = k == GJX_SCAN_ITEMS - 1 ? c + 1 : c;   // o[7] = h1 = the fold of thread c + 1
  if (cf == GJX_BLOCK) {
    rec[0] = __int_as_float(-1);
    return;
  }
#pragma unroll
  for (int b = 7; b >= 0; --b)
    if (cf & (1 << b)) rec[1 + cnt++] = tree[lvl_off(b) + ((cf >> (b + 1)) << 1)];
  switch (k) {   // o[k] of apply_tile, left to right
    case 0: rec[1 + cnt++] = r.e[0]; break;
    case 1: rec[1 + cnt++] = r.p01; break;
    case 2: rec[1 + cnt++] = r.p01; rec[1 + cnt++] = r.e[2]; break;
    case 3: rec[1 + cnt++] = r.q0; break;
    case 4: rec[1 + cnt++] = r.q0; rec[1 + cnt++] = r.e[4]; break;
    case 5: rec[1 + cnt++] = r.q0; rec[1 + cnt++] = r.p45; break;
    case 6: rec[1 + cnt++] = r.q0; rec[1 + cnt++] = r.p45; rec[1 + cnt++] = r.e[6]; break;
    default: break;
  }
  rec[0] = __int_as_float(cnt);
}

// the whole message of a rank in ONE launch: blocks [0, msg_blocks) form the tile totals and copy the two raw tiles,
// the p_cap blocks behind them write the probe records (two launches until round 3: ~5 us of the step's chain)
__global__ void __launch_bounds__(GJX_BLOCK)
    k_shard_msg(const float* __restrict__ logw, const float* __restrict__ d_lse, int64_t n, float* __restrict__ msg,
                int msg_blocks, int64_t shard_lo, const int32_t* __restrict__ nodes, const int32_t* __restrict__ mids,
                float* __restrict__ recs) {
  __shared__ float tree[512];
  if ((int)blockIdx.x < msg_blocks)
    shard_msg_block(tree, (int)blockIdx.x, msg_blocks, logw, d_lse, n, msg);
  else
    probe_record_block(tree, (int)blockIdx.x - msg_blocks, logw, d_lse, n, shard_lo, nodes, mids, recs);
}

GJX_API int32_t gjx_shard_cdf_begin(const float* d_logw, const float* d_lse, int64_t n, int32_t rank,
                                    const int32_t* d_probe_nodes, const int32_t* d_probe_mids, int32_t p_cap,
                                    float* d_msg, void* stream) {
  GJX_REQUIRE(d_logw && d_lse && d_msg, GJX_EINVAL, "gjx_shard_cdf_begin: null argument");
  GJX_REQUIRE(n > 0 && n % GJX_SCAN_TILE == 0, GJX_EINVAL,
              "gjx_shard_cdf_begin: shard size %lld is not a positive multiple of %d", (long long)n,
              GJX_SCAN_TILE);
  GJX_REQUIRE(rank >= 0 && p_cap >= 0 && (p_cap == 0 || (d_probe_nodes && d_probe_mids)), GJX_EINVAL,
              "gjx_shard_cdf_begin: bad probe list");
  hipStream_t s = (hipStream_t)stream;
  const int64_t ntiles = n / GJX_SCAN_TILE;
  const int msg_blocks = scan_grid(ntiles) + 1;
  k_shard_msg<<<msg_blocks + p_cap, GJX_BLOCK, 0, s>>>(d_logw, d_lse, n, d_msg, msg_blocks, (int64_t)rank * n,
                                                       d_probe_nodes, d_probe_mids, d_msg + ntiles + 2 * GJX_STRIP);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// gjx_cumsum_f32 of the tile totals of all shards, read where the all_gather left them (rank r's nt totals at
// msg_all + r stride), by ONE block: the totals are few (<= GJX_TOTALS_ONE_BLOCK), and the launches of the
// hierarchical scan cost more than its work.  Same loads-to-sums, same trees, same folds as cumsum_rec, level by
// level: the same bits.  Also writes the compacted totals.
constexpr int64_t GJX_TOTALS_ONE_BLOCK = 16 * GJX_SCAN_TILE;

__device__ __forceinline__ void totals_load(const float* __restrict__ msg_all, int64_t nt, int64_t stride, int64_t m,
                                            int64_t base, TileRegs& r, float* __restrict__ compact) {
  const int64_t i0 = base + (int64_t)threadIdx.x * GJX_SCAN_ITEMS;
#pragma unroll
  for (int k = 0; k < GJX_SCAN_ITEMS; ++k) {
    const int64_t i = i0 + k;
    r.e[k] = i < m ? msg_all[(i / nt) * stride + (i % nt)] : 0.0f;
    if (compact && i < m) compact[i] = r.e[k];
  }
  tile_sums(r);
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_totals_scan(const float* __restrict__ msg_all, int64_t nt, int64_t stride, int64_t m,
                  float* __restrict__ all_totals, float* __restrict__ scanned) {
  __shared__ float tree[512];
  __shared__ __align__(16) float lvl_t[GJX_TOTALS_ONE_BLOCK / GJX_SCAN_TILE], lvl_s[GJX_TOTALS_ONE_BLOCK / GJX_SCAN_TILE];
  TileRegs r;
  if (m <= GJX_SCAN_TILE) {
    totals_load(msg_all, nt, stride, m, 0, r, all_totals);
    apply_tile(r, tree, nullptr, 0, m, 0, scanned);
    return;
  }
  const int64_t tiles = (m + GJX_SCAN_TILE - 1) / GJX_SCAN_TILE;
  for (int64_t t = 0; t < tiles; ++t) {            // k_scan_totals
    totals_load(msg_all, nt, stride, m, t * GJX_SCAN_TILE, r, all_totals);
    tile_tree(tree, r.tt);
    if (threadIdx.x == 0) lvl_t[t] = tree[lvl_off(8)];
    __syncthreads();
  }
  tile_load<false>(lvl_t, 0.0f, tiles, 0, r);       // the one-tile scan of the level above
  apply_tile(r, tree, nullptr, 0, tiles, 0, lvl_s);
  for (int64_t t = 0; t < tiles; ++t) {            // k_scan_apply
    totals_load(msg_all, nt, stride, m, t * GJX_SCAN_TILE, r, nullptr);
    apply_tile(r, tree, lvl_s, t, m, t * GJX_SCAN_TILE, scanned);
  }
}

// own slice, then (two more blocks) the last tile of the previous shard and the first tile of the next one, from
// their raw log-weights in the message: the same fold of the same elements against the same scanned totals
struct ProbeTable {   // arguments of the table blocks of k_shard_cdf_apply (table_blocks = 0: none)
  int64_t rec_off;
  const int32_t* nodes_all;
  const int32_t* mids_all;
  int p_cap, table_blocks;
  float* table;
};

__device__ __forceinline__ void table_from_records(int block, int n_blocks, const float* __restrict__ msg_all,
                                                   int64_t stride, int n_ranks, const float* __restrict__ scanned,
                                                   const ProbeTable& pt);

// blocks [0, own): the own slice; the next two: the neighbour strips; the last pt.table_blocks: the coarse table
// (its own launch until round 3)
__global__ void __launch_bounds__(GJX_BLOCK)
    k_shard_cdf_apply(const float* __restrict__ logw, const float* __restrict__ d_lse, int64_t n,
                      const float* __restrict__ scanned, const float* __restrict__ msg_all, int64_t stride, int rank,
                      int n_ranks, float* __restrict__ cdf_ext, ProbeTable pt) {
  __shared__ float tree[512];
  const float lse = d_lse[0];
  const int64_t nt = n / GJX_SCAN_TILE;
  const int own = (int)gridDim.x - 2 - pt.table_blocks;
  if ((int)blockIdx.x >= own + 2) {
    table_from_records((int)blockIdx.x - own - 2, pt.table_blocks, msg_all, stride, n_ranks, scanned, pt);
    return;
  }
  TileRegs r;
  if ((int)blockIdx.x >= own) {
    const bool left = (int)blockIdx.x == own;
    if (left ? rank == 0 : rank + 1 >= n_ranks) return;
    const float* in = left ? msg_all + (int64_t)(rank - 1) * stride + nt + GJX_STRIP
                           : msg_all + (int64_t)(rank + 1) * stride + nt;
    tile_load<true>(in, lse, GJX_STRIP, 0, r);
    apply_tile(r, tree, scanned, left ? (int64_t)rank * nt - 1 : (int64_t)(rank + 1) * nt, GJX_STRIP, 0,
               left ? cdf_ext : cdf_ext + GJX_STRIP + n);
    return;
  }
  for (int64_t tile = blockIdx.x; tile < nt; tile += own) {
    const int64_t base = tile * GJX_SCAN_TILE;
    tile_load<true>(logw, lse, n, base, r);
    apply_tile(r, tree, scanned, tile + (int64_t)rank * nt, n, base, cdf_ext + GJX_STRIP);
  }
}

// table[node] = cdf[mid(node)] for every probe of every rank, from the owners' records and the scan of all totals
__device__ __forceinline__ void table_from_records(int block, int n_blocks, const float* __restrict__ msg_all,
                                                   int64_t stride, int n_ranks, const float* __restrict__ scanned,
                                                   const ProbeTable& pt) {
  const int total = pt.p_cap * n_ranks, p_cap = pt.p_cap;
  const int64_t rec_off = pt.rec_off;
  const int32_t* __restrict__ nodes_all = pt.nodes_all;
  const int32_t* __restrict__ mids_all = pt.mids_all;
  float* __restrict__ table = pt.table;
  for (int idx = block * GJX_BLOCK + threadIdx.x; idx < total; idx += n_blocks * GJX_BLOCK) {
    const int node = nodes_all[idx];
    if (node == 0) continue;
    const float* rec = msg_all + (int64_t)(idx / p_cap) * stride + rec_off + (int64_t)(idx % p_cap) * GJX_PROBE_REC;
    const int cnt = __float_as_int(rec[0]);
    const int64_t gt = (int64_t)mids_all[idx] / GJX_SCAN_TILE;
    float v;
    if (cnt < 0) {
      v = scanned[gt];
    } else {
      v = gt > 0 ? scanned[gt - 1] : 0.0f;   // apply_tile's carry
      for (int q = 0; q < cnt; ++q) v = v + rec[1 + q];
    }
    table[node] = v;
  }
  if (block == 0 && threadIdx.x == 0) table[0] = 0.0f;
}

GJX_API int32_t gjx_shard_cdf_finish(gjx_ctx* ctx, const float* d_logw, const float* d_lse, int64_t n,
                                     const float* d_msg_all, int32_t n_ranks, int32_t rank,
                                     const int32_t* d_probe_nodes_all, const int32_t* d_probe_mids_all,
                                     int32_t p_cap, float* d_all_totals, float* d_scanned, float* d_cdf_ext,
                                     int32_t lc, float* d_coarse, void* stream) {
  GJX_REQUIRE(ctx && d_logw && d_lse && d_msg_all && d_all_totals && d_scanned && d_cdf_ext && d_coarse,
              GJX_EINVAL, "gjx_shard_cdf_finish: null argument");
  GJX_REQUIRE(p_cap >= 0 && (p_cap == 0 || (d_probe_nodes_all && d_probe_mids_all)), GJX_EINVAL,
              "gjx_shard_cdf_finish: bad probe lists");
  GJX_REQUIRE(n > 0 && n % GJX_SCAN_TILE == 0 && n_ranks >= 1 && n_ranks <= GJX_MAX_RANKS && rank >= 0 &&
                  rank < n_ranks && n * n_ranks < ((int64_t)1 << 31),
              GJX_EINVAL, "gjx_shard_cdf_finish: bad sizes (n=%lld, %d ranks, rank %d)", (long long)n, n_ranks, rank);
  const int64_t n_cdf = n * n_ranks, nt = n / GJX_SCAN_TILE, rec_off = nt + 2 * GJX_STRIP;
  const int64_t stride = rec_off + (int64_t)GJX_PROBE_REC * p_cap;
  GJX_REQUIRE(lc >= 0 && lc <= 20 && lc <= bisect_levels(n_cdf), GJX_ELIMIT,
              "gjx_shard_cdf_finish: lc=%d not in [0, min(20, levels)]", lc);
  GJX_REQUIRE(((int64_t)1 << lc) - 1 <= (int64_t)p_cap * n_ranks, GJX_EINVAL,
              "gjx_shard_cdf_finish: %d probe slots per rank cannot hold the %lld probes of %d levels", p_cap,
              (long long)(((int64_t)1 << lc) - 1), lc);
  hipStream_t s = (hipStream_t)stream;
  // the tile totals of all shards in global tile order, and their scan
  // one block while the totals fit ONE scan tile (2048: e.g. 2 ranks x 2^21 particles); beyond that the block would walk
  // its tiles one after the other -- measured on the single-device scan of 8192 totals: ~31 us against ~13 us for the
  // three small launches of the hierarchical scan -- so more totals take the copy + gjx_cumsum_f32 path
#ifndef GJX_TOTALS_ONE_BLOCK_TILES
#define GJX_TOTALS_ONE_BLOCK_TILES 1
#endif
  if (nt * n_ranks <= (int64_t)GJX_TOTALS_ONE_BLOCK_TILES * GJX_SCAN_TILE) {
    k_totals_scan<<<1, GJX_BLOCK, 0, s>>>(d_msg_all, nt, stride, nt * n_ranks, d_all_totals, d_scanned);
    GJX_LAUNCH_CHECK();
  } else {
    GJX_CHECK_HIP(hipMemcpy2DAsync(d_all_totals, sizeof(float) * nt, d_msg_all, sizeof(float) * stride,
                                   sizeof(float) * nt, n_ranks, hipMemcpyDeviceToDevice, s));
    int32_t rc = gjx_cumsum_f32(ctx, d_all_totals, nt * n_ranks, d_scanned, stream);
    if (rc != GJX_OK) return rc;
  }
  // ... and, in the same launch, the COMPLETE coarse table on every rank (no reduction over ranks): owners' records
  // + the scanned totals
  const int total = p_cap * n_ranks;
  const ProbeTable pt{rec_off, d_probe_nodes_all, d_probe_mids_all, p_cap,
                      total > 0 ? (total + GJX_BLOCK - 1) / GJX_BLOCK : 1, d_coarse};
  k_shard_cdf_apply<<<scan_grid(nt) + 2 + pt.table_blocks, GJX_BLOCK, 0, s>>>(d_logw, d_lse, n, d_scanned, d_msg_all,
                                                                             stride, rank, n_ranks, d_cdf_ext, pt);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// min { j in [0, n_total] : pos_j > c }  (pos_j is non-decreasing in j; ALU only)
__device__ __forceinline__ int64_t pos_split(float c, float u, float nf, int64_t n_total) {
  int64_t lo = 0, hi = n_total;
  while (lo < hi) {
    const int64_t m = (lo + hi) >> 1;
    if (systematic_pos(m, u, nf) > c) hi = m; else lo = m + 1;
  }
  return lo;
}

// min { j : the interval (low_j, high_j] of position j after the lc coarse levels has high_j >= T }.
// Walks the coarse tree towards the first leaf with high >= T, keeping the range [a, b) of positions
// that reach the current node (the decisions of a node split the positions at pos_split(table[node])).
// The PATH of that walk depends on T and the interval arithmetic alone, not on the table: so a half-wave does it in
// one go -- lane l recomputes the path down to level l (ALU), loads that node's table entry and bisects its split
// (all levels side by side: one memory round trip and one 31-step bisection instead of lc of each in a row), and the
// clamped ranges are then folded level by level over shuffles.  `half` selects lanes [32 half, 32 half + 32).
// Measured with 2, 4, 8 ranks of 2^21 particles: 25 - 31 us for the serial walk on one lane, ~4 us this way
// (benchmarks/tools/shard_calls_timing.py); with one rank both targets return at once, which hid it.
__device__ __forceinline__ int64_t first_pos_reaching_wave(const float* __restrict__ coarse, int lc, uint32_t n_cdf,
                                                           float u, float nf, int64_t n_total, int64_t T, int half) {
  const int lane = threadIdx.x & 63, l_mine = lane & 31, base = half * 32;
  const bool mine = (lane >> 5) == half;
  const bool trivial_lo = T <= 0, trivial_hi = T > (int64_t)n_cdf;
  // the path down to this lane's level
  uint32_t low = 0u, high = n_cdf, node = 1u;
  bool left = false;
  for (int l = 0; l <= l_mine && l < lc; ++l) {
    const uint32_t mid = (low + high) >> 1;
    left = (int64_t)mid >= T;   // the left subtree ends at high = mid >= T: the first such leaf is in it
    if (l == l_mine) break;
    if (left) {
      high = mid;
      node = 2u * node;
    } else {
      low = mid;
      node = 2u * node + 1u;
    }
  }
  int64_t sp = 0;
  if (mine && l_mine < lc && !trivial_lo && !trivial_hi) sp = pos_split(coarse[node], u, nf, n_total);
  int64_t a = 0, b = n_total;
  for (int l = 0; l < lc; ++l) {   // (wave-uniform trip count; every lane takes part in the shuffles)
    const int src = base + l;
    int64_t s_l = ((int64_t)__shfl((int)(sp >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)sp, src);
    const bool left_l = __shfl((int)left, src) != 0;
    s_l = s_l < a ? a : (s_l > b ? b : s_l);
    if (left_l) b = s_l; else a = s_l;
  }
  return trivial_lo ? 0 : (trivial_hi ? n_total : a);
}

// range[0], range[1] = the candidate positions [ja, jb) of this rank: every position whose ancestor can
// lie in [shard_lo, shard_lo + n) (high_j >= shard_lo, and not yet high_j >= shard_hi + 2048: intervals
// are at most 2048 long after the coarse levels, so from there on low_j >= shard_hi).  A superset: the
// fine search decides.  All their fine probes lie in [shard_lo - 2048, shard_hi + 2048).
__global__ void k_shard_ranges(const float* __restrict__ coarse, int lc, uint32_t n_cdf, float u, int64_t shard_lo,
                               int64_t n, int64_t* __restrict__ range, int32_t* __restrict__ counts, int n_ranks) {
  const float nf = (float)n_cdf;
  if (threadIdx.x < 64) {   // one wave: lanes 0..31 find range[0], lanes 32..63 range[1]
    const int half = (threadIdx.x & 63) >> 5;
    const int64_t T0 = shard_lo, T1 = shard_lo + n + GJX_STRIP;
    // (both calls run on the whole wave: each half computes its own target, the shuffles stay inside a half)
    const int64_t r0 = first_pos_reaching_wave(coarse, lc, n_cdf, u, nf, (int64_t)n_cdf, half ? T1 : T0, half);
    if ((threadIdx.x & 31) == 0) range[half] = r0;
  }
  for (int d = threadIdx.x; d < n_ranks; d += blockDim.x) counts[d] = 0;
}

// run start: candidate i (position j = ja + i) selects a particle of this shard for ANOTHER rank's
// position, and the previous position of the same destination selected a different one.  `edge` = the
// one multiple of n a 2048-candidate tile can hold (n >= 2048): no 64-bit divisions per candidate.
__device__ __forceinline__ bool shard_run_start(const int32_t* __restrict__ idx, int64_t i, int64_t ja,
                                                int64_t shard_lo, int64_t n, int64_t edge) {
  const int64_t a = idx[i];
  if (a < shard_lo || a >= shard_lo + n) return false;
  const int64_t j = ja + i;
  if (j >= shard_lo && j < shard_lo + n) return false;   // own position: resolved in place, nothing to send
  return i == 0 || idx[i - 1] != (int32_t)a || j == edge;
}

__device__ __forceinline__ int block_sum_i32(int v, int* sm) {
  v = gjx_wave_sum_i32(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < GJX_BLOCK / 64; ++w) t += sm[w];
  __syncthreads();
  return t;
}

// rows per 2048-candidate tile, and rows per destination rank (integer atomics: order-free)
__global__ void __launch_bounds__(GJX_BLOCK)
    k_shard_count(const int32_t* __restrict__ idx, const int64_t* __restrict__ range, int64_t shard_lo, int64_t n,
                  int rank, int32_t* __restrict__ tile_counts, int32_t* __restrict__ counts) {
  __shared__ int sm[GJX_BLOCK / 64];
  const int64_t ja = range[0], m = range[1] - range[0];
  for (int64_t tile = blockIdx.x; tile * GJX_SCAN_TILE < m; tile += gridDim.x) {
    const int64_t i0 = tile * GJX_SCAN_TILE + (int64_t)threadIdx.x * GJX_SCAN_ITEMS;
    const int64_t j_tile = ja + tile * GJX_SCAN_TILE;
    if (j_tile >= shard_lo && j_tile + GJX_SCAN_TILE <= shard_lo + n) {   // own positions only: nothing to send
      if (threadIdx.x == 0) tile_counts[tile] = 0;
      continue;
    }
    const int64_t d0 = j_tile / n;                          // a tile spans at most two destinations (n >= 2048)
    const int64_t edge = j_tile % n == 0 ? j_tile : (d0 + 1) * n;
    int c0 = 0, c1 = 0;
#pragma unroll
    for (int k = 0; k < GJX_SCAN_ITEMS; ++k)
      if (i0 + k < m && shard_run_start(idx, i0 + k, ja, shard_lo, n, edge)) {
        if (ja + i0 + k < (d0 + 1) * n) ++c0; else ++c1;
      }
    const int t0 = block_sum_i32(c0, sm), t1 = block_sum_i32(c1, sm);
    if (threadIdx.x == 0) {
      tile_counts[tile] = t0 + t1;
      if (t0) atomicAdd(&counts[d0], t0);
      if (t1) atomicAdd(&counts[d0 + 1], t1);
    }
  }
}

// exclusive scan of the tile counts by one block (thread t owns a contiguous chunk)
__global__ void __launch_bounds__(GJX_FOLD_THREADS)
    k_shard_scan_tiles(const int64_t* __restrict__ range, const int32_t* __restrict__ tile_counts,
                       int32_t* __restrict__ tile_offsets) {
  __shared__ int sm[GJX_FOLD_THREADS];
  const int64_t m = range[1] - range[0];
  const int64_t nt = m > 0 ? (m + GJX_SCAN_TILE - 1) / GJX_SCAN_TILE : 0;
  const int64_t per = (nt + GJX_FOLD_THREADS - 1) / GJX_FOLD_THREADS;
  const int64_t t0 = (int64_t)threadIdx.x * per, t1 = t0 + per < nt ? t0 + per : nt;
  int sum = 0;
  for (int64_t t = t0; t < t1; ++t) sum += tile_counts[t];
  sm[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < GJX_FOLD_THREADS; o <<= 1) {   // Hillis-Steele inclusive scan
    const int v = (int)threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += v;
    __syncthreads();
  }
  int run = sm[threadIdx.x] - sum;
  for (int64_t t = t0; t < t1; ++t) {
    tile_offsets[t] = run;
    run += tile_counts[t];
  }
}

struct ShardCols {
  const float* c[6];
};

// rows in position order (= destination order): {first position (int bits), six columns of the ancestor}
__global__ void __launch_bounds__(GJX_BLOCK)
    k_shard_pack(const int32_t* __restrict__ idx, const int64_t* __restrict__ range, int64_t shard_lo, int64_t n,
                 int rank, const int32_t* __restrict__ tile_offsets, ShardCols cols, float* __restrict__ rows) {
  __shared__ int sm[GJX_BLOCK];
  const int64_t ja = range[0], m = range[1] - range[0];
  for (int64_t tile = blockIdx.x; tile * GJX_SCAN_TILE < m; tile += gridDim.x) {
    const int64_t i0 = tile * GJX_SCAN_TILE + (int64_t)threadIdx.x * GJX_SCAN_ITEMS;
    const int64_t j_tile = ja + tile * GJX_SCAN_TILE;
    if (j_tile >= shard_lo && j_tile + GJX_SCAN_TILE <= shard_lo + n) continue;   // own positions only
    const int64_t edge = j_tile % n == 0 ? j_tile : (j_tile / n + 1) * n;
    bool f[GJX_SCAN_ITEMS];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < GJX_SCAN_ITEMS; ++k) {
      f[k] = i0 + k < m && shard_run_start(idx, i0 + k, ja, shard_lo, n, edge);
      cnt += f[k] ? 1 : 0;
    }
    sm[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 1; o < GJX_BLOCK; o <<= 1) {
      const int v = (int)threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
      __syncthreads();
      sm[threadIdx.x] += v;
      __syncthreads();
    }
    int64_t slot = (int64_t)tile_offsets[tile] + sm[threadIdx.x] - cnt;
#pragma unroll
    for (int k = 0; k < GJX_SCAN_ITEMS; ++k)
      if (f[k]) {
        const int64_t a = (int64_t)idx[i0 + k] - shard_lo;
        float* row = rows + slot * GJX_SHARD_ROW;
        row[0] = __int_as_float((int32_t)(ja + i0 + k));
#pragma unroll
        for (int c = 0; c < 6; ++c) row[1 + c] = cols.c[c][a];
        ++slot;
      }
    __syncthreads();
  }
}

static inline int shard_tile_grid(int64_t cap) {
  const int64_t t = (cap + GJX_SCAN_TILE - 1) / GJX_SCAN_TILE;
  return (int)(t > GJX_MAX_GRID ? GJX_MAX_GRID : (t < 1 ? 1 : t));
}

int32_t gjx_shard_search_only(const float* d_coarse, int32_t lc, const float* d_cdf_ext, int32_t n_ranks, int32_t rank,
                              int64_t n, float u, int64_t cand_cap, int64_t* d_range, int32_t* d_cand_idx,
                              int32_t* d_counts, void* stream);

GJX_API int32_t gjx_shard_search(const float* d_coarse, int32_t lc, const float* d_cdf_ext, int32_t n_ranks,
                                 int32_t rank, int64_t n, float u, int64_t cand_cap, int64_t* d_range,
                                 int32_t* d_cand_idx, int32_t* d_tile_counts, int32_t* d_counts,
                                 void* stream) {
  GJX_REQUIRE(d_tile_counts != nullptr, GJX_EINVAL, "gjx_shard_search: null argument");
  int32_t rc = gjx_shard_search_only(d_coarse, lc, d_cdf_ext, n_ranks, rank, n, u, cand_cap, d_range, d_cand_idx,
                                     d_counts, stream);
  if (rc != GJX_OK) return rc;
  return gjx_shard_count(d_cand_idx, d_range, n_ranks, rank, n, d_tile_counts, d_counts, stream);
}

// the counting half of gjx_shard_search on its own: nothing of the main move launch needs the row counts, so the
// one-call front end (gjx_shard_front) leaves them to the side stream
GJX_API int32_t gjx_shard_count(const int32_t* d_cand_idx, const int64_t* d_range, int32_t n_ranks, int32_t rank,
                                int64_t n, int32_t* d_tile_counts, int32_t* d_counts, void* stream) {
  GJX_REQUIRE(d_cand_idx && d_range && d_tile_counts && d_counts, GJX_EINVAL, "gjx_shard_count: null argument");
  GJX_REQUIRE(n > 0 && n % GJX_SCAN_TILE == 0 && n_ranks >= 1 && n_ranks <= GJX_MAX_RANKS && rank >= 0 &&
                  rank < n_ranks && n * n_ranks < ((int64_t)1 << 31),
              GJX_EINVAL, "gjx_shard_count: bad sizes (n=%lld, %d ranks, rank %d)", (long long)n, n_ranks, rank);
  k_shard_count<<<shard_tile_grid(n * n_ranks), GJX_BLOCK, 0, (hipStream_t)stream>>>(
      d_cand_idx, d_range, (int64_t)rank * n, n, rank, d_tile_counts, d_counts);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// candidate range + fine search (d_counts is only zeroed here: gjx_shard_count fills it)
int32_t gjx_shard_search_only(const float* d_coarse, int32_t lc, const float* d_cdf_ext, int32_t n_ranks, int32_t rank,
                              int64_t n, float u, int64_t cand_cap, int64_t* d_range, int32_t* d_cand_idx,
                              int32_t* d_counts, void* stream) {
  GJX_REQUIRE(d_coarse && d_cdf_ext && d_range && d_cand_idx && d_counts, GJX_EINVAL,
              "gjx_shard_search: null argument");
  GJX_REQUIRE(n > 0 && n % GJX_SCAN_TILE == 0 && n_ranks >= 1 && n_ranks <= GJX_MAX_RANKS && rank >= 0 &&
                  rank < n_ranks && n * n_ranks < ((int64_t)1 << 31),
              GJX_EINVAL, "gjx_shard_search: bad sizes (n=%lld, %d ranks, rank %d)", (long long)n, n_ranks, rank);
  const int64_t n_cdf = n * n_ranks;
  // a rank can own the ancestors of EVERY position (all the weight in its shard)
  GJX_REQUIRE(cand_cap >= n_cdf, GJX_ELIMIT,
              "gjx_shard_search: the candidate buffers hold %lld positions, up to n_global=%lld can select this shard",
              (long long)cand_cap, (long long)n_cdf);
  GJX_REQUIRE(lc >= 0 && lc <= 20 && bisect_levels(n_cdf) - lc <= 11, GJX_EINVAL,
              "gjx_shard_search: lc=%d leaves intervals longer than one %d-tile", lc, GJX_SCAN_TILE);
  hipStream_t s = (hipStream_t)stream;
  const int64_t lo = (int64_t)rank * n;
  k_shard_ranges<<<1, 64, 0, s>>>(d_coarse, lc, (uint32_t)n_cdf, u, lo, n, d_range, d_counts, n_ranks);
  GJX_LAUNCH_CHECK();
  // the resident CDF: [lo - 2048, lo + n + 2048) clipped to [0, n_cdf)
  const int64_t base = rank > 0 ? lo - GJX_STRIP : 0;
  const float* slab = rank > 0 ? d_cdf_ext : d_cdf_ext + GJX_STRIP;
  k_searchsorted<SlabSrc><<<search_grid(n_cdf), GJX_BLOCK, 0, s>>>(
      SlabSrc{slab, base}, d_coarse, lc, (uint32_t)n_cdf, u, 0, 0, (float)n_cdf, bisect_levels(n_cdf),
      d_cand_idx, d_range);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_shard_pack(const int32_t* d_cand_idx, const int64_t* d_range, const int32_t* d_tile_counts,
                               int32_t* d_tile_offsets, int32_t n_ranks, int32_t rank, int64_t n,
                               const float* const* d_cols6, float* d_rows, void* stream) {
  GJX_REQUIRE(d_cand_idx && d_range && d_tile_counts && d_tile_offsets && d_cols6 && d_rows, GJX_EINVAL,
              "gjx_shard_pack: null argument");
  GJX_REQUIRE(n > 0 && n % GJX_SCAN_TILE == 0 && n_ranks >= 1 && n_ranks <= GJX_MAX_RANKS && rank >= 0 &&
                  rank < n_ranks,
              GJX_EINVAL, "gjx_shard_pack: bad sizes");
  ShardCols cols;
  for (int c = 0; c < 6; ++c) {
    GJX_REQUIRE(d_cols6[c] != nullptr, GJX_EINVAL, "gjx_shard_pack: column %d is null", c);
    cols.c[c] = d_cols6[c];
  }
  hipStream_t s = (hipStream_t)stream;
  k_shard_scan_tiles<<<1, GJX_FOLD_THREADS, 0, s>>>(d_range, d_tile_counts, d_tile_offsets);
  GJX_LAUNCH_CHECK();
  k_shard_pack<<<shard_tile_grid(n * n_ranks), GJX_BLOCK, 0, s>>>(d_cand_idx, d_range, (int64_t)rank * n, n, rank,
                                                                  d_tile_offsets, cols, d_rows);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

struct ShardDst {
  float* c[6];
};

__global__ void __launch_bounds__(GJX_BLOCK)
    k_shard_unpack(const float* __restrict__ rows, int64_t n_recv, ShardDst halo) {
  for (int64_t k = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; k < n_recv; k += (int64_t)gridDim.x * GJX_BLOCK) {
    const float* row = rows + k * GJX_SHARD_ROW;
#pragma unroll
    for (int c = 0; c < 6; ++c) halo.c[c][k] = row[1 + c];
  }
}

// local ancestor index of every own position: own-sourced -> [0, n) straight from the own search;
// otherwise n + (the last received row whose first position is <= j).  *err != 0: a position found no source
// (cannot happen when every rank ran the same search on the same table; checked, never clamped silently).
__global__ void __launch_bounds__(GJX_BLOCK)
    k_shard_expand(const int32_t* __restrict__ idx, const int64_t* __restrict__ range, const float* __restrict__ rows,
                   int64_t n_recv, int64_t shard_lo, int64_t n, int32_t* __restrict__ anc, int32_t* __restrict__ err) {
  const int64_t ja = range[0], m = range[1] - range[0];
  for (int64_t jl = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; jl < n; jl += (int64_t)gridDim.x * GJX_BLOCK) {
    const int64_t j = shard_lo + jl, i = j - ja;
    if (i >= 0 && i < m) {
      const int64_t a = idx[i];
      if (a >= shard_lo && a < shard_lo + n) {
        anc[jl] = (int32_t)(a - shard_lo);
        continue;
      }
    }
    int64_t lo = 0, hi = n_recv;   // first row whose first position is > j
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)__float_as_int(rows[mid * GJX_SHARD_ROW]) <= j) lo = mid + 1; else hi = mid;
    }
    if (lo == 0) {
      atomicOr(err, 1);
      anc[jl] = 0;
    } else {
      anc[jl] = (int32_t)(n + lo - 1);
    }
  }
}

int32_t gjx_shard_expand(const int32_t* d_cand_idx, const int64_t* d_range, const float* d_rows, int64_t n_recv,
                         int32_t rank, int64_t n, float* const* d_cols6, int64_t halo_cap, int32_t* d_anc,
                         int32_t* d_err, hipStream_t s) {
  GJX_REQUIRE(d_cand_idx && d_range && d_cols6 && d_anc && d_err, GJX_EINVAL, "gjx_shard_expand: null argument");
  GJX_REQUIRE(n > 0 && n_recv >= 0 && (n_recv == 0 || d_rows), GJX_EINVAL, "gjx_shard_expand: bad sizes");
  GJX_REQUIRE(n_recv <= halo_cap, GJX_ELIMIT, "gjx_shard_expand: %lld received rows exceed the halo of %lld",
              (long long)n_recv, (long long)halo_cap);
  if (n_recv > 0) {
    ShardDst halo;
    for (int c = 0; c < 6; ++c) {
      GJX_REQUIRE(d_cols6[c] != nullptr, GJX_EINVAL, "gjx_shard_expand: column %d is null", c);
      halo.c[c] = d_cols6[c] + n;
    }
    k_shard_unpack<<<gjx_grid_for(n_recv), GJX_BLOCK, 0, s>>>(d_rows, n_recv, halo);
    GJX_LAUNCH_CHECK();
  }
  k_shard_expand<<<gjx_grid_for(n), GJX_BLOCK, 0, s>>>(d_cand_idx, d_range, d_rows, n_recv, (int64_t)rank * n, n,
                                                       d_anc, d_err);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// The overlapped form of the exchange (gjx_shard_moves_main / gjx_shard_moves_fixup in gjx_smc.hip): the
// positions of a rank whose ancestor is its own particle need nothing from the all_to_all, so their moves start
// before the host has even sized it.  Positions and ancestors are both increasing: the own-sourced positions are
// one run [p0, p1), the others a prefix and a suffix.
// ---------------------------------------------------------------------------
// split = {p0 + 1 (0: no own-sourced position), p1, go, flags}; flags are the move launch's (kept here).  The candidates' ancestors are non-decreasing in the
// position, so [p0, p1) comes from two bisections over them by one thread each; the move launch re-derives every
// position's side from its own ancestor and flags a disagreement.  go = the resampling decision
// ess < n_global / 2 of rejuvenation_smc (smc.py:752-760) taken where the ESS already is.  The acceptance
// counters of the step start at zero here: the main and the fix-up move launches both add.
__global__ void k_shard_bounds(const int32_t* __restrict__ idx, const int64_t* __restrict__ range, int64_t shard_lo,
                               int64_t n, const float* __restrict__ gstats, int64_t half, int32_t* __restrict__ split,
                               int64_t* __restrict__ accept_counts, int n_moves) {
  __shared__ int64_t b[2];
  const int64_t ja = range[0], m = range[1] - range[0];
  // candidates that are own positions: i in [i_lo, i_hi)
  const int64_t i_lo = shard_lo > ja ? shard_lo - ja : 0;
  const int64_t i_hi = shard_lo + n - ja < m ? shard_lo + n - ja : m;
  if (threadIdx.x < 2) {
    const int64_t T = threadIdx.x == 0 ? shard_lo : shard_lo + n;   // first candidate with idx >= T
    int64_t lo = i_lo, hi = i_hi > i_lo ? i_hi : i_lo;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)idx[mid] >= T) hi = mid; else lo = mid + 1;
    }
    b[threadIdx.x] = lo + ja - shard_lo;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool any = b[1] > b[0];
    split[0] = any ? (int32_t)b[0] + 1 : 0;
    split[1] = any ? (int32_t)b[1] : 0;
    split[2] = (double)gstats[2] < (double)half ? 1 : 0;
  }
  if (accept_counts)
    for (int k = threadIdx.x; k < n_moves; k += blockDim.x) accept_counts[k] = 0;
}

GJX_API int32_t gjx_shard_split(const int32_t* d_cand_idx, const int64_t* d_range, int32_t rank, int64_t n,
                                const float* d_gstats, int64_t n_global, int32_t* d_split, int64_t* d_accept_counts,
                                int32_t n_moves, void* stream) {
  GJX_REQUIRE(d_cand_idx && d_range && d_gstats && d_split, GJX_EINVAL, "gjx_shard_split: null argument");
  GJX_REQUIRE(n > 0 && rank >= 0 && n_global >= n && n_global < ((int64_t)1 << 31) && n_moves >= 0 &&
                  n_moves <= GJX_MAX_MOVES,
              GJX_EINVAL, "gjx_shard_split: bad sizes (n=%lld, n_global=%lld, rank %d, %d moves)", (long long)n,
              (long long)n_global, rank, n_moves);
  k_shard_bounds<<<1, 64, 0, (hipStream_t)stream>>>(d_cand_idx, d_range, (int64_t)rank * n, n, d_gstats, n_global / 2,
                                                    d_split, d_accept_counts, n_moves);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ancestors (halo indices n + k) of the positions [0, p0) and [p1, n): work item t -> position t or p1 + t - p0
__global__ void __launch_bounds__(GJX_BLOCK)
    k_shard_fix_anc(const float* __restrict__ rows, int64_t n_recv, int64_t shard_lo, int64_t n, int64_t p0,
                    int64_t p1, int32_t* __restrict__ anc, int32_t* __restrict__ err) {
  const int64_t n_fix = p0 + (n - p1);
  for (int64_t t = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; t < n_fix; t += (int64_t)gridDim.x * GJX_BLOCK) {
    const int64_t jl = t < p0 ? t : p1 + (t - p0), j = shard_lo + jl;
    int64_t lo = 0, hi = n_recv;   // first row whose first position is > j
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)__float_as_int(rows[mid * GJX_SHARD_ROW]) <= j) lo = mid + 1; else hi = mid;
    }
    if (lo == 0) {
      atomicOr(err, 1);
      anc[jl] = 0;
    } else {
      anc[jl] = (int32_t)(n + lo - 1);
    }
  }
}

// received rows -> halo, ancestors of the positions outside [p0, p1) -> d_anc_fix (a buffer of its own: the main
// move launch may still be reading the split's)
int32_t gjx_shard_expand_fix(const float* d_rows, int64_t n_recv, int32_t rank, int64_t n, int64_t p0, int64_t p1,
                             float* const* d_cols6, int64_t halo_cap, int32_t* d_anc_fix, int32_t* d_err,
                             hipStream_t s) {
  GJX_REQUIRE(d_cols6 && d_anc_fix && d_err, GJX_EINVAL, "gjx_shard_expand_fix: null argument");
  GJX_REQUIRE(n > 0 && 0 <= p0 && p0 <= p1 && p1 <= n, GJX_EINVAL, "gjx_shard_expand_fix: bad split [%lld, %lld) of %lld",
              (long long)p0, (long long)p1, (long long)n);
  const int64_t n_fix = p0 + (n - p1);
  GJX_REQUIRE(n_recv >= 0 && (n_recv == 0 || d_rows), GJX_EINVAL, "gjx_shard_expand_fix: bad sizes");
  GJX_REQUIRE(n_recv <= halo_cap, GJX_ELIMIT, "gjx_shard_expand_fix: %lld received rows exceed the halo of %lld",
              (long long)n_recv, (long long)halo_cap);
  if (n_recv > 0) {
    ShardDst halo;
    for (int c = 0; c < 6; ++c) {
      GJX_REQUIRE(d_cols6[c] != nullptr, GJX_EINVAL, "gjx_shard_expand_fix: column %d is null", c);
      halo.c[c] = d_cols6[c] + n;
    }
    k_shard_unpack<<<gjx_grid_for(n_recv), GJX_BLOCK, 0, s>>>(d_rows, n_recv, halo);
    GJX_LAUNCH_CHECK();
  }
  if (n_fix > 0) {
    k_shard_fix_anc<<<gjx_grid_for(n_fix), GJX_BLOCK, 0, s>>>(d_rows, n_recv, (int64_t)rank * n, n, p0, p1, d_anc_fix,
                                                              d_err);
    GJX_LAUNCH_CHECK();
  }
  return GJX_OK;
}
