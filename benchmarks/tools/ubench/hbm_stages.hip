// hbm_stages.hip -- the HBM-bound stages of the smc step timed one launch at a time (HIP events), outside Python:
// the normaliser (tile sums + in-kernel fold), the association-exact cumsum (totals, apply) and the bisection.
// Links against the library's own objects (the kernels under test are the product's, not copies):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude benchmarks/tools/ubench/hbm_stages.hip \
//         genjax_amd/_lib/obj/*.o -o benchmarks/tools/ubench/hbm_stages
//   ./hbm_stages [log2 n = 24] [repeats = 20] [scale = 1: log-weights are -scale * chi^2_8 / 2; ESS / N falls with it]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../../genjax_amd/csrc/gjx_common.h"

#define CK(e)                                                                         \
  do {                                                                                \
    hipError_t _e = (e);                                                              \
    if (_e != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(_e)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)
#define CG(e)                                                               \
  do {                                                                      \
    int32_t _r = (e);                                                       \
    if (_r != 0) {                                                          \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, gjx_last_error()); \
      exit(1);                                                              \
    }                                                                       \
  } while (0)

template <class P, class F>
static void timeit(const char* name, double bytes, int reps, hipStream_t s, P&& prep, F&& launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int r = 0; r < reps + 3; ++r) {
    prep();
    CK(hipEventRecord(a, s));
    launch();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float t;
    CK(hipEventElapsedTime(&t, a, b));
    if (r >= 3) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double med = ms[ms.size() / 2], best = ms[0];
  printf("%-34s median %8.2f us  best %8.2f us  %7.1f GB/s at the median (%.1f MB algorithmic)\n", name, med * 1e3,
         best * 1e3, bytes / (med * 1e-3) / 1e9, bytes / 1e6);
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
}

// reference points: an empty launch between two events, and plain streaming reads of the same vector
__global__ void k_empty() {}
template <int LOADS>   // LOADS 16-byte loads in flight per lane, one chunk of LOADS * 1 KiB * 4 ... per wave, grid-stride
__global__ void __launch_bounds__(256) k_stream(const float4* __restrict__ a, int64_t n4, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  float acc = 0.0f;
  for (int64_t c = wave; c * (64 * LOADS) < n4; c += nwaves) {
    float4 v[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) v[k] = a[c * (64 * LOADS) + k * 64 + lane];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 24, reps = argc > 2 ? atoi(argv[2]) : 20;
  const float scale = argc > 3 ? (float)atof(argv[3]) : 1.0f;
  const int64_t n = (int64_t)1 << lg;
  gjx_ctx* ctx = nullptr;
  CG(gjx_ctx_create(0, n, &ctx));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  std::vector<float> h(n);
  uint32_t x = 12345u;
  float mx = -INFINITY;
  // log-weights shaped like a step's: -chi^2_8 / 2 (eight squared standard residuals), ESS / N = (9 / 16)^4 = 0.32
  auto unif = [&]() {
    x = x * 1664525u + 1013904223u;
    return ((float)(x >> 8) + 0.5f) / 16777216.0f;
  };
  for (int64_t i = 0; i < n; ++i) {
    float acc = 0.0f;
    for (int k = 0; k < 4; ++k) acc += -logf(unif());      // -log(U1 U2 U3 U4) = chi^2_8 / 2
    h[i] = -acc * scale - 3.0f;
    mx = fmaxf(mx, h[i]);
  }
  float *d_logw, *d_stats, *d_cdf, *d_big;
  int32_t* d_anc;
  CK(hipMalloc(&d_logw, sizeof(float) * n));
  CK(hipMalloc(&d_cdf, sizeof(float) * n));
  CK(hipMalloc(&d_anc, sizeof(int32_t) * n));
  CK(hipMalloc(&d_stats, sizeof(float) * 8));
  CK(hipMalloc(&d_big, (size_t)1 << 30));   // written between launches: the stage starts from cold caches
  CK(hipMemcpy(d_logw, h.data(), sizeof(float) * n, hipMemcpyHostToDevice));
  float* cell = ctx->d_scalars + GJX_MAX_CELL;
  CK(hipMemset(d_big, 0, (size_t)1 << 30));
  // cold = 1 GiB of OTHER data read since the last use (clean lines: nothing to write back while the stage runs)
  auto flush = [&]() { k_stream<8><<<4096, 256, 0, s>>>((const float4*)d_big, ((int64_t)1 << 30) / 16, d_stats + 6); };

  timeit("empty launch", 0.0, reps, s, [&]() {}, [&]() { k_empty<<<1, 64, 0, s>>>(); });
  // dirty = 1 GiB of other data WRITTEN right before the launch (what extend leaves behind in the step: the stage's reads
  // share the memory system with the write-back of those lines)
  auto dirty = [&]() { CK(hipMemsetAsync(d_big, 1, (size_t)1 << 30, s)); };
  timeit("normaliser after 1 GiB of writes", 4.0 * n, reps, s,
         [&]() {
           CK(hipMemcpyAsync(cell, &mx, sizeof(float), hipMemcpyHostToDevice, s));
           dirty();
         },
         [&]() { CG(gjx_sumexp_f32(ctx, d_logw, cell, n, d_stats + 3, (void*)s)); });
  for (int cold = 0; cold < 2; ++cold) {
    printf("---- n = 2^%d, %s caches\n", lg, cold ? "cold (1 GiB of other data read before every launch)" : "warm (back to back)");
    for (int grid : {1024, 2048, 4096, 8192}) {
      char nm[64];
      snprintf(nm, sizeof(nm), "stream 8 x 16 B per lane, grid %d", grid);
      timeit(nm, 4.0 * n, reps, s, [&]() { if (cold) flush(); },
             [&]() { k_stream<8><<<grid, 256, 0, s>>>((const float4*)d_logw, n / 4, d_stats + 6); });
    }
    timeit("stream 4 x 16 B per lane, grid 4096", 4.0 * n, reps, s, [&]() { if (cold) flush(); },
           [&]() { k_stream<4><<<4096, 256, 0, s>>>((const float4*)d_logw, n / 4, d_stats + 6); });
    timeit("stream 2 x 16 B per lane, grid 8192", 4.0 * n, reps, s, [&]() { if (cold) flush(); },
           [&]() { k_stream<2><<<8192, 256, 0, s>>>((const float4*)d_logw, n / 4, d_stats + 6); });
    timeit("normaliser (one launch)", 4.0 * n, reps, s,
           [&]() {
             CK(hipMemcpyAsync(cell, &mx, sizeof(float), hipMemcpyHostToDevice, s));
             if (cold) flush();
           },
           [&]() { CG(gjx_normalise_from_max_partials(ctx, d_logw, n, cell, 1, d_stats, s)); });
    timeit("normaliser (sums launch + fold launch)", 4.0 * n, reps, s,
           [&]() {
             CK(hipMemcpyAsync(cell, &mx, sizeof(float), hipMemcpyHostToDevice, s));
             if (cold) flush();
           },
           [&]() { CG(gjx_sumexp_f32(ctx, d_logw, cell, n, d_stats + 3, (void*)s)); });
    timeit("cumsum of exp(logw - lse)", 12.0 * n, reps, s, [&]() { if (cold) flush(); },
           [&]() { CG(gjx_cumsum_normalized_f32(ctx, d_logw, d_stats + 1, n, d_cdf, (void*)s)); });
    timeit("searchsorted (systematic)", 8.0 * n, reps, s, [&]() { if (cold) flush(); },
           [&]() { CG(gjx_searchsorted_systematic(d_cdf, n, 0.37f, 0, n, n, d_anc, (void*)s)); });
  }
  float st[5];
  CK(hipMemcpy(st, d_stats, sizeof(st), hipMemcpyDeviceToHost));
  printf("stats: max %.6g lse %.9g ess %.9g (ESS / N = %.4f)\n", st[0], st[1], st[2], st[2] / (double)n);
  return 0;
}
