// Microbenchmark: issue rate of v_fma_f32, v_pk_fma_f32, v_rcp_f32, integer add/xor/alignbit on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  if (MODE == 0) {  // scalar fma, 8 independent chains
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      }
    }
  } else if (MODE == 1) {  // packed fma: 4 independent v2f chains (8 VGPRs), 8 instructions per group
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, aa = {a, a}, bb = {b, b};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                     "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(aa), "v"(bb));
      }
    }
    x0 = p0.x + p0.y; x1 = p1.x + p1.y; x2 = p2.x + p2.y; x3 = p3.x + p3.y;
  } else if (MODE == 2) {  // rcp
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                     "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      }
    }
  } else if (MODE == 3) {  // integer: add / alignbit / xor mix (threefry round), 8 per group
    unsigned y0 = threadIdx.x, y1 = y0 * 3 + 1, y2 = y0 + 5, y3 = y0 * 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n v_add_u32 %2, %2, %3\n"
                     "v_alignbit_b32 %3, %3, %3, 17\n v_xor_b32 %3, %3, %2\n v_add_u32 %0, %0, %3\n v_add_u32 %2, %2, %1\n"
                     : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
      }
    }
    x0 = y0; x1 = y1; x2 = y2; x3 = y3;
  } else if (MODE == 4) {  // cndmask + cmp
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %1, %1, %3, vcc\n"
                     "v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %0, vcc\n v_cmp_lt_f32 vcc, %3, %0\n v_cndmask_b32 %3, %3, %1, vcc\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "vcc");
      }
    }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, float* d) {
  const int iters = 2000, cus = 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<MODE><<<cus * blocks_per_cu, 256>>>(d, 10, 1.0001f, 0.5f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<MODE><<<cus * blocks_per_cu, 256>>>(d, iters, 1.0001f, 0.5f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // instructions per SIMD: blocks_per_cu blocks x 4 waves / 4 SIMDs = blocks_per_cu waves per SIMD, each 64 * iters instr
  double instr_per_simd = (double)blocks_per_cu * 64.0 * iters;
  printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (cycles @2.4GHz: %.2f)\n", name,
         blocks_per_cu, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
  float* d; CHECK(hipMalloc(&d, sizeof(float) * 256 * 256 * 16));
  for (int b : {1, 2, 4, 8}) {
    if (b == 1) { run<0>("fma", 1, d); run<1>("pk_fma", 1, d); run<2>("rcp", 1, d); run<3>("int", 1, d); run<4>("cmp+cnd", 1, d); }
    if (b == 2) { run<0>("fma", 2, d); run<1>("pk_fma", 2, d); run<2>("rcp", 2, d); run<3>("int", 2, d); run<4>("cmp+cnd", 2, d); }
    if (b == 4) { run<0>("fma", 4, d); run<1>("pk_fma", 4, d); run<2>("rcp", 4, d); run<3>("int", 4, d); run<4>("cmp+cnd", 4, d); }
    if (b == 8) { run<0>("fma", 8, d); run<1>("pk_fma", 8, d); run<2>("rcp", 8, d); run<3>("int", 8, d); run<4>("cmp+cnd", 8, d); }
  }
  return 0;
}
