// Microbenchmark: per-opcode issue cost on gfx950 at 8 waves/SIMD (ns per wave-instruction per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define OP8(fmt) asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
  : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c0), "v"(c1), "s"(sm) : "vcc")

#define F_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define F_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define F_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 19\n"
#define F_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 13, %8\n"
#define F_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define F_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define F_LSHR(i) "v_lshrrev_b32 %" #i ", 13, %" #i "\n"
#define F_MINU(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define F_MINI(i) "v_min_i32 %" #i ", %" #i ", %8\n"
#define F_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, %10\n"
#define F_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define F_CMPU(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n"
#define F_MULF(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define F_SUBF(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define F_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define F_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define F_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
#define F_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define F_MINF(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define F_LOG(i) "v_log_f32 %" #i ", %" #i "\n"
#define F_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define F_SIN(i) "v_sin_f32 %" #i ", %" #i "\n"
#define F_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define F_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define F_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define F_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define F_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 9\n"
#define F_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define F_CVT(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define F_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define F_MULABS(i) "v_mul_f32 %" #i ", |%" #i "|, -%8\n"

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters, unsigned c0, unsigned c1) {
  unsigned r[8];
  for (int j = 0; j < 8; ++j) r[j] = threadIdx.x * (j + 3) + j;
  unsigned long long sm = 0x5555AAAA5555AAAAull;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) OP8(F_ADD);
      if (MODE == 1) OP8(F_XOR);
      if (MODE == 2) OP8(F_ALIGN);
      if (MODE == 3) OP8(F_LSHLOR);
      if (MODE == 4) OP8(F_ADD3);
      if (MODE == 5) OP8(F_XAD);
      if (MODE == 6) OP8(F_LSHR);
      if (MODE == 7) OP8(F_MINU);
      if (MODE == 8) OP8(F_MINI);
      if (MODE == 9) OP8(F_CND);
      if (MODE == 10) OP8(F_CMP);
      if (MODE == 11) OP8(F_CMPU);
      if (MODE == 12) OP8(F_MULF);
      if (MODE == 13) OP8(F_SUBF);
      if (MODE == 14) OP8(F_FMA);
      if (MODE == 15) OP8(F_FMAC);
      if (MODE == 16) OP8(F_MIN3);
      if (MODE == 17) OP8(F_MED3);
      if (MODE == 18) OP8(F_MINF);
      if (MODE == 19) OP8(F_LOG);
      if (MODE == 20) OP8(F_RCP);
      if (MODE == 21) OP8(F_SIN);
      if (MODE == 22) OP8(F_SQRT);
      if (MODE == 23) OP8(F_MULLO);
      if (MODE == 24) OP8(F_MAD24);
      if (MODE == 25) OP8(F_PERM);
      if (MODE == 26) OP8(F_BFE);
      if (MODE == 27) OP8(F_ANDOR);
      if (MODE == 28) OP8(F_CVT);
      if (MODE == 29) OP8(F_MOV);
      if (MODE == 30) OP8(F_MULABS);
    }
  }
  unsigned s = 0;
  for (int j = 0; j < 8; ++j) s += r[j];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

static const char* names[] = {"v_add_u32", "v_xor_b32", "v_alignbit_b32", "v_lshl_or_b32", "v_add3_u32", "v_xad_u32",
  "v_lshrrev_b32", "v_min_u32", "v_min_i32", "v_cndmask(sgpr)", "v_cmp_lt_f32", "v_cmp_lt_u32", "v_mul_f32", "v_sub_f32",
  "v_fma_f32", "v_fmac_f32", "v_min3_f32", "v_med3_f32", "v_min_f32", "v_log_f32", "v_rcp_f32", "v_sin_f32", "v_sqrt_f32",
  "v_mul_lo_u32", "v_mad_u32_u24", "v_perm_b32", "v_bfe_u32", "v_and_or_b32", "v_cvt_f32_u32", "v_mov_b32", "v_mul_f32 abs/neg"};

template <int MODE>
void run(unsigned* d) {
  const int iters = 1000, cus = 256, bpc = 8;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<MODE><<<cus * bpc, 256>>>(d, 10, 12345u, 999u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<MODE><<<cus * bpc, 256>>>(d, iters, 12345u, 999u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double instr_per_simd = (double)bpc * 64.0 * iters;
  printf("%-20s %.3f ns per wave-instruction per SIMD\n", names[MODE], ms * 1e6 / instr_per_simd);
  if constexpr (MODE + 1 < 31) run<MODE + 1>(d);
}

int main() {
  unsigned* d; CHECK(hipMalloc(&d, sizeof(unsigned) * 256 * 256 * 16));
  run<0>(d);
  return 0;
}
