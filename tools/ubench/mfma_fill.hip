#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// FILL: 0 none, 1 2x v_max, 2 2x v_fmamk(literal), 3 cvt_f32_f16 + sdwa, 4 cvt_pk_f16, 5 2x v_add_e64 neg, 6 mix (max,max,cvtpk,cvt,sdwa,add,add spread over 3 MFMAs), 7 ds_read_b128 every 3rd MFMA (consumed 12 MFMAs later), 8 = 6+7
template <int FILL>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int iters) {
  __shared__ uint4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_uint4(i, i * 3, i * 7, 0x3c003c00);
  __syncthreads();
  half8 a[4], b;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.001f * (threadIdx.x + i + j));
  for (int j = 0; j < 8; ++j) b[j] = (_Float16)(0.002f * (threadIdx.x * 3 + j));
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float x0 = threadIdx.x * 0.5f, x1 = threadIdx.x * 0.25f, y0 = 1.f, y1 = 2.f; unsigned pk = 0;
  const uint4* lp = lds + (threadIdx.x & 63);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 12; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(r + i) & 3], b, acc[i], 0, 0, 0);
        if (FILL == 1) asm volatile("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1" : "+v"(x0), "+v"(x1));
        if (FILL == 2) asm volatile("v_fmamk_f32 %0, %2, 0x3b800000, %3\n\tv_fmamk_f32 %1, %3, 0x3b800000, %2" : "=&v"(y0), "=&v"(y1) : "v"(x0), "v"(x1));
        if (FILL == 3) asm volatile("v_cvt_f32_f16 %0, %2\n\tv_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=&v"(y0), "=&v"(y1) : "v"(pk));
        if (FILL == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(x0), "v"(x1));
        if (FILL == 5) asm volatile("v_add_f32_e64 %0, %2, -%3\n\tv_add_f32_e64 %1, %3, -%2" : "=&v"(y0), "=&v"(y1) : "v"(x0), "v"(x1));
        if (FILL == 6 || FILL == 8) {
          if (i == 0) asm volatile("v_fmamk_f32 %0, %2, 0x3b800000, %3\n\tv_fmamk_f32 %1, %3, 0x3b800000, %2" : "=&v"(y0), "=&v"(y1) : "v"(x0), "v"(x1));
          if (i == 1) asm volatile("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1" : "+v"(y0), "+v"(y1));
          if (i == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(y0), "v"(y1));
          if (i == 3) asm volatile("v_cvt_f32_f16 %0, %2\n\tv_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=&v"(x0), "=&v"(x1) : "v"(pk));
        }
        if ((FILL == 7 || FILL == 8) && i == 3) {
          uint4 v = lp[((r * 5 + it) & 31) * 64];
          a[r & 3] = __builtin_bit_cast(half8, v);    // consumed 12+ MFMAs later (next time this slot is used)
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = x0 + x1 + y0 + y1 + (float)pk;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int FILL>
void run(const char* name) {
  float* out; unsigned long long* cyc; hipMalloc(&out, 4 * 256 * 256); hipMalloc(&cyc, 8);
  int iters = 2000;
  hipLaunchKernelGGL((k<FILL>), dim3(256), dim3(256), 0, 0, out, cyc, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<FILL>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double n = 48.0 * iters;
  printf("%-44s ticks/MFMA=%.1f  ns/MFMA=%.2f\n", name, c / n, ms * 1e6 / n);
}
int main() {
  run<0>("no filler"); run<1>("2x v_max per MFMA"); run<2>("2x v_fmamk (literal) per MFMA"); run<3>("cvt_f32_f16 + sdwa per MFMA");
  run<4>("v_cvt_pk_f16_f32 per MFMA"); run<5>("2x v_add_e64(neg) per MFMA"); run<6>("conversion mix (avg 1.75 VALU/MFMA)");
  run<7>("ds_read_b128 every 4th MFMA"); run<8>("mix + ds_read");
  return 0;
}
