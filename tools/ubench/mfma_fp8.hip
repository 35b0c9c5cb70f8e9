// MFMA issue rates on gfx950 under a full chip: f16 32x32x16 against the fp8 shapes a mixed "f16 main term + fp8 cross terms" split of
// the fp32-grade product would use.  hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_fp8.hip -o mfma_fp8
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int KIND>      // 0: f16 32x32x16, 1: fp8 32x32x16, 2: f8f6f4 32x32x64 (fp8 x fp8, unit scales), 3: mix per product = 1 f16 + 2 fp8 32x32x16
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int iters) {
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x * 3 + j)); }
  long a8 = 0x3839404142434445L + threadIdx.x, b8 = 0x3031323334353637L + 3 * threadIdx.x;      // eight e4m3 values each (non-zero patterns)
  i32x8 A, Bv;
  for (int j = 0; j < 8; ++j) { A[j] = 0x38394041 + threadIdx.x * (j + 1); Bv[j] = 0x30313233 + threadIdx.x * (j + 3); }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < (KIND >= 4 ? 0 : 6); ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        else if constexpr (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a8, b8, acc[i], 0, 0, 0);
        else if constexpr (KIND == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, Bv, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        else if constexpr (KIND == 3) {
          if (r % 3 == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
          else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a8, b8, acc[i], 0, 0, 0);
        }
      }
    // dependent orders of the f16f8 engine (48 MFMAs per trip as above): G accumulators take their f16 term, then their two fp8 terms
    if constexpr (KIND >= 4) {
      constexpr int G = KIND == 4 ? 2 : (KIND == 5 ? 4 : 8);
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int g0 = 0; g0 < 8; g0 += G) {
#pragma unroll
          for (int i = 0; i < G; ++i) acc[g0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[g0 + i], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < G; ++i) acc[g0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a8, b8, acc[g0 + i], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < G; ++i) acc[g0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a8 + 1, b8, acc[g0 + i], 0, 0, 0);
        }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND>
void run(const char* name, int blocks, double flop_per_mfma) {
  float* out; unsigned long long* cyc; hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 8);
  int iters = 2000;
  hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double n = 48.0 * iters;
  printf("%-44s blocks=%d  memtime ticks/MFMA=%.1f  ns/MFMA=%.2f  -> %.0f TF/s chip-equivalent\n", name, blocks, c / n, ms * 1e6 / n,
         (double)blocks * 4 * n * flop_per_mfma / (ms * 1e-3) / 1e12);
}
int main() {
  for (int blocks : {1, 256}) {
    run<0>("f16 32x32x16", blocks, 32768);
    run<1>("fp8 32x32x16", blocks, 32768);
    run<2>("f8f6f4 32x32x64 (fp8 x fp8)", blocks, 131072);
    run<3>("1 f16 + 2 fp8 32x32x16 per product", blocks, 32768);
    run<4>("f16, then 2 fp8 ONTO it: 2 accumulators at a time", blocks, 32768);
    run<5>("f16, then 2 fp8 ONTO it: 4 accumulators at a time", blocks, 32768);
    run<6>("f16, then 2 fp8 ONTO it: 8 accumulators at a time", blocks, 32768);
  }
  return 0;
}
