#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool VARY_A>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int iters) {
  half8 a[8], b;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.001f * (threadIdx.x + i + j));
  for (int j = 0; j < 8; ++j) b[j] = (_Float16)(0.002f * (threadIdx.x * 3 + j));
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 48 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(VARY_A ? a[(r + i) & 7] : a[0], b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, bool VARY_A>
void run(const char* name, int blocks) {
  float* out; unsigned long long* cyc; hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 8);
  int iters = 2000;
  hipLaunchKernelGGL((k<NACC, VARY_A>), dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, VARY_A>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double n = 48.0 * iters;
  printf("%-28s blocks=%d  memtime ticks/MFMA=%.1f  ns/MFMA=%.2f  -> %.0f TF/s chip-equivalent\n", name, blocks, c / n, ms * 1e6 / n,
         (double)blocks * 4 * n * 32768 / (ms * 1e-3) / 1e12);
}
int main() {
  run<2, false>("2 acc, same A", 256); run<4, false>("4 acc, same A", 256); run<8, false>("8 acc, same A", 256); run<16, false>("16 acc same A", 256);
  run<2, true>("2 acc, vary A", 256); run<4, true>("4 acc, vary A", 256); run<8, true>("8 acc, vary A", 256);
  run<8, true>("8 acc, vary A, 1 block", 1); run<8, true>("8 acc vary A, 512 blocks", 512);
  return 0;
}
