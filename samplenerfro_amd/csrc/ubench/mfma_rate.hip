// Sustained MFMA issue rate of THIS chip, measured in the run that quotes it (bench.py `roofline.sustained_mfma_tflops`): every CU runs
// one 4-wave workgroup issuing back-to-back v_mfma_f32_32x32x16_{f16,bf16} onto 8 alternating accumulators (48 MFMAs per loop trip, no
// memory traffic) — the matrix pipe's ceiling under the whole chip's power limit, which is what bounds the NerfMLP engines (3 such
// MFMAs per product in the fp32-grade modes).  Built into librnerf_ubench.so (NOT the product library): measurement infrastructure.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>      // 0: f16, 1: bf16
__global__ void __launch_bounds__(256, 1) mfma_rate_kernel(float* out, int iters) {
  half8 a, b;
  bf16x8 ab, bb;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x * 3 + j));
    ab[j] = (__bf16)(0.001f * (threadIdx.x + j)); bb[j] = (__bf16)(0.002f * (threadIdx.x * 3 + j));
  }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
      }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// One launch: `blocks` workgroups x 4 waves x iters x 48 MFMAs of 2 x 32 x 32 x 16 flop.  out: device scratch of blocks * 256 floats.
// Returns the flop count of the launch through *flop (so the caller's clock / this count is the rate); 0 = ok.
extern "C" int rnerf_ubench_mfma(int kind, int blocks, int iters, float* out, double* flop, void* stream) {
  if (blocks < 1 || iters < 1 || !out || (kind != 0 && kind != 1)) return -1;
  if (kind == 0) hipLaunchKernelGGL(mfma_rate_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  else hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  if (hipGetLastError() != hipSuccess) return -2;
  if (flop) *flop = (double)blocks * 4.0 * (double)iters * 48.0 * 2.0 * 32.0 * 32.0 * 16.0;
  return 0;
}
