// Operand layout and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3) on gfx950, probed from the host
// (tools/ubench/f8f6f4_probe.py drives it): one wave computes D = A x B from per-lane operand bytes and per-lane scale bytes.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/ubench/f8f6f4_probe.hip -o libf8probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(64) probe_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, const int* __restrict__ sa,
                                                   const int* __restrict__ sb, float* __restrict__ d) {
  const int l = threadIdx.x;
  i32x8 A, B;
  for (int j = 0; j < 8; ++j) { A[j] = ((const int*)(a + 32 * l))[j]; B[j] = ((const int*)(b + 32 * l))[j]; }
  f32x16 acc;
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0 /* A fp8 e4m3 */, 0 /* B fp8 e4m3 */, 0, sa[l], 0, sb[l]);
  for (int j = 0; j < 16; ++j) d[16 * l + j] = acc[j];
}

extern "C" int f8_probe(const void* a, const void* b, const void* sa, const void* sb, void* d) {
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, (const uint8_t*)a, (const uint8_t*)b, (const int*)sa, (const int*)sb, (float*)d);
  return (int)hipDeviceSynchronize();
}
