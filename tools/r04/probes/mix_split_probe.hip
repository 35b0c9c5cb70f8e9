// Probe: the three-instruction hi + lo f16 split (v_cvt_pk_f16_f32 + v_fma_mixlo_f16 + v_fma_mixhi_f16) against the plain one.
// build: hipcc --offload-arch=gfx950 -O2 tools/r04/probes/mix_split_probe.hip -o /tmp/mix_split_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__global__ void k(const float* a, uint32_t* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a[2 * i], y = a[2 * i + 1];
  uint32_t hi, lo;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(x));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(y));
  half2v h = {(_Float16)x, (_Float16)y};
  half2v l = {(_Float16)(x - (float)h[0]), (_Float16)(y - (float)h[1])};
  o[4 * i] = hi; o[4 * i + 1] = lo; o[4 * i + 2] = __builtin_bit_cast(uint32_t, h); o[4 * i + 3] = __builtin_bit_cast(uint32_t, l);
}
int main() {
  const int n = 1 << 16;
  std::vector<float> a(2 * n);
  uint32_t s = 12345;
  for (auto& v : a) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 23) * 300.f; }
  a[0] = 1.0f; a[1] = -0.3f; a[2] = 70000.f; a[3] = 1e-6f;
  float* da; uint32_t* d;
  hipMalloc(&da, a.size() * 4); hipMalloc(&d, n * 16);
  hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, d, n);
  std::vector<uint32_t> o(4 * n);
  hipMemcpy(o.data(), d, n * 16, hipMemcpyDeviceToHost);
  int bad_hi = 0, bad_lo = 0;
  for (int i = 0; i < n; ++i) { bad_hi += o[4 * i] != o[4 * i + 2]; bad_lo += o[4 * i + 1] != o[4 * i + 3]; }
  printf("pairs %d: hi words differing %d, lo words differing %d\n", n, bad_hi, bad_lo);
  for (int i = 0; i < 3; ++i) printf("  x %g y %g: asm hi %08x lo %08x | plain hi %08x lo %08x\n", a[2 * i], a[2 * i + 1], o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
  return 0;
}
