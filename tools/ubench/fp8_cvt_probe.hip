// What v_cvt_pk_fp8_f32 / v_cvt_scalef32_pk_fp8_f32 produce on gfx950 (format, saturation, direction of the scale) and whether
// v_mfma_f32_32x32x16_fp8_fp8 reads the same format: groundwork for the f16 + fp8-cross-term precision of the NerfMLP forward.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/fp8_cvt_probe.hip -o /tmp/fp8_cvt_probe && /tmp/fp8_cvt_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v2i16 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void cvt(const float* in, int n, float scale, int* plain, int* scaled) {
  const int i = threadIdx.x;
  if (i >= n) return;
  plain[i] = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], 0.f, 0, false) & 0xff;
  v2i16 o = {0, 0};
  o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, in[i], 0.f, scale, false);
  scaled[i] = o[0] & 0xff;
}
__global__ void mm(const float* a, const float* b, float* d) {      // lane l: A row l % 32, B column l % 32, k = 8 (l / 32) + j
  const int l = threadIdx.x;
  long A = 0, B = 0;
  for (int j = 0; j < 8; j += 2) {
    const int wa = __builtin_amdgcn_cvt_pk_fp8_f32(a[8 * l + j], a[8 * l + j + 1], 0, false) & 0xffff;
    const int wb = __builtin_amdgcn_cvt_pk_fp8_f32(b[8 * l + j], b[8 * l + j + 1], 0, false) & 0xffff;
    A |= (long)wa << (8 * j); B |= (long)wb << (8 * j);
  }
  f32x16 acc;
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(A, B, acc, 0, 0, 0);
  for (int j = 0; j < 16; ++j) d[16 * l + j] = acc[j];
}
int main() {
  const float vals[] = {1.0f, 1.5f, 0.5f, 448.f, 449.f, 480.f, 1000.f, 1e9f, 0.015625f, 0.001953125f, 0.0009f, -2.0f, 0.3f, 3.3f, 17.f, 240.f};
  const int n = sizeof(vals) / sizeof(float);
  float* din; int *dp, *ds; hipMalloc(&din, sizeof(vals)); hipMalloc(&dp, 4 * n); hipMalloc(&ds, 4 * n);
  hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
  for (float scale : {1.0f, 4.0f, 0.25f}) {
    hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, din, n, scale, dp, ds);
    int hp[32], hs[32]; hipMemcpy(hp, dp, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, 4 * n, hipMemcpyDeviceToHost);
    printf("scale operand %g:\n", scale);
    for (int i = 0; i < n; ++i) printf("  %-12g plain 0x%02x   scaled 0x%02x\n", vals[i], hp[i], hs[i]);
  }
  // A[i][k] = (i % 5 + 1) * 0.5 for k = 3, else 0;  B[k][j] = (j % 7 + 1) * 0.25 for k = 3 and k = 12: D[i][j] = A[i][3] * B[3][j]
  float ha[64 * 8] = {0}, hb[64 * 8] = {0}, hd[64 * 16];
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
    const int k = 8 * (l / 32) + j, r = l % 32;
    ha[8 * l + j] = k == 3 ? (r % 5 + 1) * 0.5f : (k == 12 ? 2.0f : 0.f);
    hb[8 * l + j] = k == 3 ? (r % 7 + 1) * 0.25f : (k == 12 ? 0.5f : 0.f);
  }
  float *da, *db, *dd; hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, sizeof(hd));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
    const float want = (i % 5 + 1) * 0.5f * (j % 7 + 1) * 0.25f + 2.0f * 0.5f;
    if (hd[16 * l + r] != want) { if (bad < 5) printf("  D[%d][%d] = %g, want %g\n", i, j, hd[16 * l + r], want); ++bad; }
  }
  printf("fp8 32x32x16 MFMA on operands made by v_cvt_pk_fp8_f32, lane l = (row | column) l %% 32, k = 8 (l / 32) + byte: %s (%d mismatches)\n", bad ? "MISMATCH" : "exact", bad);
  return 0;
}
