// Background MLP (rnerf/models.py:181-191: MLP(128, 4, skip 2, out 3) on pos_enc(dir, 0, 4)): parameter / saved-tensor layouts shared by the
// exact-fp32 kernels (csrc/mlp.hip) and the f16 hi + lo ones (csrc/bkgd16.hip).
#pragma once
#include "nerfmlp_layout.h"

namespace rnerf {

__host__ __device__ constexpr DenseShape bkgd_dense(int d) {
  constexpr DenseShape t[5] = {{27, 128}, {128, 128}, {128, 128}, {155, 128}, {128, 3}};
  return t[d];
}
__host__ __device__ constexpr int bkgd_koff(int d) {
  int o = 0;
  for (int i = 0; i < d; ++i) o += bkgd_dense(i).in * bkgd_dense(i).out + bkgd_dense(i).out;
  return o;
}
__host__ __device__ constexpr int bkgd_boff(int d) { return bkgd_koff(d) + bkgd_dense(d).in * bkgd_dense(d).out; }
static_assert(bkgd_koff(5) == RNERF_BKGDMLP_PARAMS, "bkgd MLP parameter count");

// K=2 MFMA steps over the 27-d direction encoding: step q (0..13): half 0 / half 1 feature
__host__ __device__ constexpr int dir_feature(int q, int h) { return q < 12 ? (h ? 15 + q : 3 + q) : (q == 12 ? (h ? 2 : 0) : (h ? -1 : 1)); }

// save (training forward), fp32 row-major: [enc: n x 28][X1: n x 128][X2][X3][X4][out: n x 3]  (X_k = ReLU'd input of Dense_k)
__host__ __device__ constexpr size_t bkgd_save_floats(long long n) { return (size_t)n * (28 + 4 * 128 + 3); }
// scratch: [dY0..dY3: n x 128][d raw: n x 4, padded to n x 128 so that dY_k = base + k*n*128][wgrad partials: chunks x params]
__host__ __device__ constexpr size_t bkgd_dy_floats(long long n) { return (size_t)n * (5 * 128) + (size_t)((n + 255) / 256) * RNERF_BKGDMLP_PARAMS; }

// csrc/bkgd16.hip
int launch_bkgd16_fwd(bool train, const float* params, const float* dirs, int dir_stride, long long n, float pad_scale, float pad, float* out_rgb,
                      float* save, hipStream_t st);

}  // namespace rnerf
