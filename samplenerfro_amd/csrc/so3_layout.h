// so3_mlp (rnerf/ior_utils.py:148-152: MLP(128, 4, skip 2, out 3) on annealed_pos_enc(x)): parameter / saved-tensor layouts and the windowed
// encoding, shared by the exact-fp32 kernels (csrc/mlp.hip, csrc/ior_train_kernels.inc) and the f16 hi + lo training forward (csrc/bkgd16.hip).
#pragma once
#include "nerfmlp_layout.h"

namespace rnerf {

__host__ __device__ constexpr DenseShape so3_dense(int d) {
  constexpr DenseShape t[5] = {{60, 128}, {128, 128}, {128, 128}, {188, 128}, {128, 3}};
  return t[d];
}
__host__ __device__ constexpr int so3_koff(int d) {
  int o = 0;
  for (int i = 0; i < d; ++i) o += so3_dense(i).in * so3_dense(i).out + so3_dense(i).out;
  return o;
}
__host__ __device__ constexpr int so3_boff(int d) { return so3_koff(d) + so3_dense(d).in * so3_dense(d).out; }
static_assert(so3_koff(5) == RNERF_SO3MLP_PARAMS, "so3 MLP parameter count");

struct So3Window { float w[10]; };   // cosine_easing_window(0, 9, 10, annealed_alpha * 10), computed by the host (model_utils.py:218-233)

// save (training forward), fp32 row-major: [enc: n x 60][X1: n x 128][X2][X3][X4][raw: n x 4]
__host__ __device__ constexpr size_t so3_save_floats(long long n) { return (size_t)n * (60 + 4 * 128 + 4); }
// scratch of the backward: [dY0..dY3: nb x 128][d raw: nb x 4, padded to nb x 128][wgrad partials: chunks x params]
__host__ __device__ constexpr size_t so3_dy_floats(long long nb) { return (size_t)nb * (5 * 128) + (size_t)((nb + 255) / 256) * RNERF_SO3MLP_PARAMS; }

__device__ __forceinline__ void so3_encode(float px, float py, float pz, const So3Window& win, int h, float (&enc)[30]) {
  const float HALF_PI = 1.5707963705062866f;
#pragma unroll
  for (int p = 0; p < 30; ++p) {
    const int d = p / 3, k = p % 3;
    const float x = k == 0 ? (h ? py : px) : (k == 1 ? (h ? px : pz) : (h ? pz : py));
    const float phase = k == 0 ? 0.f : (k == 1 ? (h ? HALF_PI : 0.f) : HALF_PI);
    const float xb = fmul(x, (float)(1 << d));
    enc[p] = fmul(pe_sin(k == 0 ? xb : fadd(xb, phase)), win.w[d]);      // the march's own sine (so3_eval_wg): the adjoint linearises the function that ran
  }
}

// csrc/bkgd16.hip: the training forward on f16 hi + lo MFMAs (same saved layout)
int launch_so3_16_fwd_train(const float* params, So3Window win, const float* pts4, long long n, float* save, hipStream_t st);

}  // namespace rnerf
