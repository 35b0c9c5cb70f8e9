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
// ... followed by the ReLU masks the dgrad chain reads INSTEAD of the activations (round 5): uint32[4 layers][n][4] = 128 bits per row and
// layer, bit f <=> X_k[row][f] > 0.  (The dgrad used to re-read the fp32 activations — 2 KB per row, 512-byte strides between a wave's
// lanes — only for their signs: the three Jacobian passes of a stage-all* step were bound by exactly that traffic.)
__host__ __device__ constexpr size_t so3_mask_words(long long n) { return (size_t)n * 16; }
__host__ __device__ constexpr size_t so3_save_bytes_total(long long n) { return (so3_save_floats(n) + so3_mask_words(n)) * 4; }
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

// the training forwards' mask store: xx = the layer's post-ReLU outputs in the accumulator layout (register r of tile t, half h <-> feature
// 32 t + 8 (r >> 2) + 4 h + (r & 3)); the two half-lanes of a row combine their 16 bits per dword, lane h == 0 writes the row's 16 bytes
template <typename X4>
__device__ __forceinline__ void so3_store_mask(float* save, long long n, long long row, int k, const X4& xx, int h, bool ok) {
  uint32_t w[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    uint32_t b = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) b |= (xx[t][r] > 0.f ? 1u : 0u) << ((r & 3) + 8 * (r >> 2) + 4 * h);
    w[t] = b | (uint32_t)__shfl_xor((int)b, 32);
  }
  if (ok && h == 0) {
    uint32_t* m = (uint32_t*)(save + so3_save_floats(n)) + ((size_t)(k - 1) * (size_t)n + (size_t)row) * 4;
    *(uint4*)m = make_uint4(w[0], w[1], w[2], w[3]);
  }
}
__device__ __forceinline__ uint4 so3_load_mask(const float* save, long long n, long long row, int k) {
  return *(const uint4*)((const uint32_t*)(save + so3_save_floats(n)) + ((size_t)(k - 1) * (size_t)n + (size_t)row) * 4);
}

// csrc/bkgd16.hip: the training forward on f16 hi + lo MFMAs (same saved layout)
int launch_so3_16_fwd_train(const float* params, So3Window win, const float* pts4, long long n, float* save, hipStream_t st);

}  // namespace rnerf
