// The background MLP on the f16 matrix cores (round 4): forward (inference / training) of MLP(128, 4, skip 2, out 3) on pos_enc(dir, 0, 4)
// + the rgb activation.  Reference: rnerf/models.py:181-191 (forward_envmap), :303, :336-337; rnerf/model_utils.py:93-140 (MLP), :187-214.
// The exact-fp32 kernels of the same network (the arbiter of this file: RNERF_BKGD_EXACT=1) and the backward kernels live in csrc/mlp.hip.
#include "mfma_ops.h"
#include "bkgd_layout.h"
#include "so3_layout.h"

namespace rnerf {

// ---- the background MLP on the f16 matrix cores (round 4) ---------------------------------------------------------------------------
// The exact-fp32 chain above (v_mfma_f32_32x32x2_f32: 64 cycles per K = 2 step, ~900 per 32 rows) is bound by the latency of its one wave
// per row block: 64 us for the 20 480 rows of a bench step, at the head of every step.  The same network with the f16 hi + lo split of the
// NerfMLP engine (3 x v_mfma_f32_32x32x16_f16 per tile: 340 MFMAs of 32 cycles per 32 rows, fp32 accumulate, weights x 2^8 so that the lo
// parts stay normal; error class 2^-22 like f16x3) — the weights are converted on the fly from the flat fp32 buffer (every wave reads the
// 57 k parameters through L1 / L2: no packed stream, no change to the C ABI), the transposed chain keeps the activations in registers
// (accumulator layout = next layer's B operand: prev_feature / view_feature slot maps as in the NerfMLP engine).  Saved tensors of the
// training forward: the same fp32 layout as bkgd_fwd_kernel (the backward kernels do not care which arithmetic produced X_k).
// RNERF_BKGD_EXACT=1 selects the exact-fp32 kernels (they stay the arbiter of this one: tests/test_gpu_parity.py).
// (A three-instruction split — v_cvt_pk_f16_f32, then v_fma_mixlo_f16 / v_fma_mixhi_f16 subtracting the f16 half straight from the packed
//  word — gives the same bits alone (tools/r04/probes/mix_split_probe.hip) but NaNs in this kernel: inline asm hides the partial-register
//  writes from the compiler's hazard recogniser.  It bought 1 us of 21: the kernel is bound by latency, not by its conversions.)
__device__ __forceinline__ void bkgd16_split2(float a, float b, uint32_t& hi, uint32_t& lo) { split2<true>(a, b, hi, lo); }
__device__ __forceinline__ void bkgd16_split8(const float (&w)[8], uint4& hi, uint4& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) bkgd16_split2(w[2 * p], w[2 * p + 1], h[p], l[p]);
  hi = make_uint4(h[0], h[1], h[2], h[3]); lo = make_uint4(l[0], l[1], l[2], l[3]);
}
// One layer's matrix part: acc[t] (+)= W^T(k-step s, n-tile t) x^T(s) for s = 0 .. NS-1, 3 MFMAs per tile (FIRST: the first k-step starts
// from zero accumulators).  wload(s, t, j): the A operand's slot j of lane (n = 32 t + (lane & 31), half = lane >> 5) — the element [row of
// slot j][n] of the layer's kernel, a pure load from a per-lane base pointer + a CONSTANT (immediate) offset: 32 dwords per lane and
// k-step, 128 consecutive bytes per half-wave and load; live(s, j) = 0: zero padding (0: never, 1: always, 2: ask lanes(s, j)).  The BIAS is
// one more such row, met by a 1.0 in the B operand (the free slot 14 of the direction encoding's second k-step, or a ninth k-step of its
// own): split into hi + lo like every weight, no separate add.  The loads run DEPTH k-steps ahead of the MFMAs that consume them (left to
// itself hipcc waits for every load where it is issued: an L2 round trip per product, the same disease as mfma_f32_stream's).
// xop(s, bh, bl): the hi / lo B operands of k-step s.
template <int NS, bool FIRST, typename WF, typename LF, typename XF>
__device__ __forceinline__ void bkgd16_layer(f32x16 (&acc)[4], WF wload, LF live, bool half0, XF xop) {
  constexpr int DEPTH = 2;
  f32x2 w[DEPTH + 1][16];
  auto load = [&](int s, f32x2 (&dst)[16]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (live(s, j) != 0) dst[4 * t + (j >> 1)][j & 1] = wload(s, t, j);
  };
#pragma unroll
  for (int s = 0; s < DEPTH && s < NS; ++s) load(s, w[s]);
  f32x16 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (s + DEPTH < NS) load(s + DEPTH, w[(s + DEPTH) % (DEPTH + 1)]);
    RNERF_PIN();
    uint4 bh, bl;
    xop(s, bh, bl);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint32_t hh[4], ll[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int l0 = live(s, 2 * p), l1 = live(s, 2 * p + 1);      // 0: padding, 1: every lane, 2: half 0 only
        if (l0 == 0 && l1 == 0) { hh[p] = 0u; ll[p] = 0u; continue; }
        f32x2 q = w[s % (DEPTH + 1)][4 * t + p];
        if (l0 == 0) q[0] = 0.f;
        if (l1 == 0) q[1] = 0.f;
        q = q * 256.0f;                                               // x 2^8 as a packed fp32 multiply (one instruction per pair)
        if (l0 == 2) q[0] = half0 ? q[0] : 0.f;
        if (l1 == 2) q[1] = half0 ? q[1] : 0.f;
        bkgd16_split2(q[0], q[1], hh[p], ll[p]);
      }
      const uint4 ah = make_uint4(hh[0], hh[1], hh[2], hh[3]), al = make_uint4(ll[0], ll[1], ll[2], ll[3]);
      acc[t] = mfma16<true>(ah, bh, FIRST && s == 0 ? zero : acc[t]);
      acc[t] = mfma16<true>(ah, bl, acc[t]);
      acc[t] = mfma16<true>(al, bh, acc[t]);
    }
    RNERF_PIN();
  }
}
// per-lane base pointers into the flat parameters (computed once): every load of the kernel is one of them + an immediate offset
struct Bkgd16Lane {
  const float* p0;      // params + m
  const float* prev;    // + 4 h rows: prev_feature(s, h, j) = 16 s + 8 (j >> 2) + (j & 3)  + 4 h
  const float* dir;     // + 12 h rows: view_feature(q < 12, h) = 3 + q + 12 h
  const float* sp;      // + 2 h rows: view_feature(12, h) = 2 h
  bool half0;
};
__device__ __forceinline__ Bkgd16Lane bkgd16_lane(const float* __restrict__ params, int m, int h) {
  return {params + m, params + m + 4 * 128 * h, params + m + 12 * 128 * h, params + m + 2 * 128 * h, h == 0};
}
// acc = W[0..127][:] x + bias for a 128-wide previous activation x (fp32, accumulator layout); BIAS: with the ninth k-step [1 | 0 ...] x bias
template <bool BIAS>
__device__ __forceinline__ void bkgd16_prev_layer(f32x16 (&acc)[4], const f32x16 (&x)[4], const Bkgd16Lane& L, int koff, int boff) {
  bkgd16_layer<BIAS ? 9 : 8, true>(acc,
      [&](int s, int t, int j) { return s < 8 ? L.prev[koff + (16 * s + 8 * (j >> 2) + (j & 3)) * 128 + 32 * t] : L.p0[boff + 32 * t]; },
      [&](int s, int j) { return s < 8 ? 1 : (j == 0 ? 2 : 0); }, L.half0,
      [&](int s, uint4& bh, uint4& bl) {
    if (s < 8) {
      float xv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j] = x[s >> 1][8 * (s & 1) + j];
      bkgd16_split8(xv, bh, bl);
    } else {
      bh = make_uint4(L.half0 ? 0x3c00u : 0u, 0u, 0u, 0u);      // f16 1.0 in slot 0 of half 0
      bl = make_uint4(0u, 0u, 0u, 0u);
    }
  });
}
// acc (+)= W[rows of the 27-d direction encoding][:] enc + bias: two k-steps in the slot order of view_feature (= dir_feature for the 14 used
// slots: q < 12: rows 3 + q | 15 + q, q = 12: rows 0 | 2, q = 13: row 1 | nothing), the bias in slot 14 of half 0
template <bool FIRST>
__device__ __forceinline__ void bkgd16_dir_layer(f32x16 (&acc)[4], const float (&enc)[14], const Bkgd16Lane& L, int koff, int boff) {
  bkgd16_layer<2, FIRST>(acc,
      [&](int s, int t, int j) {
        const int q = 8 * s + j;
        return q < 12 ? L.dir[koff + (3 + q) * 128 + 32 * t] : (q == 12 ? L.sp[koff + 32 * t] : (q == 13 ? L.p0[koff + 128 + 32 * t] : L.p0[boff + 32 * t]));
      },
      [&](int s, int j) { const int q = 8 * s + j; return q < 13 ? 1 : (q < 15 ? 2 : 0); }, L.half0,
      [&](int s, uint4& bh, uint4& bl) {
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = 8 * s + j < 14 ? enc[8 * s + j] : (8 * s + j == 14 && L.half0 ? 1.0f : 0.f);
    bkgd16_split8(xv, bh, bl);
  });
}

template <bool TRAIN>
__global__ void __launch_bounds__(64) bkgd16_fwd_kernel(const float* __restrict__ params, const float* __restrict__ dirs, int dir_stride,
                                                        long long n, float pad_scale, float pad, float* __restrict__ out_rgb,
                                                        float* __restrict__ save) {
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  long long row = (long long)blockIdx.x * 32 + m;
  const bool ok = row < n;
  if (!ok) row = n - 1;
  const float v0 = dirs[row * dir_stride], v1 = dirs[row * dir_stride + 1], v2 = dirs[row * dir_stride + 2];
  float enc[14];                                     // as bkgd_fwd_kernel: the same values in the same slots
  const float phase = h ? 1.5707963705062866f : 0.0f;
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const int d = q / 3, c = q % 3;
    const float x = c == 0 ? v0 : (c == 1 ? v1 : v2);
    enc[q] = sinf(fadd(fmul(x, (float)(1 << d)), phase));
  }
  enc[12] = h ? v2 : v0;
  enc[13] = h ? 0.f : v1;
  auto save_x = [&](int k, const f32x16 (&xx)[4]) {     // X_k[row][f], f = 32t + 8g + 4h + i
    if constexpr (TRAIN) {
      if (ok) {
        float* dst = save + (size_t)n * 28 + (size_t)(k - 1) * n * 128 + (size_t)row * 128;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *(float4*)(dst + 32 * t + 8 * g + 4 * h) = make_float4(xx[t][4 * g], xx[t][4 * g + 1], xx[t][4 * g + 2], xx[t][4 * g + 3]);
      }
    }
  };
  if constexpr (TRAIN) {
    if (ok) {
#pragma unroll
      for (int q = 0; q < 14; ++q) { const int f = h ? dir_feature(q, 1) : dir_feature(q, 0); if (f >= 0) save[(size_t)row * 28 + f] = enc[q]; }
      if (h == 1) save[(size_t)row * 28 + 27] = 0.f;
    }
  }
  constexpr float INV = 1.0f / 256.0f;
  f32x16 acc[4], x[4];
  auto relu_to_x = [&]() {      // x = ReLU(acc 2^-8)   (not fmaxf: a NaN of an out-of-range f16 operand must reach the output)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float y = acc[t][r] * INV; x[t][r] = y < 0.f ? 0.f : y; }
  };
  const Bkgd16Lane L = bkgd16_lane(params, m, h);
  // Dense_0: 27 -> 128, ReLU
  bkgd16_dir_layer<true>(acc, enc, L, bkgd_koff(0), bkgd_boff(0));
  relu_to_x();
  save_x(1, x);
  // Dense_1, Dense_2: 128 -> 128, ReLU
  bkgd16_prev_layer<true>(acc, x, L, bkgd_koff(1), bkgd_boff(1));
  relu_to_x();
  save_x(2, x);
  bkgd16_prev_layer<true>(acc, x, L, bkgd_koff(2), bkgd_boff(2));
  relu_to_x();
  save_x(3, x);
  // Dense_3: [x(128), inputs(27)] -> 128, ReLU  (skip concat after i == 2, rnerf/model_utils.py:131-132)
  bkgd16_prev_layer<false>(acc, x, L, bkgd_koff(3), 0);
  bkgd16_dir_layer<false>(acc, enc, L, bkgd_koff(3) + 128 * 128, bkgd_boff(3));
  relu_to_x();
  if constexpr (TRAIN) save_x(4, x);
  // Dense_4: 128 -> 3 on the VALU, then sigmoid*(1+2p)-p (rnerf/models.py:336-337)
  float o[3] = {0.f, 0.f, 0.f};
  const float* __restrict__ k4 = params + bkgd_koff(4);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = x[t][r];
      const int f = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
      o[0] = fmaf(v, k4[f * 3 + 0], o[0]); o[1] = fmaf(v, k4[f * 3 + 1], o[1]); o[2] = fmaf(v, k4[f * 3 + 2], o[2]);
    }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o[c] = o[c] + __shfl_xor(o[c], 32) + params[bkgd_boff(4) + c];
    o[c] = fsub(fmul(fdiv(1.0f, fadd(1.0f, expf(-o[c]))), pad_scale), pad);
  }
  if (ok && h == 0) {
    out_rgb[3 * row] = o[0]; out_rgb[3 * row + 1] = o[1]; out_rgb[3 * row + 2] = o[2];
    if constexpr (TRAIN) { float* so = save + (size_t)n * (28 + 4 * 128) + (size_t)row * 3; so[0] = o[0]; so[1] = o[1]; so[2] = o[2]; }
  }
}

int launch_bkgd16_fwd(bool train, const float* params, const float* dirs, int dir_stride, long long n, float pad_scale, float pad, float* out_rgb,
                      float* save, hipStream_t st) {
  const dim3 grid((unsigned)((n + 31) / 32));
  if (train) hipLaunchKernelGGL(bkgd16_fwd_kernel<true>, grid, dim3(64), 0, st, params, dirs, dir_stride, n, pad_scale, pad, out_rgb, save);
  else hipLaunchKernelGGL(bkgd16_fwd_kernel<false>, grid, dim3(64), 0, st, params, dirs, dir_stride, n, pad_scale, pad, out_rgb, save);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

// ---- so3_mlp: the training forward of stage "all*" on the same arithmetic --------------------------------------------------------------
// MLP(128, 4, skip 2, out 3) on the 60 windowed encoding features (rnerf/ior_utils.py:148-152, rnerf/model_utils.py:236-245): lane half h
// holds feature 2 p + h in enc[p] (so3_encode), i.e. four k-steps of 8 slots with p = 8 s + j, slot p = 30 free for the bias.  The prev-layer
// blocks are the background MLP's (same [128][128] kernels, same slot map).  Saved tensors: the layout of so3_fwd_train_kernel (the exact-fp32
// kernel of csrc/ior_train_kernels.inc, which stays selectable with RNERF_BKGD_EXACT=1 and is what the backward kernels were written against).
template <bool FIRST>
__device__ __forceinline__ void so3_16_enc_layer(f32x16 (&acc)[4], const float (&enc)[30], const float* __restrict__ pe, const Bkgd16Lane& L, int koff, int boff) {
  bkgd16_layer<4, FIRST>(acc,
      [&](int s, int t, int j) { const int p = 8 * s + j; return p < 30 ? pe[koff + 2 * p * 128 + 32 * t] : L.p0[boff + 32 * t]; },
      [&](int s, int j) { const int p = 8 * s + j; return p < 30 ? 1 : (p == 30 ? 2 : 0); }, L.half0,
      [&](int s, uint4& bh, uint4& bl) {
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int p = 8 * s + j; xv[j] = p < 30 ? enc[p < 30 ? p : 0] : ((p == 30 && L.half0) ? 1.0f : 0.f); }
    bkgd16_split8(xv, bh, bl);
  });
}

__global__ void __launch_bounds__(64) so3_16_fwd_train_kernel(const float* __restrict__ params, So3Window win, const float4* __restrict__ pts, long long n,
                                                              float* __restrict__ save) {
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  long long row = (long long)blockIdx.x * 32 + m;
  const bool ok = row < n;
  if (!ok) row = n - 1;
  const float4 pt = pts[row];
  float enc[30];
  so3_encode(pt.x, pt.y, pt.z, win, h, enc);
  if (ok) {
#pragma unroll
    for (int p = 0; p < 30; ++p) save[(size_t)row * 60 + 2 * p + h] = enc[p];
  }
  auto save_x = [&](int k, const f32x16 (&xx)[4]) {     // X_k[row][f], f = 32t + 8g + 4h + i
    so3_store_mask(save, n, row, k, xx, h, ok);         // + the 128 sign bits the dgrad reads (every lane takes part in the half-lane exchange)
    if (ok) {
      float* dst = save + (size_t)n * 60 + (size_t)(k - 1) * n * 128 + (size_t)row * 128;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(dst + 32 * t + 8 * g + 4 * h) = make_float4(xx[t][4 * g], xx[t][4 * g + 1], xx[t][4 * g + 2], xx[t][4 * g + 3]);
    }
  };
  constexpr float INV = 1.0f / 256.0f;
  f32x16 acc[4], x[4];
  auto relu_to_x = [&]() {      // x = ReLU(acc 2^-8)   (not fmaxf: a NaN of an out-of-range f16 operand must reach the output)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float y = acc[t][r] * INV; x[t][r] = y < 0.f ? 0.f : y; }
  };
  const Bkgd16Lane L = bkgd16_lane(params, m, h);
  const float* __restrict__ pe = params + m + 128 * h;       // kernel row 2 p + h of an encoding block = pe[block + 2 p * 128]
  so3_16_enc_layer<true>(acc, enc, pe, L, so3_koff(0), so3_boff(0));
  relu_to_x();
  save_x(1, x);
  bkgd16_prev_layer<true>(acc, x, L, so3_koff(1), so3_boff(1));
  relu_to_x();
  save_x(2, x);
  bkgd16_prev_layer<true>(acc, x, L, so3_koff(2), so3_boff(2));
  relu_to_x();
  save_x(3, x);
  bkgd16_prev_layer<false>(acc, x, L, so3_koff(3), 0);                                   // Dense_3: [x(128), inputs(60)] (skip concat after i == 2)
  so3_16_enc_layer<false>(acc, enc, pe, L, so3_koff(3) + 128 * 128, so3_boff(3));
  relu_to_x();
  save_x(4, x);
  float o[3] = {0.f, 0.f, 0.f};
  const float* __restrict__ k4 = params + so3_koff(4);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
      o[0] = fmaf(x[t][r], k4[f * 3 + 0], o[0]); o[1] = fmaf(x[t][r], k4[f * 3 + 1], o[1]); o[2] = fmaf(x[t][r], k4[f * 3 + 2], o[2]);
    }
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = o[c] + __shfl_xor(o[c], 32) + params[so3_boff(4) + c];
  if (ok && h == 0) *(float4*)(save + (size_t)n * (60 + 4 * 128) + (size_t)row * 4) = make_float4(o[0], o[1], o[2], 0.f);
}

int launch_so3_16_fwd_train(const float* params, So3Window win, const float* pts4, long long n, float* save, hipStream_t st) {
  hipLaunchKernelGGL(so3_16_fwd_train_kernel, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, st, params, win, (const float4*)pts4, n, save);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

}  // namespace rnerf
