// Shared pieces of the matrix-core kernels (csrc/mlp.hip: NerfMLP engines; csrc/bkgd16.hip: the background MLP on f16 hi + lo operands):
// vector types, the 16-bit hi / lo split, the 32x32x16 MFMA wrapper and the operand-slot -> feature maps of the transposed chain
// (lane (row m = lane & 31, half h = lane >> 5) of an accumulator holds features n = 32 t + (r & 3) + 8 (r >> 2) + 4 h — see mlp.hip).
#pragma once
#include "common.h"

namespace rnerf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// ---- 16-bit packing -------------------------------------------------------------------------------------------------
template <bool F16>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  f32x2 v = {a, b};
  if constexpr (F16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, half2v));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
template <bool F16>
__device__ __forceinline__ void unpack2(uint32_t p, float& a, float& b) {
  if constexpr (F16) {
    half2v hv = __builtin_bit_cast(half2v, p);
    a = (float)hv[0]; b = (float)hv[1];
  } else {
    a = __uint_as_float(p << 16); b = __uint_as_float(p & 0xffff0000u);
  }
}
// hi = round16(x), lo = round16(x - hi)
template <bool F16>
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack2<F16>(a, b);
  float ha, hb;
  unpack2<F16>(hi, ha, hb);
  lo = pack2<F16>(a - ha, b - hb);
}

template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const uint4 a, const uint4 b, const f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- operand slot -> input feature maps (shared by the pack kernel and the forward kernel) -----------------------------
// previous-layer activations: k-step s, half h, slot j  ->  feature index
__host__ __device__ constexpr int prev_feature(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }
// 63-d position encoding [x(3) | sin(2^d x)(30) | sin(2^d x + pi/2)(30)] (rnerf/model_utils.py:211-214):
// slot q = 8*s + j (0..31); half 0 carries the sin block, half 1 the cos block, the identity terms ride in q = 30, 31.
__host__ __device__ constexpr int pe_feature(int q, int h) { return q < 30 ? (h ? 33 + q : 3 + q) : (q == 30 ? (h ? 2 : 0) : (h ? -1 : 1)); }
// 27-d view encoding [d(3) | sin(2^k d)(12) | sin(2^k d + pi/2)(12)], slot q = 0..15
__host__ __device__ constexpr int view_feature(int q, int h) { return q < 12 ? (h ? 15 + q : 3 + q) : (q == 12 ? (h ? 2 : 0) : (q == 13 ? (h ? -1 : 1) : -1)); }

#define RNERF_PIN() __builtin_amdgcn_sched_barrier(0)

}  // namespace rnerf
