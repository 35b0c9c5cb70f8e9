// Positional encoding + MLP evaluation on the matrix cores (P1, N1, N2).
// Reference: rnerf/model_utils.py:187-214 (pos_enc), :30-90 (NerfMLP), :93-140 (MLP); call sites rnerf/models.py:257,289,
// 303-308,394,426,441 and :181-191 (forward_envmap).
//
// Design (gfx950, wave64, v_mfma_f32_32x32x16_{f16,bf16} / v_mfma_f32_32x32x2_f32):
//   * "Transposed chain": every layer is computed as  Y^T[n_out][row] = W^T[n_out][k] * X^T[k][row], i.e. the WEIGHTS are
//     the MFMA A operand and the 32 sample rows owned by a wave are the B operand / the lane index of the accumulator.
//     The C/D layout (lane = row, registers = output features) is then exactly a B-operand layout for the next layer: the
//     8 k-values a lane supplies per MFMA may be ANY 8 features as long as the A operand (pre-packed weights) uses the
//     same assignment, so activations never leave the register file between layers — no LDS round trip, no transpose,
//     no cross-lane exchange.  Lane (row m = lane&31, half h = lane>>5) holds features
//         n = 32*t + (r&3) + 8*(r>>2) + 4*h          (t = n-tile, r = accumulator register 0..15)
//     and feeds k-step s, slot j with feature 16*s + 8*(j>>2) + 4*h + (j&3)  (= accumulator (t=s>>1, r=8*(s&1)+j)).
//   * fp32 parity: the reference computes the MLP in fp32.  F16X3/BF16X3 split both operands into hi+lo 16-bit parts and
//     issue 3 MFMAs per tile (hi*hi + hi*lo + lo*hi, fp32 accumulate).  F16/BF16 issue one.
//   * A workgroup = 4 waves = 128 consecutive sample rows; the weight stream (2.3 MB for X3) is read from L2 through a
//     double-buffered LDS ring with global_load_lds (16 B/lane), one slab = 2 k-steps of one layer, shared by the 4 waves.
//   * Persistent grid: each workgroup walks row tiles with stride gridDim.x and prefetches across layer and tile seams.
#include "common.h"

#include <stdlib.h>

namespace rnerf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// ------------------------------------------------------------------------------------------------------------------
// Parameter layout of the flat fp32 NerfMLP buffer (flax creation order, rnerf/model_utils.py:58-89).
// ------------------------------------------------------------------------------------------------------------------
struct DenseShape { int in, out; };
__host__ __device__ constexpr DenseShape nerf_dense(int d) {
  constexpr DenseShape t[12] = {{63, 256},  {256, 256}, {256, 256}, {256, 256}, {256, 256}, {319, 256},
                                {256, 256}, {256, 256}, {256, 1},   {256, 256}, {283, 128}, {128, 3}};
  return t[d];
}
__host__ __device__ constexpr int nerf_koff(int d) {
  int o = 0;
  for (int i = 0; i < d; ++i) o += nerf_dense(i).in * nerf_dense(i).out + nerf_dense(i).out;
  return o;
}
__host__ __device__ constexpr int nerf_boff(int d) { return nerf_koff(d) + nerf_dense(d).in * nerf_dense(d).out; }
static_assert(nerf_koff(12) == RNERF_NERFMLP_PARAMS, "NerfMLP parameter count");

// The 10 MFMA layers: Dense_0..Dense_7 (trunk), Dense_9 (bottleneck), Dense_10 (view layer).
// kind: 0 = input is the 63-d positional encoding; 1 = previous activations; 2 = previous + PE (skip concat,
// rnerf/model_utils.py:68-69); 3 = previous (bottleneck) + 27-d view encoding (:82-83).
struct MfmaLayer { int dense, ks, nt, kind; };
__host__ __device__ constexpr MfmaLayer mfma_layer(int l) {
  constexpr MfmaLayer t[10] = {{0, 4, 8, 0},  {1, 16, 8, 1}, {2, 16, 8, 1}, {3, 16, 8, 1}, {4, 16, 8, 1},
                               {5, 20, 8, 2}, {6, 16, 8, 1}, {7, 16, 8, 1}, {9, 16, 8, 1}, {10, 18, 4, 3}};
  return t[l];
}
__host__ __device__ constexpr int layer_blocks_before(int l) {   // in units of (kstep, ntile) 1-KiB-per-part blocks
  int o = 0;
  for (int i = 0; i < l; ++i) o += mfma_layer(i).ks * mfma_layer(i).nt;
  return o;
}
constexpr int kTotalBlocks = layer_blocks_before(10);   // 1160

// aux section (floats) that follows the weight stream: biases in natural feature order (scaled by WSCALE), heads.
constexpr int AUX_BIAS = 0;            // 10 layers x 256
constexpr int AUX_WSIG = 2560;         // Dense_8 kernel [256]
constexpr int AUX_BSIG = 2816;         // Dense_8 bias (+3 pad)
constexpr int AUX_WRGB = 2820;         // Dense_11 kernel transposed [3][128]
constexpr int AUX_BRGB = 3204;         // Dense_11 bias (+1 pad)
constexpr int AUX_FLOATS = 3208;

template <int PREC>
struct Prec {
  static constexpr bool F16 = (PREC == RNERF_PREC_F16X3 || PREC == RNERF_PREC_F16);
  static constexpr int NP = (PREC == RNERF_PREC_F16X3 || PREC == RNERF_PREC_BF16X3) ? 2 : 1;
  // power-of-two weight scale: keeps the lo part of an f16 split out of the f16 subnormal range
  static constexpr float WSCALE = F16 ? 256.f : 1.f;
  static constexpr size_t STREAM_BYTES = (size_t)kTotalBlocks * NP * 1024;
  static constexpr int SLAB8 = 2 * 8 * NP * 1024;   // one slab = 2 k-steps of an N=256 layer
  static constexpr int SLAB4 = 2 * 4 * NP * 1024;   // ... of the N=128 view layer
  static constexpr size_t PACKED_BYTES = STREAM_BYTES + (size_t)AUX_FLOATS * 4;
};

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

// input feature (row of the Dense kernel) for MFMA layer l, k-step s, half h, slot j; -1 = zero padding
__host__ __device__ constexpr int in_feature(int l, int s, int h, int j) {
  const int kind = mfma_layer(l).kind;
  if (kind == 0) return pe_feature(8 * s + j, h);
  if (s < 16) return prev_feature(s, h, j);
  if (kind == 2) { const int f = pe_feature(8 * (s - 16) + j, h); return f < 0 ? -1 : 256 + f; }
  const int f = view_feature(8 * (s - 16) + j, h);
  return f < 0 ? -1 : 256 + f;
}

// ---- pack kernel: flat fp32 params -> MFMA A-operand stream + aux ---------------------------------------------------------
template <int PREC>
__global__ void nerfmlp_pack_kernel(const float* __restrict__ params, char* __restrict__ packed) {
  using PP = Prec<PREC>;
  constexpr bool F16 = PP::F16;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (block, lane): 8 slots
  if (gid < kTotalBlocks * 64) {
    const int blk = gid >> 6, lane = gid & 63;
    int l = 0;
    while (l < 9 && blk >= layer_blocks_before(l + 1)) ++l;
    const int rel = blk - layer_blocks_before(l);
    const int nt = mfma_layer(l).nt, s = rel / nt, t = rel % nt;
    const int d = mfma_layer(l).dense, out_dim = nerf_dense(d).out;
    const int n_out = 32 * t + (lane & 31), h = lane >> 5;
    float w[8];
    for (int j = 0; j < 8; ++j) {
      const int f = in_feature(l, s, h, j);
      w[j] = f < 0 ? 0.f : params[nerf_koff(d) + f * out_dim + n_out] * PP::WSCALE;
    }
    uint32_t hi[4], lo[4];
    for (int p = 0; p < 4; ++p) split2<F16>(w[2 * p], w[2 * p + 1], hi[p], lo[p]);
    uint4* dst = (uint4*)(packed + (size_t)blk * PP::NP * 1024);
    dst[lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    if (PP::NP == 2) dst[64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
  float* aux = (float*)(packed + PP::STREAM_BYTES);
  if (gid < AUX_FLOATS) {
    float v = 0.f;
    if (gid < AUX_WSIG) {
      const int l = gid >> 8, n = gid & 255, d = mfma_layer(l).dense;
      v = n < nerf_dense(d).out ? params[nerf_boff(d) + n] * PP::WSCALE : 0.f;
    } else if (gid < AUX_BSIG) v = params[nerf_koff(8) + (gid - AUX_WSIG)];
    else if (gid == AUX_BSIG) v = params[nerf_boff(8)];
    else if (gid >= AUX_WRGB && gid < AUX_BRGB) { const int c = (gid - AUX_WRGB) / 128, n = (gid - AUX_WRGB) % 128; v = params[nerf_koff(11) + n * 3 + c]; }
    else if (gid >= AUX_BRGB && gid < AUX_BRGB + 3) v = params[nerf_boff(11) + (gid - AUX_BRGB)];
    aux[gid] = v;
  }
}

// ---- forward kernel -----------------------------------------------------------------------------------------------------
template <int BYTES>
__device__ __forceinline__ void issue_slab(const char* __restrict__ gsrc, char* lds_dst, int wave, int lane) {
  constexpr int PER_WAVE = BYTES / 4;
  constexpr int N = PER_WAVE / 1024;
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const char* g = gsrc + wave * PER_WAVE + c * 1024 + lane * 16;
    char* l = lds_dst + wave * PER_WAVE + c * 1024;   // wave-uniform; hardware adds lane*16
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)l, 16, 0, 0);
  }
}

// 2 k-steps x NT n-tiles of MFMAs out of one LDS slab; (bh0,bl0) / (bh1,bl1) are this wave's B operands of the 2 k-steps.
template <int PREC, int NT>
__device__ __forceinline__ void slab_compute(f32x16 (&acc)[8], const uint4 bh0, const uint4 bl0, const uint4 bh1, const uint4 bl1,
                                             const char* slab, int lane) {
  using PP = Prec<PREC>;
  const uint4* a = (const uint4*)slab + lane;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const uint4 bh = kk ? bh1 : bh0, bl = kk ? bl1 : bl0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const uint4 ah = a[((kk * NT + t) * PP::NP + 0) * 64];
      acc[t] = mfma16<PP::F16>(ah, bh, acc[t]);
      if constexpr (PP::NP == 2) {
        const uint4 al = a[((kk * NT + t) * PP::NP + 1) * 64];
        acc[t] = mfma16<PP::F16>(ah, bl, acc[t]);
        acc[t] = mfma16<PP::F16>(al, bh, acc[t]);
      }
    }
  }
}

template <int NT>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[8], const float* __restrict__ bias, int h) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = *(const float4*)(bias + 32 * t + 8 * g + 4 * h);
      acc[t][4 * g + 0] = b.x; acc[t][4 * g + 1] = b.y; acc[t][4 * g + 2] = b.z; acc[t][4 * g + 3] = b.w;
    }
}

// accumulators (after scale, optional ReLU) -> next layer's B operands (hi / lo)
template <int PREC, int NT>
__device__ __forceinline__ void acc_to_operands(f32x16 (&acc)[8], float inv_scale, float floor_v, uint4 (&xh)[16], uint4 (&xl)[16]) {
  using PP = Prec<PREC>;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = fmaxf(acc[t][r] * inv_scale, floor_v);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float a = acc[t][8 * half + 2 * p], b = acc[t][8 * half + 2 * p + 1];
        if constexpr (PP::NP == 2) split2<PP::F16>(a, b, hi[p], lo[p]);
        else { hi[p] = pack2<PP::F16>(a, b); lo[p] = 0; }
      }
      xh[2 * t + half] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      xl[2 * t + half] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
  }
}

// positional encoding of one 3-vector into NK k-steps of B operands (slot maps pe_feature / view_feature)
template <int PREC, int NK, int NSIN>
__device__ __forceinline__ void encode(const float v0, const float v1, const float v2, int h, uint4* eh, uint4* el) {
  using PP = Prec<PREC>;
  const float phase = h ? 1.5707963705062866f : 0.0f;   // f32(0.5*pi) (rnerf/model_utils.py:213)
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = 8 * s + j;
      if (q < NSIN) {
        const int d = q / 3, c = q % 3;
        const float x = c == 0 ? v0 : (c == 1 ? v1 : v2);
        f[j] = sinf(fadd(fmul(x, (float)(1 << d)), phase));
      } else if (q == NSIN) f[j] = h ? v2 : v0;
      else if (q == NSIN + 1) f[j] = h ? 0.f : v1;
      else f[j] = 0.f;
    }
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if constexpr (PP::NP == 2) split2<PP::F16>(f[2 * p], f[2 * p + 1], hi[p], lo[p]);
      else { hi[p] = pack2<PP::F16>(f[2 * p], f[2 * p + 1]); lo[p] = 0; }
    }
    eh[s] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    el[s] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

template <int PREC>
__global__ void __launch_bounds__(256, 1)
nerfmlp_fwd_kernel(const char* __restrict__ packed, const float4* __restrict__ rows_pd, const float4* __restrict__ rows_dr,
                   const int* __restrict__ node_of_sample, int B, long long total_rows, int n_tiles, float4* __restrict__ out_raw,
                   int dbg) {
  // dbg (profiling ablations only, RNERF_MLP_DEBUG env var; results are garbage when set):
  //   bit0 = skip the weight-stream loads, bit1 = skip ds_read + MFMA, bit2 = skip the barrier
  using PP = Prec<PREC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  const float* __restrict__ aux = (const float*)(packed + PP::STREAM_BYTES);
  constexpr float INV_SCALE = 1.0f / PP::WSCALE;
  constexpr int BUF = PP::SLAB8;
  int buf = 0;
  size_t off = 0;   // stream offset of the next slab to prefetch

  if ((int)blockIdx.x < n_tiles) { issue_slab<PP::SLAB8>(packed, smem, wave, lane); off = PP::SLAB8; }
  __syncthreads();

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const bool has_next_tile = tile + (int)gridDim.x < n_tiles;
    long long row = (long long)tile * 128 + wave * 32 + m;
    const bool row_ok = row < total_rows;
    if (!row_ok) row = total_rows - 1;
    size_t rec = (size_t)row;
    if (node_of_sample) { const long long s = row / B; rec = (size_t)node_of_sample[s] * B + (size_t)(row - s * B); }
    const float4 pd = rows_pd[rec];
    const float4 dr = rows_dr[rec];

    uint4 peh[4], pel[4], vwh[2], vwl[2];
    encode<PREC, 4, 30>(pd.x, pd.y, pd.z, h, peh, pel);   // pos_enc(pos, 0, 10)  (rnerf/models.py:257)
    encode<PREC, 2, 12>(dr.x, dr.y, dr.z, h, vwh, vwl);   // pos_enc(dir, 0, 4)   (rnerf/models.py:289-294)

    f32x16 acc[8];
    uint4 xh[16], xl[16];

#define RUN_SLAB(NT, B0H, B0L, B1H, B1L, NEXT_BYTES, DO_NEXT)                               \
  do {                                                                                      \
    if (DO_NEXT) { if (!(dbg & 1)) issue_slab<NEXT_BYTES>(packed + off, smem + (buf ^ 1) * BUF, wave, lane); off += NEXT_BYTES; } \
    if (!(dbg & 2)) slab_compute<PREC, NT>(acc, B0H, B0L, B1H, B1L, smem + buf * BUF, lane); \
    if (!(dbg & 4)) __syncthreads();                                                        \
    buf ^= 1;                                                                               \
  } while (0)

    // ---- layer 0: 63 -> 256 (Dense_0), ReLU
    init_bias<8>(acc, aux + AUX_BIAS, h);
    RUN_SLAB(8, peh[0], pel[0], peh[1], pel[1], PP::SLAB8, true);
    RUN_SLAB(8, peh[2], pel[2], peh[3], pel[3], PP::SLAB8, true);
    acc_to_operands<PREC, 8>(acc, INV_SCALE, 0.f, xh, xl);

    // ---- layers 1..8: Dense_1..Dense_7 (ReLU; Dense_5 takes the skip concat), Dense_9 = bottleneck (no activation)
    float sigma_raw = 0.f;
#pragma unroll 1
    for (int l = 1; l <= 8; ++l) {
      init_bias<8>(acc, aux + AUX_BIAS + 256 * l, h);
      RUN_SLAB(8, xh[0], xl[0], xh[1], xl[1], PP::SLAB8, true);
      RUN_SLAB(8, xh[2], xl[2], xh[3], xl[3], PP::SLAB8, true);
      RUN_SLAB(8, xh[4], xl[4], xh[5], xl[5], PP::SLAB8, true);
      RUN_SLAB(8, xh[6], xl[6], xh[7], xl[7], PP::SLAB8, true);
      RUN_SLAB(8, xh[8], xl[8], xh[9], xl[9], PP::SLAB8, true);
      RUN_SLAB(8, xh[10], xl[10], xh[11], xl[11], PP::SLAB8, true);
      RUN_SLAB(8, xh[12], xl[12], xh[13], xl[13], PP::SLAB8, true);
      if (l == 8) RUN_SLAB(8, xh[14], xl[14], xh[15], xl[15], PP::SLAB4, true);
      else RUN_SLAB(8, xh[14], xl[14], xh[15], xl[15], PP::SLAB8, true);
      if (l == 5) {   // skip concat: [x, inputs] (rnerf/model_utils.py:68-69)
        RUN_SLAB(8, peh[0], pel[0], peh[1], pel[1], PP::SLAB8, true);
        RUN_SLAB(8, peh[2], pel[2], peh[3], pel[3], PP::SLAB8, true);
      }
      acc_to_operands<PREC, 8>(acc, INV_SCALE, l == 8 ? -__builtin_inff() : 0.f, xh, xl);
      if (l == 7) {   // sigma head on the fp32 trunk output (Dense_8, rnerf/model_utils.py:70)
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 w = *(const float4*)(aux + AUX_WSIG + 32 * t + 8 * g + 4 * h);
            part = fmaf(acc[t][4 * g + 0], w.x, part); part = fmaf(acc[t][4 * g + 1], w.y, part);
            part = fmaf(acc[t][4 * g + 2], w.z, part); part = fmaf(acc[t][4 * g + 3], w.w, part);
          }
        sigma_raw = part + __shfl_xor(part, 32) + aux[AUX_BSIG];
      }
    }

    // ---- view layer: [bottleneck(256), view_enc(27)] -> 128 (Dense_10), ReLU
    init_bias<4>(acc, aux + AUX_BIAS + 256 * 9, h);
    RUN_SLAB(4, xh[0], xl[0], xh[1], xl[1], PP::SLAB4, true);
    RUN_SLAB(4, xh[2], xl[2], xh[3], xl[3], PP::SLAB4, true);
    RUN_SLAB(4, xh[4], xl[4], xh[5], xl[5], PP::SLAB4, true);
    RUN_SLAB(4, xh[6], xl[6], xh[7], xl[7], PP::SLAB4, true);
    RUN_SLAB(4, xh[8], xl[8], xh[9], xl[9], PP::SLAB4, true);
    RUN_SLAB(4, xh[10], xl[10], xh[11], xl[11], PP::SLAB4, true);
    RUN_SLAB(4, xh[12], xl[12], xh[13], xl[13], PP::SLAB4, true);
    RUN_SLAB(4, xh[14], xl[14], xh[15], xl[15], PP::SLAB4, true);
    // last slab of the tile: prefetch the first slab of the next tile (stream restarts at 0)
    if (has_next_tile) { off = 0; }
    RUN_SLAB(4, vwh[0], vwl[0], vwh[1], vwl[1], PP::SLAB8, has_next_tile);

    // ---- rgb head (Dense_11) on the fp32 view-layer output
    float pr = 0.f, pg = 0.f, pb = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaxf(acc[t][r] * INV_SCALE, 0.f);
        const int n = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
        pr = fmaf(v, aux[AUX_WRGB + n], pr);
        pg = fmaf(v, aux[AUX_WRGB + 128 + n], pg);
        pb = fmaf(v, aux[AUX_WRGB + 256 + n], pb);
      }
    pr = pr + __shfl_xor(pr, 32) + aux[AUX_BRGB];
    pg = pg + __shfl_xor(pg, 32) + aux[AUX_BRGB + 1];
    pb = pb + __shfl_xor(pb, 32) + aux[AUX_BRGB + 2];
    if (row_ok && h == 0) out_raw[row] = make_float4(pr, pg, pb, sigma_raw);
#undef RUN_SLAB
  }
}

// ------------------------------------------------------------------------------------------------------------------
// N2: the 4x128 background MLP (rnerf/models.py:116-118) in exact fp32 on v_mfma_f32_32x32x2_f32.
// One wave = 32 rays; weights are read straight from the flat fp32 buffer (kernel[in][out], coalesced along out).
// ------------------------------------------------------------------------------------------------------------------
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

__device__ __forceinline__ void small_init_bias(f32x16 (&acc)[4], const float* __restrict__ bias, int h) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = bias[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
}

// acc += W[rows f(h)][:] x operand, for the 64 (t,r) steps of a 128-wide previous activation held in `x`
__device__ __forceinline__ void small_prev_layer(f32x16 (&acc)[4], const f32x16 (&x)[4], const float* __restrict__ kern, int m, int h) {
#pragma unroll
  for (int ts = 0; ts < 4; ++ts)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * ts + (r & 3) + 8 * (r >> 2) + 4 * h;   // the feature this half holds in x[ts][r]
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kern[f * 128 + 32 * t + m], x[ts][r], acc[t], 0, 0, 0);
    }
}

__device__ __forceinline__ void small_dir_layer(f32x16 (&acc)[4], const float (&enc)[14], const float* __restrict__ kern, int m, int h) {
#pragma unroll
  for (int q = 0; q < 14; ++q) {
    const int f0 = dir_feature(q, 0), f1 = dir_feature(q, 1);
    const int f = h ? f1 : f0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float w = f < 0 ? 0.f : kern[f * 128 + 32 * t + m];
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, enc[q], acc[t], 0, 0, 0);
    }
  }
}

__global__ void __launch_bounds__(64) bkgd_fwd_kernel(const float* __restrict__ params, const float* __restrict__ dirs, int dir_stride,
                                                      long long n, float pad_scale, float pad, float* __restrict__ out_rgb) {
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  long long row = (long long)blockIdx.x * 32 + m;
  const bool ok = row < n;
  if (!ok) row = n - 1;
  const float v0 = dirs[row * dir_stride], v1 = dirs[row * dir_stride + 1], v2 = dirs[row * dir_stride + 2];
  // pos_enc(dir, 0, 4) (rnerf/model_utils.py:187-214) in the K=2 slot order of dir_feature
  float enc[14];
  const float phase = h ? 1.5707963705062866f : 0.0f;
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const int d = q / 3, c = q % 3;
    const float x = c == 0 ? v0 : (c == 1 ? v1 : v2);
    enc[q] = sinf(fadd(fmul(x, (float)(1 << d)), phase));
  }
  enc[12] = h ? v2 : v0;
  enc[13] = h ? 0.f : v1;

  f32x16 acc[4], x[4];
  // Dense_0: 27 -> 128, ReLU
  small_init_bias(acc, params + bkgd_boff(0), h);
  small_dir_layer(acc, enc, params + bkgd_koff(0), m, h);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[t][r] = fmaxf(acc[t][r], 0.f);
  // Dense_1, Dense_2: 128 -> 128, ReLU
#pragma unroll 1
  for (int l = 1; l <= 2; ++l) {
    small_init_bias(acc, params + (l == 1 ? bkgd_boff(1) : bkgd_boff(2)), h);
    small_prev_layer(acc, x, params + (l == 1 ? bkgd_koff(1) : bkgd_koff(2)), m, h);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[t][r] = fmaxf(acc[t][r], 0.f);
  }
  // Dense_3: [x(128), inputs(27)] -> 128, ReLU  (skip concat after i == 2, rnerf/model_utils.py:131-132)
  small_init_bias(acc, params + bkgd_boff(3), h);
  small_prev_layer(acc, x, params + bkgd_koff(3), m, h);
  small_dir_layer(acc, enc, params + bkgd_koff(3) + 128 * 128, m, h);
  // Dense_4: 128 -> 3 on the VALU, then sigmoid*(1+2p)-p (rnerf/models.py:336-337)
  float o[3] = {0.f, 0.f, 0.f};
  const float* __restrict__ k4 = params + bkgd_koff(4);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = fmaxf(acc[t][r], 0.f);
      const int f = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
      o[0] = fmaf(v, k4[f * 3 + 0], o[0]); o[1] = fmaf(v, k4[f * 3 + 1], o[1]); o[2] = fmaf(v, k4[f * 3 + 2], o[2]);
    }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o[c] = o[c] + __shfl_xor(o[c], 32) + params[bkgd_boff(4) + c];
    o[c] = fsub(fmul(fdiv(1.0f, fadd(1.0f, expf(-o[c]))), pad_scale), pad);
  }
  if (ok && h == 0) { out_rgb[3 * row] = o[0]; out_rgb[3 * row + 1] = o[1]; out_rgb[3 * row + 2] = o[2]; }
}

}  // namespace rnerf

using namespace rnerf;

static bool prec_ok(int p) { return p == RNERF_PREC_F16X3 || p == RNERF_PREC_BF16X3 || p == RNERF_PREC_F16 || p == RNERF_PREC_BF16; }

extern "C" size_t rnerf_nerfmlp_packed_bytes(int precision) {
  switch (precision) {
    case RNERF_PREC_F16X3: return Prec<RNERF_PREC_F16X3>::PACKED_BYTES;
    case RNERF_PREC_BF16X3: return Prec<RNERF_PREC_BF16X3>::PACKED_BYTES;
    case RNERF_PREC_F16: return Prec<RNERF_PREC_F16>::PACKED_BYTES;
    case RNERF_PREC_BF16: return Prec<RNERF_PREC_BF16>::PACKED_BYTES;
    default: set_error("rnerf_nerfmlp_packed_bytes: unsupported precision %d", precision); return 0;
  }
}

extern "C" int rnerf_nerfmlp_pack(const float* params, int precision, void* packed, void* stream) {
  RNERF_CHECK_ARG(params && packed, "rnerf_nerfmlp_pack: null pointer");
  RNERF_CHECK_ARG(prec_ok(precision), "rnerf_nerfmlp_pack: unsupported precision %d", precision);
  RNERF_CHECK_ARG(((uintptr_t)packed & 15) == 0, "rnerf_nerfmlp_pack: packed must be 16-byte aligned");
  const int threads = kTotalBlocks * 64, block = 256, grid = (threads + block - 1) / block;
  hipStream_t st = (hipStream_t)stream;
  switch (precision) {
    case RNERF_PREC_F16X3: hipLaunchKernelGGL(nerfmlp_pack_kernel<RNERF_PREC_F16X3>, dim3(grid), dim3(block), 0, st, params, (char*)packed); break;
    case RNERF_PREC_BF16X3: hipLaunchKernelGGL(nerfmlp_pack_kernel<RNERF_PREC_BF16X3>, dim3(grid), dim3(block), 0, st, params, (char*)packed); break;
    case RNERF_PREC_F16: hipLaunchKernelGGL(nerfmlp_pack_kernel<RNERF_PREC_F16>, dim3(grid), dim3(block), 0, st, params, (char*)packed); break;
    default: hipLaunchKernelGGL(nerfmlp_pack_kernel<RNERF_PREC_BF16>, dim3(grid), dim3(block), 0, st, params, (char*)packed); break;
  }
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

static int mlp_debug_flags() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("RNERF_MLP_DEBUG"); v = e ? atoi(e) : 0; }
  return v;
}

template <int PREC>
static int launch_fwd(const void* packed, const float* rows_pd, const float* rows_dr, const int32_t* node_of_sample, int32_t B,
                      long long total_rows, float* out_raw, hipStream_t st) {
  using PP = Prec<PREC>;
  const int n_tiles = (int)((total_rows + 127) / 128);
  int dev = 0, cus = 0;
  RNERF_CHECK_HIP(hipGetDevice(&dev));
  RNERF_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int grid = n_tiles < cus ? n_tiles : cus;
  const size_t lds = 2 * (size_t)PP::SLAB8;
  static bool attr_set = false;
  if (!attr_set) {
    RNERF_CHECK_HIP(hipFuncSetAttribute((const void*)nerfmlp_fwd_kernel<PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(nerfmlp_fwd_kernel<PREC>, dim3(grid), dim3(256), lds, st, (const char*)packed, (const float4*)rows_pd,
                     (const float4*)rows_dr, node_of_sample, B, total_rows, n_tiles, (float4*)out_raw, mlp_debug_flags());
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

extern "C" int rnerf_nerfmlp_forward(const void* packed, int precision, const float* rows_pd, const float* rows_dr,
                                     const int32_t* node_of_sample, int32_t S, int32_t B, float* out_raw, void* stream) {
  RNERF_CHECK_ARG(packed && rows_pd && rows_dr && out_raw, "rnerf_nerfmlp_forward: null pointer");
  RNERF_CHECK_ARG(prec_ok(precision), "rnerf_nerfmlp_forward: unsupported precision %d", precision);
  RNERF_CHECK_ARG(S >= 1 && B >= 1, "rnerf_nerfmlp_forward: need S >= 1 and B >= 1");
  RNERF_CHECK_ARG((((uintptr_t)packed | (uintptr_t)rows_pd | (uintptr_t)rows_dr | (uintptr_t)out_raw) & 15) == 0,
                  "rnerf_nerfmlp_forward: buffers must be 16-byte aligned");
  const long long total = (long long)S * B;
  hipStream_t st = (hipStream_t)stream;
  switch (precision) {
    case RNERF_PREC_F16X3: return launch_fwd<RNERF_PREC_F16X3>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st);
    case RNERF_PREC_BF16X3: return launch_fwd<RNERF_PREC_BF16X3>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st);
    case RNERF_PREC_F16: return launch_fwd<RNERF_PREC_F16>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st);
    default: return launch_fwd<RNERF_PREC_BF16>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st);
  }
}

extern "C" int rnerf_bkgd_forward(const float* params, const float* dirs, int32_t dir_stride, int64_t n, double rgb_padding,
                                  float* out_rgb, void* stream) {
  RNERF_CHECK_ARG(params && dirs && out_rgb, "rnerf_bkgd_forward: null pointer");
  RNERF_CHECK_ARG(dir_stride >= 3, "rnerf_bkgd_forward: dir_stride must be >= 3");
  RNERF_CHECK_ARG(n >= 1, "rnerf_bkgd_forward: n must be >= 1");
  hipLaunchKernelGGL(bkgd_fwd_kernel, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, (hipStream_t)stream, params, dirs, dir_stride,
                     (long long)n, (float)(1 + 2 * rgb_padding), (float)rgb_padding, out_rgb);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}
