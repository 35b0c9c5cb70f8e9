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
//   * A workgroup = 4 waves = 256 consecutive sample rows (64 per wave, two 32-row m-tiles); the weight stream (2.3 MB
//     for X3) is read from L2 through a double-buffered LDS ring with global_load_lds (16 B/lane), one slab = one k-step
//     of one layer, shared by the 4 waves: 1 KB of L2->LDS traffic per row and layer.
//   * Persistent grid: each workgroup walks row tiles with stride gridDim.x and prefetches across layer and tile seams.
#include "common.h"
#include "nerfmlp_layout.h"
#include "mfma_ops.h"
#include "bkgd_layout.h"
#include "so3_layout.h"

#include <stdlib.h>
#include <type_traits>
#include <vector>
#include <stdio.h>

namespace rnerf {


#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

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

// aux section (floats) that follows the weight stream: fp32 biases in natural feature order, sigma/rgb heads.
constexpr int AUX_BIAS = 0;            // 10 layers x 256
constexpr int AUX_WSIG = 2560;         // Dense_8 kernel [256]
constexpr int AUX_BSIG = 2816;         // Dense_8 bias (+3 pad)
constexpr int AUX_WRGB = 2820;         // Dense_11 kernel transposed [3][128]
constexpr int AUX_BRGB = 3204;         // Dense_11 bias (+1 pad)
constexpr int AUX_ZERO = 3208;         // 256 zeros: the sigma weights of every layer but the trunk output (PrevConv SIG)
constexpr int AUX_FLAG = 3464;         // f16 modes: non-zero = a weight left the f16 range of the scaled stream (|W| >= 256 for the 2^8-scaled streams: the
                                       // forward returns NaN, never a plausible wrong colour; |W| >= 3.99 for f16f8's 2^14: the launch falls back to f16x3)
constexpr int AUX_FLOATS = 3468;

template <int PREC>
struct Prec {
  static constexpr bool F16 = (PREC == RNERF_PREC_F16X3 || PREC == RNERF_PREC_F16 || PREC == RNERF_PREC_F16X2 || PREC == RNERF_PREC_F16F8);
  // F8X: f16 main term + the two cross terms of the hi/lo split on v_mfma_f32_32x32x16_fp8_fp8 (e4m3).  The cross terms carry 2^-11 of the
  // product and need 4 bits, not 11: same instruction count and operand bytes as f16x3, but the fp8 MFMA draws less power, and with every CU
  // busy the engine is power-limited (tools/ubench/mfma_fp8.hip: 16.8 instead of 19.4 ns per MFMA for a 1 f16 + 2 fp8 mix).  Second block of a
  // weight tile = [fp8(W_lo) x 8 | fp8(W 2^-10) x 8] per lane, second operand part = [fp8(x) x 8 | fp8(x_lo 2^10) x 8]: with the weights
  // scaled by 2^14 both factors of both terms sit in the middle of e4m3's range without any block scale (W_lo ~ 4 W, x_lo 2^10 ~ x / 4).
  static constexpr bool F8X = PREC == RNERF_PREC_F16F8;
  static constexpr int NP = (PREC == RNERF_PREC_F16X3 || PREC == RNERF_PREC_BF16X3 || PREC == RNERF_PREC_F16X2 || F8X) ? 2 : 1;   // parts of a WEIGHT
  // MFMA passes per tile of a two-part stream: 3 = W_hi x_hi + W_lo x_hi + W_hi x_lo (fp32-grade);
  // 22 = (W_hi + W_lo) x_hi (f16x2: exact weights, activations rounded to f16 — 2/3 of the matrix work)
  // 38 = W_hi x_hi (f16) + W_lo x (fp8) + W x_lo (fp8)
  static constexpr int PASSES = PREC == RNERF_PREC_F16X2 ? 22 : (F8X ? 38 : 3);
  // power-of-two weight scale: keeps the lo part of an f16 split out of the f16 subnormal range (f16f8: |W| must stay below 3.99)
  static constexpr float WSCALE = F8X ? 16384.f : (F16 ? 256.f : 1.f);
  static constexpr size_t STREAM_BYTES = (size_t)kTotalBlocks * NP * 1024;
  static constexpr int SLAB = 8 * NP * 1024;        // one slab = 1 k-step of an N=256 layer = 2 k-steps of the N=128 view layer
  static constexpr size_t PACKED_BYTES = STREAM_BYTES + (size_t)AUX_FLOATS * 4;
};

// ---- fp8 (e4m3) cross-term operands of the f16f8 precision ------------------------------------------------------------------
constexpr float F8X_LO_SCALE = 1024.f;       // x_lo is multiplied, the W image of the second cross term divided by it
// two values -> two fp8 bytes in half `HI` of `old` (the other half is kept).  Written as the instruction: the LOW half is a plain output
// (the builtin takes the old value as an input, and an "old" that does not exist yet is an undefined register the allocator spills and
// reloads); SCALED = v_cvt_scalef32_pk_fp8_f32, which divides by its scale operand (the power of two 1 / F8X_LO_SCALE).  Overflow: with
// MODE.FP16_OVFL set (f8x_mode()) the conversion clamps to +-448, without it a value above ~464 becomes NaN (probed: fp8_cvt_probe.hip).
// ORDER CONTRACT: the LOW half of a word must be converted BEFORE its HIGH half (the LOW form starts a fresh register and would drop a high
// half written earlier).  Every caller walks the value pairs p = 0, 1, 2, 3 upwards (f8x_lo_part, PrevConv / EncWork chunks, the pack
// kernel); f8_word() below builds a whole word in the right order for new callers.
template <bool HI, bool SCALED = false>
__device__ __forceinline__ uint32_t f8_pair(uint32_t old, float a, float b) {
  uint32_t r = old;
  const float inv = 1.0f / F8X_LO_SCALE;
  if constexpr (!SCALED) {
    if constexpr (HI) asm("v_cvt_pk_fp8_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(r) : "v"(a), "v"(b));
    else asm("v_cvt_pk_fp8_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  } else {
    if constexpr (HI) asm("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(r) : "v"(a), "v"(b), "v"(inv));
    else asm("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(inv));
  }
  return r;
}
// one whole fp8 word from four values, low half first (the order contract of f8_pair in one place)
template <bool SCALED = false>
__device__ __forceinline__ uint32_t f8_word(float a, float b, float c, float d) {
  return f8_pair<true, SCALED>(f8_pair<false, SCALED>(0u, a, b), c, d);
}
// fp8 conversions clamp instead of producing NaN: MODE.FP16_OVFL (bit 23); set once at the head of an f16f8 kernel
__device__ __forceinline__ void f8x_mode() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
__device__ __forceinline__ f32x16 mfma_f8(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, const f32x16 c) {
  const long A = (long)(((unsigned long long)a1 << 32) | a0), B = (long)(((unsigned long long)b1 << 32) | b0);
  return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(A, B, c, 0, 0, 0);
}
// lo part of an operand in the f16f8 layout from 8 values and their f16 hi parts: [fp8(x) x 8 | fp8((x - hi) 2^10) x 8]
__device__ __forceinline__ uint4 f8x_lo_part(const float (&x)[8], const uint32_t (&h)[4]) {
  uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float ha, hb;
    unpack2<true>(h[p], ha, hb);
    const float r0 = x[2 * p] - ha, r1 = x[2 * p + 1] - hb;
    if (p & 1) { o[p >> 1] = f8_pair<true>(o[p >> 1], x[2 * p], x[2 * p + 1]); o[2 + (p >> 1)] = f8_pair<true, true>(o[2 + (p >> 1)], r0, r1); }
    else { o[p >> 1] = f8_pair<false>(o[p >> 1], x[2 * p], x[2 * p + 1]); o[2 + (p >> 1)] = f8_pair<false, true>(o[2 + (p >> 1)], r0, r1); }
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// ---- 8-bit lo planes of the training tensors (backward mode RNERF_BWD_F16X3_LO8) ------------------------------------------------------
// The lo part of an operand (x - f16(x): at most half an ulp of the hi part) is STORED as e4m3 of lo / scale and decoded back to f16 by the
// wgrad (v_cvt_scalef32_pk_f16_fp8 multiplies by the same scale operand; probed: tools/ubench/tr8_cvt_probe.hip): hi 2 B + lo 1 B per
// value.  Fixed power-of-two scales: saved activations 2^-13 (full 3-bit lo significand for 0.004 <= |x| < 128, absolute resolution
// 2^-22 below, the lo part clamped — f16-hi precision — above), gradients 2^-14 (rows are normalised to 32 <= max |d raw| < 64: full
// precision for 2^-7 <= |dY| < 64).  Overflow: v_cvt_scalef32_pk_fp8_f16 returns NaN above 448 unless MODE.FP16_OVFL is set — the training
// forward sets it (its range watch then takes f16's clamp value as "left the range", like f16f8), the dgrad clamps the f16 lo parts first
// (its f16 overflow must stay inf: that is how a gradient chain that left its headroom shows).
constexpr float LO8_SCALE_X = 1.0f / 8192.f, LO8_SCALE_D = 1.0f / 16384.f;
template <bool HI>
__device__ __forceinline__ uint32_t lo8_pair(uint32_t old, uint32_t f16x2, float scale) {      // ORDER CONTRACT of f8_pair: low half first
  uint32_t r = old;
  if constexpr (HI) asm("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(r) : "v"(f16x2), "v"(scale));
  else asm("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2" : "=v"(r) : "v"(f16x2), "v"(scale));
  return r;
}
template <bool CLAMP>
__device__ __forceinline__ uint2 lo8_pack(const uint4& lo, float scale) {
  uint32_t w[4] = {lo.x, lo.y, lo.z, lo.w};
  if constexpr (CLAMP) {      // |lo| <= 448 scale as packed f16 (448 * 2^-14 = 0.02734375 = 0x2700)
    static_assert(LO8_SCALE_D == 1.0f / 16384.f, "clamp constant");
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      asm("v_pk_min_f16 %0, %0, %1" : "+v"(w[p]) : "v"(0x27002700u));
      asm("v_pk_max_f16 %0, %0, %1" : "+v"(w[p]) : "v"(0xA700A700u));
    }
  }
  uint2 o;
  o.x = lo8_pair<true>(lo8_pair<false>(0u, w[0], scale), w[1], scale);
  o.y = lo8_pair<true>(lo8_pair<false>(0u, w[2], scale), w[3], scale);
  return o;
}
typedef int int2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half8 lo8_decode(const int2v v, float scale) {      // 8 e4m3 bytes -> 8 f16 (x scale)
  const half2v a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((unsigned)v.x, scale, false), b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((unsigned)v.x, scale, true);
  const half2v c = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((unsigned)v.y, scale, false), d = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((unsigned)v.y, scale, true);
  return half8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}


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
      if constexpr (PP::F16) { if (!(fabsf(w[j]) <= 65504.f)) ((float*)(packed + PP::STREAM_BYTES))[AUX_FLAG] = 1.0f; }   // zeroed by the launcher
    }
    uint32_t hi[4], lo[4];
    for (int p = 0; p < 4; ++p) split2<F16>(w[2 * p], w[2 * p + 1], hi[p], lo[p]);
    if constexpr (PP::F8X) {      // [fp8(W_lo) x 8 | fp8(W / 2^10) x 8]
      f8x_mode();
      float wl[8], ws[8];
      for (int p = 0; p < 4; ++p) {
        float ha, hb;
        unpack2<true>(hi[p], ha, hb);
        wl[2 * p] = w[2 * p] - ha; wl[2 * p + 1] = w[2 * p + 1] - hb;
        ws[2 * p] = w[2 * p] * (1.0f / F8X_LO_SCALE); ws[2 * p + 1] = w[2 * p + 1] * (1.0f / F8X_LO_SCALE);
      }
      lo[0] = f8_pair<true>(f8_pair<false>(0u, wl[0], wl[1]), wl[2], wl[3]); lo[1] = f8_pair<true>(f8_pair<false>(0u, wl[4], wl[5]), wl[6], wl[7]);
      lo[2] = f8_pair<true>(f8_pair<false>(0u, ws[0], ws[1]), ws[2], ws[3]); lo[3] = f8_pair<true>(f8_pair<false>(0u, ws[4], ws[5]), ws[6], ws[7]);
    }
    uint4* dst = (uint4*)(packed + (size_t)blk * PP::NP * 1024);
    dst[lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    if constexpr (PP::F8X) { uint2* d2 = (uint2*)(dst + 64); d2[lane] = make_uint2(lo[0], lo[1]); d2[64 + lane] = make_uint2(lo[2], lo[3]); }
    else if (PP::NP == 2) dst[64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
  float* aux = (float*)(packed + PP::STREAM_BYTES);
  if (gid < AUX_FLAG) {
    float v = 0.f;
    if (gid < AUX_WSIG) {
      const int l = gid >> 8, n = gid & 255, d = mfma_layer(l).dense;
      v = n < nerf_dense(d).out ? params[nerf_boff(d) + n] : 0.f;
    } else if (gid < AUX_BSIG) v = params[nerf_koff(8) + (gid - AUX_WSIG)];
    else if (gid == AUX_BSIG) v = params[nerf_boff(8)];
    else if (gid >= AUX_WRGB && gid < AUX_BRGB) { const int c = (gid - AUX_WRGB) / 128, n = (gid - AUX_WRGB) % 128; v = params[nerf_koff(11) + n * 3 + c]; }
    else if (gid >= AUX_BRGB && gid < AUX_BRGB + 3) v = params[nerf_boff(11) + (gid - AUX_BRGB)];
    aux[gid] = v;
  }
}

// ---- forward kernel -----------------------------------------------------------------------------------------------------
// Weight-stream DMA (global -> LDS, 16 B per lane, 1 KiB per wave-instruction).  Written as inline asm on purpose: for the
// __builtin_amdgcn_global_load_lds form hipcc orders every later ds_read behind the DMA (it cannot prove the reads hit the
// OTHER ring slot) and emits s_waitcnt vmcnt(0) right after the issue, which serialises the prefetch with the compute.
// The asm form is invisible to the compiler's counters; slab_wait_dma() before the slab-end barrier is the only wait.
// lds_off: wave-uniform LDS byte address (dynamic LDS starts at 0: the kernel has no static __shared__).
__device__ __forceinline__ void glds16(const char* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_off) : "memory");
}
// the same with the non-temporal policy: operand streams that are read exactly once (wgrad)
__device__ __forceinline__ void glds16_nt(const char* gsrc, unsigned lds_off) {
  unsigned keep;
#ifdef RNERF_WGTR_NO_NT
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
#else
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
#endif
               : "=&s"(keep) : "v"(gsrc), "s"(lds_off) : "memory");
}
// the same with a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset: the address arithmetic of a stream that advances
// by a uniform stride runs on the scalar unit and the lane part costs one VGPR instead of two
__device__ __forceinline__ void glds16_nt_s(const char* sbase, unsigned voff, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}
// vmcnt(0) as the BUILTIN (imm: vmcnt=0, expcnt=7, lgkmcnt=15): hipcc folds an explicit s_waitcnt into its own scoreboard,
// so it also knows that every load it tracks itself (bias prefetch) has landed and emits no stricter wait later.
__device__ __forceinline__ void slab_wait_dma() { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); }

template <int BYTES>
__device__ __forceinline__ void issue_slab(const char* __restrict__ gsrc, unsigned lds_off, int wave, int lane) {
  constexpr int PER_WAVE = BYTES / 4;
  constexpr int N = PER_WAVE / 1024;
  static_assert(N >= 1 && N * 4096 == BYTES, "slab must be a multiple of 4 KiB");
#pragma unroll
  for (int c = 0; c < N; ++c)
    glds16(gsrc + wave * PER_WAVE + c * 1024 + lane * 16,
           __builtin_amdgcn_readfirstlane(lds_off + wave * PER_WAVE + c * 1024));   // hardware adds lane*16
}

// B operands of one k-step for the wave's two 32-row m-tiles
// min(u16, 1) of both halves of a dword = the non-zero flags of two non-negative 16-bit floats.  Written as the instruction:
// hipcc expands __builtin_elementwise_min(u16x2, {1,1}) into ~20 SDWA compares / scalar mask ops per dword.
__device__ __forceinline__ uint32_t pk_maxu(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t pk_min1(uint32_t w) {
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(w), "v"(0x00010001u));
  return r;
}

// 16-byte load through an explicitly GLOBAL pointer (a pointer that went through an asm operand is generic to hipcc: flat_load + 64-bit
// address arithmetic per load)
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gload4(const float* p) {
  const f32x4v v = *(const __attribute__((address_space(1))) f32x4v*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}

// B operands of one k-step for the two 32-row m-tiles of a wave.  ONE: the wave runs a single m-tile (128-row workgroup tiles for launches
// that would leave more than half of the CUs without a 256-row tile): everything of m-tile 1 — MFMAs, conversions, LDS state, saves — is
// compiled out; h1 / l1 are never read.
template <bool ONE_> struct KOpsT { static constexpr bool ONE = ONE_; uint4 h0, l0, h1, l1; };
typedef KOpsT<false> KOps;

// ReLU as an INTEGER max on the bit pattern: max_i32(bits(x), 0) is x for x >= +0, +0 for every negative x (and -0), and — unlike
// v_max_f32, which returns the non-NaN operand — it keeps +inf and NaN.  That matters because the activations travel as f16 parts: a hidden
// activation above 65504 becomes hi = inf, lo = x - inf = -inf, the next layer's sums inf - inf = NaN, and with fmaxf(NaN, 0) = 0 every unit
// of that layer silently read 0 — finite, plausible, wrong outputs (found in round 4; tools/r04/dbg_hot.py).  With the integer max the NaN
// reaches the outputs: out of range means non-finite, never plausible.  NO_FLOOR (= INT_MIN) turns the max into the identity (bottleneck).
constexpr int RELU_FLOOR = 0, NO_FLOOR = (int)0x80000000u;
__device__ __forceinline__ float relu_keep(float x, int floor_bits) {
  const int b = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, b > floor_bits ? b : floor_bits);
}

// streaming accesses of the training tensors (written once, read once by a later kernel): keep them out of the L2 working set
#ifdef RNERF_NO_NT
__device__ __forceinline__ void stream_store(uint4* p, const uint4 v) { *p = v; }
__device__ __forceinline__ uint4 stream_load(const uint4* p) { return *p; }
#else
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store(uint4* p, const uint4 v) {
  u32x4_t t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, (u32x4_t*)p);
}
__device__ __forceinline__ uint4 stream_load(const uint4* p) {
  const u32x4_t t = __builtin_nontemporal_load((const u32x4_t*)p);
  return make_uint4(t.x, t.y, t.z, t.w);
}
#endif
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stream_store8(uint2* p, const uint2 v) {
#ifdef RNERF_NO_NT
  *p = v;
#else
  u32x2_t t = {v.x, v.y};
  __builtin_nontemporal_store(t, (u32x2_t*)p);
#endif
}

template <int PREC>
__device__ __forceinline__ void split8(const float (&x)[8], uint4& hi, uint4& lo) {
  using PP = Prec<PREC>;
  uint32_t h[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if constexpr (PP::NP == 2 && !PP::F8X) split2<PP::F16>(x[2 * p], x[2 * p + 1], h[p], l[p]);
    else { h[p] = pack2<PP::F16>(x[2 * p], x[2 * p + 1]); l[p] = 0; }
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  if constexpr (PP::F8X) lo = f8x_lo_part(x, h);
  else lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// Conversion of the previous layer's outputs into the B operands of the NEXT k-step, cut into small chunks that are issued
// between the MFMAs of the CURRENT k-step (each MFMA occupies the matrix pipe for 32 cycles; an in-order wave can issue
// ~5 independent VALU ops in that shadow).  hipcc does not build this interleave by itself (it emits the ~110 VALU ops as
// one clump ahead of the 48 MFMAs and the matrix pipe idles), so the order is written out and pinned with sched_barrier.
//   pair pi in [0,8): m-tile pi>>2, value pair pi&3 -> chunk 0: bias+scale+ReLU, chunk 1: hi + residual, chunk 2: lo
// SIG: also accumulate sg0/sg1 += x * ws[j] (the sigma head, Dense_8, rides on the conversion of the trunk output that feeds Dense_9: the
// same relu(acc / scale + b) values; for every other layer ws points at zeros — no branch in the shared loop body).
template <int PREC, int S, bool MASK = false, bool SIG = false, bool ONE = false>
struct PrevConv {
  using PP = Prec<PREC>;
  float ws[8];
  float sg0 = 0.f, sg1 = 0.f;
  uint32_t nz[2] = {0, 0};   // MASK (training forward): non-zero flags of the 8 hi values per m-tile, bits p and 16 + p for pair p
  const f32x16& p0;   // m-tile 0: the accumulator registers holding features 32*(S>>1) .. +31 of the previous layer
  float v1[8];        // m-tile 1: raw accumulator values of the 8 features of k-step S (from LDS)
  float b[8];         // fp32 bias of the producing layer
  __device__ __forceinline__ PrevConv(const f32x16& p) : p0(p) {}
  int floor_v;     // RELU_FLOOR (ReLU) or NO_FLOOR (bottleneck: no activation) — see relu_keep()
  uint32_t hi[2][4], lo[2][4];
  // the pair in flight, per m-tile (a 4-tile k-step runs one pair of each m-tile at a time: PairOfPairs).  Four scalars, not arrays:
  // the inline asm below takes them by "+v", and an array member whose element goes into an asm operand stays in memory (scratch)
  float x0m0, x1m0, x0m1, x1m1;

  // five dependent stages of one value pair, one stage per MFMA slot (each stage = 1-2 independent VALU ops)
  template <int C, int PI>
  __device__ __forceinline__ void chunk() {
    constexpr int mt = PI >> 2, p = PI & 3;
    if constexpr (ONE && mt == 1) return;
    constexpr float INV_SCALE = 1.0f / PP::WSCALE;
    float& x0 = mt == 0 ? x0m0 : x0m1;
    float& x1 = mt == 0 ? x1m0 : x1m1;
    if constexpr (C == 0) {
      const float r0 = mt == 0 ? p0[8 * (S & 1) + 2 * p] : v1[2 * p];
      const float r1 = mt == 0 ? p0[8 * (S & 1) + 2 * p + 1] : v1[2 * p + 1];
      x0 = fmaf(r0, INV_SCALE, b[2 * p]);
      x1 = fmaf(r1, INV_SCALE, b[2 * p + 1]);
    } else if constexpr (C == 1) {
      x0 = relu_keep(x0, floor_v);
      x1 = relu_keep(x1, floor_v);
      if constexpr (SIG) {
        // volatile asm: as plain fmaf() the two FMAs are sunk below the residual stage (which overwrites x0 / x1 in place), and the
        // copies of x0 / x1 that keeps alive are spilled — 16 scratch stores per k-step
        float& sg = mt == 0 ? sg0 : sg1;
        asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(sg) : "v"(x0), "v"(ws[2 * p]));
        asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(sg) : "v"(x1), "v"(ws[2 * p + 1]));
      }
    } else if constexpr (C == 2) {
      hi[mt][p] = pack2<PP::F16>(x0, x1);
      if constexpr (PP::F8X) lo[mt][p >> 1] = f8_pair<(p & 1) != 0>(lo[mt][p >> 1], x0, x1);       // fp8(x): the operand of the W_lo term
    } else if constexpr (C == 3) {
      if constexpr (PP::NP == 2) {
        if constexpr (PP::F16) {   // x - (float)hi as one v_fma_mix_f32 per value (exact: the residual is representable)
          // written as the instruction: hipcc lowers fmaf((float)half, -1, x) to v_cvt_f32_f16 + v_add_f32 (2 ops per value)
          const uint32_t hw = hi[mt][p];
          asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x0) : "v"(hw));
          asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x1) : "v"(hw));
        } else {
          float ha, hb;
          unpack2<false>(hi[mt][p], ha, hb);
          x0 -= ha; x1 -= hb;
        }
      }
    } else {
      if constexpr (PP::F8X) lo[mt][2 + (p >> 1)] = f8_pair<(p & 1) != 0, true>(lo[mt][2 + (p >> 1)], x0, x1);   // fp8(x_lo 2^10)
      else if constexpr (PP::NP == 2) lo[mt][p] = pack2<PP::F16>(x0, x1);
      else lo[mt][p] = 0;
#ifndef RNERF_FWD_NOMASK
      if constexpr (MASK) nz[mt] |= pk_min1(hi[mt][p]) << p;   // post-ReLU values are non-negative: min(u16, 1) = non-zero flag of each half
#endif
    }
  }
  __device__ __forceinline__ KOpsT<ONE> result() const {
    KOpsT<ONE> o;
    o.h0 = make_uint4(hi[0][0], hi[0][1], hi[0][2], hi[0][3]); o.l0 = make_uint4(lo[0][0], lo[0][1], lo[0][2], lo[0][3]);
    if constexpr (ONE) { o.h1 = o.h0; o.l1 = o.l0; }
    else { o.h1 = make_uint4(hi[1][0], hi[1][1], hi[1][2], hi[1][3]); o.l1 = make_uint4(lo[1][0], lo[1][1], lo[1][2], lo[1][3]); }
    return o;
  }
};

struct NoWork {
  template <int C, int PI>
  __device__ __forceinline__ void chunk() {}
};
// a k-step of 4 tiles (the view layer's N = 128) has half the MFMA slots: tile T carries pair T of BOTH m-tiles
template <typename W>
struct PairOfPairs {
  W& w;
  __device__ __forceinline__ PairOfPairs(W& w_) : w(w_) {}
  template <int C, int PI>
  __device__ __forceinline__ void chunk() {
    if constexpr (PI < 4) { w.template chunk<C, PI>(); w.template chunk<C, PI + 4>(); }
  }
};

// Position-encoding operands of k-step S (slot map pe_feature: slot q = 8 S + j < NSIN is sin(2^(q/3) x_(q%3) + phase(h)), then the
// identity terms) for both m-tiles, one value pair per MFMA tile like PrevConv: stage 0 argument + reduction index, 1 reduced argument,
// 2 sine polynomial, 3 cosine polynomial + selection, 4 hi / lo split.  Same arithmetic, same order as enc_ops() / pe_sin().
template <int PREC, int S, int NSIN, bool ONE = false>
struct EncWork {
  using PP = Prec<PREC>;
  const float4 (&v)[2];
  const int h;
  uint32_t hi[2][4], lo[2][4];
  PeSin e0, e1;
  float f0, f1;
  __device__ __forceinline__ EncWork(const float4 (&v_)[2], int h_) : v(v_), h(h_) {}
  template <int MT, int Q>
  __device__ __forceinline__ float coord() const {      // x, y or z of m-tile MT for sine slot Q
    constexpr int c = Q % 3;
    return c == 0 ? v[MT].x : (c == 1 ? v[MT].y : v[MT].z);
  }
  template <int MT, int Q>
  __device__ __forceinline__ float plain() const {      // the non-sine slots
    if constexpr (Q == NSIN) return h ? v[MT].z : v[MT].x;
    else if constexpr (Q == NSIN + 1) return h ? 0.f : v[MT].y;
    else return 0.f;
  }
  template <int C, int PI>
  __device__ __forceinline__ void chunk() {
    constexpr int mt = PI >> 2, p = PI & 3, q0 = 8 * S + 2 * p, q1 = q0 + 1;
    if constexpr (ONE && mt == 1) return;
    const float phase = h ? 1.5707963705062866f : 0.0f;   // f32(0.5*pi) (rnerf/model_utils.py:213)
    if constexpr (C == 0) {
      if constexpr (q0 < NSIN) e0.s0(fadd(fmul(coord<mt, q0>(), (float)(1 << (q0 / 3))), phase));
      if constexpr (q1 < NSIN) e1.s0(fadd(fmul(coord<mt, q1>(), (float)(1 << (q1 / 3))), phase));
    } else if constexpr (C == 1) {
      if constexpr (q0 < NSIN) e0.s1();
      if constexpr (q1 < NSIN) e1.s1();
    } else if constexpr (C == 2) {
      if constexpr (q0 < NSIN) e0.s2();
      if constexpr (q1 < NSIN) e1.s2();
    } else if constexpr (C == 3) {
      if constexpr (q0 < NSIN) f0 = e0.s3(); else f0 = plain<mt, q0>();
      if constexpr (q1 < NSIN) f1 = e1.s3(); else f1 = plain<mt, q1>();
    } else {
      if constexpr (PP::F8X) {
        hi[mt][p] = pack2<true>(f0, f1);
        float ha, hb;
        unpack2<true>(hi[mt][p], ha, hb);
        lo[mt][p >> 1] = f8_pair<(p & 1) != 0>(lo[mt][p >> 1], f0, f1);
        lo[mt][2 + (p >> 1)] = f8_pair<(p & 1) != 0, true>(lo[mt][2 + (p >> 1)], f0 - ha, f1 - hb);
      } else if constexpr (PP::NP == 2) split2<PP::F16>(f0, f1, hi[mt][p], lo[mt][p]);
      else { hi[mt][p] = pack2<PP::F16>(f0, f1); lo[mt][p] = 0; }
    }
  }
  __device__ __forceinline__ KOpsT<ONE> result() const {
    KOpsT<ONE> o;
    o.h0 = make_uint4(hi[0][0], hi[0][1], hi[0][2], hi[0][3]); o.l0 = make_uint4(lo[0][0], lo[0][1], lo[0][2], lo[0][3]);
    if constexpr (ONE) { o.h1 = o.h0; o.l1 = o.l0; }
    else { o.h1 = make_uint4(hi[1][0], hi[1][1], hi[1][2], hi[1][3]); o.l1 = make_uint4(lo[1][0], lo[1][1], lo[1][2], lo[1][3]); }
    return o;
  }
};

// Work of the LAST k-step of a layer (the layer seam), in the shadow of its MFMAs: once the MFMAs of tile t have been issued, tile
// t - 1 is final, so in tile t's slots
//   * its m-tile 0 outputs move to prev0[t - 1], its m-tile 1 outputs to the wave's LDS state (128 register moves + 32 ds_write_b128
//     per wave, which used to run after the k-step with the matrix pipe idle), and
//   * the operands of the NEXT layer's k-step 0 (features 0..15 = tile 0, registers 0..7) are converted from the final accumulators
//     (used to be prev_ops(0) at the head of the next layer, exposed together with its bias load).
// Tile 7 is handed over by finish() after the k-step.
template <int PREC, bool MASK, bool ONE = false>
struct SeamWork {
  f32x16 (&acc0)[8];
  f32x16 (&acc1)[8];
  f32x16 (&prev0)[8];
  float4* st1;
  PrevConv<PREC, 0, MASK, true, ONE> cv;      // pairs 0..3: m-tile 0 from acc0[0], pairs 4..7: m-tile 1 from acc1[0]
  __device__ __forceinline__ SeamWork(f32x16 (&a0)[8], f32x16 (&a1)[8], f32x16 (&p0)[8], float4* s1) : acc0(a0), acc1(a1), prev0(p0), st1(s1), cv(a0[0]) {}
  template <int C, int T>
  __device__ __forceinline__ void move() {            // piece C of the hand-over of tile T
    // (the m-tile 0 moves are plain assignments: hipcc sinks them to the head of the next layer, ~1 k clocks per layer; pinning them here
    // with volatile v_accvgpr_read asm made the allocator spill in the steady k-steps: 1.57 -> 1.79 ms)
    if constexpr (C == 0) prev0[T] = acc0[T];
    else if constexpr (!ONE) st1[(T * 4 + C - 1) * 64] = make_float4(acc1[T][4 * (C - 1)], acc1[T][4 * (C - 1) + 1], acc1[T][4 * (C - 1) + 2], acc1[T][4 * (C - 1) + 3]);
  }
  template <int C, int Q>
  __device__ __forceinline__ void conv() {
    if constexpr (ONE && Q >= 4) return;
    if constexpr (C == 0 && Q >= 4) { cv.v1[2 * (Q & 3)] = acc1[0][2 * (Q & 3)]; cv.v1[2 * (Q & 3) + 1] = acc1[0][2 * (Q & 3) + 1]; }
    cv.template chunk<C, Q>();
  }
  template <int C, int PI>
  __device__ __forceinline__ void chunk() {
    if constexpr (PI >= 1) {
      move<C, PI - 1>();
      conv<C, PI - 1>();
      // pair 7 after pair 6 (PrevConv keeps ONE pair in flight: x0 / x1)
      if constexpr (PI == 7 && C == 4) { conv<0, 7>(); conv<1, 7>(); conv<2, 7>(); conv<3, 7>(); conv<4, 7>(); }
    }
  }
  __device__ __forceinline__ void finish() {
    move<0, 7>(); move<1, 7>(); move<2, 7>(); move<3, 7>(); move<4, 7>();
  }
};


// one n-tile of one k-step: 6 (X3) or 2 MFMAs with the conversion chunks of pair PI in their shadow
// PASSES (X3 modes): 3 = hi*hi + hi*lo + lo*hi (fp32-grade), 2 = drop the lo(weight) term, 1 = hi*hi only
template <int PREC, bool FIRST, int PI, typename W, int PASSES = Prec<PREC>::PASSES, typename BOPS = KOps>
__device__ __forceinline__ void tile_mfma(f32x16& a0, f32x16& a1, const uint4 ah, const uint4 al, const BOPS& b, W& work) {
  using PP = Prec<PREC>;
  constexpr bool TWO = !BOPS::ONE;      // m-tile 1 exists
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  a0 = mfma16<PP::F16>(ah, b.h0, FIRST ? zero : a0);
  work.template chunk<0, PI>();
  RNERF_PIN();
  if constexpr (TWO) a1 = mfma16<PP::F16>(ah, b.h1, FIRST ? zero : a1);
  work.template chunk<1, PI>();
  RNERF_PIN();
  if constexpr (PP::NP == 2 && PASSES == 3) {
    a0 = mfma16<PP::F16>(ah, b.l0, a0);
    work.template chunk<2, PI>();
    RNERF_PIN();
    if constexpr (TWO) a1 = mfma16<PP::F16>(ah, b.l1, a1);
    work.template chunk<3, PI>();
    RNERF_PIN();
    a0 = mfma16<PP::F16>(al, b.h0, a0);
    work.template chunk<4, PI>();
    RNERF_PIN();
    if constexpr (TWO) a1 = mfma16<PP::F16>(al, b.h1, a1);
  } else if constexpr (PP::NP == 2 && PASSES == 38) {      // f16 main term + the two cross terms on the fp8 MFMA
    a0 = mfma_f8(al.x, al.y, b.l0.x, b.l0.y, a0);            // fp8(W_lo) * fp8(x)
    work.template chunk<2, PI>();
    RNERF_PIN();
    if constexpr (TWO) a1 = mfma_f8(al.x, al.y, b.l1.x, b.l1.y, a1);
    work.template chunk<3, PI>();
    RNERF_PIN();
    a0 = mfma_f8(al.z, al.w, b.l0.z, b.l0.w, a0);            // fp8(W 2^-10) * fp8(x_lo 2^10)
    work.template chunk<4, PI>();
    RNERF_PIN();
    if constexpr (TWO) a1 = mfma_f8(al.z, al.w, b.l1.z, b.l1.w, a1);
  } else if constexpr (PP::NP == 2 && PASSES == 2) {
    a0 = mfma16<PP::F16>(ah, b.l0, a0);
    work.template chunk<2, PI>();
    work.template chunk<3, PI>();
    RNERF_PIN();
    if constexpr (TWO) a1 = mfma16<PP::F16>(ah, b.l1, a1);
    work.template chunk<4, PI>();
  } else if constexpr (PP::NP == 2 && PASSES == 22) {      // (W_hi + W_lo) * x_hi: the operand is a single 16-bit part
    a0 = mfma16<PP::F16>(al, b.h0, a0);
    work.template chunk<2, PI>();
    work.template chunk<3, PI>();
    RNERF_PIN();
    if constexpr (TWO) a1 = mfma16<PP::F16>(al, b.h1, a1);
    work.template chunk<4, PI>();
  } else {
    work.template chunk<2, PI>();
    work.template chunk<3, PI>();
    work.template chunk<4, PI>();
  }
  RNERF_PIN();
}

// one k-step (block KOFF of the slab) of MFMAs for NT n-tiles x 2 m-tiles; FIRST: accumulators start from 0.
// The A fragments of tile t+1 are read from LDS before the MFMAs of tile t are issued.
struct NoDma { __device__ __forceinline__ void operator()() const {} };
// Training kernels: the four operand / gradient stores of a k-step (hi and lo parts of both m-tiles) used to be issued at the head of the
// k-step, right behind the slab barrier and in front of the first MFMA — four 1 KiB vector stores with the matrix pipe idle (the ISA of every
// steady k-step begins `S S S S ds_read x 10 s_waitcnt M ...`).  RNERF_SPREAD_STORES: one store behind each of tiles 2..5 instead (the store
// hook `st` of kstep_mfma), in the MFMAs' shadow like the conversion chunks.  Measured (round 6, tools/r06/ab_wgrad.py, one box, alternating):
// training forward 2.016 / 1.974 -> 1.963 / 1.972 ms (inside the noise), dgrad 1.771 / 1.783 -> 1.838 / 1.842 ms (its spills grow from 259 to
// 301 registers) — the seventh re-placement of work in these engines that does not pay: OFF.
#ifndef RNERF_SPREAD_STORES
#define RNERF_SPREAD_STORES 0
#endif
struct NoStore { template <int K> __device__ __forceinline__ void part() const {} };

// `dma` is invoked after tile 1: the weight DMA of the next slab is issued while MFMAs are already in the matrix pipe and
// the A fragments of tiles 0..3 are already on their way (an LDS-DMA instruction costs ~100 issue cycles).
// A fragments are read FRAG_DEPTH tiles ahead of their MFMAs: with one wave per SIMD the LDS read rate is set by the
// number of reads in flight (latency x concurrency), not by the 256 B/clk peak.
constexpr int FRAG_DEPTH = 4;

template <int PREC, int NT, int KOFF, bool FIRST, typename W, bool NOREAD = false, typename D = NoDma, int PASSES = Prec<PREC>::PASSES, typename BOPS = KOps, typename ST = NoStore>
__device__ __forceinline__ void kstep_mfma(f32x16 (&acc0)[8], f32x16 (&acc1)[8], const BOPS& b, const char* slab, int lane, W& work, D dma = D(), const ST& st = ST()) {
  using PP = Prec<PREC>;
  const uint4* a = (const uint4*)slab + lane + (size_t)KOFF * NT * PP::NP * 64;
  uint4 fh[FRAG_DEPTH], fl[FRAG_DEPTH];
  // second block of a tile: 16 bytes per lane; f16f8: two 8-byte planes [fp8(W_lo): 64 lanes][fp8(W 2^-10): 64 lanes], read as two
  // conflict-free 8-byte loads (read as one 16-byte value whose halves are used at different times, hipcc re-reads the first half later
  // with a second, narrower load right in front of its MFMA)
  auto load_lo = [&](int tile) -> uint4 {
    if constexpr (PP::F8X) {
      const uint2* a2 = (const uint2*)((const uint4*)slab + ((size_t)KOFF * NT + tile) * PP::NP * 64 + 64) + lane;
      const uint2 u = a2[0], v = a2[64];
      return make_uint4(u.x, u.y, v.x, v.y);
    } else return a[(tile * PP::NP + PP::NP - 1) * 64];
  };
#pragma unroll
  for (int t = 0; t < FRAG_DEPTH; ++t) {
    const int tt = t < NT ? t : NT - 1;
    fh[t] = a[(tt * PP::NP) * 64];
    fl[t] = load_lo(tt);
  }
#define RNERF_TILE(T)                                                                                     \
  if constexpr (T < NT) {                                                                                 \
    uint4 ah = fh[T % FRAG_DEPTH], al = fl[T % FRAG_DEPTH];                                               \
    if constexpr (NOREAD) { asm volatile("" : "+v"(ah.x), "+v"(al.x)); }  /* ablation: opaque, so tiles are not CSE'd */ \
    tile_mfma<PREC, FIRST, (T & 7), W, PASSES, BOPS>(acc0[T], acc1[T], ah, al, b, work);                  \
    if constexpr (T + FRAG_DEPTH < NT && !NOREAD) {                                                       \
      fh[T % FRAG_DEPTH] = a[((T + FRAG_DEPTH) * PP::NP) * 64];                                           \
      fl[T % FRAG_DEPTH] = load_lo(T + FRAG_DEPTH);                                                       \
    }                                                                                                     \
  }
  RNERF_TILE(0) RNERF_TILE(1)
  dma();
  RNERF_TILE(2)
  if constexpr (NT == 8) { st.template part<0>(); RNERF_PIN(); }
  RNERF_TILE(3)
  if constexpr (NT == 8) { st.template part<1>(); RNERF_PIN(); }
  RNERF_TILE(4)
  if constexpr (NT == 8) { st.template part<2>(); RNERF_PIN(); }
  RNERF_TILE(5)
  if constexpr (NT == 8) { st.template part<3>(); RNERF_PIN(); }
  RNERF_TILE(6) RNERF_TILE(7)
#undef RNERF_TILE
}

// Workgroup = 4 waves = 256 consecutive sample rows; wave w owns rows [64w, 64w+64) as two 32-row m-tiles.
//   registers : acc0/acc1 (fp32 accumulators of both m-tiles, 2 x 128), prev0 (fp32 outputs of the previous layer, m-tile 0)
//   LDS       : 2 x SLAB weight ring (shared by the 4 waves) + 32 KiB per wave holding the previous layer's fp32 outputs
//               of m-tile 1 ([n-tile][4-register group][lane] float4) = 160 KiB for the X3 modes.
// The previous layer's outputs become this layer's B operands JUST IN TIME, one k-step ahead of use: bias + ReLU +
// hi/lo split of 8 values per lane per m-tile, issued in the shadow of the 48 MFMAs of the current k-step.
// Saved-activation layout of the training forward (consumed by the dgrad / wgrad kernels): slot-major uint4[SAVE_SLOTS][R][2]
// (R = rows padded to the 256-row tile), one uint4 = the 8 hi-part operand values lane (row, half h) fed to one k-step, i.e.
// exactly the B operands of the forward MFMAs.  slot 0..3: position encoding; 4 + 16*(l-1) + s: input k-step s of MFMA layer
// l = 1..9; 148..149: view encoding; 150..157: ReLU'd output of the view layer (= rgb head input).
// slots 158..166: ReLU bit masks for the dgrad chain, one uint4 per (row, half) per mask set (set l-1 = the input of MFMA layer
// l = 1..8, set 8 = the rgb-head input), stored word-major uint32[9][4][R][2]: word w holds k-steps 4w..4w+3 (written as soon as
// they are converted: no register lives across the layer), byte s&3, bit (j>>1) + 4*(j&1) <=> element j of k-step s is non-zero.
constexpr int SAVE_PE = 0, SAVE_L1 = 4, SAVE_VIEW = 148, SAVE_RGBIN = 150, SAVE_SLOTS = 158, SAVE_MASK = 158, SAVE_TOTAL = 167;
// Memory layout of the saved operands (and of the dY planes below): TILE-major — [32-row tile][slot][row in tile][half] uint4, so that
// the slots a wave touches for its 32 rows are consecutive 1 KiB blocks: the training forward / the dgrad write, and the wgrad reads,
// 16-32 KiB contiguous per (wave, chunk) instead of 1 KiB pieces 16 MB apart (one DRAM page activation per KiB).  The save buffer is
// [hi plane: SAVE_SLOTS slots][ReLU masks: 36 dword-slots = 576 uint4 per tile][lo plane: SAVE_SLOTS slots (hi + lo mode only)].
__host__ __device__ constexpr size_t sv_mask0(long long R) { return (size_t)(R / 32) * SAVE_SLOTS * 64; }                  // uint4 units
__host__ __device__ constexpr size_t sv_lo0(long long R) { return sv_mask0(R) + (size_t)(R / 32) * 576; }
__host__ __device__ constexpr size_t sv_addr(int q, long long t32, int m, int h) { return (((size_t)t32 * SAVE_SLOTS + q) * 32 + m) * 2 + h; }
__host__ __device__ constexpr size_t sv_mask_addr(int sw, long long t32, int m, int h) { return (((size_t)t32 * 36 + sw) * 32 + m) * 2 + h; }   // dword units

// non-zero flags of the 8 post-ReLU (non-negative) 16-bit values of one operand: bits 0..3 = even elements, 16..19 = odd elements
__device__ __forceinline__ uint32_t nz_nibbles(const uint4& o) {
  return pk_min1(o.x) | (pk_min1(o.y) << 1) | (pk_min1(o.z) << 2) | (pk_min1(o.w) << 3);
}
__device__ __forceinline__ uint32_t nz_byte(uint32_t nib) { return (nib & 0xFu) | (nib >> 12); }   // even flags | odd flags << 4

// The saves of one k-step one at a time (kstep_mfma's store hook, RNERF_SPREAD_STORES): part 0 / 1 = the hi parts of m-tile 0 / 1, 2 / 3 = the
// lo parts (TRAIN = 2: f16, 3: e4m3 bytes).  (Namespace scope: a local class cannot have a member template.)
template <int TRAIN, bool ONE>
struct SaveParts {
  uint4* save; long long save_rows, t32_0; int q, m, h; const KOpsT<ONE>& o;
  template <int K> __device__ __forceinline__ void part() const {
    if constexpr (TRAIN != 0) {
#ifndef RNERF_FWD_NOSAVE
      uint4* dst = save + sv_addr(q, t32_0, m, h);
      if constexpr (K == 0) stream_store(dst, o.h0);
      if constexpr (K == 1 && !ONE) stream_store(dst + (size_t)SAVE_SLOTS * 64, o.h1);
      if constexpr (TRAIN == 2) {
        uint4* dl = dst + sv_lo0(save_rows);
        if constexpr (K == 2) stream_store(dl, o.l0);
        if constexpr (K == 3 && !ONE) stream_store(dl + (size_t)SAVE_SLOTS * 64, o.l1);
      }
      if constexpr (TRAIN == 3) {
        uint2* dl = (uint2*)(save + sv_lo0(save_rows)) + sv_addr(q, t32_0, m, h);
        if constexpr (K == 2) stream_store8(dl, lo8_pack<false>(o.l0, LO8_SCALE_X));
        if constexpr (K == 3 && !ONE) stream_store8(dl + (size_t)SAVE_SLOTS * 64, lo8_pack<false>(o.l1, LO8_SCALE_X));
      }
#endif
    }
  }
};

// TRAIN: 0 = evaluation, 1 = training forward keeping the hi 16-bit operand parts, 2 = hi and lo parts (fp32-grade backward),
// 3 = hi parts + the lo parts as e4m3 bytes (RNERF_BWD_F16X3_LO8: lo8 plane = uint2[...] at the lo plane's place, same (tile, slot, row, half) order)
template <int PREC, int dbg, int TRAIN, bool ONE = false>
__global__ void __launch_bounds__(256, 1)
nerfmlp_fwd_kernel(const char* __restrict__ packed, const float4* __restrict__ rows_pd, const float4* __restrict__ rows_dr,
                   const int* __restrict__ node_of_sample, int B, long long total_rows, int n_tiles, float4* __restrict__ out_raw,
                   uint4* __restrict__ save, long long save_rows, int* __restrict__ tileq, const float* __restrict__ gate, int gate_skip_if) {
  // gate (nullable): a range flag raised by a pack kernel.  The launch does nothing when (flag set) == gate_skip_if: the f16f8 forward steps
  // aside when one of ITS weights left the range of its 2^14-scaled stream, and the f16x3 launch queued right behind it (same outputs,
  // gate_skip_if = 0) does the work instead — a per-launch fallback decided on the device, no host round trip (rnerf_nerfmlp_forward).
  if (gate != nullptr && ((gate[0] != 0.f) == (gate_skip_if != 0))) return;
  // tileq != nullptr (training forward on a capped grid): tiles beyond the first round are handed out by an atomic counter (tileq[0],
  // zeroed by the launcher; tileq[1 + workgroup] passes the draw from thread 0 to the other waves) — with fewer workgroups than CUs the
  // static stride would leave most of the chip idle in a last partial round (2048 tiles over 248 workgroups: 9 rounds instead of 8.26).
  // dbg != 0: profiling ablations, only instantiated with -DRNERF_MLP_ABLATE (results are garbage):
  //   bit0 = skip the weight-stream loads, bit1 = skip ds_read + MFMA, bit2 = skip the barrier.
  // Compile-time on purpose: a runtime branch per k-step would split the scheduling region and stop the compiler from
  // interleaving the operand conversion (VALU) with the MFMAs.
  // ONE: 128-row tiles — every wave runs ONE 32-row m-tile (rows 128 tile + 32 wave + m): half the serial work per tile, for launches whose
  // 256-row tiles would leave more than half of the CUs idle.  n_tiles then counts 128-row tiles; the save layout (32-row tiles) is the same.
  using PP = Prec<PREC>;
  using KOps = KOpsT<ONE>;
  constexpr int SLAB = PP::SLAB;
  constexpr int WROWS = ONE ? 32 : 64, TROWS = 4 * WROWS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool OVFL_MODE = PP::F8X || TRAIN == 3;      // fp8 conversions clamp; f16 overflow clamps to 65504 too (see `watch`)
  if constexpr (OVFL_MODE) f8x_mode();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  const float* __restrict__ aux = (const float*)(packed + PP::STREAM_BYTES);
  float4* __restrict__ st1 = (float4*)(smem + 2 * SLAB + wave * 32768) + lane;   // + (t*4 + rq)*64
  constexpr float INV_SCALE = 1.0f / PP::WSCALE;
  int buf = 0;
  size_t off = 0;   // stream offset of the next slab to prefetch
  float prof_dma = 0.f, prof_bar = 0.f, prof_tot = 0.f, prof_n = 0.f;
  unsigned long long prof_last = __builtin_amdgcn_s_memtime();
  // dbg & 256: shader clocks per phase of a tile (PH(k) books the time since the previous mark on phase k):
  //   0 row loads  1 layer 0 (encoding slabs)  2 first k-step of a hidden layer (incl. its operand conversion)  3 k-steps 1..15
  //   4 layer end (state to registers / LDS)  5 skip-concat slabs  6 sigma head  7 view layer  8 rgb head + store
  float prof_ph[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define PH(K) do { if constexpr ((dbg & 256) != 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_ph[K] += (float)(t_ - prof_last); prof_last = t_; } } while (0)

  // dbg & 512 — REDO, the range-safe second pass (rnerf_nerfmlp_forward queues it behind every f16-based evaluation launch, in bf16x3: fp32's
  // exponent range): it walks the same tiles but works only on those that hold a row the first pass returned as NaN (an activation above
  // f16's 65504 or a weight >= 256, see `watch` / AUX_FLAG below) and rewrites only those rows.  The flags ARE the first pass's outputs: no
  // scratch, nothing to clear, and a launch without such a row costs one read of out_raw (8 tiles' rows in flight per barrier).
  constexpr bool REDO = (dbg & 512) != 0;
  // bit k of `todo`: this workgroup's k-th tile (blockIdx.x + k gridDim.x) holds a NaN row.  Found in the prologue, while the LDS is still
  // free (four 8-byte words for the cross-wave OR; the kernel owns all 160 KiB later).  A workgroup with more than 64 tiles works on every
  // tile beyond the 64th (correct — only NaN rows are rewritten — just not skipped: > 4 M rows per launch on this chip).
  unsigned long long todo = ~0ull;
  if constexpr (REDO) {
    unsigned long long mine = 0;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
      const long long tk = (long long)blockIdx.x + (long long)k * gridDim.x;
      const long long r = tk * TROWS + tid;
      const bool bad = tk < n_tiles && r < total_rows && __builtin_isnan(out_raw[r].x);
      if (__ballot(bad) != 0ull) mine |= 1ull << k;
    }
    unsigned long long* red = (unsigned long long*)smem;
    if (lane == 0) red[wave] = mine;
    __syncthreads();
    todo = red[0] | red[1] | red[2] | red[3];
    __syncthreads();
  }
  auto find_next = [&](int t) -> int {      // first tile >= t of this workgroup's sequence that is to be worked on; n_tiles if none
    if constexpr (REDO) {
      while (t < n_tiles) {
        const int k = (t - (int)blockIdx.x) / (int)gridDim.x;
        if (k >= 64 || ((todo >> k) & 1ull)) return t;
        t += (int)gridDim.x;
      }
      return n_tiles;
    } else {
      return t;
    }
  };
  const int first_tile = find_next((int)blockIdx.x);
  if (first_tile < n_tiles) { if (!(dbg & 1)) issue_slab<SLAB>(packed, 0u, wave, lane); off = SLAB; }
  slab_wait_dma();
  __syncthreads();

  for (int tile = first_tile; tile < n_tiles;) {
    if (tileq != nullptr && tid == 0)      // read back by every wave at the end of the tile, ~145 barriers later
      __hip_atomic_store(tileq + 1 + blockIdx.x, (int)gridDim.x + atomicAdd(tileq, 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The aux vectors (biases, head weights) are read at lane-dependent but tile-invariant addresses.  Left alone, hipcc hoists the ~100
    // 64-bit per-lane addresses out of the tile loop, spills them, and in the heads / the view layer reloads each one from scratch right
    // before its load, one after the other (measured: 52 k clocks for the rgb head, 5.4 k per view-layer slab).  An opaque per-tile copy
    // of the base keeps the address arithmetic next to the loads (scalar base + lane offset + immediate).
    const float* __restrict__ auxt = aux;
    asm volatile("" : "+s"(auxt));
    int next_tile = n_tiles;
    long long row[2]; bool row_ok[2];
    float4 pd[2], dr[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (ONE && mt == 1) { row[1] = row[0]; row_ok[1] = false; pd[1] = pd[0]; dr[1] = dr[0]; continue; }
      row[mt] = (long long)tile * TROWS + wave * WROWS + mt * 32 + m;
      row_ok[mt] = row[mt] < total_rows;
      if (!row_ok[mt]) row[mt] = total_rows - 1;
      size_t rec = (size_t)row[mt];
      if (node_of_sample) { const long long sidx = row[mt] / B; rec = (size_t)node_of_sample[sidx] * B + (size_t)(row[mt] - sidx * B); }
      pd[mt] = rows_pd[rec];
      dr[mt] = rows_dr[rec];
    }

    f32x16 acc0[8], acc1[8], prev0[8];
    PH(0);

    // training forward: keep the hi parts of the operands of slot q (see SAVE_* above); padded rows are written too
    const long long t32_0 = (long long)tile * (TROWS / 32) + wave * (WROWS / 32);      // 32-row tile of m-tile 0 (m-tile 1: + 1)
    auto save_ops = [&](int q, const KOps& o) {
      if constexpr (TRAIN != 0) {
#ifndef RNERF_FWD_NOSAVE      /* profiling ablation */
        uint4* dst = save + sv_addr(q, t32_0, m, h);
        stream_store(dst, o.h0);
        if constexpr (!ONE) stream_store(dst + (size_t)SAVE_SLOTS * 64, o.h1);          // m-tile 1 = the next 32-row tile
        if constexpr (TRAIN == 2) {            // lo plane
          uint4* dl = dst + sv_lo0(save_rows);
          stream_store(dl, o.l0);
          if constexpr (!ONE) stream_store(dl + (size_t)SAVE_SLOTS * 64, o.l1);
        }
        if constexpr (TRAIN == 3) {            // lo plane as e4m3 bytes: 8 B per (row, half)
          uint2* dl = (uint2*)(save + sv_lo0(save_rows)) + sv_addr(q, t32_0, m, h);
          stream_store8(dl, lo8_pack<false>(o.l0, LO8_SCALE_X));
          if constexpr (!ONE) stream_store8(dl + (size_t)SAVE_SLOTS * 64, lo8_pack<false>(o.l1, LO8_SCALE_X));
        }
#endif
      }
    };

    // Range watch of the f16 operand parts (RNERF_FWD_NORANGE: ablation).  A hidden activation above f16's 65504 becomes hi = inf; the next
    // layer's sums are inf - inf = NaN (sign bit set on this hardware), which the ReLU turns into 0 — every unit of that layer would read 0
    // and the outputs would be finite, plausible and wrong (found in round 4, tools/r04/dbg_hot.py).  So the largest |hi| bit pattern per lane
    // is kept (one v_pk_max_u16 per operand dword: positive f16 order like their bits, inf = 0x7C00) and a row that met inf / NaN returns NaN.
    uint32_t ovf0 = 0, ovf1 = 0;
    auto watch = [&](const KOps& o, bool is_signed) {
#ifndef RNERF_FWD_NORANGE
      {                                                                 // (bf16 modes too — round 5: a non-finite position used to come out FINITE there)
        const uint32_t am = is_signed ? 0x7FFF7FFFu : 0xFFFFFFFFu;      // the bottleneck has no ReLU: drop the sign bits
        ovf0 = pk_maxu(pk_maxu(ovf0, o.h0.x & am), pk_maxu(pk_maxu(o.h0.y & am, o.h0.z & am), o.h0.w & am));
        if constexpr (!ONE) ovf1 = pk_maxu(pk_maxu(ovf1, o.h1.x & am), pk_maxu(pk_maxu(o.h1.y & am, o.h1.z & am), o.h1.w & am));
      }
#endif
    };

    uint32_t mw0 = 0, mw1 = 0;                                                // mask bytes of up to 4 k-steps, then one dword store
    auto save_mask = [&](int set, int s, uint32_t nib0, uint32_t nib1) {      // nib: nz_nibbles() of the hi operands of k-step s
#ifdef RNERF_FWD_NOMASK       /* profiling ablation */
      return;
#endif
      if constexpr (TRAIN != 0) {
        if ((s & 3) == 0) { mw0 = nz_byte(nib0); mw1 = nz_byte(nib1); }
        else { mw0 |= nz_byte(nib0) << (8 * (s & 3)); mw1 |= nz_byte(nib1) << (8 * (s & 3)); }
        if ((s & 3) == 3) {      // word-major: a wave's 64 dwords are contiguous (full-line writes)
          uint32_t* dst = (uint32_t*)(save + sv_mask0(save_rows)) + sv_mask_addr(set * 4 + (s >> 2), t32_0, m, h);
          __builtin_nontemporal_store(mw0, dst);
          if constexpr (!ONE) __builtin_nontemporal_store(mw1, dst + 36 * 64);
        }
      }
    };

    // positional-encoding operands of k-step s (slot maps pe_feature / view_feature), for both m-tiles
    auto enc_ops = [&](const float4 (&v)[2], int s, int nsin) -> KOps {
      const float phase = h ? 1.5707963705062866f : 0.0f;   // f32(0.5*pi) (rnerf/model_utils.py:213)
      float f[2][8];
#pragma unroll
      for (int mt = 0; mt < (ONE ? 1 : 2); ++mt)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int q = 8 * s + j;
          if (q < nsin) {
            const int d = q / 3, c = q % 3;
            const float x = c == 0 ? v[mt].x : (c == 1 ? v[mt].y : v[mt].z);
            f[mt][j] = pe_sin(fadd(fmul(x, (float)(1 << d)), phase));
          } else if (q == nsin) f[mt][j] = h ? v[mt].z : v[mt].x;
          else if (q == nsin + 1) f[mt][j] = h ? 0.f : v[mt].y;
          else f[mt][j] = 0.f;
        }
      KOps o;
      split8<PREC>(f[0], o.h0, o.l0);
      if constexpr (ONE) { o.h1 = o.h0; o.l1 = o.l0; } else split8<PREC>(f[1], o.h1, o.l1);
      return o;
    };

    // operands of k-step s from the previous layer's outputs: x = max(acc * inv_scale + bias, floor), hi/lo split.
    auto prev_ops = [&](int s, const float* __restrict__ bias, int floor_v) -> KOps {
      const float4 b0 = *(const float4*)(bias + 16 * s + 4 * h), b1 = *(const float4*)(bias + 16 * s + 8 + 4 * h);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float r1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if constexpr (!ONE) {
        const float4 u0 = st1[((s >> 1) * 4 + 2 * (s & 1)) * 64], u1 = st1[((s >> 1) * 4 + 2 * (s & 1) + 1) * 64];
        r1[0] = u0.x; r1[1] = u0.y; r1[2] = u0.z; r1[3] = u0.w; r1[4] = u1.x; r1[5] = u1.y; r1[6] = u1.z; r1[7] = u1.w;
      }
      float x0[8], x1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x0[j] = relu_keep(fmaf(prev0[s >> 1][8 * (s & 1) + j], INV_SCALE, bb[j]), floor_v);
        x1[j] = relu_keep(fmaf(r1[j], INV_SCALE, bb[j]), floor_v);
      }
      KOps o;
      split8<PREC>(x0, o.h0, o.l0);
      if constexpr (ONE) { o.h1 = o.h0; o.l1 = o.l0; } else split8<PREC>(x1, o.h1, o.l1);
      return o;
    };

    // bias of the 8 features of k-step s (global, L2/L1 resident).  Loaded one slab EARLIER than it is consumed: the
    // vmcnt(0) of the barrier that ends the slab (needed for the weight DMA anyway) then also covers this load.
    auto load_bias8 = [&](int s, const float* __restrict__ bias, float (&b)[8]) {
      const float4 b0 = *(const float4*)(bias + 16 * s + 4 * h), b1 = *(const float4*)(bias + 16 * s + 8 + 4 * h);
      b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
    };
    // raw accumulators of m-tile 1 for k-step s (this wave's LDS region)
    auto load_state8 = [&](int s, float (&v)[8]) {
      if constexpr (ONE) return;
      const float4 u0 = st1[((s >> 1) * 4 + 2 * (s & 1)) * 64], u1 = st1[((s >> 1) * 4 + 2 * (s & 1) + 1) * 64];
      v[0] = u0.x; v[1] = u0.y; v[2] = u0.z; v[3] = u0.w; v[4] = u1.x; v[5] = u1.y; v[6] = u1.z; v[7] = u1.w;
    };
    NoWork nowork;

#define SLAB_PREFETCH(DO_NEXT)                                                                                       \
  do { if (DO_NEXT) { if (!(dbg & 1)) issue_slab<SLAB>(packed + off, (unsigned)((buf ^ 1) * SLAB), wave, lane); off += SLAB; } } while (0)
#define SLAB_DONE()                                                                                                  \
  do {                                                                                                               \
    if constexpr ((dbg & 64) != 0) {   /* profiling: where does a slab's time go (shader clocks, summed per wave) */   \
      const unsigned long long ta = __builtin_amdgcn_s_memtime();                                                    \
      slab_wait_dma();                                                                                               \
      const unsigned long long tb = __builtin_amdgcn_s_memtime();                                                    \
      __syncthreads();                                                                                               \
      const unsigned long long tc2 = __builtin_amdgcn_s_memtime();                                                   \
      prof_dma += (float)(tb - ta); prof_bar += (float)(tc2 - tb); prof_tot += (float)(tc2 - prof_last); prof_last = tc2; prof_n += 1.f; \
    } else {                                                                                                         \
      slab_wait_dma(); if (!(dbg & 4)) __syncthreads();                                                              \
    }                                                                                                                \
    buf ^= 1;                                                                                                        \
  } while (0)

    float sig0 = 0.f, sig1 = 0.f;

    // ---- layer 0: pos_enc(pos, 0, 10) (63) -> 256 (Dense_0)  (rnerf/models.py:257)
    KOps cur;                                   // operands of the next k-step; across a layer seam: k-step 0 of the next layer (SeamWork)
    {
      cur = enc_ops(pd, 0, 30);
      SeamWork<PREC, TRAIN != 0, ONE> seam(acc0, acc1, prev0, st1);
      seam.cv.floor_v = RELU_FLOOR;
      load_bias8(0, auxt + AUX_BIAS, seam.cv.b);
      load_bias8(0, auxt + AUX_ZERO, seam.cv.ws);
      // k-steps 0..2 convert the position-encoding operands of the next k-step in their own MFMA shadows (EncWork), k-step 3 carries the seam
#define RNERF_PE_KSTEP(S, FIRST_, SAVE_)                                                                             \
      {                                                                                                              \
        if (SAVE_) save_ops(SAVE_PE + S, cur);                                                                       \
        watch(cur, true);      /* a non-finite position (a caller's NaN) must not come out as a colour either */    \
        SLAB_PREFETCH(true);   /* glds first: it is a scheduling boundary, conversion + MFMAs must share the region after it */ \
        if constexpr (S < 3) {                                                                                       \
          EncWork<PREC, S + 1, 30, ONE> ew(pd, h);                                                                        \
          if (!(dbg & 2)) kstep_mfma<PREC, 8, 0, FIRST_>(acc0, acc1, cur, smem + buf * SLAB, lane, ew);              \
          cur = ew.result();                                                                                         \
        } else {                                                                                                     \
          if (!(dbg & 2)) kstep_mfma<PREC, 8, 0, false>(acc0, acc1, cur, smem + buf * SLAB, lane, seam);             \
        }                                                                                                            \
        SLAB_DONE();                                                                                                 \
      }
      RNERF_PE_KSTEP(0, true, true) RNERF_PE_KSTEP(1, false, true) RNERF_PE_KSTEP(2, false, true) RNERF_PE_KSTEP(3, false, true)
      PH(1);
      seam.finish();
      cur = seam.cv.result();
      PH(4);
    }

    // bias / sigma weights of the conversion that runs in the shadow of the next hidden k-step (fetched one slab ahead, across layers too)
    float bnext[8], wnext[8];
    load_bias8(1, auxt + AUX_BIAS, bnext);
    load_bias8(1, auxt + AUX_ZERO, wnext);

    // ---- layers 1..8: Dense_1..Dense_7 (inputs ReLU'd; Dense_5 also takes the skip concat), Dense_9 = bottleneck
#pragma unroll 1
    for (int l = 1; l <= 8; ++l) {
      const float* __restrict__ bias = auxt + AUX_BIAS + 256 * (l - 1);
      // the seam of this layer: hand-over of the outputs + conversion of the next layer's k-step 0 (bias of THIS layer; the bottleneck
      // Dense_9 = layer 8 has no activation), run by the layer's last k-step
      SeamWork<PREC, TRAIN != 0, ONE> seam(acc0, acc1, prev0, st1);
      seam.cv.floor_v = l == 8 ? NO_FLOOR : RELU_FLOOR;
      // sigma head (Dense_8, rnerf/model_utils.py:70) = sum over the trunk output x7 = the inputs of layer 8: k-step 0 in layer 7's seam,
      // k-steps 1..15 in layer 8's conversions.  (The seam's bias / sigma weights are fetched at k-step 13: 16 registers less in the steady state.)
      const float* __restrict__ wseam = auxt + (l == 7 ? AUX_WSIG : AUX_ZERO);
      const float* __restrict__ wsel = auxt + (l == 8 ? AUX_WSIG : AUX_ZERO);
#define RNERF_KSTEP(S)                                                                                              \
      {                                                                                                              \
        const SaveParts<TRAIN, ONE> sp{save, save_rows, t32_0, SAVE_L1 + 16 * (l - 1) + S, m, h, cur};                            \
        if constexpr (!(RNERF_SPREAD_STORES && TRAIN != 0)) save_ops(SAVE_L1 + 16 * (l - 1) + S, cur);               \
        watch(cur, false);                                                                                           \
        if constexpr (TRAIN != 0 && S == 0) save_mask(l - 1, 0, nz_nibbles(cur.h0), nz_nibbles(cur.h1));                  \
        if constexpr (S == 13) { load_bias8(0, bias + 256, seam.cv.b); load_bias8(0, wseam, seam.cv.ws); }           \
        float bnn[8], wnn[8];                                                                                        \
        if constexpr (S + 2 < 16) { load_bias8(S + 2, bias, bnn); load_bias8(S + 2, wsel, wnn); }   /* consumed in the NEXT slab */ \
        else if constexpr (S == 14) { load_bias8(1, bias + 256, bnn); load_bias8(1, wseam, wnn); }   /* k-step 1 of the NEXT layer */ \
        auto dma = [&]() { SLAB_PREFETCH(true); };                                                                   \
        if constexpr (S + 1 < 16) {                                                                                  \
          PrevConv<PREC, S + 1, TRAIN != 0, true, ONE> cv(prev0[(S + 1) >> 1]);                                           \
          cv.floor_v = RELU_FLOOR;                                                                                        \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) { cv.b[j] = bnext[j]; cv.ws[j] = wnext[j]; }                 \
          load_state8(S + 1, cv.v1);                                                                                 \
          if (dbg & 8) { kstep_mfma<PREC, 8, 0, S == 0, NoWork, (dbg & 16) != 0>(acc0, acc1, cur, smem + buf * SLAB, lane, nowork, dma); } \
          else { if (!(dbg & 2)) { if constexpr (RNERF_SPREAD_STORES && TRAIN != 0) kstep_mfma<PREC, 8, 0, S == 0, PrevConv<PREC, S + 1, TRAIN != 0, true, ONE>, (dbg & 16) != 0, decltype(dma), Prec<PREC>::PASSES, KOps, SaveParts<TRAIN, ONE>>(acc0, acc1, cur, smem + buf * SLAB, lane, cv, dma, sp); else \
            kstep_mfma<PREC, 8, 0, S == 0, PrevConv<PREC, S + 1, TRAIN != 0, true, ONE>, (dbg & 16) != 0, decltype(dma)>(acc0, acc1, cur, smem + buf * SLAB, lane, cv, dma); }              \
          cur = cv.result();                                                                                         \
          sig0 += cv.sg0; sig1 += cv.sg1;                                                                            \
          save_mask(l - 1, S + 1, cv.nz[0], cv.nz[1]); }                                                             \
        } else {                                                                                                     \
          /* also for l == 5, whose last k-step is the 4th skip slab: that one runs the seam again on the final sums (no branch here: */ \
          /* a run-time choice of the work functor splits the accumulators' live ranges and hipcc spills them around it) */ \
          if (!(dbg & 2)) { if constexpr (RNERF_SPREAD_STORES && TRAIN != 0) kstep_mfma<PREC, 8, 0, false, SeamWork<PREC, TRAIN != 0, ONE>, false, decltype(dma), Prec<PREC>::PASSES, KOps, SaveParts<TRAIN, ONE>>(acc0, acc1, cur, smem + buf * SLAB, lane, seam, dma, sp); else \
            kstep_mfma<PREC, 8, 0, false, SeamWork<PREC, TRAIN != 0, ONE>, false, decltype(dma)>(acc0, acc1, cur, smem + buf * SLAB, lane, seam, dma); } \
        }                                                                                                            \
        if constexpr (S + 2 <= 16) { _Pragma("unroll") for (int j = 0; j < 8; ++j) { bnext[j] = bnn[j]; wnext[j] = wnn[j]; } } \
        SLAB_DONE();                                                                                                 \
      }
      RNERF_KSTEP(0) PH(2); RNERF_KSTEP(1) RNERF_KSTEP(2) RNERF_KSTEP(3) RNERF_KSTEP(4) RNERF_KSTEP(5) RNERF_KSTEP(6) RNERF_KSTEP(7)
      RNERF_KSTEP(8) RNERF_KSTEP(9) RNERF_KSTEP(10) RNERF_KSTEP(11) RNERF_KSTEP(12) RNERF_KSTEP(13) RNERF_KSTEP(14) RNERF_KSTEP(15)
#undef RNERF_KSTEP
      PH(3);
      if (l == 5) {   // skip concat: [x, inputs] (rnerf/model_utils.py:68-69)
        // the encoding is recomputed here on purpose (in the MFMA shadow); with the position opaque the compiler cannot keep layer 0's
        // encoded values alive across five layers instead (f16f8: it kept all 64, through scratch — 8 k clocks per skip slab instead of 4)
        asm volatile("" : "+v"(pd[0].x), "+v"(pd[0].y), "+v"(pd[0].z), "+v"(pd[1].x), "+v"(pd[1].y), "+v"(pd[1].z));
        cur = enc_ops(pd, 0, 30);
        RNERF_PE_KSTEP(0, false, false) RNERF_PE_KSTEP(1, false, false) RNERF_PE_KSTEP(2, false, false) RNERF_PE_KSTEP(3, false, false)
        PH(5);
      }
      seam.finish();
      cur = seam.cv.result();
      PH(4);
      sig0 += seam.cv.sg0; sig1 += seam.cv.sg1;
    }

    // ---- view layer: [bottleneck(256) (no activation), pos_enc(dir, 0, 4) (27)] -> 128 (Dense_10)  (rnerf/models.py:289-294)
    {
      const float* __restrict__ bias = auxt + AUX_BIAS + 256 * 8;
      KOps c0 = cur;                             // converted by layer 8's seam
      KOps c1 = prev_ops(1, bias, NO_FLOOR);
      float bA[8], bB[8];                        // biases of the two k-steps converted in the shadow of the current slab
      load_bias8(2, bias, bA);
      load_bias8(3, bias, bB);
#define RNERF_VSLAB(SL)                                                                                              \
      {                                                                                                              \
        save_ops(SAVE_L1 + 16 * 8 + 2 * SL, c0);                                                                     \
        save_ops(SAVE_L1 + 16 * 8 + 2 * SL + 1, c1);                                                                 \
        watch(c0, true); watch(c1, true);                                                                            \
        SLAB_PREFETCH(true);                                                                                         \
        if constexpr (SL + 1 < 8) {                                                                                  \
          float bA2[8], bB2[8];                                                                                      \
          if constexpr (SL + 2 < 8) { load_bias8(2 * SL + 4, bias, bA2); load_bias8(2 * SL + 5, bias, bB2); }        \
          PrevConv<PREC, 2 * SL + 2, false, false, ONE> cvA(prev0[SL + 1]);                                                      \
          PrevConv<PREC, 2 * SL + 3, false, false, ONE> cvB(prev0[SL + 1]);                                                      \
          cvA.floor_v = NO_FLOOR; cvB.floor_v = NO_FLOOR;                                                                \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) { cvA.b[j] = bA[j]; cvB.b[j] = bB[j]; }                      \
          load_state8(2 * SL + 2, cvA.v1);                                                                           \
          load_state8(2 * SL + 3, cvB.v1);                                                                           \
          PairOfPairs<decltype(cvA)> wA(cvA);                                                                        \
          PairOfPairs<decltype(cvB)> wB(cvB);                                                                        \
          if (!(dbg & 2)) {                                                                                          \
            kstep_mfma<PREC, 4, 0, SL == 0>(acc0, acc1, c0, smem + buf * SLAB, lane, wA);                            \
            kstep_mfma<PREC, 4, 1, false>(acc0, acc1, c1, smem + buf * SLAB, lane, wB);                              \
          }                                                                                                          \
          c0 = cvA.result(); c1 = cvB.result();                                                                      \
          if constexpr (SL + 2 < 8) { _Pragma("unroll") for (int j = 0; j < 8; ++j) { bA[j] = bA2[j]; bB[j] = bB2[j]; } } \
        } else {                                                                                                     \
          const KOps n0 = enc_ops(dr, 0, 12), n1 = enc_ops(dr, 1, 12);                                               \
          if (!(dbg & 2)) {                                                                                          \
            kstep_mfma<PREC, 4, 0, false>(acc0, acc1, c0, smem + buf * SLAB, lane, nowork);                          \
            kstep_mfma<PREC, 4, 1, false>(acc0, acc1, c1, smem + buf * SLAB, lane, nowork);                          \
          }                                                                                                          \
          c0 = n0; c1 = n1;                                                                                          \
        }                                                                                                            \
        SLAB_DONE();                                                                                                 \
      }
      RNERF_VSLAB(0) RNERF_VSLAB(1) RNERF_VSLAB(2) RNERF_VSLAB(3) RNERF_VSLAB(4) RNERF_VSLAB(5) RNERF_VSLAB(6) RNERF_VSLAB(7)
#undef RNERF_VSLAB
      // last slab of the tile (the two view-encoding k-steps): prefetch the first slab of the next tile (stream restarts)
      save_ops(SAVE_VIEW, c0);
      save_ops(SAVE_VIEW + 1, c1);
      watch(c0, true); watch(c1, true);
      next_tile = tileq != nullptr ? __builtin_amdgcn_readfirstlane(__hip_atomic_load(tileq + 1 + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                   : find_next(tile + (int)gridDim.x);
      const bool has_next_tile = next_tile < n_tiles;
      if (has_next_tile) off = 0;
      SLAB_PREFETCH(has_next_tile);
      if (!(dbg & 2)) { kstep_mfma<PREC, 4, 0, false>(acc0, acc1, c0, smem + buf * SLAB, lane, nowork); kstep_mfma<PREC, 4, 1, false>(acc0, acc1, c1, smem + buf * SLAB, lane, nowork); }
      SLAB_DONE();
      PH(7);
    }

    // ---- heads: sigma (Dense_8, accumulated above) and rgb (Dense_11) on the fp32 view-layer output
    {
      // the 16 weight / bias vectors of n-tile t + 1 are fetched while n-tile t is reduced (left to itself hipcc loads two vectors, waits,
      // uses them, loads the next two: 32 exposed round trips, 18 k clocks per tile)
      const float* __restrict__ hw = auxt + 4 * h;
      auto head_load = [&](int t, float4 (&W)[16]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = 32 * t + 8 * g;
          W[4 * g] = gload4(hw + AUX_BIAS + 256 * 9 + n);
          W[4 * g + 1] = gload4(hw + AUX_WRGB + n); W[4 * g + 2] = gload4(hw + AUX_WRGB + 128 + n); W[4 * g + 3] = gload4(hw + AUX_WRGB + 256 + n);
        }
      };
      float p0[3] = {0.f, 0.f, 0.f}, p1[3] = {0.f, 0.f, 0.f};
      float4 Wc[16];
      head_load(0, Wc);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 Wn[16];
        if (t + 1 < 4) head_load(t + 1, Wn);
        RNERF_PIN();
        float rv0[16], rv1[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bb = Wc[4 * g], wr = Wc[4 * g + 1], wg = Wc[4 * g + 2], wb = Wc[4 * g + 3];
          const float bv[4] = {bb.x, bb.y, bb.z, bb.w}, wrv[4] = {wr.x, wr.y, wr.z, wr.w}, wgv[4] = {wg.x, wg.y, wg.z, wg.w}, wbv[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v0 = relu_keep(fmaf(acc0[t][4 * g + i], INV_SCALE, bv[i]), RELU_FLOOR);
            p0[0] = fmaf(v0, wrv[i], p0[0]); p0[1] = fmaf(v0, wgv[i], p0[1]); p0[2] = fmaf(v0, wbv[i], p0[2]);
            rv0[4 * g + i] = v0;
            if constexpr (!ONE) {
              const float v1 = relu_keep(fmaf(acc1[t][4 * g + i], INV_SCALE, bv[i]), RELU_FLOOR);
              p1[0] = fmaf(v1, wrv[i], p1[0]); p1[1] = fmaf(v1, wgv[i], p1[1]); p1[2] = fmaf(v1, wbv[i], p1[2]);
              rv1[4 * g + i] = v1;
            } else rv1[4 * g + i] = v0;
          }
        }
        if (t + 1 < 4) {
#pragma unroll
          for (int q = 0; q < 16; ++q) Wc[q] = Wn[q];
        }
        if constexpr (TRAIN != 0) {   // rgb-head input (ReLU'd view-layer output) in operand-slot order: k-step 2t + half, slot j = reg & 7
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            KOps o;
            float a0[8], a1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { a0[j] = rv0[8 * half + j]; a1[j] = rv1[8 * half + j]; }
            if constexpr (TRAIN >= 2) { split8<PREC>(a0, o.h0, o.l0); split8<PREC>(a1, o.h1, o.l1); }
            else {
              o.h0 = make_uint4(pack2<PP::F16>(a0[0], a0[1]), pack2<PP::F16>(a0[2], a0[3]), pack2<PP::F16>(a0[4], a0[5]), pack2<PP::F16>(a0[6], a0[7]));
              o.h1 = make_uint4(pack2<PP::F16>(a1[0], a1[1]), pack2<PP::F16>(a1[2], a1[3]), pack2<PP::F16>(a1[4], a1[5]), pack2<PP::F16>(a1[6], a1[7]));
            }
            save_ops(SAVE_RGBIN + 2 * t + half, o);
            save_mask(8, 2 * t + half, nz_nibbles(o.h0), nz_nibbles(o.h1));
          }
        }
      }
      const float bsig = auxt[AUX_BSIG];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        p0[c] = p0[c] + __shfl_xor(p0[c], 32) + auxt[AUX_BRGB + c];
        p1[c] = p1[c] + __shfl_xor(p1[c], 32) + auxt[AUX_BRGB + c];
      }
      sig0 = sig0 + __shfl_xor(sig0, 32) + bsig;
      sig1 = sig1 + __shfl_xor(sig1, 32) + bsig;
#ifndef RNERF_FWD_NORANGE
      {      // an activation outside the operand type's range somewhere along this row's chain (either half of its features); bf16: inf / NaN only
        ovf0 = pk_maxu(ovf0, (uint32_t)__shfl_xor((int)ovf0, 32));
        ovf1 = pk_maxu(ovf1, (uint32_t)__shfl_xor((int)ovf1, 32));
        const float qn = __builtin_nanf("");
        // f16f8 runs with MODE.FP16_OVFL (f8x_mode: the fp8 conversions must clamp) — and that bit also makes every f16 conversion clamp an
        // overflow to 65504 (0x7BFF) instead of producing inf: the watch has to take the clamp value itself as "left the range" there (found
        // in round 5 by the second-pass test: hidden activations of 3e5 rendered finite, wrong colours in the default eval precision)
        // f16f8 also feeds fp8(x) into its W_lo cross term: an activation beyond e4m3's 448 is clamped there (a silent 2^-12 relative error per
        // product that compounds over the layers — measured 1e-2 RGB on hidden kernels x 4, round 6): the watch hands every row with an
        // operand >= 448 (0x5F00) to the range-safe second pass as well
        constexpr uint32_t OVF = PP::F8X ? 0x5F00u : (OVFL_MODE ? 0x7BFFu : (PP::F16 ? 0x7C00u : 0x7F80u));
        if ((ovf0 & 0xFFFFu) >= OVF || (ovf0 >> 16) >= OVF) { p0[0] = p0[1] = p0[2] = qn; sig0 = qn; }
        if ((ovf1 & 0xFFFFu) >= OVF || (ovf1 >> 16) >= OVF) { p1[0] = p1[1] = p1[2] = qn; sig1 = qn; }
      }
#endif
      if constexpr (PP::F16 && !PP::F8X) {      // a weight outside the range of this precision's operand stream (|W| >= 256): fail loudly, not plausibly
        if (auxt[AUX_FLAG] != 0.f) { const float qn = __builtin_nanf(""); p0[0] = p0[1] = p0[2] = qn; p1[0] = p1[1] = p1[2] = qn; sig0 = qn; sig1 = qn; }
      }
      if (h == 0) {
        if constexpr (REDO) {      // only the rows the first pass gave up on: every other row keeps the first pass's bits
          if (row_ok[0] && __builtin_isnan(out_raw[row[0]].x)) out_raw[row[0]] = make_float4(p0[0], p0[1], p0[2], sig0);
          if (row_ok[1] && __builtin_isnan(out_raw[row[1]].x)) out_raw[row[1]] = make_float4(p1[0], p1[1], p1[2], sig1);
        } else {
          if (row_ok[0]) out_raw[row[0]] = make_float4(p0[0], p0[1], p0[2], sig0);
          if (row_ok[1]) out_raw[row[1]] = make_float4(p1[0], p1[1], p1[2], sig1);
        }
      }
      PH(8);
    }
#undef RNERF_PE_KSTEP
#undef SLAB_PREFETCH
#undef SLAB_DONE
    tile = next_tile;
  }
  if constexpr ((dbg & 64) != 0) {
    if (lane == 0) out_raw[blockIdx.x * 4 + wave] = make_float4(prof_tot, prof_dma, prof_bar, prof_n);
  }
  if constexpr ((dbg & 256) != 0) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) out_raw[(blockIdx.x * 4 + wave) * 3 + i] = make_float4(prof_ph[4 * i], prof_ph[4 * i + 1], prof_ph[4 * i + 2], prof_ph[4 * i + 3]);
    }
  }
#undef PH
}

// ------------------------------------------------------------------------------------------------------------------
// T1 (MLP part): backward of the NerfMLP.  Replaces jax.value_and_grad through NerfMLP.__call__ (train.py:164 /
// rnerf/model_utils.py:30-90).
//   dgrad chain (this kernel): the same transposed-chain engine as the forward, with the weights transposed:
//       dX^T[k][row] = W[k][n] dY^T[n][row]   (A = W rows, B = dY in accumulator layout, bf16 hi/lo split -> 3 MFMAs)
//   dY_{l-1} = dX_l * 1[X_l > 0] uses the operands saved by the training forward as the ReLU mask (a saved operand is the
//   ReLU output; 0 <=> inactive).  Every dY_l (bf16 hi parts, the very B operands of the next dgrad layer) is written
//   slot-major for the wgrad kernel:  slot 16*l + s (l = 0..8), 144 + s (l = 9, 8 k-steps), 152 = head grads
//   (half 0: slot 0 = d raw_sigma, slots 1..3 = d raw_rgb).
// ------------------------------------------------------------------------------------------------------------------
constexpr int DY_L9 = 144, DY_HEADS = 152, DY_SLOTS = 153;
// dgrad MFMA passes per product: 22 = (W_hi + W_lo) * dY_hi, i.e. exact weights times bf16-rounded gradients — the same rounding the
// wgrad applies to dY anyway (measured worst gradient error 6.0e-3 of the tensor maximum; 3 passes: 5.1e-3 at +0.3 ms;
// 2 = (dY_hi + dY_lo) * W_hi: 6.8e-3 at the same speed)
#ifndef RNERF_DGRAD_PASSES
#define RNERF_DGRAD_PASSES 22
#endif
constexpr int DGRAD_PASSES = RNERF_DGRAD_PASSES;
// Internal dgrad variant of backward F16 (not an enum rnerf_backward value): ONE MFMA per product — the weights rounded to f16 like the
// gradients — selected when the FORWARD is the single-pass f16 one too.  The step is then one MFMA per product end to end (the arithmetic
// north_star names); behind the f16x3 forward the F16 backward keeps its exact (hi + lo) weights (passes 22): there the rounded weights
// would be the largest error left (gradient 4.9e-6 -> 4.1e-5 of max|g| from the default's on the bench batch), behind the f16 forward they
// change nothing measurable (9.2e-5 both ways) and take 0.19 ms off the dgrad (1.21 -> 1.03 ms; tools/r05/t_dg1.sh).
constexpr int kBwdF16OnePass = 3;
constexpr int kBwdBlocks = 8 * 8 + 8 * 16 * 8;   // (k-steps over n) x (8 input-feature tiles): L9 then L8..L1

// dgrad layer order: index 0 = MFMA layer 9 (Dense_10), 1 = layer 8 (Dense_9), 2..8 = layers 7..1 (Dense_7..Dense_1)
__host__ __device__ constexpr int bwd_dense(int i) { return i == 0 ? 10 : (i == 1 ? 9 : 9 - i); }
__host__ __device__ constexpr int bwd_blocks_before(int i) { return i == 0 ? 0 : 64 + (i - 1) * 128; }

template <int PREC>
__global__ void nerfmlp_pack_bwd_kernel(const float* __restrict__ params, char* __restrict__ packed) {
  using PP = Prec<PREC>;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= kBwdBlocks * 64) return;
  const int blk = gid >> 6, lane = gid & 63;
  int i = 0;
  while (i < 8 && blk >= bwd_blocks_before(i + 1)) ++i;
  const int rel = blk - bwd_blocks_before(i);
  const int s = rel / 8, kt = rel % 8;
  const int d = bwd_dense(i), out_dim = nerf_dense(d).out;
  const int kf = 32 * kt + (lane & 31), h = lane >> 5;        // input feature (row of the Dense kernel), < 256 always
  float w[8];
  for (int j = 0; j < 8; ++j) {
    const int n = prev_feature(s, h, j);
    w[j] = n < out_dim ? params[nerf_koff(d) + kf * out_dim + n] * PP::WSCALE : 0.f;
  }
  uint32_t hi[4], lo[4];
  for (int p = 0; p < 4; ++p) split2<PP::F16>(w[2 * p], w[2 * p + 1], hi[p], lo[p]);
  uint4* dst = (uint4*)(packed + (size_t)blk * PP::NP * 1024);
  dst[lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  if (PP::NP == 2) dst[64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

// dgrad twin of PrevConv: the B operands of k-step S of the NEXT dgrad GEMM = (state / WSCALE [+ d sigma * w_sigma]) * ReLU mask, split
// into 16-bit hi/lo parts, computed pair by pair in the shadow of the current k-step's MFMAs.
template <int PREC, int S, bool NEED_LO, bool ONE = false>
struct GradConv {
  using PP = Prec<PREC>;
  const f32x16& p0;   // m-tile 0 state (accumulator registers of the previous dgrad layer)
  float v1[8];        // m-tile 1 state (from LDS)
  float t0[8], t1[8]; // d raw_sigma * w_sigma of these 8 features for the two m-tiles (0 unless this layer receives d sigma)
  uint32_t w0, w1;    // mask word S>>2 of both m-tiles (all ones: no ReLU)
  uint32_t hi[2][4], lo[2][4];
  float x0, x1;
  __device__ __forceinline__ GradConv(const f32x16& p) : p0(p) {}
  template <int C, int PI>
  __device__ __forceinline__ void chunk() {
    constexpr int mt = PI >> 2, p = PI & 3;
    if constexpr (ONE && mt == 1) return;
    constexpr float INV_SCALE = 1.0f / PP::WSCALE;
    if constexpr (C == 0) {
      const float r0 = mt == 0 ? p0[8 * (S & 1) + 2 * p] : v1[2 * p];
      const float r1 = mt == 0 ? p0[8 * (S & 1) + 2 * p + 1] : v1[2 * p + 1];
      x0 = fmaf(r0, INV_SCALE, mt == 0 ? t0[2 * p] : t1[2 * p]);
      x1 = fmaf(r1, INV_SCALE, mt == 0 ? t0[2 * p + 1] : t1[2 * p + 1]);
    } else if constexpr (C == 1) {
      const uint32_t w = mt == 0 ? w0 : w1;      // v_bfe_i32 spreads the flag into a 0 / ~0 word, v_and applies it: 2 ops per value
      constexpr int pos0 = 8 * (S & 3) + p, pos1 = pos0 + 4;
      uint32_t m0, m1;                           // (asm: hipcc rewrites the builtin form back into v_and + v_cmp + v_cndmask)
      asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m0) : "v"(w), "n"(pos0));
      asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m1) : "v"(w), "n"(pos1));
      x0 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x0) & m0);
      x1 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x1) & m1);
    } else if constexpr (C == 2) {
      hi[mt][p] = pack2<PP::F16>(x0, x1);
    } else if constexpr (C == 3) {
      if constexpr (NEED_LO) {
        float ha, hb;
        unpack2<PP::F16>(hi[mt][p], ha, hb);
        x0 -= ha; x1 -= hb;
      }
    } else {
      if constexpr (NEED_LO) lo[mt][p] = pack2<PP::F16>(x0, x1);
      else lo[mt][p] = 0;
    }
  }
  __device__ __forceinline__ KOpsT<ONE> result() const {
    KOpsT<ONE> o;
    o.h0 = make_uint4(hi[0][0], hi[0][1], hi[0][2], hi[0][3]); o.l0 = make_uint4(lo[0][0], lo[0][1], lo[0][2], lo[0][3]);
    if constexpr (ONE) { o.h1 = o.h0; o.l1 = o.l0; }
    else { o.h1 = make_uint4(hi[1][0], hi[1][1], hi[1][2], hi[1][3]); o.l1 = make_uint4(lo[1][0], lo[1][1], lo[1][2], lo[1][3]); }
    return o;
  }
};

// Backward arithmetic (enum rnerf_backward): BF16 = gradients rounded to bf16 (8-bit significand), one 16-bit part per operand;
// F16 / F16X2 = every row's gradient chain is normalised by a power of two m_row >= max |d raw[row]| (the chain is linear in d raw, so
// the normalised values sit in f16's range whatever the loss scale), operands are f16 (11-bit significand) or f16 hi + lo (22 bits);
// the wgrad multiplies the saved activations by m_row / m_ref again (m_ref = max m_row, collected with an atomic).
template <int BWD> struct Bwd {
  static constexpr bool F16 = BWD != RNERF_BWD_BF16;
  static constexpr bool LO8 = BWD == RNERF_BWD_F16X3_LO8;                                 // the lo plane is stored as e4m3 bytes (lo8_pack)
  static constexpr int NP = (BWD == RNERF_BWD_F16X2 || LO8) ? 2 : 1;                      // parts stored per gradient / consumed per activation
  static constexpr int PREC = F16 ? RNERF_PREC_F16X3 : RNERF_PREC_BF16X3;                 // operand type + weight split of the dgrad MFMAs
  static constexpr int PASSES = (BWD == RNERF_BWD_F16X2 || LO8) ? 3 : (BWD == RNERF_BWD_F16 ? 22 : (BWD == kBwdF16OnePass ? 1 : DGRAD_PASSES));
  static constexpr bool NEED_LO = PASSES != 22 && PASSES != 1;
};
// dy buffer: uint4[DY_SLOTS * NP][R][2] operand planes (hi, then lo), then float row_scale[R] and uint32 m_ref bits (F16 modes)
__host__ __device__ constexpr size_t dy_plane_uint4(long long R, int np) { return (size_t)DY_SLOTS * np * (size_t)R * 2; }
__host__ __device__ constexpr size_t dy_addr(int q, long long t32, int m, int h) { return (((size_t)t32 * DY_SLOTS + q) * 32 + m) * 2 + h; }     // tile-major, see sv_addr

// the dY stores of one dgrad k-step one at a time (see SaveParts)
template <bool LO8, int NP, bool ONE>
struct DyParts {
  uint4* dy; long long save_rows, t32_0; int q, m, h; const KOpsT<ONE>& o;
  template <int K> __device__ __forceinline__ void part() const {
#ifndef RNERF_DGRAD_NOSTORE
    uint4* dst = dy + dy_addr(q, t32_0, m, h);
    if constexpr (K == 0) stream_store(dst, o.h0);
    if constexpr (K == 1 && !ONE) stream_store(dst + (size_t)DY_SLOTS * 64, o.h1);
    if constexpr (LO8) {
      uint2* dl = (uint2*)(dy + dy_plane_uint4(save_rows, 1)) + dy_addr(q, t32_0, m, h);
      if constexpr (K == 2) stream_store8(dl, lo8_pack<true>(o.l0, LO8_SCALE_D));
      if constexpr (K == 3 && !ONE) stream_store8(dl + (size_t)DY_SLOTS * 64, lo8_pack<true>(o.l1, LO8_SCALE_D));
    } else if constexpr (NP == 2) {
      uint4* dl = dst + dy_plane_uint4(save_rows, 1);
      if constexpr (K == 2) stream_store(dl, o.l0);
      if constexpr (K == 3 && !ONE) stream_store(dl + (size_t)DY_SLOTS * 64, o.l1);
    }
#endif
  }
};

template <int BWD, bool ONE = false>
__global__ void __launch_bounds__(256, 1)
nerfmlp_dgrad_kernel(const char* __restrict__ packed_bwd, const float* __restrict__ fwd_aux, const uint4* __restrict__ saved,
                     long long save_rows, const float4* __restrict__ d_raw, long long total_rows, int n_tiles,
                     uint4* __restrict__ dy) {
  // ONE: 128-row tiles, one m-tile per wave (see nerfmlp_fwd_kernel); n_tiles then counts 128-row tiles
  using BW = Bwd<BWD>;
  using KOps = KOpsT<ONE>;
  constexpr int PREC = BW::PREC;
  constexpr int DGP = BW::PASSES;
  constexpr bool NEED_LO = BW::NEED_LO;
  using PP = Prec<PREC>;
  constexpr int SLAB = PP::SLAB;
  constexpr int WROWS = ONE ? 32 : 64, TROWS = 4 * WROWS;
  constexpr float INV_SCALE = 1.0f / PP::WSCALE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  float4* __restrict__ st1 = (float4*)(smem + 2 * SLAB + wave * 32768) + lane;
  int buf = 0;
  size_t off = 0;
  NoWork nowork;
  constexpr int dbg = 0;

  if ((int)blockIdx.x < n_tiles) { issue_slab<SLAB>(packed_bwd, 0u, wave, lane); off = SLAB; }
  slab_wait_dma();
  __syncthreads();

  float mref = 0.f;
  // -DRNERF_DGRAD_PROFILE (ablation build): clocks per phase of a tile, written over the head of the dy buffer (results are garbage)
  //   0 rows + head gradients  1 first k-step operands of a layer (grad_ops(0), masks)  2 k-steps  3 layer end  4 dY_0 record
#ifdef RNERF_DGRAD_PROFILE
  float dprof[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  unsigned long long dlast = __builtin_amdgcn_s_memtime();
#define DPH(K) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); dprof[K] += (float)(t_ - dlast); dlast = t_; } while (0)
#else
#define DPH(K) do {} while (0)
#endif
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const bool has_next_tile = tile + (int)gridDim.x < n_tiles;
    const long long srow0 = (long long)tile * TROWS + wave * WROWS + m;
    const long long t32_0 = (long long)tile * (TROWS / 32) + wave * (WROWS / 32);
    float4 g[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (ONE && mt == 1) { g[1] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
      const long long row = srow0 + 32 * mt;
      g[mt] = row < total_rows ? d_raw[row] : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (BW::F16) {          // row normalisation
        const float mx = fmaxf(fmaxf(fabsf(g[mt].x), fabsf(g[mt].y)), fmaxf(fabsf(g[mt].z), fabsf(g[mt].w)));
        const uint32_t ex = __builtin_bit_cast(uint32_t, mx) >> 23;
        // m_row = 2^(exponent(max |g|) - 5): 32 <= max |g / m_row| < 64 — 2^10 of headroom below f16's 65504 for growth along the chain,
        // and every value above 0.2 % of the row's maximum keeps a NORMAL f16 lo part.  Rows with no (or a subnormal) gradient
        // contribute nothing: m_row = 0 (a non-finite d raw still yields NaN gradients: NaN * 0).
        const bool okx = ex >= 8 && ex <= 250;
        const float m_row = okx ? __builtin_bit_cast(float, (ex - 5) << 23) : 0.0f;
        const float inv = okx ? __builtin_bit_cast(float, (259 - ex) << 23) : 0.0f;
        g[mt].x *= inv; g[mt].y *= inv; g[mt].z *= inv; g[mt].w *= inv;
        float* rs = (float*)(dy + dy_plane_uint4(save_rows, BW::NP));
        if (h == 0) rs[row] = m_row;
        mref = fmaxf(mref, m_row);
      }
    }
    f32x16 acc0[8], acc1[8], prev0[8];

    auto dy_store = [&](int q, const KOps& o) {
#ifndef RNERF_DGRAD_NOSTORE   /* profiling ablation */
      uint4* dst = dy + dy_addr(q, t32_0, m, h);
      stream_store(dst, o.h0);
      if constexpr (!ONE) stream_store(dst + (size_t)DY_SLOTS * 64, o.h1);
      if constexpr (BW::LO8) {
        uint2* dl = (uint2*)(dy + dy_plane_uint4(save_rows, 1)) + dy_addr(q, t32_0, m, h);
        stream_store8(dl, lo8_pack<true>(o.l0, LO8_SCALE_D));
        if constexpr (!ONE) stream_store8(dl + (size_t)DY_SLOTS * 64, lo8_pack<true>(o.l1, LO8_SCALE_D));
      } else if constexpr (BW::NP == 2) {
        uint4* dl = dst + dy_plane_uint4(save_rows, 1);
        stream_store(dl, o.l0);
        if constexpr (!ONE) stream_store(dl + (size_t)DY_SLOTS * 64, o.l1);
      }
#endif
    };
    // ReLU masks: one uint4 of non-zero flags per (row, half) per layer (SAVE_MASK), fetched one layer ahead
    auto mask_at = [&](int set, uint4& a, uint4& b) {
      const uint32_t* src = (const uint32_t*)(saved + sv_mask0(save_rows)) + sv_mask_addr(set * 4, t32_0, m, h);
      constexpr size_t ws = 64, mt1 = 36 * 64;          // next mask word of the set / the wave's second 32-row tile
#define RNERF_NTL(P) __builtin_nontemporal_load(P)
      a = make_uint4(RNERF_NTL(src), RNERF_NTL(src + ws), RNERF_NTL(src + 2 * ws), RNERF_NTL(src + 3 * ws));
      if constexpr (ONE) b = make_uint4(0, 0, 0, 0);
      else b = make_uint4(RNERF_NTL(src + mt1), RNERF_NTL(src + ws + mt1), RNERF_NTL(src + 2 * ws + mt1), RNERF_NTL(src + 3 * ws + mt1));
#undef RNERF_NTL
    };
    // operands of k-step s from the state (prev0 / st1): x = (state + dsig * wadd) * 1[mask != 0]
    auto grad_ops = [&](int s, bool use_mask, const uint4 mk0, const uint4 mk1, const float* __restrict__ wadd) -> KOps {
      float r1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if constexpr (!ONE) {
        const float4 u0 = st1[((s >> 1) * 4 + 2 * (s & 1)) * 64], u1 = st1[((s >> 1) * 4 + 2 * (s & 1) + 1) * 64];
        r1[0] = u0.x; r1[1] = u0.y; r1[2] = u0.z; r1[3] = u0.w; r1[4] = u1.x; r1[5] = u1.y; r1[6] = u1.z; r1[7] = u1.w;
      }
      const uint32_t w0 = (s >> 2) == 0 ? mk0.x : ((s >> 2) == 1 ? mk0.y : ((s >> 2) == 2 ? mk0.z : mk0.w));
      const uint32_t w1 = (s >> 2) == 0 ? mk1.x : ((s >> 2) == 1 ? mk1.y : ((s >> 2) == 2 ? mk1.z : mk1.w));
      float x0[8], x1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = prev0[s >> 1][8 * (s & 1) + j] * INV_SCALE, b = r1[j] * INV_SCALE;
        if (wadd) { const float w = wadd[16 * s + 8 * (j >> 2) + 4 * h + (j & 3)]; a = fmaf(g[0].w, w, a); b = fmaf(g[1].w, w, b); }
        if (use_mask) {
          const uint32_t bit = 1u << (8 * (s & 3) + (j >> 1) + 4 * (j & 1));
          a = (w0 & bit) ? a : 0.f;
          b = (w1 & bit) ? b : 0.f;
        }
        x0[j] = a; x1[j] = b;
      }
      KOps o;
      if constexpr (ONE) { split8<PREC>(x0, o.h0, o.l0); if constexpr (!(NEED_LO || BW::NP == 2)) o.l0 = make_uint4(0, 0, 0, 0); o.h1 = o.h0; o.l1 = o.l0; }
      else if constexpr (NEED_LO || BW::NP == 2) { split8<PREC>(x0, o.h0, o.l0); split8<PREC>(x1, o.h1, o.l1); }
      else {
        o.h0 = make_uint4(pack2<PP::F16>(x0[0], x0[1]), pack2<PP::F16>(x0[2], x0[3]), pack2<PP::F16>(x0[4], x0[5]), pack2<PP::F16>(x0[6], x0[7]));
        o.h1 = make_uint4(pack2<PP::F16>(x1[0], x1[1]), pack2<PP::F16>(x1[2], x1[3]), pack2<PP::F16>(x1[4], x1[5]), pack2<PP::F16>(x1[6], x1[7]));
        o.l0 = make_uint4(0, 0, 0, 0); o.l1 = o.l0;
      }
      return o;
    };
    auto layer_end = [&]() {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        prev0[t] = acc0[t];
        if constexpr (!ONE) {
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
            st1[(t * 4 + rq) * 64] = make_float4(acc1[t][4 * rq], acc1[t][4 * rq + 1], acc1[t][4 * rq + 2], acc1[t][4 * rq + 3]);
        }
      }
    };
#define SLAB_PREFETCH(DO_NEXT)                                                                                       \
  do { if (DO_NEXT) { issue_slab<SLAB>(packed_bwd + off, (unsigned)((buf ^ 1) * SLAB), wave, lane); off += SLAB; } } while (0)
#define SLAB_DONE() do { slab_wait_dma(); __syncthreads(); buf ^= 1; } while (0)

    // ---- head gradients: record them for the wgrad kernel, push d raw_rgb through the rgb head (Dense_11) into the state
    {
      KOps o;
      float hz0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hz1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (h == 0) { hz0[0] = g[0].w; hz0[1] = g[0].x; hz0[2] = g[0].y; hz0[3] = g[0].z; hz1[0] = g[1].w; hz1[1] = g[1].x; hz1[2] = g[1].y; hz1[3] = g[1].z; }
      split8<PREC>(hz0, o.h0, o.l0);
      if constexpr (ONE) { o.h1 = o.h0; o.l1 = o.l0; } else split8<PREC>(hz1, o.h1, o.l1);
      dy_store(DY_HEADS, o);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float a = 0.f, b = 0.f;
          if (t < 4) {
            const int n = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float wr = fwd_aux[AUX_WRGB + n], wg = fwd_aux[AUX_WRGB + 128 + n], wb = fwd_aux[AUX_WRGB + 256 + n];
            a = (g[0].x * wr + g[0].y * wg + g[0].z * wb) * PP::WSCALE;       // the state is kept in the accumulators' units (x WSCALE)
            b = (g[1].x * wr + g[1].y * wg + g[1].z * wb) * PP::WSCALE;
          }
          acc0[t][r] = a;
          if constexpr (!ONE) acc1[t][r] = b;
        }
      DPH(0);
      layer_end();
      DPH(3);
    }

    // ---- dgrad of MFMA layer 9 (Dense_10): k-steps over its 128 outputs; state masked by the saved rgb-head input
    {
      uint4 ma, mb;
      mask_at(8, ma, mb);
      KOps cur = grad_ops(0, true, ma, mb, nullptr);
      DPH(1);
#define RNERF_DG_KSTEP(S, NSTEPS, SLOT0, PREFETCH_STMT)                                                                       \
      {                                                                                                                         \
        const DyParts<BW::LO8, BW::NP, ONE> dp{dy, save_rows, t32_0, (SLOT0) + S, m, h, cur};                                    \
        if constexpr (!RNERF_SPREAD_STORES) dy_store((SLOT0) + S, cur);                                                         \
        auto dma = [&]() { PREFETCH_STMT; };   /* issued after the first two tiles' MFMAs (an LDS-DMA instruction costs ~100 issue cycles) */ \
        if constexpr (S + 1 < NSTEPS) {                                                                                         \
          GradConv<PREC, S + 1, NEED_LO, ONE> cv(prev0[(S + 1) >> 1]);                                                                   \
          if constexpr (!ONE) {                                                                                                 \
            const float4 u0 = st1[(((S + 1) >> 1) * 4 + 2 * ((S + 1) & 1)) * 64], u1 = st1[(((S + 1) >> 1) * 4 + 2 * ((S + 1) & 1) + 1) * 64]; \
            cv.v1[0] = u0.x; cv.v1[1] = u0.y; cv.v1[2] = u0.z; cv.v1[3] = u0.w; cv.v1[4] = u1.x; cv.v1[5] = u1.y; cv.v1[6] = u1.z; cv.v1[7] = u1.w; \
          }                                                                                                                     \
          if (wadd) {   /* ONE test per k-step (per value it was 8 branches), two vector loads; only layer 7 takes it */         \
            const float4 wa = *(const float4*)(wadd + 16 * (S + 1) + 4 * h), wb = *(const float4*)(wadd + 16 * (S + 1) + 8 + 4 * h); \
            const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};                                               \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) { cv.t0[j] = g[0].w * wv[j]; cv.t1[j] = g[1].w * wv[j]; }             \
          } else {                                                                                                              \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) { cv.t0[j] = 0.f; cv.t1[j] = 0.f; }                                   \
          }                                                                                                                     \
          cv.w0 = ((S + 1) >> 2) == 0 ? ma.x : (((S + 1) >> 2) == 1 ? ma.y : (((S + 1) >> 2) == 2 ? ma.z : ma.w));             \
          cv.w1 = ((S + 1) >> 2) == 0 ? mb.x : (((S + 1) >> 2) == 1 ? mb.y : (((S + 1) >> 2) == 2 ? mb.z : mb.w));             \
          if constexpr (RNERF_SPREAD_STORES) kstep_mfma<PREC, 8, 0, S == 0, GradConv<PREC, S + 1, NEED_LO, ONE>, false, decltype(dma), DGP, KOps, DyParts<BW::LO8, BW::NP, ONE>>(acc0, acc1, cur, smem + buf * SLAB, lane, cv, dma, dp); else \
          kstep_mfma<PREC, 8, 0, S == 0, GradConv<PREC, S + 1, NEED_LO, ONE>, false, decltype(dma), DGP>(acc0, acc1, cur, smem + buf * SLAB, lane, cv, dma); \
          cur = cv.result();                                                                                                    \
        } else {                                                                                                                \
          if constexpr (RNERF_SPREAD_STORES) kstep_mfma<PREC, 8, 0, false, NoWork, false, decltype(dma), DGP, KOps, DyParts<BW::LO8, BW::NP, ONE>>(acc0, acc1, cur, smem + buf * SLAB, lane, nowork, dma, dp); else \
          kstep_mfma<PREC, 8, 0, false, NoWork, false, decltype(dma), DGP>(acc0, acc1, cur, smem + buf * SLAB, lane, nowork, dma);  \
        }                                                                                                                       \
        SLAB_DONE();                                                                                                            \
      }
      {
        const float* __restrict__ wadd = nullptr;
        RNERF_DG_KSTEP(0, 8, DY_L9, SLAB_PREFETCH(true)) RNERF_DG_KSTEP(1, 8, DY_L9, SLAB_PREFETCH(true))
        RNERF_DG_KSTEP(2, 8, DY_L9, SLAB_PREFETCH(true)) RNERF_DG_KSTEP(3, 8, DY_L9, SLAB_PREFETCH(true))
        RNERF_DG_KSTEP(4, 8, DY_L9, SLAB_PREFETCH(true)) RNERF_DG_KSTEP(5, 8, DY_L9, SLAB_PREFETCH(true))
        RNERF_DG_KSTEP(6, 8, DY_L9, SLAB_PREFETCH(true)) RNERF_DG_KSTEP(7, 8, DY_L9, SLAB_PREFETCH(true))
      }
      DPH(2);
      layer_end();
      DPH(3);
    }

    // ---- dgrad of MFMA layers 8..1: dY_l = (dX_{l+1} [+ d sigma * w_sigma for l = 7]) * 1[X_{l+1} > 0]  (no mask for the bottleneck l = 8)
    uint4 nma = make_uint4(0, 0, 0, 0), nmb = nma;      // mask of the NEXT iteration (set l-1), loaded a whole layer ahead
#pragma unroll 1
    for (int l = 8; l >= 1; --l) {
      const bool use_mask = l != 8;
      const float* __restrict__ wadd = (l == 7) ? fwd_aux + AUX_WSIG : nullptr;
      uint4 ma = nma, mb = nmb;                  // set l = the ReLU mask of the input of layer l+1
      if (!use_mask) { ma = make_uint4(~0u, ~0u, ~0u, ~0u); mb = ma; }
      mask_at(l - 1, nma, nmb);
      KOps cur = grad_ops(0, true, ma, mb, wadd);
      const int slot0 = 16 * l;
      DPH(1);
#define RNERF_DG_PF(S) do { if ((S) == 15 && l == 1) { if (has_next_tile) off = 0; SLAB_PREFETCH(has_next_tile); } else SLAB_PREFETCH(true); } while (0)
      RNERF_DG_KSTEP(0, 16, slot0, RNERF_DG_PF(0)) RNERF_DG_KSTEP(1, 16, slot0, RNERF_DG_PF(1)) RNERF_DG_KSTEP(2, 16, slot0, RNERF_DG_PF(2))
      RNERF_DG_KSTEP(3, 16, slot0, RNERF_DG_PF(3)) RNERF_DG_KSTEP(4, 16, slot0, RNERF_DG_PF(4)) RNERF_DG_KSTEP(5, 16, slot0, RNERF_DG_PF(5))
      RNERF_DG_KSTEP(6, 16, slot0, RNERF_DG_PF(6)) RNERF_DG_KSTEP(7, 16, slot0, RNERF_DG_PF(7)) RNERF_DG_KSTEP(8, 16, slot0, RNERF_DG_PF(8))
      RNERF_DG_KSTEP(9, 16, slot0, RNERF_DG_PF(9)) RNERF_DG_KSTEP(10, 16, slot0, RNERF_DG_PF(10)) RNERF_DG_KSTEP(11, 16, slot0, RNERF_DG_PF(11))
      RNERF_DG_KSTEP(12, 16, slot0, RNERF_DG_PF(12)) RNERF_DG_KSTEP(13, 16, slot0, RNERF_DG_PF(13)) RNERF_DG_KSTEP(14, 16, slot0, RNERF_DG_PF(14))
      RNERF_DG_KSTEP(15, 16, slot0, RNERF_DG_PF(15))
#undef RNERF_DG_PF
      DPH(2);
      layer_end();
      DPH(3);
    }
#undef RNERF_DG_KSTEP

    // ---- dY_0 = dX_1 * 1[X_1 > 0]: only recorded (layer 0's inputs are constants)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const KOps o = grad_ops(s, true, nma, nmb, nullptr);
      dy_store(s, o);
    }
    DPH(4);
#undef SLAB_PREFETCH
#undef SLAB_DONE
  }
  if constexpr (BW::F16) {      // m_ref = max m_row over all rows (positive floats order like their bit patterns)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mref = fmaxf(mref, __shfl_xor(mref, o));
    if (lane == 0 && mref > 0.f)
      atomicMax((unsigned int*)((float*)(dy + dy_plane_uint4(save_rows, BW::NP)) + save_rows), __builtin_bit_cast(unsigned int, mref));
  }
#ifdef RNERF_DGRAD_PROFILE
  if (lane == 0) {
    float4* pr = (float4*)dy + (blockIdx.x * 4 + wave) * 2;
    pr[0] = make_float4(dprof[0], dprof[1], dprof[2], dprof[3]);
    pr[1] = make_float4(dprof[4], dprof[5], dprof[6], dprof[7]);
  }
#endif
#undef DPH
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad: dW[k][n] = sum_rows X[row][k] * dY[row][n] for one (X slots, dY slots) job.  The saved tensors hold, per lane, 8
// FEATURES of one ROW (the forward's B-operand layout); an MFMA contracts over the 8 values a lane holds, so both operands
// must first be transposed to "lane = feature, slots = rows".  The transposition is itself done on the matrix cores:
//     D = X_frag(k-step 2t) * I_lo + X_frag(k-step 2t+1) * I_hi        (A = the saved operand, B = a shifted identity)
// puts feature 16a + 8h + j of the two k-steps in output LANE 16a+8h+j and the wave's 32 rows in the 16 accumulator
// registers (exact: one non-zero product per output) -> repacked to bf16 they are wgrad operands.  Each wave transposes
// its own 32 rows, publishes the fragments in LDS, and after a barrier accumulates its share of the dW tiles over the 128
// rows of the workgroup.  Partials per workgroup are summed (and the slot permutation undone) by wgrad_reduce_kernel.
// ------------------------------------------------------------------------------------------------------------------
template <bool F16>
__device__ __forceinline__ uint4 shifted_identity(int lane, int a) {   // B operand: 1 at slot (h, j) of column 16a + 8h + j
  const int col = lane & 31, h = lane >> 5;
  uint32_t w[4] = {0, 0, 0, 0};
  if ((col >> 4) == a && ((col >> 3) & 1) == h) { const int j = col & 7; w[j >> 1] = (F16 ? 0x3C00u : 0x3F80u) << (16 * (j & 1)); }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ void pack_rows_bf16(const f32x16& d, uint4& u0, uint4& u1) {
  u0 = make_uint4(pack2<false>(d[0], d[1]), pack2<false>(d[2], d[3]), pack2<false>(d[4], d[5]), pack2<false>(d[6], d[7]));
  u1 = make_uint4(pack2<false>(d[8], d[9]), pack2<false>(d[10], d[11]), pack2<false>(d[12], d[13]), pack2<false>(d[14], d[15]));
}

// job description for the reduction: where the slot-ordered partial rows/columns land in the flat fp32 gradient buffer
struct WgradJob {
  int dense;      // Dense_k index in the flat parameter buffer
  int xkind;      // 0 = position encoding slots, 1 = previous-layer slots, 2 = view encoding slots
  int row_off;    // kernel row offset of this X block (256 for the concat parts)
  int dkind;      // 0 = layer outputs (prev_feature order), 1 = sigma head (slot 0 of the head grads), 2 = rgb head (slots 1..3)
  int KT, NT, write_bias;
  // merged jobs of the transpose-read kernel (an operand stream that two Dense blocks share is read once): the k-tiles from KTa on are
  // a second X segment (xkind2 / row_off2: the concat rows of the same Dense), the n-tiles from NTa on a second dY segment (dense2 /
  // dkind2: the sigma head beside Dense_9).  Single-segment jobs have KTa = KT, NTa = NT.
  int KTa, xkind2, row_off2, NTa, dense2, dkind2;
  int bias_sigma;  // rgb-head job: column 0 of its bias row is the sigma head's bias gradient (Dense_8), written from here
};
// all jobs of one NerfMLP in ONE launch: workgroups [wg0[j], wg0[j+1]) belong to job j
struct WgradTable {
  int n;
  int qx[16], KSx[16], qd[16], KSd[16], wg0[17];
  int qx2[16], qd2[16], KSd2[16];         // slot bases of the second segments, k-steps of the second dY segment
  long long poff[16], pboff[16];          // float offsets of the job's partial blocks in the workspace
  WgradJob job[16];
};

// One job shape (KT k-tiles of X, KSd k-steps = ceil(KSd/2) n-tiles of dY) is a compile-time instance: with run-time shapes the
// compiler guards every load with a branch and a vmcnt(0), which serialises the 32 loads of a chunk (measured: 21 us per chunk).
// Workgroup = 8 waves = 2 per SIMD (<= 256 registers each): waves 0..3 fetch + transpose the X slots of row groups 0..3, waves
// 4..7 the dY slots; in the accumulate phase wave w owns k-tile w (all n-tiles).  One wave's load waits / packing VALU overlap the
// partner wave's MFMAs, and the prefetch registers per wave halve (the 4-wave form with 2 k-tiles per wave took 1.92 ms, this 1.42).
template <bool X_F16, int KT, int KSd>
__device__ __forceinline__ void wgrad_body(const uint4* __restrict__ saved, const uint4* __restrict__ dy, long long R, long long total_rows,
                                            int n_chunks, float* __restrict__ pg, float* __restrict__ pbias, int qx, int qd, int g, int G,
                                            char* smem) {
  constexpr int NT = (KSd + 1) / 2;
  constexpr int NOP = 2 * (KT > NT ? KT : NT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, rw = wave & 3;
  const int m = lane & 31, h = lane >> 5;
  uint4* myT = (uint4*)(smem + rw * 32768) + lane + (grp ? 16 * 64 : 0);   // slot*64: X^T fragments 0..15, dY^T fragments 16..31
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const uint4 il = grp ? shifted_identity<false>(lane, 0) : shifted_identity<X_F16>(lane, 0);
  const uint4 ih = grp ? shifted_identity<false>(lane, 1) : shifted_identity<X_F16>(lane, 1);
  const uint32_t one2 = (m == 0) ? 0x3F803F80u : 0u;
  const uint4 ones = make_uint4(one2, one2, one2, one2);
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  f32x16 acc[NT], accb = zero;
#pragma unroll
  for (int b = 0; b < NT; ++b) acc[b] = zero;
  uint4 opr[NOP];
  auto load_chunk = [&](int chunk) {
    const long long t32 = (long long)chunk * 4 + rw;
    if (grp == 0) {
      const uint4* base = saved + sv_addr(qx, t32, m, h);
#pragma unroll
      for (int t = 0; t < 2 * KT; ++t) opr[t] = stream_load(base + t * 64);
    } else {
      const uint4* base = dy + dy_addr(qd, t32, m, h);
#pragma unroll
      for (int t = 0; t < 2 * NT; ++t) opr[t] = t < KSd ? stream_load(base + t * 64) : z4;
    }
  };
  if (g < n_chunks) load_chunk(g);
  for (int chunk = g; chunk < n_chunks; chunk += G) {
    const bool ok = (long long)chunk * 128 + rw * 32 + m < total_rows;
    if (grp == 0) {
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        f32x16 d = mfma16<X_F16>(opr[2 * t], il, zero);
        d = mfma16<X_F16>(opr[2 * t + 1], ih, d);
        uint4 u0, u1;
        pack_rows_bf16(d, u0, u1);
        myT[(2 * t) * 64] = u0; myT[(2 * t + 1) * 64] = u1;
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const uint4 a0 = ok ? opr[2 * t] : z4, a1 = ok ? opr[2 * t + 1] : z4;   // padded rows carry replayed data: they must not contribute
        f32x16 d = mfma16<false>(a0, il, zero);
        d = mfma16<false>(a1, ih, d);
        uint4 u0, u1;
        pack_rows_bf16(d, u0, u1);
        myT[(2 * t) * 64] = u0; myT[(2 * t + 1) * 64] = u1;
      }
    }
    __syncthreads();
    if (chunk + G < n_chunks) load_chunk(chunk + G);
    if (wave < KT || wave < NT) {
#pragma unroll 1
      for (int v = 0; v < 4; ++v) {
        const uint4* T = (const uint4*)(smem + v * 32768) + lane;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (wave < KT) {
            const uint4 a = T[(2 * wave + u) * 64];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16<false>(a, T[(16 + 2 * nt + u) * 64], acc[nt]);
          }
          if (wave < NT) accb = mfma16<false>(ones, T[(16 + 2 * wave + u) * 64], accb);
        }
      }
    }
    __syncthreads();
  }
  constexpr size_t ldn = (size_t)NT * 32;
  if (wave < KT) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        pg[(size_t)(wave * 32 + i) * ldn + nt * 32 + m] = acc[nt][r];
      }
    }
  }
  if (wave < NT && h == 0) pbias[wave * 32 + m] = accb[0];
}

// ---- wgrad of the row-normalised f16 modes (RNERF_BWD_F16: NP = 1, RNERF_BWD_F16X2: NP = 2): no transposition phase ------------------------
// The saved tensors are row-major per slot ([row][half][8 features] = 32 B per row); an MFMA operand wants, per lane, 8 ROWS of one
// feature.  gfx950's LDS transpose read (ds_read_b64_tr_b16: within a 16-lane group, destination lane i, element j receives element
// (i & 3) of the 8 bytes addressed by source lane 4 j + (i >> 2); probed by tools/ubench/tr_read.hip) does that on the way out of LDS, so
//   * the operand slots travel HBM -> LDS by DMA (global_load_lds_dwordx4, no registers), 16 rows (one MFMA k-step) at a time, through a
//     ring of 4 (hi + lo) or 8 steps: all but one in flight while one is consumed — the kernel is paced by HBM (intensity 192 flop/B in the hi + lo mode);
//   * one DMA instruction builds one 1 KiB "block" = 16 rows x 32 features (two slots: a k-tile of X or an n-tile of dY) as
//     [row r][slot parity a][32 B]: lane L = 4 r + 2 a + h fetches the 16 B (slot 2 t + a, row r, half h).  A transpose read of a
//     32-lane half then covers 256 contiguous bytes: conflict-free;
//   * A operand of k-tile t: lane (m = l & 31, kh = l >> 5) reads rows 8 kh .. 8 kh + 7 of feature position m (= 16 a + p, p the position in
//     the slot's row — the order wgrad_reduce_kernel already undoes) with two transpose reads; B operands of dY the same way;
//   * the per-row scale m_row / m_ref (a power of two <= 1) multiplies the X operands as packed f16 (exact above the subnormal range; rows
//     that far below the largest carry nothing, as before); the bias row sum_rows m_row dY^[row][n] is an MFMA whose A operand holds the
//     scales in row m = 0;
//   * 8 waves = WK x WN, each accumulating TK x TN output tiles over all rows of its workgroup: per k-step TK + TN operand fetches feed
//     TK TN MFMAs (x 3 in the hi + lo mode).
// (Round 2's first form transposed on the matrix cores like wgrad_body below, through registers and LDS writes: 3.5 ms for the hi + lo
// mode on the bench workload against 2.2 ms for this one, which runs at ~5.2 TB/s of operand stream.)
// NT == 9 (Dense_9 + the sigma head): the 8 x 8 split, and the ninth n-tile as ONE extra tile per wave (k-tile 2 wk + wn, one of its own)
template <int KT, int NT> struct WgTrShape {
  static constexpr int EXTRA = NT == 9 ? 1 : 0, NTR = NT - EXTRA;
  static constexpr int WK = KT == 10 ? 2 : (KT == 9 ? 3 : (KT >= 8 ? (NTR >= 4 ? 4 : 8) : (KT >= 4 ? 4 : KT)));
  static constexpr int WN = (8 / WK) < NTR ? (8 / WK) : NTR;
  static constexpr int TK = KT / WK, TN = NTR / WN;
  static_assert(WK * TK == KT && WN * TN == NTR && WK * WN <= 8 && (!EXTRA || (TK == 2 && WN == 2)), "tile split");
};
#ifndef RNERF_WGTR_LATE_DMA
#define RNERF_WGTR_LATE_DMA 0      /* round 6: launched ALONE it pays (tools/r06/ab_wgrad.py: f16x3 wgrad 2.008 / 2.010 -> 1.987 / 1.973 ms), in the STEP — the next batch's \
                                      march co-resident on the same SIMDs — it does not (tools/r06/ab_step.py, three alternating pairs: 6.352 / 6.352 / 6.355 ms with it, 6.348 / 6.319 / 6.319 without): off */
#endif
#ifndef RNERF_WGTR_NCH
#define RNERF_WGTR_NCH 2      /* n-tiles of B fragments fetched at a time when the A side is resident */
#endif
template <int NP> constexpr int wgtr_nbuf() { return NP == 2 ? 4 : 8; }       // ring depth: what fits 160 KiB
template <int NP> constexpr int wgtr_lds_bytes() { return wgtr_nbuf<NP>() * ((10 + 8) * NP + 1) * 1024; }       // largest job: 10 k-tiles + 8 n-tiles

#if defined(RNERF_WGTR_ABL) && (RNERF_WGTR_ABL & 2)   /* profiling ablation: no LDS operand reads */
__device__ __forceinline__ half8 tr_read8(const char* p) { const _Float16 v = (_Float16)(float)((size_t)p & 7); return half8{v, v, v, v, v, v, v, v}; }
#else
__device__ __forceinline__ half8 tr_read8(const char* p) {      // rows k .. k+3 at p, rows k+4 .. k+7 at p + 256 (4 rows x 64 B)
  typedef short short4v __attribute__((ext_vector_type(4)));
  typedef short short8v __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) short4v lds_short4;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)p);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(p + 256));
  return __builtin_bit_cast(half8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
#endif
#if defined(RNERF_WGTR_ABL) && (RNERF_WGTR_ABL & 1)   /* profiling ablation: the wgrad without its MFMAs (results are garbage) */
__device__ __forceinline__ f32x16 mfma_h8(const half8 a, const half8 b, f32x16 c) { c[0] += (float)a[0] * (float)b[0]; return c; }
#else
__device__ __forceinline__ f32x16 mfma_h8(const half8 a, const half8 b, const f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

struct WgTrSegs { int qx, KTa, qx2, qd, NTa, KSd, qd2, KSd2; };    // slot bases / extents of the (up to two) X and dY segments of a job

template <int NP, int KT, int NT>
__device__ __forceinline__ void wgrad_body_tr(const uint4* __restrict__ saved, const uint4* __restrict__ dy, long long R, float* __restrict__ pg,
                                               float* __restrict__ pbias, const WgTrSegs sg, int g, int G, char* smem) {
  using SH = WgTrShape<KT, NT>;
  constexpr int TK = SH::TK, TN = SH::TN;
  constexpr int NBLK = (KT + NT) * NP, NDMA = (NBLK + 7) / 8, STEP_BYTES = (NBLK + 1) * 1024;   // + the row-scale block (64 B used)
  constexpr int WGTR_NBUF = wgtr_nbuf<NP>(), AHEAD = WGTR_NBUF - 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / SH::WN, wn = wave % SH::WN;
  const bool active = wave < SH::WK * SH::WN;
  const float* __restrict__ rs = (const float*)(dy + dy_plane_uint4(R, NP));
  const float mref = rs[R];                                     // written by the dgrad's atomicMax
  const float inv_mref = mref > 0.f ? 1.0f / mref : 0.f;        // m_ref is a power of two: exact
  // DMA roles: block b = wave + 8 i (clamped: a surplus instruction re-fetches the last block, same bytes to the same place).  The
  // source address of a block is (wave-uniform base of its slot pair and part, advanced by a uniform stride per 32-row tile: SGPRs) +
  // (lane part: row r, slot parity a, half h: one 32-bit VGPR per block)
  const char* sbase[NDMA];
  unsigned tstride[NDMA], voff[NDMA];
  unsigned lds_blk[NDMA];
  {
    const int r = lane >> 2, a = (lane >> 1) & 1, h = lane & 1;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      int b = wave + 8 * i;
      b = b < NBLK ? b : NBLK - 1;
      const int t = b / NP, part = b % NP;
      int slot0, slot1;              // the slots lane parity a = 0 / 1 fetch
      const uint4* base;
      if (t < KT) {
        slot0 = t < sg.KTa ? sg.qx + 2 * t : sg.qx2 + 2 * (t - sg.KTa);
        slot1 = slot0 + 1;
        base = saved + (part ? sv_lo0(R) : 0);
        tstride[i] = (unsigned)(SAVE_SLOTS * 64 * sizeof(uint4));
      } else {     // a slot beyond the segment's k-steps repeats its last one: those columns are dropped by the reduction
        const int nt = t - KT;
        const int rel = nt < sg.NTa ? 2 * nt : 2 * (nt - sg.NTa);
        const int qb = nt < sg.NTa ? sg.qd : sg.qd2, ks = nt < sg.NTa ? sg.KSd : sg.KSd2;
        slot0 = qb + (rel < ks ? rel : ks - 1);
        slot1 = qb + (rel + 1 < ks ? rel + 1 : ks - 1);
        base = dy + (part ? dy_plane_uint4(R, 1) : 0);
        tstride[i] = (unsigned)(DY_SLOTS * 64 * sizeof(uint4));
      }
      static_assert(sv_addr(1, 0, 0, 0) == dy_addr(1, 0, 0, 0) && sv_addr(0, 0, 1, 1) == dy_addr(0, 0, 1, 1), "one lane-offset formula for both tensors");
      sbase[i] = (const char*)(base + sv_addr(__builtin_amdgcn_readfirstlane(slot0), 0, 0, 0));
      voff[i] = (unsigned)((sv_addr(a ? __builtin_amdgcn_readfirstlane(slot1 - slot0) : 0, 0, r, h)) * sizeof(uint4));
      lds_blk[i] = (unsigned)b * 1024u;
    }
  }
  // the 16 row scales of a step ride the same ring (every VMEM operation of the loop is a DMA: one counter discipline): wave 0 fetches
  // them as one more DMA instruction, lanes 0..3 carry 4 floats each (the other lanes repeat them)
  const float* rs_src = rs + 4 * (lane & 3);
  const long long n_t32 = R / 32;
  const int my_tiles = g < n_t32 ? (int)((n_t32 - 1 - g) / G) + 1 : 0;
  const int n_steps = 2 * my_tiles;
  auto issue = [&](int s) {                                     // DMA of local step s into ring slot s & 3 (steps past the end: the last one again)
    const int sc = s < n_steps ? s : n_steps - 1;
    const size_t t32 = (size_t)g + (size_t)(sc >> 1) * G;
    const unsigned ring = (unsigned)(s & (WGTR_NBUF - 1)) * STEP_BYTES;
#pragma unroll
    for (int i = 0; i < NDMA; ++i)
      glds16_nt_s(sbase[i] + t32 * tstride[i] + (sc & 1) * (32 * sizeof(uint4)), voff[i], __builtin_amdgcn_readfirstlane(ring + lds_blk[i]));
    if (wave == 0) glds16_nt((const char*)(rs_src + t32 * 32 + (sc & 1) * 16), __builtin_amdgcn_readfirstlane(ring + NBLK * 1024u));
  };
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 acc[TK][TN], acce = zero;                         // acce: the extra tile (k-tile wk TK + wn) x (n-tile NT - 1)
  float accb = 0.f;                                        // bias row: this lane's share of sum_rows (m_row / m_ref) dY^[row][n] (VALU dot products)
#pragma unroll
  for (int i = 0; i < TK; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = zero;
  if (n_steps > 0) {
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) issue(s);
    // lane part of every transpose-read address: row (8 kh + (i >> 2)) * 64 + slot parity a * 32 + (i & 3) * 8, i = lane & 15
    const unsigned lane_off = (unsigned)((8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);
    const int kh = lane >> 5;
    for (int s = 0; s < n_steps; ++s) {
      if (wave == 0) wait_vmcnt<(AHEAD - 1) * (NDMA + 1)>(); else
      wait_vmcnt<(AHEAD - 1) * NDMA>();                         // this wave's part of step s has landed (the later steps may be in flight)
      __syncthreads();                                          // ... everybody's has, and everybody is done with the ring slot of step s - 1
      // An LDS-DMA instruction costs ~100 issue cycles and a wave issues NDMA of them per step.  Right behind the barrier all eight waves
      // would do that at once, with the matrix pipe idle: the second wave of every SIMD (waves 4..7) issues its share after its first chunk
      // of MFMAs instead, so that one wave's DMA issue overlaps its partner's MFMAs (RNERF_WGTR_LATE_DMA; the slot written is the one of
      // step s - 1, free since the barrier)
      const bool dma_late = RNERF_WGTR_LATE_DMA && wave >= 4 && active;
      if (!dma_late) issue(s + AHEAD);
      if (active) {
        const char* ring0 = smem + (s & (WGTR_NBUF - 1)) * STEP_BYTES;
        const char* ring = ring0 + lane_off;
        const float4 s0 = *(const float4*)(ring0 + NBLK * 1024 + kh * 32), s1 = *(const float4*)(ring0 + NBLK * 1024 + kh * 32 + 16);
        const half8 sc8 = {(_Float16)(s0.x * inv_mref), (_Float16)(s0.y * inv_mref), (_Float16)(s0.z * inv_mref), (_Float16)(s0.w * inv_mref),
                           (_Float16)(s1.x * inv_mref), (_Float16)(s1.y * inv_mref), (_Float16)(s1.z * inv_mref), (_Float16)(s1.w * inv_mref)};
        // Operands are fetched and consumed in chunks so that the live set beside the TK x TN accumulators stays small enough for
        // RNERF_WGRAD_VGPRS without spilling: the smaller operand side stays resident for the step, the other side streams through in
        // chunks (A fragments one k-tile at a time when TK > TN, else B fragments NCH n-tiles at a time).
        half8 beh, bel;
        if constexpr (SH::EXTRA) {
          const char* p = ring + (KT + NT - 1) * NP * 1024;
          beh = tr_read8(p);
          if constexpr (NP == 2) bel = tr_read8(p + 1024);
        }
        static_assert(SH::WK >= TN, "bias owners");
        auto read_a = [&](int i, half8& h, half8& l) {
          const char* p = ring + (wk * TK + i) * NP * 1024;
          h = tr_read8(p) * sc8;
          if constexpr (NP == 2) l = tr_read8(p + 1024) * sc8;
        };
        auto read_b = [&](int j, half8& h, half8& l) {
          const char* p = ring + (KT + wn * TN + j) * NP * 1024;
          h = tr_read8(p);
          if constexpr (NP == 2) l = tr_read8(p + 1024);
        };
        // bias row: wave (wk, wn) owns it for n-tile wn TN + wk, wk < TN (the 9th n-tile of the Dense_9 + sigma job has no owner: the
        // sigma bias comes out of the rgb-head job, which reads the same head-gradient slot).  A B fragment holds, per lane, 8 rows of one
        // column and sc8 the scales of the same 8 rows: four v_dot2_f32_f16 per fragment instead of an MFMA with a one-row A operand and
        // a 16-register accumulator (the kernel has to stay within RNERF_WGRAD_VGPRS).
        auto dot8 = [&](const half8& x, float c) -> float {
          typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int k = 0; k < 4; ++k) c = __builtin_amdgcn_fdot2(half2v{sc8[2 * k], sc8[2 * k + 1]}, half2v{x[2 * k], x[2 * k + 1]}, c, false);
          return c;
        };
        auto bias_row = [&](int j, const half8& h, const half8& l) {
          if (wk == j) {
            accb = dot8(h, accb);
            if constexpr (NP == 2) accb = dot8(l, accb);
          }
        };
        half8 ae_h, ae_l;                   // the A fragments of the extra tile's k-tile (EXTRA: TK == 2, A side resident)
        if constexpr (TK > TN) {            // B resident, A streams
          half8 bh[TN], bl[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) read_b(j, bh[j], bl[j]);
          half8 an_h, an_l;
          read_a(0, an_h, an_l);
#pragma unroll
          for (int i = 0; i < TK; ++i) {
            const half8 ah = an_h, al = an_l;
            if (i + 1 < TK) read_a(i + 1, an_h, an_l);         // the next k-tile's fragments are on their way while this one's MFMAs issue
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma_h8(ah, bh[j], acc[i][j]);
            if constexpr (NP == 2) {
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = mfma_h8(ah, bl[j], acc[i][j]);
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = mfma_h8(al, bh[j], acc[i][j]);
            }
            if (i == 0 && dma_late) issue(s + AHEAD);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) bias_row(j, bh[j], bl[j]);
        } else {                            // A resident, B streams in chunks of NCH n-tiles
          constexpr int NCH = RNERF_WGTR_NCH < TN ? RNERF_WGTR_NCH : TN;
          static_assert(TN % NCH == 0, "n-tile chunks");
          half8 ah[TK], al[TK];
#pragma unroll
          for (int i = 0; i < TK; ++i) read_a(i, ah[i], al[i]);
#pragma unroll
          for (int c = 0; c < TN / NCH; ++c) {
            half8 bh[NCH], bl[NCH];
#pragma unroll
            for (int jj = 0; jj < NCH; ++jj) read_b(c * NCH + jj, bh[jj], bl[jj]);
#pragma unroll
            for (int i = 0; i < TK; ++i)
#pragma unroll
              for (int jj = 0; jj < NCH; ++jj) acc[i][c * NCH + jj] = mfma_h8(ah[i], bh[jj], acc[i][c * NCH + jj]);
            if constexpr (NP == 2) {
#pragma unroll
              for (int i = 0; i < TK; ++i)
#pragma unroll
                for (int jj = 0; jj < NCH; ++jj) acc[i][c * NCH + jj] = mfma_h8(ah[i], bl[jj], acc[i][c * NCH + jj]);
#pragma unroll
              for (int i = 0; i < TK; ++i)
#pragma unroll
                for (int jj = 0; jj < NCH; ++jj) acc[i][c * NCH + jj] = mfma_h8(al[i], bh[jj], acc[i][c * NCH + jj]);
            }
#pragma unroll
            for (int jj = 0; jj < NCH; ++jj) bias_row(c * NCH + jj, bh[jj], bl[jj]);
            if (c == 0 && dma_late) issue(s + AHEAD);
          }
          if constexpr (SH::EXTRA) { ae_h = wn ? ah[1] : ah[0]; if constexpr (NP == 2) ae_l = wn ? al[1] : al[0]; }
        }
        if constexpr (SH::EXTRA) {          // wn picks which of the wave's two k-tiles (a select on registers, no branch)
          static_assert(!SH::EXTRA || TK <= TN, "the extra tile reads the resident A fragments");
          acce = mfma_h8(ae_h, beh, acce);
          if constexpr (NP == 2) {
            acce = mfma_h8(ae_h, bel, acce);
            acce = mfma_h8(ae_l, beh, acce);
          }
        }
      }
    }
    wait_vmcnt<0>();                                            // the surplus prefetches must not outlive the workgroup's LDS
  }
  constexpr size_t ldn = (size_t)NT * 32;
  const int m = lane & 31, h = lane >> 5;
  if (active) {
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          pg[(size_t)((wk * TK + i) * 32 + row) * ldn + (wn * TN + j) * 32 + m] = acc[i][j][r];
        }
    if constexpr (SH::EXTRA) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        pg[(size_t)((wk * TK + wn) * 32 + row) * ldn + (NT - 1) * 32 + m] = acce[r];
      }
    }
    const float bsum = accb + __shfl_xor(accb, 32);          // the two 8-row halves of a k-step live in lanes m and m + 32
    if (wk < TN && h == 0) pbias[(wn * TN + wk) * 32 + m] = bsum;
  }
}

// ---- the same body for RNERF_BWD_F16X3_LO8: f16 hi planes as above, the lo planes as e4m3 bytes -------------------------------------------
// A lo tile of a 16-row step is 512 B in LDS, [row r][slot parity a][16 B = half 0 | half 1]; ONE DMA instruction builds TWO of them (lane
// L: tile L >> 5, row (L & 31) >> 1, parity L & 1 — 16 B = both halves of (slot, row), contiguous in the lo8 plane).  The 8-bit transpose
// read ds_read_b64_tr_b8 (within a 16-lane group destination lane i, byte j receives byte i & 7 of the 8 bytes addressed by source lane
// 2 j + (i >> 3): tools/ubench/tr8_cvt_probe.hip) hands lane (m = 16 a + i, kh) the 8 rows 8 kh .. 8 kh + 7 of feature position m in ONE
// read; four v_cvt_scalef32_pk_f16_fp8 turn them into the f16 lo fragment, and the three f16 MFMAs per product are those of the f16 lo
// planes — with the exact f16 hi parts in both cross terms.  Per step: (KT + NT) KiB of hi + (KT + NT) / 2 KiB of lo blocks.
template <int KT, int NT> struct WgTr8 {
  static constexpr int NHI = KT + NT, NLX = (KT + 1) / 2, NLD = (NT + 1) / 2;
  static constexpr int NBLK = NHI + NLX + NLD, NDMA = (NBLK + 7) / 8, STEP_BYTES = (NBLK + 1) * 1024;
  static constexpr int LOX = NHI * 1024, LOD = (NHI + NLX) * 1024, SCALES = NBLK * 1024;
};
#ifndef RNERF_WGTR8_NBUF
#define RNERF_WGTR8_NBUF 4
#endif
constexpr int wgtr8_lds_bytes() { return RNERF_WGTR8_NBUF * WgTr8<10, 8>::STEP_BYTES; }      // largest job: 10 k-tiles + 8 n-tiles
#if defined(RNERF_WGTR_ABL) && (RNERF_WGTR_ABL & 2)
__device__ __forceinline__ int2v tr_read8b(const char* p) { const int v = (int)((size_t)p & 7); return int2v{v, v}; }
#else
__device__ __forceinline__ int2v tr_read8b(const char* p) {
  typedef __attribute__((address_space(3))) int2v lds_int2;
  return __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_int2*)p);
}
#endif

template <int KT, int NT>
__device__ __forceinline__ void wgrad_body_tr8(const uint4* __restrict__ saved, const uint4* __restrict__ dy, long long R, float* __restrict__ pg,
                                                float* __restrict__ pbias, const WgTrSegs sg, int g, int G, char* smem) {
  using SH = WgTrShape<KT, NT>;
  using L8 = WgTr8<KT, NT>;
  constexpr int TK = SH::TK, TN = SH::TN;
  constexpr int NHI = L8::NHI, NBLK = L8::NBLK, NDMA = L8::NDMA, STEP_BYTES = L8::STEP_BYTES;
  constexpr int WGTR_NBUF = RNERF_WGTR8_NBUF, AHEAD = WGTR_NBUF - 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / SH::WN, wn = wave % SH::WN;
  const bool active = wave < SH::WK * SH::WN;
  const float* __restrict__ rs = (const float*)(dy + dy_plane_uint4(R, 2));
  const float mref = rs[R];
  const float inv_mref = mref > 0.f ? 1.0f / mref : 0.f;
  // slot pair of X tile t / dY tile nt (see wgrad_body_tr)
  auto x_slots = [&](int t, int& s0, int& s1) { s0 = t < sg.KTa ? sg.qx + 2 * t : sg.qx2 + 2 * (t - sg.KTa); s1 = s0 + 1; };
  auto d_slots = [&](int nt, int& s0, int& s1) {
    const int rel = nt < sg.NTa ? 2 * nt : 2 * (nt - sg.NTa);
    const int qb = nt < sg.NTa ? sg.qd : sg.qd2, ks = nt < sg.NTa ? sg.KSd : sg.KSd2;
    s0 = qb + (rel < ks ? rel : ks - 1);
    s1 = qb + (rel + 1 < ks ? rel + 1 : ks - 1);
  };
  const char* sbase[NDMA];
  unsigned tstride[NDMA], hstride[NDMA], voff[NDMA], lds_blk[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    int b = wave + 8 * i;
    b = b < NBLK ? b : NBLK - 1;
    lds_blk[i] = (unsigned)b * 1024u;
    if (b < NHI) {                      // f16 hi block of tile b: lane = 4 r + 2 a + h, 16 B = (slot a, row r, half h)
      const int r = lane >> 2, a = (lane >> 1) & 1, h = lane & 1;
      int s0, s1;
      const bool is_x = b < KT;
      if (is_x) x_slots(b, s0, s1); else d_slots(b - KT, s0, s1);
      sbase[i] = (const char*)((is_x ? saved : dy) + sv_addr(__builtin_amdgcn_readfirstlane(s0), 0, 0, 0));
      voff[i] = (unsigned)((sv_addr(a ? __builtin_amdgcn_readfirstlane(s1 - s0) : 0, 0, r, h)) * sizeof(uint4));
      tstride[i] = (unsigned)((is_x ? SAVE_SLOTS : DY_SLOTS) * 64 * sizeof(uint4));
      hstride[i] = (unsigned)(32 * sizeof(uint4));
    } else {                            // e4m3 lo block: two tiles, lane = 32 (tile parity) + 2 r + a, 16 B = (slot a, row r, both halves)
      const int lb = b - NHI;
      const bool is_x = lb < L8::NLX;
      const int t0 = 2 * (is_x ? lb : lb - L8::NLX) + (lane >> 5);
      const int r = (lane & 31) >> 1, a = lane & 1;
      int s0, s1;
      if (is_x) x_slots(t0 < KT ? t0 : KT - 1, s0, s1); else d_slots(t0 < NT ? t0 : NT - 1, s0, s1);
      sbase[i] = is_x ? (const char*)(saved + sv_lo0(R)) : (const char*)(dy + dy_plane_uint4(R, 1));
      voff[i] = (unsigned)(((a ? s1 : s0) * 32 + r) * 16);
      tstride[i] = (unsigned)((is_x ? SAVE_SLOTS : DY_SLOTS) * 32 * 16);
      hstride[i] = 16u * 16u;
    }
  }
  const float* rs_src = rs + 4 * (lane & 3);
  const long long n_t32 = R / 32;
  const int my_tiles = g < n_t32 ? (int)((n_t32 - 1 - g) / G) + 1 : 0;
  const int n_steps = 2 * my_tiles;
  auto issue = [&](int s) {
    const int sc = s < n_steps ? s : n_steps - 1;
    const size_t t32 = (size_t)g + (size_t)(sc >> 1) * G;
    const unsigned ring = (unsigned)(s % WGTR_NBUF) * STEP_BYTES;
#pragma unroll
    for (int i = 0; i < NDMA; ++i)
      glds16_nt_s(sbase[i] + t32 * tstride[i] + (sc & 1) * hstride[i], voff[i], __builtin_amdgcn_readfirstlane(ring + lds_blk[i]));
    if (wave == 0) glds16_nt((const char*)(rs_src + t32 * 32 + (sc & 1) * 16), __builtin_amdgcn_readfirstlane(ring + L8::SCALES));
  };
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 acc[TK][TN], acce = zero;
  float accb = 0.f;
#pragma unroll
  for (int i = 0; i < TK; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = zero;
  if (n_steps > 0) {
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) issue(s);
    const unsigned lane_off = (unsigned)((8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);
    // lo tile: row (8 kh + (i >> 1)) * 32 + slot parity a * 16 + (i & 1) * 8, i = lane & 15
    const unsigned lane_off8 = (unsigned)((8 * (lane >> 5) + ((lane & 15) >> 1)) * 32 + ((lane >> 4) & 1) * 16 + (lane & 1) * 8);
    const int kh = lane >> 5;
    for (int s = 0; s < n_steps; ++s) {
      if (wave == 0) wait_vmcnt<(AHEAD - 1) * (NDMA + 1)>(); else
      wait_vmcnt<(AHEAD - 1) * NDMA>();
      __syncthreads();
      const bool dma_late = RNERF_WGTR_LATE_DMA && wave >= 4 && active;      // see wgrad_body_tr
      if (!dma_late) issue(s + AHEAD);
      if (active) {
        const char* ring0 = smem + (s % WGTR_NBUF) * STEP_BYTES;
        const char* ring = ring0 + lane_off;
        const char* ring8 = ring0 + lane_off8;
        const float4 s0 = *(const float4*)(ring0 + L8::SCALES + kh * 32), s1 = *(const float4*)(ring0 + L8::SCALES + kh * 32 + 16);
        const half8 sc8 = {(_Float16)(s0.x * inv_mref), (_Float16)(s0.y * inv_mref), (_Float16)(s0.z * inv_mref), (_Float16)(s0.w * inv_mref),
                           (_Float16)(s1.x * inv_mref), (_Float16)(s1.y * inv_mref), (_Float16)(s1.z * inv_mref), (_Float16)(s1.w * inv_mref)};
        half8 beh, bel;
        if constexpr (SH::EXTRA) {
          beh = tr_read8(ring + (KT + NT - 1) * 1024);
          bel = lo8_decode(tr_read8b(ring8 + L8::LOD + (NT - 1) * 512), LO8_SCALE_D);
        }
        static_assert(SH::WK >= TN, "bias owners");
        auto read_a = [&](int i, half8& h, half8& l) {
          const int t = wk * TK + i;
          h = tr_read8(ring + t * 1024) * sc8;
          l = lo8_decode(tr_read8b(ring8 + L8::LOX + t * 512), LO8_SCALE_X) * sc8;
        };
        auto read_b = [&](int j, half8& h, half8& l) {
          const int nt = wn * TN + j;
          h = tr_read8(ring + (KT + nt) * 1024);
          l = lo8_decode(tr_read8b(ring8 + L8::LOD + nt * 512), LO8_SCALE_D);
        };
        auto dot8 = [&](const half8& x, float c) -> float {
#pragma unroll
          for (int k = 0; k < 4; ++k) c = __builtin_amdgcn_fdot2(half2v{sc8[2 * k], sc8[2 * k + 1]}, half2v{x[2 * k], x[2 * k + 1]}, c, false);
          return c;
        };
        auto bias_row = [&](int j, const half8& h, const half8& l) {
          if (wk == j) { accb = dot8(h, accb); accb = dot8(l, accb); }
        };
        half8 ae_h, ae_l;
        if constexpr (TK > TN) {            // B resident, A streams
          half8 bh[TN], bl[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) read_b(j, bh[j], bl[j]);
          half8 an_h, an_l;
          read_a(0, an_h, an_l);
#pragma unroll
          for (int i = 0; i < TK; ++i) {
            const half8 ah = an_h, al = an_l;
            if (i + 1 < TK) read_a(i + 1, an_h, an_l);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma_h8(ah, bh[j], acc[i][j]);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma_h8(ah, bl[j], acc[i][j]);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma_h8(al, bh[j], acc[i][j]);
            if (i == 0 && dma_late) issue(s + AHEAD);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) bias_row(j, bh[j], bl[j]);
        } else {                            // A resident, B streams in chunks of NCH n-tiles
          constexpr int NCH = RNERF_WGTR_NCH < TN ? RNERF_WGTR_NCH : TN;
          static_assert(TN % NCH == 0, "n-tile chunks");
          half8 ah[TK], al[TK];
#pragma unroll
          for (int i = 0; i < TK; ++i) read_a(i, ah[i], al[i]);
#pragma unroll
          for (int c = 0; c < TN / NCH; ++c) {
            half8 bh[NCH], bl[NCH];
#pragma unroll
            for (int jj = 0; jj < NCH; ++jj) read_b(c * NCH + jj, bh[jj], bl[jj]);
#pragma unroll
            for (int i = 0; i < TK; ++i)
#pragma unroll
              for (int jj = 0; jj < NCH; ++jj) acc[i][c * NCH + jj] = mfma_h8(ah[i], bh[jj], acc[i][c * NCH + jj]);
#pragma unroll
            for (int i = 0; i < TK; ++i)
#pragma unroll
              for (int jj = 0; jj < NCH; ++jj) acc[i][c * NCH + jj] = mfma_h8(ah[i], bl[jj], acc[i][c * NCH + jj]);
#pragma unroll
            for (int i = 0; i < TK; ++i)
#pragma unroll
              for (int jj = 0; jj < NCH; ++jj) acc[i][c * NCH + jj] = mfma_h8(al[i], bh[jj], acc[i][c * NCH + jj]);
#pragma unroll
            for (int jj = 0; jj < NCH; ++jj) bias_row(c * NCH + jj, bh[jj], bl[jj]);
            if (c == 0 && dma_late) issue(s + AHEAD);
          }
          if constexpr (SH::EXTRA) { ae_h = wn ? ah[1] : ah[0]; ae_l = wn ? al[1] : al[0]; }
        }
        if constexpr (SH::EXTRA) {
          static_assert(!SH::EXTRA || TK <= TN, "the extra tile reads the resident A fragments");
          acce = mfma_h8(ae_h, beh, acce);
          acce = mfma_h8(ae_h, bel, acce);
          acce = mfma_h8(ae_l, beh, acce);
        }
      }
    }
    wait_vmcnt<0>();
  }
  constexpr size_t ldn = (size_t)NT * 32;
  const int m = lane & 31, h = lane >> 5;
  if (active) {
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          pg[(size_t)((wk * TK + i) * 32 + row) * ldn + (wn * TN + j) * 32 + m] = acc[i][j][r];
        }
    if constexpr (SH::EXTRA) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        pg[(size_t)((wk * TK + wn) * 32 + row) * ldn + (NT - 1) * 32 + m] = acce[r];
      }
    }
    const float bsum = accb + __shfl_xor(accb, 32);
    if (wk < TN && h == 0) pbias[(wn * TN + wk) * 32 + m] = bsum;
  }
}

// At most 224 VGPRs per wave: two of these waves per SIMD then leave 64 registers — one wave of the march kernel — on every SIMD, so
// the next batch's march (a latency-bound chain that needs a wave slot on every CU, no LDS) can be co-resident with this HBM-paced kernel
// instead of waiting for whole CUs to drain.
#ifndef RNERF_WGRAD_VGPRS
#define RNERF_WGRAD_VGPRS 112   /* the attribute counts half of the unified VGPR + AGPR file on gfx90a+: 112 -> 224 registers */
#endif
template <int NP>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(RNERF_WGRAD_VGPRS)))
nerfmlp_wgrad_tr_kernel(const uint4* __restrict__ saved, const uint4* __restrict__ dy, long long R, float* __restrict__ workspace, const WgradTable tab,
                        long long* __restrict__ trace) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#if defined(RNERF_WGTR_PRIO) && RNERF_WGTR_PRIO > 0
  __builtin_amdgcn_s_setprio(RNERF_WGTR_PRIO);      // above the co-resident march wave in the SIMD's issue arbitration
#endif
  if (trace && threadIdx.x == 0) trace[2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
  int j = 0;
  while ((int)blockIdx.x >= tab.wg0[j + 1]) ++j;
  const WgradJob job = tab.job[j];
  const int g = blockIdx.x - tab.wg0[j], G = tab.wg0[j + 1] - tab.wg0[j];
  const int KT = job.KT, NT = job.NT;
  float* pg = workspace + tab.poff[j] + (size_t)g * (size_t)KT * 32 * NT * 32;
  float* pb = workspace + tab.pboff[j] + (size_t)g * NT * 32;
  const WgTrSegs sg = {tab.qx[j], job.KTa, tab.qx2[j], tab.qd[j], job.NTa, tab.KSd[j], tab.qd2[j], tab.KSd2[j]};
#define RNERF_WGRAD_CASE(KT_, NT_)                                                                                                  \
  if (KT == (KT_) && NT == (NT_)) {                                                                                                 \
    wgrad_body_tr<NP, KT_, NT_>(saved, dy, R, pg, pb, sg, g, G, smem);                                                              \
    if (trace && threadIdx.x == 0) trace[2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();                         \
    return;                                                                                                                         \
  }
  RNERF_WGRAD_CASE(8, 8) RNERF_WGRAD_CASE(2, 8) RNERF_WGRAD_CASE(10, 8) RNERF_WGRAD_CASE(8, 9) RNERF_WGRAD_CASE(9, 4) RNERF_WGRAD_CASE(4, 1)
#undef RNERF_WGRAD_CASE
  __builtin_trap();
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(RNERF_WGRAD_VGPRS)))
nerfmlp_wgrad_tr8_kernel(const uint4* __restrict__ saved, const uint4* __restrict__ dy, long long R, float* __restrict__ workspace, const WgradTable tab,
                         long long* __restrict__ trace) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (trace && threadIdx.x == 0) trace[2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
  int j = 0;
  while ((int)blockIdx.x >= tab.wg0[j + 1]) ++j;
  const WgradJob job = tab.job[j];
  const int g = blockIdx.x - tab.wg0[j], G = tab.wg0[j + 1] - tab.wg0[j];
  const int KT = job.KT, NT = job.NT;
  float* pg = workspace + tab.poff[j] + (size_t)g * (size_t)KT * 32 * NT * 32;
  float* pb = workspace + tab.pboff[j] + (size_t)g * NT * 32;
  const WgTrSegs sg = {tab.qx[j], job.KTa, tab.qx2[j], tab.qd[j], job.NTa, tab.KSd[j], tab.qd2[j], tab.KSd2[j]};
#define RNERF_WGRAD_CASE(KT_, NT_)                                                                                                  \
  if (KT == (KT_) && NT == (NT_)) {                                                                                                 \
    wgrad_body_tr8<KT_, NT_>(saved, dy, R, pg, pb, sg, g, G, smem);                                                                 \
    if (trace && threadIdx.x == 0) trace[2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();                         \
    return;                                                                                                                         \
  }
  RNERF_WGRAD_CASE(8, 8) RNERF_WGRAD_CASE(2, 8) RNERF_WGRAD_CASE(10, 8) RNERF_WGRAD_CASE(8, 9) RNERF_WGRAD_CASE(9, 4) RNERF_WGRAD_CASE(4, 1)
#undef RNERF_WGRAD_CASE
  __builtin_trap();
}

template <bool X_F16>
__global__ void __launch_bounds__(512)
nerfmlp_wgrad_kernel(const uint4* __restrict__ saved, const uint4* __restrict__ dy, long long R, long long total_rows, int n_chunks,
                      float* __restrict__ workspace, const WgradTable tab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int j = 0;
  while ((int)blockIdx.x >= tab.wg0[j + 1]) ++j;
  const int qx = tab.qx[j], KSx = tab.KSx[j], qd = tab.qd[j], KSd = tab.KSd[j];
  const int g = blockIdx.x - tab.wg0[j], G = tab.wg0[j + 1] - tab.wg0[j];
  const int KT = KSx >> 1, NT = (KSd + 1) >> 1;
  float* pg = workspace + tab.poff[j] + (size_t)g * (size_t)KT * 32 * NT * 32;
  float* pb = workspace + tab.pboff[j] + (size_t)g * NT * 32;
#define RNERF_WGRAD_CASE(KT_, KSD_)                                                                                                 \
  if (KSx == 2 * (KT_) && KSd == (KSD_)) { wgrad_body<X_F16, KT_, KSD_>(saved, dy, R, total_rows, n_chunks, pg, pb, qx, qd, g, G, smem); return; }
  RNERF_WGRAD_CASE(8, 16) RNERF_WGRAD_CASE(2, 16) RNERF_WGRAD_CASE(8, 1) RNERF_WGRAD_CASE(8, 8) RNERF_WGRAD_CASE(1, 8) RNERF_WGRAD_CASE(4, 1)
#undef RNERF_WGRAD_CASE
  __builtin_trap();
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ workspace, const WgradTable tab, float* __restrict__ grads,
                                                           const float* __restrict__ out_scale) {
  const float osc = out_scale ? *out_scale : 1.0f;      // m_ref of the row-normalised modes
  const int jb = blockIdx.y;
  const WgradJob job = tab.job[jb];
  const int n_wg = tab.wg0[jb + 1] - tab.wg0[jb];
  const float* __restrict__ partial = workspace + tab.poff[jb];
  const float* __restrict__ partial_bias = workspace + tab.pboff[jb];
  const int ldn = job.NT * 32, rows = job.KT * 32;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  // column J of the partial -> (Dense index, output feature) through the dY segment it lies in
  auto out_feature = [&](int J, int& dense) -> int {
    int nt = J >> 5;
    const int c = J & 31, a = c >> 4, hh = (c >> 3) & 1, j = c & 7;
    const bool segb = nt >= job.NTa;
    const int dkind = segb ? job.dkind2 : job.dkind;
    dense = segb ? job.dense2 : job.dense;
    if (segb) nt -= job.NTa;
    if (dkind == 0) { const int n = prev_feature(2 * nt + a, hh, j); return n < nerf_dense(dense).out ? n : -1; }
    if (a != 0 || hh != 0 || nt != 0) return -1;
    if (dkind == 1) return j == 0 ? 0 : -1;
    return (j >= 1 && j <= 3) ? j - 1 : -1;
  };
  auto sum_over = [&](const float* __restrict__ p, size_t stride) -> float {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 4 <= n_wg; g += 4) { s0 += p[(size_t)g * stride]; s1 += p[(size_t)(g + 1) * stride]; s2 += p[(size_t)(g + 2) * stride]; s3 += p[(size_t)(g + 3) * stride]; }
    for (; g < n_wg; ++g) s0 += p[(size_t)g * stride];
    return (s0 + s1) + (s2 + s3);
  };
  if (e < rows * ldn) {
    const int I = e / ldn, J = e % ldn;
    int kt = I >> 5;
    const int c = I & 31, a = c >> 4, hh = (c >> 3) & 1, j = c & 7;
    const bool segb = kt >= job.KTa;
    const int xkind = segb ? job.xkind2 : job.xkind, row_off = segb ? job.row_off2 : job.row_off;
    if (segb) kt -= job.KTa;
    const int s = 2 * kt + a;
    int fin = xkind == 1 ? prev_feature(s, hh, j) : (xkind == 0 ? pe_feature(8 * s + j, hh) : view_feature(8 * s + j, hh));
    int dense;
    const int fout = out_feature(J, dense);
    if (fin >= 0 && fout >= 0) {
      fin += row_off;
      if (fin < nerf_dense(dense).in) grads[nerf_koff(dense) + fin * nerf_dense(dense).out + fout] = sum_over(partial + e, (size_t)rows * ldn) * osc;
    }
  }
  if (job.write_bias && e < ldn) {
    int dense;
    const int fout = out_feature(e, dense);
    if (fout >= 0 && (e >> 5) < job.NTa) grads[nerf_boff(dense) + fout] = sum_over(partial_bias + e, (size_t)ldn) * osc;
    if (job.bias_sigma && e == 0) grads[nerf_boff(8)] = sum_over(partial_bias, (size_t)ldn) * osc;      // column 0 of the head slot = d raw_sigma
  }
}

// ------------------------------------------------------------------------------------------------------------------
// N2: the 4x128 background MLP (rnerf/models.py:116-118) in exact fp32 on v_mfma_f32_32x32x2_f32.
// One wave = 32 rays; weights are read straight from the flat fp32 buffer (kernel[in][out], coalesced along out).
// ------------------------------------------------------------------------------------------------------------------
// acc[t] += W(step, t) * x(step) over NSTEP K=2 steps of v_mfma_f32_32x32x2_f32 for NT n-tiles, with the per-lane weight dwords fetched one
// BATCH (4 steps) ahead of the MFMAs that consume them.  Left to itself hipcc emits load -> s_waitcnt vmcnt(0) -> MFMA for every single
// product, re-using one register: a full L2 round trip (~650 cycles) per 64-cycle MFMA — the small-MLP kernels ran at a tenth of the
// matrix rate (so3_fwd_train_kernel 1.77 ms for 208 k rows).  wload(step, t) must be a pure load of a lane-dependent address, x(step) a
// register operand; both are called with constants after unrolling.  Same products in the same order: same bits.
template <int NSTEP, int NT, typename WF, typename XF>
__device__ __forceinline__ void mfma_f32_stream(f32x16* acc, WF wload, XF xop) {
  constexpr int BS = 4, NB = (NSTEP + BS - 1) / BS;
  float w[2][BS * NT];
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) w[0][i * NT + t] = i < NSTEP ? wload(i, t) : 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b + 1 < NB) {
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) w[(b + 1) & 1][i * NT + t] = (b + 1) * BS + i < NSTEP ? wload((b + 1) * BS + i, t) : 0.f;
    }
    RNERF_PIN();
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      if (b * BS + i < NSTEP) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[b & 1][i * NT + t], xop(b * BS + i), acc[t], 0, 0, 0);
      }
    }
    RNERF_PIN();
  }
}

// The same with the weights of a batch's 4 steps in ONE 16-byte load per n-tile (wload4(batch, t)): for the transposed products of the
// dgrad chains a lane's 4 consecutive steps read 4 consecutive floats of one kernel row, while the lanes of a load are 512 bytes apart
// (64 cache lines per instruction: the address unit, not the matrix pipe, paced these kernels with one dword per load).
template <int NSTEP, int NT, typename WF4, typename XF>
__device__ __forceinline__ void mfma_f32_stream4(f32x16* acc, WF4 wload4, XF xop) {
  static_assert(NSTEP % 4 == 0, "whole batches");
  constexpr int NB = NSTEP / 4;
  float4 w[2][NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) w[0][t] = wload4(0, t);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b + 1 < NB) {
#pragma unroll
      for (int t = 0; t < NT; ++t) w[(b + 1) & 1][t] = wload4(b + 1, t);
    }
    RNERF_PIN();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 q = w[b & 1][t];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(i == 0 ? q.x : (i == 1 ? q.y : (i == 2 ? q.z : q.w)), xop(4 * b + i), acc[t], 0, 0, 0);
      }
    RNERF_PIN();
  }
}

__device__ __forceinline__ void small_init_bias(f32x16 (&acc)[4], const float* __restrict__ bias, int h) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = bias[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
}

// acc += W[rows f(h)][:] x operand, for the 64 (t,r) steps of a 128-wide previous activation held in `x`
__device__ __forceinline__ void small_prev_layer(f32x16 (&acc)[4], const f32x16 (&x)[4], const float* __restrict__ kern, int m, int h) {
  const float* __restrict__ kl = kern + 4 * h * 128 + m;           // step (ts, r): the feature this half holds in x[ts][r] = 32 ts + (r & 3) + 8 (r >> 2) + 4 h
  mfma_f32_stream<64, 4>(acc, [&](int st, int t) { return kl[(32 * (st >> 4) + (st & 3) + 8 * ((st & 15) >> 2)) * 128 + 32 * t]; },
                         [&](int st) { return x[st >> 4][st & 15]; });
}

__device__ __forceinline__ void small_dir_layer(f32x16 (&acc)[4], const float (&enc)[14], const float* __restrict__ kern, int m, int h) {
  mfma_f32_stream<14, 4>(acc, [&](int q, int t) {
    const int f = h ? dir_feature(q, 1) : dir_feature(q, 0);
    const float w = kern[(f < 0 ? 0 : f) * 128 + 32 * t + m];
    return f < 0 ? 0.f : w;
  }, [&](int q) { return enc[q]; });
}

template <bool TRAIN>
__global__ void __launch_bounds__(64) bkgd_fwd_kernel(const float* __restrict__ params, const float* __restrict__ dirs, int dir_stride,
                                                      long long n, float pad_scale, float pad, float* __restrict__ out_rgb,
                                                      float* __restrict__ save) {
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

  f32x16 acc[4], x[4];
  // Dense_0: 27 -> 128, ReLU
  small_init_bias(acc, params + bkgd_boff(0), h);
  small_dir_layer(acc, enc, params + bkgd_koff(0), m, h);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[t][r] = fmaxf(acc[t][r], 0.f);
  save_x(1, x);
  // Dense_1, Dense_2: 128 -> 128, ReLU
#pragma unroll 1
  for (int l = 1; l <= 2; ++l) {
    small_init_bias(acc, params + (l == 1 ? bkgd_boff(1) : bkgd_boff(2)), h);
    small_prev_layer(acc, x, params + (l == 1 ? bkgd_koff(1) : bkgd_koff(2)), m, h);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[t][r] = fmaxf(acc[t][r], 0.f);
    save_x(l + 1, x);
  }
  // Dense_3: [x(128), inputs(27)] -> 128, ReLU  (skip concat after i == 2, rnerf/model_utils.py:131-132)
  small_init_bias(acc, params + bkgd_boff(3), h);
  small_prev_layer(acc, x, params + bkgd_koff(3), m, h);
  small_dir_layer(acc, enc, params + bkgd_koff(3) + 128 * 128, m, h);
  // Dense_4: 128 -> 3 on the VALU, then sigmoid*(1+2p)-p (rnerf/models.py:336-337)
  if constexpr (TRAIN) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[t][r] = fmaxf(acc[t][r], 0.f);
    save_x(4, x);
  }
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
  if (ok && h == 0) {
    out_rgb[3 * row] = o[0]; out_rgb[3 * row + 1] = o[1]; out_rgb[3 * row + 2] = o[2];
    if constexpr (TRAIN) { float* so = save + (size_t)n * (28 + 4 * 128) + (size_t)row * 3; so[0] = o[0]; so[1] = o[1]; so[2] = o[2]; }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// G4 / P2 / E1 (stage "all"): so3_mlp = MLP(128, 4, skip 2, out 3) on annealed_pos_enc(x) (rnerf/ior_utils.py:148-152, :283;
// rnerf/model_utils.py:236-245), the Rodrigues rotation of grad n by that axis-angle (ior_utils.py:305-312), and the march
// that uses it (rnerf/eikonal_utils.py:34-39).  Same exact-fp32 MFMA chain as the background MLP; one wave = 32 points.
// ------------------------------------------------------------------------------------------------------------------
// K=2 steps over the 60 annealed features: step p, half h -> feature 2p + h = 6d + 3*is_cos + c with d = p / 3 for both halves
__device__ __forceinline__ void so3_enc_layer(f32x16 (&acc)[4], const float (&enc)[30], const float* __restrict__ kern, int m, int h) {
  const float* __restrict__ kl = kern + h * 128 + m;
  mfma_f32_stream<30, 4>(acc, [&](int p, int t) { return kl[2 * p * 128 + 32 * t]; }, [&](int p) { return enc[p]; });
}

// raw axis-angle of one point per lane pair (lanes m and m + 32 hold the same point); all 64 lanes must call it
__device__ __forceinline__ void so3_eval(const float* __restrict__ params, float px, float py, float pz, const So3Window& win, int m, int h,
                                         float (&raw)[3]) {
  float enc[30];
  const float HALF_PI = 1.5707963705062866f;
#pragma unroll
  for (int p = 0; p < 30; ++p) {
    const int d = p / 3, k = p % 3;
    const float x = k == 0 ? (h ? py : px) : (k == 1 ? (h ? px : pz) : (h ? pz : py));
    const float phase = k == 0 ? 0.f : (k == 1 ? (h ? HALF_PI : 0.f) : HALF_PI);
    const float xb = fmul(x, (float)(1 << d));
    enc[p] = fmul(sinf(k == 0 ? xb : fadd(xb, phase)), win.w[d]);
  }
  f32x16 acc[4], x[4];
  small_init_bias(acc, params + so3_boff(0), h);
  so3_enc_layer(acc, enc, params + so3_koff(0), m, h);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[t][r] = fmaxf(acc[t][r], 0.f);
#pragma unroll 1
  for (int l = 1; l <= 2; ++l) {
    small_init_bias(acc, params + (l == 1 ? so3_boff(1) : so3_boff(2)), h);
    small_prev_layer(acc, x, params + (l == 1 ? so3_koff(1) : so3_koff(2)), m, h);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[t][r] = fmaxf(acc[t][r], 0.f);
  }
  small_init_bias(acc, params + so3_boff(3), h);                     // Dense_3: [x(128), inputs(60)] (skip concat after i == 2)
  small_prev_layer(acc, x, params + so3_koff(3), m, h);
  so3_enc_layer(acc, enc, params + so3_koff(3) + 128 * 128, m, h);
  float o[3] = {0.f, 0.f, 0.f};
  const float* __restrict__ k4 = params + so3_koff(4);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = fmaxf(acc[t][r], 0.f);
      const int f = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
      o[0] = fmaf(v, k4[f * 3 + 0], o[0]); o[1] = fmaf(v, k4[f * 3 + 1], o[1]); o[2] = fmaf(v, k4[f * 3 + 2], o[2]);
    }
#pragma unroll
  for (int c = 0; c < 3; ++c) raw[c] = o[c] + __shfl_xor(o[c], 32) + params[so3_boff(4) + c];
}

// ---- so3_mlp on the 16-bit matrix cores (f16 hi + lo split of both operands, 3 MFMAs per product tile, fp32 accumulate: the arithmetic of
// the NerfMLP engine) for the march of stage "all*", where it runs once per node of the boundary shell.  One WORKGROUP of four waves
// evaluates it for 16 rays on v_mfma_f32_16x16x32_f16 (transposed chain: weights = A, rows = output features; activations = B, columns =
// rays): lane (ray c = lane & 15, k-group g = lane >> 4) holds D rows 4 g + i of a 16-feature tile.  Wave w owns the output tiles 2 w and
// 2 w + 1 (32 features) of every layer, so after a layer it holds, per lane, the 8 values
//     slot j < 4: feature 32 w + 4 g + j (tile 2 w)       slot j >= 4: feature 32 w + 16 + 4 g + (j - 4) (tile 2 w + 1)
// which is exactly what lane (c, g) must supply as B operand of k-step w (K = 32) of the next layer when the packed weights use the same
// slot -> feature map.  Packed stream: 16 k-steps (Dense_0: 2 encoding; Dense_1, Dense_2: 4; Dense_3: 4 + 2 encoding) x 8 tiles; block
// (ks, t) = 64 lanes x uint4 hi, then lo (2 KiB).  Encoding k-steps use feature 32 s + 8 g + j of annealed_pos_enc (zero past 59).
constexpr int kSo3KSteps = 16;
constexpr int kSo3Blocks = kSo3KSteps * 8;                       // 128 blocks x 2 KiB
constexpr float SO3_WSCALE = 256.f;                              // keeps the lo parts of the f16 split normal (as Prec<F16X3>::WSCALE)
// kernel row (input feature) of Dense_l fed by slot j of lane group g in k-step ks, or -1 (zero padding); l through *layer
__host__ __device__ constexpr int so3s_row(int ks, int g, int j, int* layer) {
  const int l = ks < 2 ? 0 : (ks < 6 ? 1 : (ks < 10 ? 2 : 3));
  const int s = ks - (l == 0 ? 0 : (l == 1 ? 2 : (l == 2 ? 6 : 10)));
  *layer = l;
  const bool enc = l == 0 || (l == 3 && s >= 4);
  if (enc) { const int f = 32 * (l == 0 ? s : s - 4) + 8 * g + j; return f < 60 ? (l == 0 ? f : 128 + f) : -1; }
  return 32 * s + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
}

__global__ void so3_pack16_kernel(const float* __restrict__ params, uint4* __restrict__ packed) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= kSo3Blocks * 64) return;
  const int blk = gid >> 6, lane = gid & 63;
  const int ks = blk >> 3, t = blk & 7;
  const int n_out = 16 * t + (lane & 15), g = lane >> 4;
  float w[8];
  for (int j = 0; j < 8; ++j) {
    int l = 0;
    const int row = so3s_row(ks, g, j, &l);
    w[j] = row < 0 ? 0.f : params[(l == 0 ? so3_koff(0) : (l == 1 ? so3_koff(1) : (l == 2 ? so3_koff(2) : so3_koff(3)))) + row * 128 + n_out] * SO3_WSCALE;
  }
  uint32_t hi[4], lo[4];
  for (int p = 0; p < 4; ++p) split2<true>(w[2 * p], w[2 * p + 1], hi[p], lo[p]);
  packed[(size_t)blk * 128 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  packed[(size_t)blk * 128 + 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

struct So3Ops { uint4 h, l; };
__device__ __forceinline__ So3Ops so3_split_ops(const float (&x)[8]) {
  uint32_t hh[4], ll[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) split2<true>(x[2 * p], x[2 * p + 1], hh[p], ll[p]);
  So3Ops o; o.h = make_uint4(hh[0], hh[1], hh[2], hh[3]); o.l = make_uint4(ll[0], ll[1], ll[2], ll[3]);
  return o;
}

// LDS of one 16-ray workgroup.  Every exchange has its own region, so one barrier per exchange is enough (a region is rewritten only after
// the barriers of a whole evaluation have been passed by every wave).
struct So3Shared {
  uint4 enc[2][2][64];          // encoding operands: k-step, (hi, lo), lane
  uint4 x[3][4][2][64];         // inputs of Dense_1, Dense_2, Dense_3: k-step, (hi, lo), lane
  float4 out[4][16];            // Dense_4 partial sums per wave and ray
  float4 pos[16];               // the points to evaluate (written by the marching wave)
  float bias[4][128];           // Dense_0 .. Dense_3
  float k4[128][4];             // Dense_4 kernel rows (3 outputs, padded)
  float tail[16];               // Dense_4 bias (3), the annealing window (10) at [4..13]
  int done;                     // set by the marching wave before its last barrier
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 so3_mfma16(const uint4 a, const uint4 b, const f32x4 c) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void so3_shared_init(So3Shared& sh, const float* __restrict__ params, const So3Window& win, int tid) {
  for (int i = tid; i < 4 * 128; i += 256) { const int l = i >> 7, n = i & 127; sh.bias[l][n] = params[(l == 0 ? so3_boff(0) : (l == 1 ? so3_boff(1) : (l == 2 ? so3_boff(2) : so3_boff(3)))) + n]; }
  for (int i = tid; i < 128 * 3; i += 256) sh.k4[i / 3][i % 3] = params[so3_koff(4) + i];
  if (tid < 3) sh.tail[tid] = params[so3_boff(4) + tid];
  if (tid < 10) sh.tail[4 + tid] = win.w[tid];
  if (tid == 0) sh.done = 0;
}

// One evaluation of so3_mlp at the 16 points sh.pos[] by the four waves of the workgroup (all must call it; 5 barriers).  Wave w keeps
// its slice of the packed stream — tiles 2 w, 2 w + 1 of the 16 k-steps, hi and lo = 256 registers — for the whole march: an evaluation
// reads no weight from memory.  Between layers the waves exchange the converted operands through LDS: wave w converts its 32 outputs
// (bias + ReLU + hi/lo split) = k-step w of the next layer; the encoding is split the same way (wave w: slots 4 (w & 1) .. + 3 of k-step
// w >> 1, four sines per lane).  All operand reads of a layer are issued ahead of its MFMAs, which alternate between the wave's two
// accumulators.  The result — Dense_4 (128 -> 3, fp32 VALU) as four 32-feature partials — is left in sh.out[wave][ray].
__device__ __forceinline__ void so3_eval_wg(So3Shared& sh, const uint4 (&wh)[kSo3KSteps][2], const uint4 (&wl)[kSo3KSteps][2], int wave, int lane) {
  const int c = lane & 15, g = lane >> 4;
  const float HALF_PI = 1.5707963705062866f;
  const f32x4 zero = {0, 0, 0, 0};
  constexpr float INV = 1.0f / SO3_WSCALE;
  const float4 pt = sh.pos[c];
  {   // annealed_pos_enc (model_utils.py:236-245): feature f = 6 d + 3 is_cos + c3, slots 4 (w & 1) .. + 3 of k-step w >> 1
    const int s = wave >> 1, j0 = 4 * (wave & 1);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = 32 * s + 8 * g + j0 + j;
      const bool in = f < 60;
      const int fc = in ? f : 0;
      const int d = fc / 6, jj = fc - 6 * d, c3 = jj >= 3 ? jj - 3 : jj;
      const float x = c3 == 0 ? pt.x : (c3 == 1 ? pt.y : pt.z);
      const float xb = fmul(x, (float)(1 << d));
      const float e = fmul(pe_sin(jj >= 3 ? fadd(xb, HALF_PI) : xb), sh.tail[4 + d]);     // ~1 ulp, 20 VALU ops (ocml sinf: ~100)
      v[j] = in ? e : 0.f;
    }
    uint32_t h0, l0, h1, l1;
    split2<true>(v[0], v[1], h0, l0); split2<true>(v[2], v[3], h1, l1);
    uint2* eh = (uint2*)&sh.enc[s][0][lane] + (wave & 1);
    uint2* el = (uint2*)&sh.enc[s][1][lane] + (wave & 1);
    *eh = make_uint2(h0, h1); *el = make_uint2(l0, l1);
  }
  __syncthreads();
  f32x4 a0 = zero, a1 = zero;
  auto kstep = [&](int ks, const uint4 bh, const uint4 bl) {        // hi*hi + hi*lo + lo*hi, the two tiles interleaved
    a0 = so3_mfma16(wh[ks][0], bh, a0); a1 = so3_mfma16(wh[ks][1], bh, a1);
    a0 = so3_mfma16(wh[ks][0], bl, a0); a1 = so3_mfma16(wh[ks][1], bl, a1);
    a0 = so3_mfma16(wl[ks][0], bh, a0); a1 = so3_mfma16(wl[ks][1], bh, a1);
  };
  auto enc_layer = [&](int ks0) {     // the encoding operands feed Dense_0 and, through the skip concat, Dense_3
    uint4 eh[2], el[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { eh[s] = sh.enc[s][0][lane]; el[s] = sh.enc[s][1][lane]; }
    RNERF_PIN();
#pragma unroll
    for (int s = 0; s < 2; ++s) kstep(ks0 + s, eh[s], el[s]);
  };
  // ReLU(acc / scale + bias) of this wave's 32 outputs = the operands of k-step w of the next layer
  auto hand_over = [&](int l) {
    const float4 b0 = *(const float4*)&sh.bias[l][32 * wave + 4 * g], b1 = *(const float4*)&sh.bias[l][32 * wave + 16 + 4 * g];
    float v[8];
    v[0] = fmaxf(fmaf(a0[0], INV, b0.x), 0.f); v[1] = fmaxf(fmaf(a0[1], INV, b0.y), 0.f); v[2] = fmaxf(fmaf(a0[2], INV, b0.z), 0.f); v[3] = fmaxf(fmaf(a0[3], INV, b0.w), 0.f);
    v[4] = fmaxf(fmaf(a1[0], INV, b1.x), 0.f); v[5] = fmaxf(fmaf(a1[1], INV, b1.y), 0.f); v[6] = fmaxf(fmaf(a1[2], INV, b1.z), 0.f); v[7] = fmaxf(fmaf(a1[3], INV, b1.w), 0.f);
    const So3Ops o = so3_split_ops(v);
    sh.x[l][wave][0][lane] = o.h; sh.x[l][wave][1][lane] = o.l;
  };
  auto layer = [&](int l, int ks0) {
    uint4 bh[4], bl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { bh[s] = sh.x[l][s][0][lane]; bl[s] = sh.x[l][s][1][lane]; }
    RNERF_PIN();
    a0 = zero; a1 = zero;
#pragma unroll
    for (int s = 0; s < 4; ++s) kstep(ks0 + s, bh[s], bl[s]);
  };
  enc_layer(0);
  hand_over(0);
  __syncthreads();
  layer(0, 2);
  hand_over(1);
  __syncthreads();
  layer(1, 6);
  hand_over(2);
  __syncthreads();
  layer(2, 10);
  enc_layer(14);                     // skip concat (model_utils.py:131-132)
  // Dense_4 (128 -> 3) on the VALU in fp32: this wave's 32 features, 8 per lane
  float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int f = 32 * wave + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
    const float v = fmaxf(fmaf(j < 4 ? a0[j & 3] : a1[j & 3], INV, sh.bias[3][f]), 0.f);
    const float4 k = *(const float4*)&sh.k4[f][0];
    o[0] = fmaf(v, k.x, o[0]); o[1] = fmaf(v, k.y, o[1]); o[2] = fmaf(v, k.z, o[2]);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) { o[q] = o[q] + __shfl_xor(o[q], 16); o[q] = o[q] + __shfl_xor(o[q], 32); }
  if (g == 0) sh.out[wave][c] = make_float4(o[0], o[1], o[2], 0.f);
  __syncthreads();
}

// pred_grad = a (cos(t) v + sin(t) e x v + (1 - cos(t)) (e . v) e),  e = raw / |raw|, v = g / |g| with safe norms (ior_utils.py:305-312)
__device__ __forceinline__ void so3_rotate(const float (&raw)[3], const float (&g)[3], float (&pred)[3]) {
  const float theta = fsqrt(fmaxf(fadd(fadd(fmul(raw[0], raw[0]), fmul(raw[1], raw[1])), fmul(raw[2], raw[2])), 1e-6f));
  const float e[3] = {fdiv(raw[0], theta), fdiv(raw[1], theta), fdiv(raw[2], theta)};
  const float a = fsqrt(fmaxf(fadd(fadd(fmul(g[0], g[0]), fmul(g[1], g[1])), fmul(g[2], g[2])), 1e-6f));
  const float v[3] = {fdiv(g[0], a), fdiv(g[1], a), fdiv(g[2], a)};
  const float ct = cosf(theta), st = sinf(theta);
  const float cr[3] = {fsub(fmul(e[1], v[2]), fmul(e[2], v[1])), fsub(fmul(e[2], v[0]), fmul(e[0], v[2])), fsub(fmul(e[0], v[1]), fmul(e[1], v[0]))};
  const float dot = fadd(fadd(fmul(e[0], v[0]), fmul(e[1], v[1])), fmul(e[2], v[2]));
  const float k = fmul(fsub(1.0f, ct), dot);
#pragma unroll
  for (int c = 0; c < 3; ++c) pred[c] = fmul(a, fadd(fadd(fmul(ct, v[c]), fmul(st, cr[c])), fmul(k, e[c])));
}

// G4: VoxMLP.__call__ for arbitrary points -> (n, grad n) and pred_grad
__global__ void __launch_bounds__(64) so3_query_kernel(const float4* __restrict__ table, GridParams gp, const float* __restrict__ params,
                                                       So3Window win, const float* __restrict__ pts, const float* __restrict__ cond, long long n,
                                                       float4* __restrict__ out4, float* __restrict__ pred_out) {
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  long long row = (long long)blockIdx.x * 32 + m;
  const bool ok = row < n;
  if (!ok) row = n - 1;
  const float px = pts[3 * row], py = pts[3 * row + 1], pz = pts[3 * row + 2];
  const float4 c = trilinear(table, gp, px, py, pz, nullptr);
  float raw[3], pred[3];
  so3_eval(params, px, py, pz, win, m, h, raw);
  // wrapper_grad_mlp (ior_utils.py:225-267) rotates a caller-supplied vector; VoxMLP.__call__ (:269-312) the looked-up gradient
  const float g[3] = {cond ? cond[3 * row] : c.y, cond ? cond[3 * row + 1] : c.z, cond ? cond[3 * row + 2] : c.w};
  so3_rotate(raw, g, pred);
  if (ok && h == 0) { out4[row] = c; pred_out[3 * row] = pred[0]; pred_out[3 * row + 1] = pred[1]; pred_out[3 * row + 2] = pred[2]; }
}

// E1/E2 with stage "all": one workgroup = 16 rays.  Wave 0 marches them exactly like march_kernel does — four lanes per ray (lane q owns
// coordinate q of the state and component (q + 1) & 3 of the table entries), the gathers of the next two nodes in flight from predicted
// cells — and at every node where any of its rays is inside the boundary shell (|grad n| > 1e-3) hands the 16 positions to LDS; the other
// three waves wait at that barrier, and all four evaluate so3_mlp together (so3_eval_wg).  Outside the shell a node costs what it costs in
// the radiance-stage march; inside, the chain is one evaluation of 96 v_mfma_f32_16x16x32_f16 per wave.
__global__ void __launch_bounds__(256, 1) march_all_kernel(const float4* __restrict__ table4, GridParams gp, const float* __restrict__ params,
                                                          const uint4* __restrict__ packed16, So3Window win, const float* __restrict__ origins, const float* __restrict__ viewdirs,
                                                          int B, float near, float step, int num_nodes, float4* __restrict__ path_pd4,
                                                          float4* __restrict__ path_dr4, float4* __restrict__ path_ior4,
                                                          // training record (nullable): raw direction + n per node, and the compacted list of
                                                          // (ray, node) pairs at which pred_grad was selected (the object's boundary shell)
                                                          float4* __restrict__ path_rdn4, int* __restrict__ pair_count, int pair_cap,
                                                          int2* __restrict__ pair_id, float4* __restrict__ pair_x4, float4* __restrict__ pair_g4,
                                                          int* __restrict__ pair_of_node, const int* __restrict__ ray_order) {
  __shared__ So3Shared sh;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  so3_shared_init(sh, params, win, threadIdx.x);
  uint4 wh[kSo3KSteps][2], wl[kSo3KSteps][2];
#pragma unroll
  for (int ks = 0; ks < kSo3KSteps; ++ks)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const uint4* __restrict__ blk = packed16 + (size_t)(8 * ks + 2 * wave + tt) * 128;
      wh[ks][tt] = blk[lane]; wl[ks][tt] = blk[64 + lane];
    }
  __syncthreads();
  if (wave != 0) {                    // the evaluation waves: wait for the marcher's next request (or the end of the march)
    for (;;) {
      __syncthreads();
      if (sh.done) return;
      so3_eval_wg(sh, wh, wl, wave, lane);
    }
  }
  // ---- the marching wave: march_kernel's node (march.hip), plus the so3 term
  const float* __restrict__ table = (const float*)table4;
  float* __restrict__ path_pd = (float*)path_pd4; float* __restrict__ path_dr = (float*)path_dr4;
  float* __restrict__ path_ior = (float*)path_ior4; float* __restrict__ path_rdn = (float*)path_rdn4;
  float* __restrict__ pair_x = (float*)pair_x4; float* __restrict__ pair_g = (float*)pair_g4;
  const int q = lane & 3, ray = lane >> 2;
  int r = blockIdx.x * 16 + ray;
  const bool live = r < B;
  if (!live) r = B - 1;               // surplus quads replay the last ray (same values to the same addresses); their pair records are suppressed
  if (ray_order) r = ray_order[r];    // the rays of a workgroup in the caller's (shell-coherent) order; records go to the ray's own index
  const int qc = q < 3 ? q : 0;
  const float nmin_q = qc == 0 ? gp.nminx : (qc == 1 ? gp.nminy : gp.nminz);
  const double rcp_q = 1.0 / (double)(qc == 0 ? gp.ndx : (qc == 1 ? gp.ndy : gp.ndz));
  const int dim_q = qc == 0 ? gp.dx : (qc == 1 ? gp.dy : gp.dz);
  const unsigned comp = (q + 1) & 3;
  const char* __restrict__ tabc = (const char*)table;
  const unsigned cofs = comp * 4u;
  const unsigned sa_q = gp.sa[qc], sb_q = gp.sb[qc];                // table addressing of this lane's axis, either layout (GridParams::sa / sb)
  const int hi_q = dim_q - 1;
  float d = q < 3 ? viewdirs[3 * r + qc] : 0.f;
  float p = q < 3 ? fadd(origins[3 * r + qc], fmul(near, d)) : 0.f;   // eikonal_utils.py:104-106
  float rt = near;
  struct Corners { float c[8]; int i0, i1; };     // 000 100 001 101 010 110 011 111 (xyz) + the indices they were gathered with
  Corners ca, cb, cc;
  auto gather = [&](int i0, int i1, Corners& o) {       // (the integer glue as in march.hip: clamp0 / floor_to_int / lerp_pk, common.h)
    o.i0 = i0; o.i1 = i1;
    const unsigned m0 = __umul24((unsigned)i0 >> 1, sa_q) + __umul24((unsigned)i0 & 1u, sb_q);
    const unsigned m1 = __umul24((unsigned)i1 >> 1, sa_q) + __umul24((unsigned)i1 & 1u, sb_q);
    const unsigned y0 = quad_bcast_i<1>(m0), y1 = quad_bcast_i<1>(m1);
    const unsigned z0 = quad_bcast_i<2>(m0) + cofs, z1 = quad_bcast_i<2>(m1) + cofs;
    const unsigned b00 = quad_bcast_i<0>(m0) + y0, b10 = quad_bcast_i<0>(m1) + y0, b01 = quad_bcast_i<0>(m0) + y1, b11 = quad_bcast_i<0>(m1) + y1;
    o.c[0] = *(const float*)(tabc + (b00 + z0)); o.c[1] = *(const float*)(tabc + (b10 + z0));
    o.c[2] = *(const float*)(tabc + (b00 + z1)); o.c[3] = *(const float*)(tabc + (b10 + z1));
    o.c[4] = *(const float*)(tabc + (b01 + z0)); o.c[5] = *(const float*)(tabc + (b11 + z0));
    o.c[6] = *(const float*)(tabc + (b01 + z1)); o.c[7] = *(const float*)(tabc + (b11 + z1));
  };
  auto predict = [&](float xq, Corners& o) {
    const int j = floor_to_int(xq);
    gather(clamp0(j, hi_q), clamp0(j + 1, hi_q), o);
  };
  float x_prev;
  {
    const float x = div_const(fsub(p, nmin_q), rcp_q);
    x_prev = div_const(fsub(fsub(p, fmul(step, d)), nmin_q), rcp_q);   // as if a vacuum step had led here
    predict(x, ca);
    predict(fadd(x, fsub(x, x_prev)), cb);
  }
  const size_t node_stride = 4 * (size_t)B;
  size_t rec = 4 * (size_t)r;                                           // float offset of this ray's record of the current node
  auto one_step = [&](int k, Corners& cn, Corners& nx) {
    // ---- VoxMLP._linear3 addressing (ior_utils.py:188-211): one coordinate per lane
    const float x = div_const(fsub(p, nmin_q), rcp_q);
    const float fx = floorf(x);
    const int i = (int)fx;
    const float t = fsub(x, fx);
    const int i0 = clamp0(i, hi_q), i1 = clamp0(i + 1, hi_q);
    if (__builtin_amdgcn_ballot_w64(i0 != cn.i0 || i1 != cn.i1) != 0) gather(i0, i1, cn);     // mispredicted somewhere in the wave
    const float dx = fsub(x, x_prev);
    predict(fadd(x, fadd(dx, dx)), nx);
    x_prev = x;
    const float xd = quad_bcast<0>(t), yd = quad_bcast<1>(t), zd = quad_bcast<2>(t);
    // ---- node record while the gathers are in flight (eikonal_utils.py:112-114), direction safe-normalised (math_utils.py:6-12)
    const float nrm = fsqrt(fmaxf(quad_sumsq3(d), 1e-6f));
    path_pd[rec + q] = q < 3 ? p : rt;
    path_dr[rec + q] = q < 3 ? fdiv(d, nrm) : 0.f;
    // ---- 7 lerps a*(1-t) + b*t (ior_utils.py:214-222)
    const f32x2_t wx = {fsub(1.0f, xd), xd}, wy = {fsub(1.0f, yd), yd}, wz = {fsub(1.0f, zd), zd};
    const float c00 = lerp_pk(cn.c[0], cn.c[1], wx);
    const float c01 = lerp_pk(cn.c[2], cn.c[3], wx);
    const float c10 = lerp_pk(cn.c[4], cn.c[5], wx);
    const float c11 = lerp_pk(cn.c[6], cn.c[7], wx);
    const float c0 = lerp_pk(c00, c10, wy);
    const float c1 = lerp_pk(c01, c11, wy);
    const float c = lerp_pk(c0, c1, wz);   // lanes 0..2: grad component q, lane 3: n
    if (path_ior) path_ior[rec + comp] = c;
    const float n = quad_bcast<3>(c);
    const float s = fdiv(step, n);
    const float np = fadd(p, fmul(s, d));                // needs the current direction only
    // |grad n| > 1e-3 (eikonal_utils.py:35): sqrt_rn(s) > f32(1e-3) <=> s > 0x358637BE, the largest f32 whose correctly rounded root is
    // still <= f32(1e-3) (sqrt is monotone): the same decision for every s without a root on the chain
    const float g2 = quad_sumsq3(c);
    const bool use = g2 > __uint_as_float(0x358637BEu);
    int idx = -1;
    if (path_rdn) {
      path_rdn[rec + q] = q < 3 ? d : n;
      if (use && q == 0 && live) idx = atomicAdd(pair_count, 1);      // consumed after the evaluation: its round trip hides behind the MLP
    }
    float pred = 0.f;
    if (__builtin_amdgcn_ballot_w64(use) != 0) {         // (a workgroup's rays are neighbours in the shell-coherent order)
      if (q < 3) ((float*)&sh.pos[ray])[q] = p;
      __syncthreads();
      so3_eval_wg(sh, wh, wl, 0, lane);
      const float4 r0 = sh.out[0][ray], r1 = sh.out[1][ray], r2 = sh.out[2][ray], r3 = sh.out[3][ray];
      const float o0 = q == 0 ? r0.x : (q == 1 ? r0.y : r0.z), o1 = q == 0 ? r1.x : (q == 1 ? r1.y : r1.z);
      const float o2 = q == 0 ? r2.x : (q == 1 ? r2.y : r2.z), o3 = q == 0 ? r3.x : (q == 1 ? r3.y : r3.z);
      const float raw = q < 3 ? ((o0 + o1) + (o2 + o3)) + sh.tail[qc] : 0.f;
      // pred_grad = a (cos(t) v + sin(t) e x v + (1 - cos(t)) (e . v) e),  e = raw / |raw|, v = g / |g| with safe norms (ior_utils.py:305-312)
      const float theta = fsqrt(fmaxf(quad_sumsq3(raw), 1e-6f));
      const float e = fdiv(raw, theta);
      const float a = fsqrt(fmaxf(g2, 1e-6f));
      const float v = q < 3 ? fdiv(c, a) : 0.f;
      float ct, st;
      pe_sincos(theta, st, ct);
      const float e0 = quad_bcast<0>(e), e1 = quad_bcast<1>(e), e2 = quad_bcast<2>(e);
      const float v0 = quad_bcast<0>(v), v1 = quad_bcast<1>(v), v2 = quad_bcast<2>(v);
      const float ea = q == 0 ? e1 : (q == 1 ? e2 : e0), eb = q == 0 ? e2 : (q == 1 ? e0 : e1);
      const float va = q == 0 ? v1 : (q == 1 ? v2 : v0), vb = q == 0 ? v2 : (q == 1 ? v0 : v1);
      const float cr = fsub(fmul(ea, vb), fmul(eb, va));
      const float dot = fadd(fadd(fmul(e0, v0), fmul(e1, v1)), fmul(e2, v2));
      const float kk = fmul(fsub(1.0f, ct), dot);
      pred = fmul(a, fadd(fadd(fmul(ct, v), fmul(st, cr)), fmul(kk, e)));
    }
    if (path_rdn) {
      idx = quad_bcast_i<0>(idx);
      if (idx >= pair_cap) idx = -1;
      if (idx >= 0) {
        if (q == 0) pair_id[idx] = make_int2(r, k);
        pair_x[4 * (size_t)idx + q] = q < 3 ? p : 0.f;
        pair_g[4 * (size_t)idx + q] = q < 3 ? c : 0.f;
      }
      if (q == 0 && live) pair_of_node[(size_t)k * B + r] = idx;
    }
    d = fadd(d, fmul(step, use ? pred : c));
    rt = fadd(rt, fsqrt(quad_sumsq3(fsub(p, np))));
    p = np;
    rec += node_stride;
  };
  int k = 0;
  for (; k + 2 < num_nodes; k += 3) {      // three nodes per trip: the corner register sets rotate instead of being copied
    one_step(k, ca, cc);
    one_step(k + 1, cb, ca);
    one_step(k + 2, cc, cb);
  }
  if (k < num_nodes) { one_step(k, ca, cc); ++k; }
  if (k < num_nodes) one_step(k, cb, ca);
  sh.done = 1;
  __syncthreads();
}

// ---- backward of the background MLP (exact fp32 on v_mfma_f32_32x32x2_f32, as the forward) --------------------------------
// dgrad chain: dX^T[k][row] = W[k][n] dY^T[n][row] with the accumulator registers as B operands; ReLU masks from the saved X_k.
__device__ __forceinline__ void small_prev_layer_T(f32x16 (&acc)[4], const f32x16 (&x)[4], const float* __restrict__ kern, int m, int h) {
  const float* __restrict__ kl = kern + m * 128 + 4 * h;           // step (ts, r): output feature n = 32 ts + 8 (r >> 2) + 4 h + (r & 3) held in x[ts][r]
  mfma_f32_stream4<64, 4>(acc, [&](int b, int t) { return *(const float4*)(kl + 32 * t * 128 + 32 * (b >> 2) + 8 * (b & 3)); },
                          [&](int st) { return x[st >> 4][st & 15]; });
}

// acc2[t2] += W[f = 32 t2 + m][n(h)] * x (contraction over the 128 outputs n held in x): the input-gradient GEMM of a small MLP, f < fin
__device__ __forceinline__ void small_input_T(f32x16 (&acc2)[2], const f32x16 (&x)[4], const float* __restrict__ kern, int fin, int m, int h) {
  mfma_f32_stream4<64, 2>(acc2, [&](int b, int t2) {
    const int f = 32 * t2 + m;
    const float4 w = *(const float4*)(kern + (f < fin ? f : fin - 1) * 128 + 32 * (b >> 2) + 8 * (b & 3) + 4 * h);
    return f < fin ? w : make_float4(0.f, 0.f, 0.f, 0.f);
  }, [&](int st) { return x[st >> 4][st & 15]; });
}

__global__ void __launch_bounds__(64) bkgd_dgrad_kernel(const float* __restrict__ params, const float* __restrict__ save,
                                                        const float* __restrict__ d_out, long long n, float pad_scale, float pad,
                                                        float* __restrict__ dy, float4* __restrict__ d_dirs) {
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  long long row = (long long)blockIdx.x * 32 + m;
  const bool ok = row < n;
  if (!ok) row = n - 1;
  const float* so = save + (size_t)n * (28 + 4 * 128) + (size_t)row * 3;
  float draw[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float sg = (so[c] + pad) / pad_scale;                       // out = sigmoid(raw) * (1+2p) - p   (rnerf/models.py:336-337)
    draw[c] = ok ? d_out[3 * row + c] * pad_scale * sg * (1.0f - sg) : 0.f;
  }
  float* dyraw = dy + (size_t)n * 4 * 128;
  if (ok && h == 0) { dyraw[4 * row] = draw[0]; dyraw[4 * row + 1] = draw[1]; dyraw[4 * row + 2] = draw[2]; dyraw[4 * row + 3] = 0.f; }
  auto Xk = [&](int k) -> const float* { return save + (size_t)n * 28 + (size_t)(k - 1) * n * 128 + (size_t)row * 128; };
  auto store_dy = [&](int k, const f32x16 (&xx)[4]) {
    if (ok) {
      float* dst = dy + (size_t)k * n * 128 + (size_t)row * 128;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(dst + 32 * t + 8 * g + 4 * h) = make_float4(xx[t][4 * g], xx[t][4 * g + 1], xx[t][4 * g + 2], xx[t][4 * g + 3]);
    }
  };
  auto mask_by = [&](int k, const f32x16 (&a)[4], f32x16 (&xx)[4]) {
    const float* xs = Xk(k);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = *(const float4*)(xs + 32 * t + 8 * g + 4 * h);
        xx[t][4 * g] = v.x > 0.f ? a[t][4 * g] : 0.f; xx[t][4 * g + 1] = v.y > 0.f ? a[t][4 * g + 1] : 0.f;
        xx[t][4 * g + 2] = v.z > 0.f ? a[t][4 * g + 2] : 0.f; xx[t][4 * g + 3] = v.w > 0.f ? a[t][4 * g + 3] : 0.f;
      }
  };
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 acc[4], x[4];
  // through Dense_4 (128 -> 3) on the VALU
  const float* __restrict__ k4 = params + bkgd_koff(4);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
      acc[t][r] = draw[0] * k4[f * 3] + draw[1] * k4[f * 3 + 1] + draw[2] * k4[f * 3 + 2];
    }
  mask_by(4, acc, x);
  store_dy(3, x);                                   // dY_3
  f32x16 denc[2] = {zero, zero};                    // d loss / d pos_enc(dir) (stage "all": the direction is a function of the so3 parameters)
  if (d_dirs) small_input_T(denc, x, params + bkgd_koff(3) + 128 * 128, 27, m, h);     // the skip-concat inputs of Dense_3
#pragma unroll 1
  for (int k = 3; k >= 1; --k) {                    // through Dense_k (only the first 128 input rows of Dense_3 carry gradient)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = zero;
    small_prev_layer_T(acc, x, params + (k == 3 ? bkgd_koff(3) : (k == 2 ? bkgd_koff(2) : bkgd_koff(1))), m, h);
    mask_by(k, acc, x);
    store_dy(k - 1, x);                             // dY_{k-1}
  }
  if (d_dirs) {
    small_input_T(denc, x, params + bkgd_koff(0), 27, m, h);
    const float* v = save + (size_t)row * 28;       // features 0..2 of the saved encoding = the direction itself
    const float dcv[3] = {v[0], v[1], v[2]};
    float gd3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = (r & 3) + 8 * (r >> 2) + 4 * h;                  // [d(3) | sin(2^k d)(12) | sin(2^k d + pi/2)(12)]
      if (f < 3) gd3[f] += denc[0][r];
      else if (f < 27) {
        const int q = (f - 3) % 12, is_cos = (f - 3) / 12, d = q / 3, c = q % 3;
        const float xb = fmul(dcv[c], (float)(1 << d));
        gd3[c] = fmaf(denc[0][r], (float)(1 << d) * cosf(is_cos ? fadd(xb, 1.5707963705062866f) : xb), gd3[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) gd3[c] += __shfl_xor(gd3[c], 32);
    if (ok && h == 0) d_dirs[row] = make_float4(gd3[0], gd3[1], gd3[2], 0.f);
  }
}

// wgrad: dW[k][n] = sum_rows X[row][k] dY[row][n]; A = X (lane = k, the half picks one of 2 rows), B = dY: row-major fp32 needs
// no transposition for the K=2 MFMA.  One launch covers every (Dense_k block, k-tile) "unit"; a workgroup = 256 rows (4 waves x 64
// rows, summed through LDS) of one unit and writes its partial in flat-gradient layout; bkgd_wgrad_reduce_kernel adds them up.
struct BkgdUnit { int xk, ldx, kin, kt, dk, ldy, nout, goff, out_dim, boff; };   // xk: 0 = enc, k = X_k; dk: 0..3 = dY_k, 4 = d raw
__constant__ BkgdUnit kBkgdUnits[18] = {
    {0, 28, 27, 0, 0, 128, 128, bkgd_koff(0), 128, bkgd_boff(0)},
    {1, 128, 128, 0, 1, 128, 128, bkgd_koff(1), 128, bkgd_boff(1)}, {1, 128, 128, 1, 1, 128, 128, bkgd_koff(1), 128, -1},
    {1, 128, 128, 2, 1, 128, 128, bkgd_koff(1), 128, -1},            {1, 128, 128, 3, 1, 128, 128, bkgd_koff(1), 128, -1},
    {2, 128, 128, 0, 2, 128, 128, bkgd_koff(2), 128, bkgd_boff(2)}, {2, 128, 128, 1, 2, 128, 128, bkgd_koff(2), 128, -1},
    {2, 128, 128, 2, 2, 128, 128, bkgd_koff(2), 128, -1},            {2, 128, 128, 3, 2, 128, 128, bkgd_koff(2), 128, -1},
    {3, 128, 128, 0, 3, 128, 128, bkgd_koff(3), 128, bkgd_boff(3)}, {3, 128, 128, 1, 3, 128, 128, bkgd_koff(3), 128, -1},
    {3, 128, 128, 2, 3, 128, 128, bkgd_koff(3), 128, -1},            {3, 128, 128, 3, 3, 128, 128, bkgd_koff(3), 128, -1},
    {0, 28, 27, 0, 3, 128, 128, bkgd_koff(3) + 128 * 128, 128, -1},
    {4, 128, 128, 0, 4, 4, 3, bkgd_koff(4), 3, bkgd_boff(4)},       {4, 128, 128, 1, 4, 4, 3, bkgd_koff(4), 3, -1},
    {4, 128, 128, 2, 4, 4, 3, bkgd_koff(4), 3, -1},                  {4, 128, 128, 3, 4, 4, 3, bkgd_koff(4), 3, -1}};

// the so3 MLP (60 -> 128 x 4 with the 60 inputs concatenated before Dense_3 -> 3, rnerf/ior_utils.py:148-152) has the same unit structure
__constant__ BkgdUnit kSo3Units[20] = {
    {0, 60, 60, 0, 0, 128, 128, so3_koff(0), 128, so3_boff(0)},      {0, 60, 60, 1, 0, 128, 128, so3_koff(0), 128, -1},
    {1, 128, 128, 0, 1, 128, 128, so3_koff(1), 128, so3_boff(1)},    {1, 128, 128, 1, 1, 128, 128, so3_koff(1), 128, -1},
    {1, 128, 128, 2, 1, 128, 128, so3_koff(1), 128, -1},             {1, 128, 128, 3, 1, 128, 128, so3_koff(1), 128, -1},
    {2, 128, 128, 0, 2, 128, 128, so3_koff(2), 128, so3_boff(2)},    {2, 128, 128, 1, 2, 128, 128, so3_koff(2), 128, -1},
    {2, 128, 128, 2, 2, 128, 128, so3_koff(2), 128, -1},             {2, 128, 128, 3, 2, 128, 128, so3_koff(2), 128, -1},
    {3, 128, 128, 0, 3, 128, 128, so3_koff(3), 128, so3_boff(3)},    {3, 128, 128, 1, 3, 128, 128, so3_koff(3), 128, -1},
    {3, 128, 128, 2, 3, 128, 128, so3_koff(3), 128, -1},             {3, 128, 128, 3, 3, 128, 128, so3_koff(3), 128, -1},
    {0, 60, 60, 0, 3, 128, 128, so3_koff(3) + 128 * 128, 128, -1},   {0, 60, 60, 1, 3, 128, 128, so3_koff(3) + 128 * 128, 128, -1},
    {4, 128, 128, 0, 4, 4, 3, so3_koff(4), 3, so3_boff(4)},          {4, 128, 128, 1, 4, 4, 3, so3_koff(4), 3, -1},
    {4, 128, 128, 2, 4, 4, 3, so3_koff(4), 3, -1},                   {4, 128, 128, 3, 4, 4, 3, so3_koff(4), 3, -1}};
template <int KIND> struct SmallNet {
  static constexpr int ENC_LD = KIND == 0 ? 28 : 60, NPARAMS = KIND == 0 ? RNERF_BKGDMLP_PARAMS : RNERF_SO3MLP_PARAMS, UNITS = KIND == 0 ? 18 : 20;
  static constexpr int CHUNKS_PER_WG = KIND == 0 ? 1 : 4;      // 256-row chunks summed by one workgroup of the weight-gradient kernel (one partial each)
};

template <int KIND>
__global__ void __launch_bounds__(256) bkgd_wgrad_kernel(const float* __restrict__ save, const float* __restrict__ dy, long long n,
                                                         float* __restrict__ partial) {
  __shared__ float red[4][4][1024 + 32];            // [wave][n-tile][acc reg * 64 + lane] (+ the bias row)
  // 1-D grid, XCD-aware: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so workgroup b = 8 slot + xcd takes
  // unit slot % UNITS of chunk 8 (slot / UNITS) + xcd: the UNITS workgroups that read the same 256 rows of dY (each layer's dY by its
  // 4-5 k-tiles) run back to back on ONE XCD and share them in its L2 (with (chunk, unit) as grid (x, y) they were `chunks` workgroups
  // apart: every re-read went to memory).
  // CPW 256-row chunks per workgroup (so3: 4 — its 208 k pair rows made 813 partials of 65 411 floats, 213 MB written and read again by the
  // reduction: 0.56 + 0.10 ms of the stage-all step's tail; the background MLP keeps 1: a few dozen chunks, and the
  // default step's bits stay what they were)
  constexpr int UNITS = SmallNet<KIND>::UNITS, CPW = SmallNet<KIND>::CHUNKS_PER_WG;
  const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const int chunk = 8 * (slot / UNITS) + xcd, unit = slot % UNITS;      // (the index of the partial: CPW consecutive 256-row chunks)
  if ((long long)chunk * 256 * CPW >= n) return;
  const BkgdUnit u = KIND == 0 ? kBkgdUnits[unit] : kSo3Units[unit];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
  const float* __restrict__ X = u.xk == 0 ? save : save + (size_t)n * SmallNet<KIND>::ENC_LD + (size_t)(u.xk - 1) * n * 128;
  const float* __restrict__ dY = dy + (size_t)u.dk * n * 128;
  const int k = 32 * u.kt + m;
  const bool bias = u.boff >= 0;
  const int NT = (u.nout + 31) / 32;
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 acc[4] = {zero, zero, zero, zero};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};               // bias row: this lane's share of sum_rows dY[row][its column] (plain VALU adds)
  long long r0 = (long long)chunk * 256 * CPW + wave * 64;
  // Operand loads: unconditional, from clamped addresses, a BATCH of 8 row pairs ahead of the MFMAs that consume them, and made opaque
  // (empty asm on the loaded registers) before the row / column predicates are applied.  Without the last step hipcc turns
  // `ok ? load : 0` back into a branch around the load followed by s_waitcnt vmcnt(0): two exposed round trips per 4 MFMAs
  // (the so3 instance ran at a sixth of the matrix rate: 1.06 ms for 208 k rows).  The n-tile count and the bias row are compile-time
  // instances.  With 4 n-tiles (a 128-wide dY) n-tile nt is the columns {4 m + nt}: one 16-byte load per lane and row feeds all four MFMAs.
  const int kc = k < u.kin ? k : u.kin - 1;
  const bool kin_ok = k < u.kin;
  auto rows = [&](auto nt_c, auto bias_c) {
    constexpr int NTC = decltype(nt_c)::value;
    constexpr bool BIAS = decltype(bias_c)::value;
    constexpr int BS = 8, NB = 32 / BS;
    float av[2][BS];
    float4 bv[2][BS];
    const int mc = m < u.nout ? m : u.nout - 1;
    auto fetch = [&](int bi, int buf) {
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const long long rr = r0 + 2 * (bi * BS + j) + h;
        const size_t rc = (size_t)(rr < n ? rr : n - 1);
        av[buf][j] = X[rc * u.ldx + kc];
        if constexpr (NTC == 4) bv[buf][j] = *(const float4*)(dY + rc * 128 + 4 * m);
        else bv[buf][j].x = dY[rc * u.ldy + mc];
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      if (bi + 1 < NB) fetch(bi + 1, (bi + 1) & 1);
      RNERF_PIN();
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const bool ok = r0 + 2 * (bi * BS + j) + h < n;
        float a = av[bi & 1][j];
        float4 q = bv[bi & 1][j];
        if constexpr (NTC == 4) asm volatile("" : "+v"(a), "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w));
        else asm volatile("" : "+v"(a), "+v"(q.x));
        a = (ok && kin_ok) ? a : 0.f;
        if constexpr (NTC == 4) {
          const float b[4] = {ok ? q.x : 0.f, ok ? q.y : 0.f, ok ? q.z : 0.f, ok ? q.w : 0.f};
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[nt], acc[nt], 0, 0, 0);
            if constexpr (BIAS) bsum[nt] += b[nt];
          }
        } else {
          const float b = (ok && m < u.nout) ? q.x : 0.f;
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
          if constexpr (BIAS) bsum[0] += b;
        }
      }
      RNERF_PIN();
    }
  };
  using I1 = std::integral_constant<int, 1>; using I4 = std::integral_constant<int, 4>;
#pragma unroll 1
  for (int cc = 0; cc < CPW; ++cc, r0 += 256) {
    if (cc > 0 && (long long)(chunk * CPW + cc) * 256 >= n) break;      // (workgroup-uniform)
    if (NT == 4) { if (bias) rows(I4{}, std::true_type{}); else rows(I4{}, std::false_type{}); }
    else { if (bias) rows(I1{}, std::true_type{}); else rows(I1{}, std::false_type{}); }
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
    if (nt < NT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave][nt][r * 64 + lane] = acc[nt][r];
      const float bs = bsum[nt] + __shfl_xor(bsum[nt], 32);        // even + odd rows
      if (h == 0) red[wave][nt][1024 + m] = bs;
    }
  __syncthreads();
  const int nt = wave;                                // wave w sums n-tile w over the 4 waves
  if (nt < NT) {
    float* pg = partial + (size_t)chunk * SmallNet<KIND>::NPARAMS;
    const int nn = NT == 4 ? 4 * m + nt : m;          // the column this lane's accumulator entries belong to
    if (nn < u.nout) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ki = 32 * u.kt + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = red[0][nt][r * 64 + lane] + red[1][nt][r * 64 + lane] + red[2][nt][r * 64 + lane] + red[3][nt][r * 64 + lane];
        if (ki < u.kin) pg[u.goff + ki * u.out_dim + nn] = v;
      }
      if (bias && h == 0) pg[u.boff + nn] = red[0][nt][1024 + m] + red[1][nt][1024 + m] + red[2][nt][1024 + m] + red[3][nt][1024 + m];
    }
  }
}

// The same wgrad as a kernel that can be CO-RESIDENT with the NerfMLP wgrad (which keeps 64-80 registers per SIMD free and nearly all
// of the LDS for itself): one wave per (256-row chunk, unit, pair of n-tiles), at most 80 registers, no LDS — the partial of a chunk is
// written by the wave that computed it.  It is slower than bkgd_wgrad_kernel when it runs alone (no unrolling headroom), which does
// not matter where it is used: on a side stream beside the 2 ms NerfMLP wgrad, off the step's critical path (csrc/pipeline.hip).
template <int KIND>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(40)))
bkgd_wgrad_co_kernel(const float* __restrict__ save, const float* __restrict__ dy, long long n, float* __restrict__ partial) {
  const BkgdUnit u = KIND == 0 ? kBkgdUnits[blockIdx.y] : kSo3Units[blockIdx.y];
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  const int NT = (u.nout + 31) / 32;
  const int np = blockIdx.z;                          // n-tiles {2 np, 2 np + 1} = columns {4 m + 2 np, 4 m + 2 np + 1}
  if (NT == 1 && np != 0) return;
  const float* __restrict__ X = u.xk == 0 ? save : save + (size_t)n * SmallNet<KIND>::ENC_LD + (size_t)(u.xk - 1) * n * 128;
  const float* __restrict__ dY = dy + (size_t)u.dk * n * 128;
  const int k = 32 * u.kt + m;
  const int kc = k < u.kin ? k : u.kin - 1;
  const bool bias = u.boff >= 0;
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 acc0 = zero, acc1 = zero;
  float bs0 = 0.f, bs1 = 0.f;
  const long long r0 = (long long)blockIdx.x * 256;
#pragma unroll 4
  for (int i = 0; i < 128; ++i) {
    const long long rr = r0 + 2 * i + h;
    const bool ok = rr < n;
    const size_t rc = (size_t)(ok ? rr : n - 1);
    const float av = X[rc * u.ldx + kc];
    const float a = (ok && k < u.kin) ? av : 0.f;
    if (NT == 4) {
      const float2 bv = *(const float2*)(dY + rc * 128 + 4 * m + 2 * np);
      const float b0 = ok ? bv.x : 0.f, b1 = ok ? bv.y : 0.f;
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
      bs0 += b0; bs1 += b1;
    } else {
      const float bv = dY[rc * u.ldy + (m < u.nout ? m : u.nout - 1)];
      const float b0 = (ok && m < u.nout) ? bv : 0.f;
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      bs0 += b0;
    }
  }
  float* pg = partial + (size_t)blockIdx.x * SmallNet<KIND>::NPARAMS;
  bs0 += __shfl_xor(bs0, 32); bs1 += __shfl_xor(bs1, 32);
  const int nn0 = NT == 4 ? 4 * m + 2 * np : m;
  if (nn0 < u.nout) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ki = 32 * u.kt + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (ki < u.kin) {
        pg[u.goff + ki * u.out_dim + nn0] = acc0[r];
        if (NT == 4) pg[u.goff + ki * u.out_dim + nn0 + 1] = acc1[r];
      }
    }
    if (bias && h == 0) {
      pg[u.boff + nn0] = bs0;
      if (NT == 4) pg[u.boff + nn0 + 1] = bs1;
    }
  }
}

template <int NPARAMS>
__global__ void __launch_bounds__(256) bkgd_wgrad_reduce_kernel(const float* __restrict__ partial, int chunks, float* __restrict__ grads) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= NPARAMS) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = 0;
  for (; c + 4 <= chunks; c += 4) {
    s0 += partial[(size_t)c * NPARAMS + e]; s1 += partial[(size_t)(c + 1) * NPARAMS + e];
    s2 += partial[(size_t)(c + 2) * NPARAMS + e]; s3 += partial[(size_t)(c + 3) * NPARAMS + e];
  }
  for (; c < chunks; ++c) s0 += partial[(size_t)c * NPARAMS + e];
  grads[e] += (s0 + s1) + (s2 + s3);
}

#include "ior_train_kernels.inc"

}  // namespace rnerf

namespace rnerf {
// csrc/mlp_f32.hip: the exact-fp32 arbiter (RNERF_PREC_F32)
int launch_fwd_f32(const void* packed, const float* rows_pd, const float* rows_dr, const int32_t* node_of_sample, int32_t B, long long total_rows,
                   float* out_raw, hipStream_t st);
}

using namespace rnerf;

// f16f8 packs TWO streams: its own, and behind it (256-byte aligned) the f16x3 stream a launch falls back to when a weight is out of its range
constexpr size_t kF8Fallback = (Prec<RNERF_PREC_F16F8>::PACKED_BYTES + 255) & ~(size_t)255;

// Every f16-based EVALUATION stream is followed (256-byte aligned) by a bf16x3 stream of the same weights: the range-safe second pass of
// rnerf_nerfmlp_forward (nerfmlp_fwd_kernel's REDO mode) recomputes, in fp32's exponent range, the rows the f16 pass returned as NaN.
constexpr size_t kX3Bytes = (Prec<RNERF_PREC_F16X3>::PACKED_BYTES + 255) & ~(size_t)255;
constexpr size_t kSafeBytes = Prec<RNERF_PREC_BF16X3>::PACKED_BYTES;
constexpr size_t kF16Bytes = (Prec<RNERF_PREC_F16>::PACKED_BYTES + 255) & ~(size_t)255;
static size_t safe_stream_offset(int precision) {      // offset of that bf16x3 stream in a packed buffer of `precision` (0: it has none)
  switch (precision) {
    case RNERF_PREC_F16X3:
    case RNERF_PREC_F16X2: return kX3Bytes;
    case RNERF_PREC_F16F8: return kF8Fallback + kX3Bytes;
    case RNERF_PREC_F16: return kF16Bytes;
    default: return 0;
  }
}

#define RNERF_TRY_(expr) do { int rc_ = (expr); if (rc_ != RNERF_OK) return rc_; } while (0)

static bool prec_ok(int p) {
  return p == RNERF_PREC_F32 || p == RNERF_PREC_F16X3 || p == RNERF_PREC_BF16X3 || p == RNERF_PREC_F16 || p == RNERF_PREC_BF16 || p == RNERF_PREC_F16X2 || p == RNERF_PREC_F16F8;
}

extern "C" size_t rnerf_nerfmlp_packed_bytes(int precision) {
  switch (precision) {
    case RNERF_PREC_F32: return (size_t)RNERF_NERFMLP_PARAMS * sizeof(float);  // the flat fp32 buffer itself
    case RNERF_PREC_F16X3:
    case RNERF_PREC_F16X2: return kX3Bytes + kSafeBytes;                      // f16x2 reads the f16x3 stream; + the range-safe bf16x3 stream
    case RNERF_PREC_F16F8: return kF8Fallback + kX3Bytes + kSafeBytes;        // its own stream, the f16x3 stream it falls back to, the bf16x3 stream
    case RNERF_PREC_BF16X3: return Prec<RNERF_PREC_BF16X3>::PACKED_BYTES;
    case RNERF_PREC_F16: return kF16Bytes + kSafeBytes;                       // the single-pass stream + the range-safe bf16x3 stream
    case RNERF_PREC_BF16: return Prec<RNERF_PREC_BF16>::PACKED_BYTES;
    default: set_error("rnerf_nerfmlp_packed_bytes: unsupported precision %d", precision); return 0;
  }
}

namespace rnerf {
void* nerfmlp_dgrad_scale_ref(int backward, void* dy, int64_t rows);      // (defined with the dgrad launcher below)
// the range flag(s) of a packed stream: 4 floats behind the aux block of every f16 stream (f16f8: its own and its f16x3 fallback's)
static int pack_flag_regions(int precision, void* packed, void** out) {
  auto at = [&](size_t off) { return (void*)((char*)packed + off + AUX_FLAG * sizeof(float)); };
  switch (precision) {
    case RNERF_PREC_F16X3:
    case RNERF_PREC_F16X2: out[0] = at(Prec<RNERF_PREC_F16X3>::STREAM_BYTES); return 1;
    case RNERF_PREC_F16F8: out[0] = at(Prec<RNERF_PREC_F16F8>::STREAM_BYTES); out[1] = at(kF8Fallback + Prec<RNERF_PREC_F16X3>::STREAM_BYTES); return 2;
    case RNERF_PREC_F16: out[0] = at(Prec<RNERF_PREC_F16>::STREAM_BYTES); return 1;
    default: return 0;
  }
}
struct FlagRegions { float4* p[12]; };
__global__ void __launch_bounds__(64) zero_flags_kernel(FlagRegions r, int n) {
  if ((int)threadIdx.x < n) *r.p[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
}
// Zeroes, in ONE launch, every 16-byte accumulator / flag word a training step needs cleared before its kernels run: the range flags of
// the operand streams it packs (`packed`, `count` <= 2) and the row-scale reference of each dgrad's dY buffer (`dy` / `dy_rows`, see
// launch_dgrad).  As hipMemsetAsync calls inside the producers each was a fill kernel of its own, ~6 us apiece with its launch gap, on
// the main stream's dependent chain (round 4; the env-map sum and sum theta^2 became fixed-order reductions that need no clearing).
int nerfmlp_step_zero(int precision, void* const* packed, int count, int backward, void* const* dy, const int64_t* dy_rows, int dy_count, hipStream_t st) {
  FlagRegions r;
  int n = 0;
  for (int i = 0; i < count && i < 2; ++i) { void* o[2]; const int k = pack_flag_regions(precision, packed[i], o); for (int q = 0; q < k; ++q) r.p[n++] = (float4*)o[q]; }
  for (int i = 0; i < dy_count && i < 4; ++i) { void* f = nerfmlp_dgrad_scale_ref(backward, dy[i], dy_rows[i]); if (f) r.p[n++] = (float4*)f; }
  if (n == 0) return RNERF_OK;
  hipLaunchKernelGGL(zero_flags_kernel, dim3(1), dim3(64), 0, st, r, n);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

// zero_flags = false: the caller has zeroed the stream's range flags on `stream` already (nerfmlp_pack_zero_flags)
// with_safe = false: leave the range-safe bf16x3 stream of an f16-based buffer unwritten (the training step's own streams: the training
// forward has no second pass — a row out of range reaches the non-finite-gradient count instead)
int nerfmlp_pack_impl(const float* params, int precision, void* packed, bool zero_flags, hipStream_t st, bool with_safe) {
  const int threads = kTotalBlocks * 64, block = 256, grid = (threads + block - 1) / block;
  if (zero_flags) {
    void* o[2];
    const int k = pack_flag_regions(precision, packed, o);
    for (int q = 0; q < k; ++q) RNERF_CHECK_HIP(hipMemsetAsync(o[q], 0, 4 * sizeof(float), st));
  }
  switch (precision) {
    case RNERF_PREC_F32:
      RNERF_CHECK_HIP(hipMemcpyAsync(packed, params, (size_t)RNERF_NERFMLP_PARAMS * sizeof(float), hipMemcpyDeviceToDevice, st));
      return RNERF_OK;
    case RNERF_PREC_F16X3:
    case RNERF_PREC_F16X2: hipLaunchKernelGGL(nerfmlp_pack_kernel<RNERF_PREC_F16X3>, dim3(grid), dim3(block), 0, st, params, (char*)packed); break;
    case RNERF_PREC_F16F8:
      hipLaunchKernelGGL(nerfmlp_pack_kernel<RNERF_PREC_F16F8>, dim3(grid), dim3(block), 0, st, params, (char*)packed);
      hipLaunchKernelGGL(nerfmlp_pack_kernel<RNERF_PREC_F16X3>, dim3(grid), dim3(block), 0, st, params, (char*)packed + kF8Fallback);
      break;
    case RNERF_PREC_F16: hipLaunchKernelGGL(nerfmlp_pack_kernel<RNERF_PREC_F16>, dim3(grid), dim3(block), 0, st, params, (char*)packed); break;
    case RNERF_PREC_BF16X3: hipLaunchKernelGGL(nerfmlp_pack_kernel<RNERF_PREC_BF16X3>, dim3(grid), dim3(block), 0, st, params, (char*)packed); break;
    default: hipLaunchKernelGGL(nerfmlp_pack_kernel<RNERF_PREC_BF16>, dim3(grid), dim3(block), 0, st, params, (char*)packed); break;
  }
  if (with_safe && safe_stream_offset(precision) != 0)
    hipLaunchKernelGGL(nerfmlp_pack_kernel<RNERF_PREC_BF16X3>, dim3(grid), dim3(block), 0, st, params, (char*)packed + safe_stream_offset(precision));
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}
}  // namespace rnerf

extern "C" int rnerf_nerfmlp_pack(const float* params, int precision, void* packed, void* stream) {
  RNERF_CHECK_ARG(params && packed, "rnerf_nerfmlp_pack: null pointer");
  RNERF_CHECK_ARG(prec_ok(precision), "rnerf_nerfmlp_pack: unsupported precision %d", precision);
  RNERF_CHECK_ARG(((uintptr_t)packed & 15) == 0, "rnerf_nerfmlp_pack: packed must be 16-byte aligned");
  return nerfmlp_pack_impl(params, precision, packed, true, (hipStream_t)stream, true);
}

// saved operands + masks of `padded` rows; behind them SAVE_QUEUE_BYTES for the dynamic tile queue of a capped training forward
constexpr size_t SAVE_QUEUE_BYTES = 8192;       // counter + one slot per workgroup (<= 2047 CUs)
static size_t save_payload_bytes(long long padded, bool with_lo) {
  return (size_t)(SAVE_TOTAL + (with_lo ? SAVE_SLOTS : 0)) * (size_t)padded * 2 * sizeof(uint4);
}

static int mlp_debug_flags() {
  static int v = -1;
  if (v < 0) { const char* e = RNERF_ENV("RNERF_MLP_DEBUG"); v = e ? atoi(e) : 0; }
  return v;
}

template <int PREC, int DBG, int TRAIN = 0>
static int launch_fwd_dbg(const void* packed, const float* rows_pd, const float* rows_dr, const int32_t* node_of_sample, int32_t B,
                          long long total_rows, float* out_raw, hipStream_t st, void* save = nullptr, int max_wg = 0, const float* gate = nullptr,
                          int gate_skip_if = 0);

template <int PREC>
static int launch_fwd(const void* packed, const float* rows_pd, const float* rows_dr, const int32_t* node_of_sample, int32_t B,
                      long long total_rows, float* out_raw, hipStream_t st, int max_wg, const float* gate = nullptr, int gate_skip_if = 0) {
#ifdef RNERF_MLP_ABLATE
  switch (mlp_debug_flags()) {
    case 1: return launch_fwd_dbg<PREC, 1>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 2: return launch_fwd_dbg<PREC, 2>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 3: return launch_fwd_dbg<PREC, 3>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 4: return launch_fwd_dbg<PREC, 4>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 5: return launch_fwd_dbg<PREC, 5>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 8: return launch_fwd_dbg<PREC, 8>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 13: return launch_fwd_dbg<PREC, 13>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 16: return launch_fwd_dbg<PREC, 16>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 48: return launch_fwd_dbg<PREC, 48>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 128: return launch_fwd_dbg<PREC, 128>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 64: return launch_fwd_dbg<PREC, 64>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 32: return launch_fwd_dbg<PREC, 32>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 29: return launch_fwd_dbg<PREC, 29>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    case 256: return launch_fwd_dbg<PREC, 256>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg);
    default: break;
  }
#endif
  return launch_fwd_dbg<PREC, 0>(packed, rows_pd, rows_dr, node_of_sample, B, total_rows, out_raw, st, nullptr, max_wg, gate, gate_skip_if);
}

template <int PREC, int DBG, int TRAIN>
static int launch_fwd_dbg(const void* packed, const float* rows_pd, const float* rows_dr, const int32_t* node_of_sample, int32_t B,
                      long long total_rows, float* out_raw, hipStream_t st, void* save, int max_wg, const float* gate, int gate_skip_if) {
  using PP = Prec<PREC>;
  const int n_tiles = (int)((total_rows + 255) / 256);
  int dev = 0, cus = 0;
  RNERF_CHECK_HIP(hipGetDevice(&dev));
  RNERF_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int lim = (max_wg > 0 && max_wg < cus) ? max_wg : cus;
  // Few rows: 128-row tiles (one m-tile per wave) when all of them fit ONE round of workgroups — a tile's time is the serial work of its
  // waves, so a launch that cannot fill the chip with 256-row tiles (a 512-ray shard's coarse level: 128 tiles on 256 CUs; both levels of a
  // 128-ray one) finishes sooner on twice as many CUs with half the work each.  (Beyond one round the 256-row tiles' better reuse of the
  // weight stream wins.)  RNERF_FWD_HALF_TILES=0 switches it off (A/B).
  if constexpr (DBG == 0 && (PREC == RNERF_PREC_F16X3 || PREC == RNERF_PREC_F16F8)) {
    static const bool half_ok = [] { const char* e = RNERF_ENV("RNERF_FWD_HALF_TILES"); return !(e && e[0] == '0'); }();
    if (half_ok && 2 * n_tiles <= lim) {
      const size_t lds1 = 2 * (size_t)PP::SLAB;
      static DeviceOnce attr1_set;
      if (attr1_set.need()) {
        RNERF_CHECK_HIP(hipFuncSetAttribute((const void*)nerfmlp_fwd_kernel<PREC, 0, TRAIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        attr1_set.set();
      }
      hipLaunchKernelGGL((nerfmlp_fwd_kernel<PREC, 0, TRAIN, true>), dim3(2 * n_tiles), dim3(256), lds1, st, (const char*)packed, (const float4*)rows_pd,
                         (const float4*)rows_dr, node_of_sample, B, total_rows, 2 * n_tiles, (float4*)out_raw, (uint4*)save, (long long)n_tiles * 256,
                         (int*)nullptr, gate, gate_skip_if);
      RNERF_CHECK_LAUNCH();
      return RNERF_OK;
    }
  }
  const int grid = n_tiles < lim ? n_tiles : lim;
  const size_t lds = 2 * (size_t)PP::SLAB + 4 * 32768;
  static DeviceOnce attr_set;
  if (attr_set.need()) {
    RNERF_CHECK_HIP(hipFuncSetAttribute((const void*)nerfmlp_fwd_kernel<PREC, DBG, TRAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.set();
  }
  int* tileq = nullptr;
  if (TRAIN != 0 && grid < cus && grid < n_tiles) {      // capped training forward: dynamic tile queue in the tail of the save buffer
    tileq = (int*)((char*)save + save_payload_bytes((long long)n_tiles * 256, TRAIN >= 2));
    RNERF_CHECK_HIP(hipMemsetAsync(tileq, 0, sizeof(int), st));
  }
  hipLaunchKernelGGL((nerfmlp_fwd_kernel<PREC, DBG, TRAIN>), dim3(grid), dim3(256), lds, st, (const char*)packed, (const float4*)rows_pd,
                     (const float4*)rows_dr, node_of_sample, B, total_rows, n_tiles, (float4*)out_raw, (uint4*)save,
                     (long long)n_tiles * 256, tileq, gate, gate_skip_if);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

extern "C" int rnerf_nerfmlp_forward(const void* packed, int precision, const float* rows_pd, const float* rows_dr,
                                     const int32_t* node_of_sample, int32_t S, int32_t B, float* out_raw, int32_t max_workgroups,
                                     void* stream) {
  RNERF_CHECK_ARG(packed && rows_pd && rows_dr && out_raw, "rnerf_nerfmlp_forward: null pointer");
  RNERF_CHECK_ARG(max_workgroups >= 0, "rnerf_nerfmlp_forward: max_workgroups must be >= 0");
  RNERF_CHECK_ARG(prec_ok(precision), "rnerf_nerfmlp_forward: unsupported precision %d", precision);
  RNERF_CHECK_ARG(S >= 1 && B >= 1, "rnerf_nerfmlp_forward: need S >= 1 and B >= 1");
  RNERF_CHECK_ARG((((uintptr_t)packed | (uintptr_t)rows_pd | (uintptr_t)rows_dr | (uintptr_t)out_raw) & 15) == 0,
                  "rnerf_nerfmlp_forward: buffers must be 16-byte aligned");
  const long long total = (long long)S * B;
  hipStream_t st = (hipStream_t)stream;
  switch (precision) {
    case RNERF_PREC_F32: return launch_fwd_f32(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st);
    // every f16-based launch is followed by the range-safe second pass: bf16x3 (fp32's exponent range) on the rows the first pass returned
    // as NaN — an activation above f16's 65504, a weight >= 256 —, nothing else touched; without such a row it reads out_raw once and ends
    case RNERF_PREC_F16X3:
      RNERF_TRY_(launch_fwd<RNERF_PREC_F16X3>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, max_workgroups));
      return launch_fwd_dbg<RNERF_PREC_BF16X3, 512>((const char*)packed + safe_stream_offset(precision), rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, nullptr, max_workgroups);
    case RNERF_PREC_F16X2:
      RNERF_TRY_(launch_fwd<RNERF_PREC_F16X2>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, max_workgroups));
      return launch_fwd_dbg<RNERF_PREC_BF16X3, 512>((const char*)packed + safe_stream_offset(precision), rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, nullptr, max_workgroups);
    case RNERF_PREC_F16F8: {
      // the f16f8 launch steps aside when its pack kernel flagged a weight outside the range of the 2^14-scaled stream (|W| >= 3.99); the f16x3
      // launch behind it (its stream sits in the same packed buffer) runs only then: a fallback per launch, decided on the device
      const float* flag = (const float*)((const char*)packed + Prec<RNERF_PREC_F16F8>::STREAM_BYTES) + AUX_FLAG;
      RNERF_TRY_(launch_fwd<RNERF_PREC_F16F8>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, max_workgroups, flag, 1));
      RNERF_TRY_(launch_fwd<RNERF_PREC_F16X3>((const char*)packed + kF8Fallback, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, max_workgroups, flag, 0));
      return launch_fwd_dbg<RNERF_PREC_BF16X3, 512>((const char*)packed + safe_stream_offset(precision), rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, nullptr, max_workgroups);
    }
    case RNERF_PREC_BF16X3: return launch_fwd<RNERF_PREC_BF16X3>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, max_workgroups);
    case RNERF_PREC_F16:
      RNERF_TRY_(launch_fwd<RNERF_PREC_F16>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, max_workgroups));
      return launch_fwd_dbg<RNERF_PREC_BF16X3, 512>((const char*)packed + safe_stream_offset(precision), rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, nullptr, max_workgroups);
    default: return launch_fwd<RNERF_PREC_BF16>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, max_workgroups);
  }
}

static bool bwd_ok(int b) { return b == RNERF_BWD_BF16 || b == RNERF_BWD_F16 || b == RNERF_BWD_F16X2 || b == RNERF_BWD_F16X3_LO8; }
static bool bwd_two_planes(int b) { return b == RNERF_BWD_F16X2 || b == RNERF_BWD_F16X3_LO8; }      // (LO8: the lo planes use half of their region)
// (forward precision, backward mode) pairs the training kernels are built for
static bool train_combo_ok(int precision, int backward) {
  return precision == RNERF_PREC_F16X3 || (precision == RNERF_PREC_F16 && !bwd_two_planes(backward)) ||
         (precision == RNERF_PREC_BF16X3 && backward == RNERF_BWD_BF16);
}
static size_t train_stream_bytes(int precision) {      // the aux floats (biases, heads) sit behind the forward's operand stream
  return precision == RNERF_PREC_F16 ? Prec<RNERF_PREC_F16>::STREAM_BYTES : (precision == RNERF_PREC_BF16X3 ? Prec<RNERF_PREC_BF16X3>::STREAM_BYTES : Prec<RNERF_PREC_F16X3>::STREAM_BYTES);
}

extern "C" size_t rnerf_nerfmlp_save_bytes(int64_t rows, int backward) {
  const int64_t padded = (rows + 255) / 256 * 256;
  return save_payload_bytes(padded, bwd_two_planes(backward)) + SAVE_QUEUE_BYTES;
}

extern "C" int rnerf_nerfmlp_forward_train(const void* packed, int precision, const float* rows_pd, const float* rows_dr,
                                           const int32_t* node_of_sample, int32_t S, int32_t B, float* out_raw, void* save,
                                           int backward, int32_t max_workgroups, void* stream) {
  RNERF_CHECK_ARG(packed && rows_pd && rows_dr && out_raw && save, "rnerf_nerfmlp_forward_train: null pointer");
  RNERF_CHECK_ARG(max_workgroups >= 0, "rnerf_nerfmlp_forward_train: max_workgroups must be >= 0");
  RNERF_CHECK_ARG(bwd_ok(backward), "rnerf_nerfmlp_forward_train: unknown backward mode %d", backward);
  // (the bf16x3 forward is an inference-only precision: its 8-bit saved operands would cap every backward mode at bf16 accuracy)
  // F16: the single-pass training forward (one MFMA per product, the hi plane IS the operand) — with the single-plane backward modes only:
  // the north-star arithmetic as a labelled bench leg, never the default (11-bit products against the reference's fp32)
  // BF16X3 + backward BF16: the RANGE-SAFE training arithmetic (fp32's exponent range end to end: bf16 hi + lo products forward, the bf16 hi
  // plane saved as it is, bf16 gradients) — what a step is re-run in when the f16-based one met a row outside f16's range (train.py)
  RNERF_CHECK_ARG(train_combo_ok(precision, backward),
                  "rnerf_nerfmlp_forward_train: the training forward is built for precision f16x3, f16 with the single-plane backward modes, and bf16x3 with backward bf16");
  RNERF_CHECK_ARG(S >= 1 && B >= 1, "rnerf_nerfmlp_forward_train: need S >= 1 and B >= 1");
  RNERF_CHECK_ARG((((uintptr_t)packed | (uintptr_t)rows_pd | (uintptr_t)rows_dr | (uintptr_t)out_raw | (uintptr_t)save) & 15) == 0,
                  "rnerf_nerfmlp_forward_train: buffers must be 16-byte aligned");
  const long long total = (long long)S * B;
  hipStream_t st = (hipStream_t)stream;
#ifdef RNERF_MLP_ABLATE      /* RNERF_MLP_DEBUG=256: per-phase clocks of the training forward (hi + lo saves) */
  if (backward == RNERF_BWD_F16X2 && mlp_debug_flags() == 256)
    return launch_fwd_dbg<RNERF_PREC_F16X3, 256, 2>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, save, max_workgroups);
#endif
  if (backward == RNERF_BWD_F16X2) return launch_fwd_dbg<RNERF_PREC_F16X3, 0, 2>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, save, max_workgroups);
  if (backward == RNERF_BWD_F16X3_LO8) return launch_fwd_dbg<RNERF_PREC_F16X3, 0, 3>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, save, max_workgroups);
  if (precision == RNERF_PREC_BF16X3) return launch_fwd_dbg<RNERF_PREC_BF16X3, 0, 1>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, save, max_workgroups);
  if (precision == RNERF_PREC_F16) return launch_fwd_dbg<RNERF_PREC_F16, 0, 1>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, save, max_workgroups);
  return launch_fwd_dbg<RNERF_PREC_F16X3, 0, 1>(packed, rows_pd, rows_dr, node_of_sample, B, total, out_raw, st, save, max_workgroups);
}

extern "C" size_t rnerf_nerfmlp_bwd_packed_bytes(void) { return (size_t)kBwdBlocks * 2 * 1024; }
extern "C" size_t rnerf_nerfmlp_dy_bytes(int64_t rows, int backward) {
  const int64_t padded = (rows + 255) / 256 * 256;
  const int np = bwd_two_planes(backward) ? 2 : 1;
  return dy_plane_uint4(padded, np) * sizeof(uint4) + (backward == RNERF_BWD_BF16 ? 0 : ((size_t)padded + 4) * sizeof(float));
}

extern "C" int rnerf_nerfmlp_pack_bwd(const float* params, int backward, void* packed_bwd, void* stream) {
  RNERF_CHECK_ARG(params && packed_bwd, "rnerf_nerfmlp_pack_bwd: null pointer");
  RNERF_CHECK_ARG(bwd_ok(backward), "rnerf_nerfmlp_pack_bwd: unknown backward mode %d", backward);
  const int threads = kBwdBlocks * 64;
  if (backward == RNERF_BWD_BF16)
    hipLaunchKernelGGL(nerfmlp_pack_bwd_kernel<RNERF_PREC_BF16X3>, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, (char*)packed_bwd);
  else
    hipLaunchKernelGGL(nerfmlp_pack_bwd_kernel<RNERF_PREC_F16X3>, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, (char*)packed_bwd);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

// the dgrad's row-scale reference (m_ref, an atomicMax accumulator behind the row scales of the f16 modes' dY buffer; nullptr: none) — it
// must be zero when the kernel starts: launch_dgrad zeroes it unless the caller already has (nerfmlp_step_zero)
void* rnerf::nerfmlp_dgrad_scale_ref(int backward, void* dy, int64_t rows) {
  const long long R = (rows + 255) / 256 * 256;
  if (bwd_two_planes(backward)) return (char*)dy + dy_plane_uint4(R, Bwd<RNERF_BWD_F16X2>::NP) * sizeof(uint4) + (size_t)R * sizeof(float);
  if (backward == RNERF_BWD_F16) return (char*)dy + dy_plane_uint4(R, Bwd<RNERF_BWD_F16>::NP) * sizeof(uint4) + (size_t)R * sizeof(float);
  return nullptr;
}

template <int BWD>
static int launch_dgrad(const void* packed_bwd, const float* fwd_aux, const void* save, const float* d_raw, int64_t rows, void* dy, hipStream_t st,
                        bool zero_ref = true, bool allow_half = true) {
  using PB = Prec<Bwd<BWD>::PREC>;
  const int n_tiles = (int)((rows + 255) / 256);
  int dev = 0, cus = 0;
  RNERF_CHECK_HIP(hipGetDevice(&dev));
  RNERF_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int grid = n_tiles < cus ? n_tiles : cus;
  const size_t lds = 2 * (size_t)PB::SLAB + 4 * 32768;
  static DeviceOnce attr_set;
  if (attr_set.need()) {
    RNERF_CHECK_HIP(hipFuncSetAttribute((const void*)nerfmlp_dgrad_kernel<BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.set();
  }
  const long long R = (long long)n_tiles * 256;
  if (Bwd<BWD>::F16 && zero_ref) RNERF_CHECK_HIP(hipMemsetAsync(nerfmlp_dgrad_scale_ref(BWD == kBwdF16OnePass ? RNERF_BWD_F16 : BWD, dy, rows), 0, 4 * sizeof(float), st));
  // Few rows: 128-row tiles when they all fit one round (see launch_fwd_dbg) — unless the caller runs another level's dgrad beside this one
  // (allow_half = false): two kernels that each own whole CUs then share the chip, and with twice the workgroups of half the work the
  // step measures the same or slower (512 rays, levels side by side: 2.05 -> 2.09 ms; alone, a 256-ray single-level step: 1.17 -> 1.12 ms).
  if constexpr (BWD == RNERF_BWD_F16X2 || BWD == RNERF_BWD_F16 || BWD == kBwdF16OnePass || BWD == RNERF_BWD_F16X3_LO8) {
    static const int half_lim = [] { const char* e = RNERF_ENV("RNERF_DGRAD_HALF_TILES"); return e ? atoi(e) : -1; }();      // 0: off, n: at most n tiles
    if (allow_half && half_lim != 0 && 2 * n_tiles <= (half_lim > 1 ? half_lim : cus)) {
      const size_t lds1 = 2 * (size_t)PB::SLAB;
      static DeviceOnce attr1_set;
      if (attr1_set.need()) {
        RNERF_CHECK_HIP(hipFuncSetAttribute((const void*)nerfmlp_dgrad_kernel<BWD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        attr1_set.set();
      }
      hipLaunchKernelGGL((nerfmlp_dgrad_kernel<BWD, true>), dim3(2 * n_tiles), dim3(256), lds1, st, (const char*)packed_bwd, fwd_aux, (const uint4*)save, R,
                         (const float4*)d_raw, (long long)rows, 2 * n_tiles, (uint4*)dy);
      RNERF_CHECK_LAUNCH();
      return RNERF_OK;
    }
  }
  hipLaunchKernelGGL(nerfmlp_dgrad_kernel<BWD>, dim3(grid), dim3(256), lds, st, (const char*)packed_bwd, fwd_aux, (const uint4*)save, R,
                     (const float4*)d_raw, (long long)rows, n_tiles, (uint4*)dy);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

namespace rnerf {
// zero_ref = false: the caller has zeroed nerfmlp_dgrad_scale_ref(backward, dy, rows) on `stream` already (nerfmlp_step_zero)
int nerfmlp_dgrad_impl(const void* packed_bwd, const void* packed_fwd, int fwd_precision, int backward, const void* save, const float* d_raw, int64_t rows,
                       void* dy, bool zero_ref, bool allow_half, hipStream_t st) {
  RNERF_CHECK_ARG(packed_bwd && packed_fwd && save && d_raw && dy, "rnerf_nerfmlp_dgrad: null pointer");
  RNERF_CHECK_ARG(train_combo_ok(fwd_precision, backward), "rnerf_nerfmlp_dgrad: forward precision must be f16x3 (f16 with a single-plane backward mode; bf16x3 with backward bf16)");
  RNERF_CHECK_ARG(bwd_ok(backward), "rnerf_nerfmlp_dgrad: unknown backward mode %d", backward);
  RNERF_CHECK_ARG(rows >= 1, "rnerf_nerfmlp_dgrad: rows must be >= 1");
  // the aux floats (biases, heads) sit behind the forward's operand stream, whose length depends on its precision
  const float* fwd_aux = (const float*)((const char*)packed_fwd + train_stream_bytes(fwd_precision));
  if (backward == RNERF_BWD_F16X2) return launch_dgrad<RNERF_BWD_F16X2>(packed_bwd, fwd_aux, save, d_raw, rows, dy, st, zero_ref, allow_half);
  if (backward == RNERF_BWD_F16X3_LO8) return launch_dgrad<RNERF_BWD_F16X3_LO8>(packed_bwd, fwd_aux, save, d_raw, rows, dy, st, zero_ref, allow_half);
  if (backward == RNERF_BWD_F16 && fwd_precision == RNERF_PREC_F16) return launch_dgrad<kBwdF16OnePass>(packed_bwd, fwd_aux, save, d_raw, rows, dy, st, zero_ref, allow_half);
  if (backward == RNERF_BWD_F16) return launch_dgrad<RNERF_BWD_F16>(packed_bwd, fwd_aux, save, d_raw, rows, dy, st, zero_ref, allow_half);
  return launch_dgrad<RNERF_BWD_BF16>(packed_bwd, fwd_aux, save, d_raw, rows, dy, st, zero_ref, allow_half);
}
}  // namespace rnerf

extern "C" int rnerf_nerfmlp_dgrad(const void* packed_bwd, const void* packed_fwd, int fwd_precision, int backward, const void* save,
                                   const float* d_raw, int64_t rows, void* dy, void* stream) {
  return nerfmlp_dgrad_impl(packed_bwd, packed_fwd, fwd_precision, backward, save, d_raw, rows, dy, true, true, (hipStream_t)stream);
}

// the wgrad jobs of one NerfMLP: {x slot base, x k-steps, dy slot base, dy k-steps, job, second-segment bases}
struct WgradPlan { int qx, KSx, qd, KSd; WgradJob job; int qx2, qd2, KSd2; };
static constexpr WgradJob job1(int dense, int xkind, int row_off, int dkind, int KT, int NT, int wb, int bias_sigma = 0) {
  return WgradJob{dense, xkind, row_off, dkind, KT, NT, wb, KT, 0, 0, NT, 0, 0, bias_sigma};
}
// 14 single-segment jobs: the bf16 body (MFMA transposition)
static const WgradPlan kWgradPlan[15] = {
    {SAVE_PE, 4, 0, 16, job1(0, 0, 0, 0, 2, 8, 1), 0, 0, 0},
    {SAVE_L1 + 0, 16, 16, 16, job1(1, 1, 0, 0, 8, 8, 1), 0, 0, 0},   {SAVE_L1 + 16, 16, 32, 16, job1(2, 1, 0, 0, 8, 8, 1), 0, 0, 0},
    {SAVE_L1 + 32, 16, 48, 16, job1(3, 1, 0, 0, 8, 8, 1), 0, 0, 0},  {SAVE_L1 + 48, 16, 64, 16, job1(4, 1, 0, 0, 8, 8, 1), 0, 0, 0},
    {SAVE_L1 + 64, 16, 80, 16, job1(5, 1, 0, 0, 8, 8, 1), 0, 0, 0},  {SAVE_PE, 4, 80, 16, job1(5, 0, 256, 0, 2, 8, 0), 0, 0, 0},
    {SAVE_L1 + 80, 16, 96, 16, job1(6, 1, 0, 0, 8, 8, 1), 0, 0, 0},  {SAVE_L1 + 96, 16, 112, 16, job1(7, 1, 0, 0, 8, 8, 1), 0, 0, 0},
    {SAVE_L1 + 112, 16, 128, 16, job1(9, 1, 0, 0, 8, 8, 1), 0, 0, 0}, {SAVE_L1 + 112, 16, DY_HEADS, 1, job1(8, 1, 0, 1, 8, 1, 1), 0, 0, 0},
    {SAVE_L1 + 128, 16, DY_L9, 8, job1(10, 1, 0, 0, 8, 4, 1), 0, 0, 0}, {SAVE_VIEW, 2, DY_L9, 8, job1(10, 2, 256, 0, 1, 4, 0), 0, 0, 0},
    {SAVE_RGBIN, 8, DY_HEADS, 1, job1(11, 1, 0, 2, 4, 1, 1), 0, 0, 0},
    {0, 0, 0, 0, job1(0, 0, 0, 0, 0, 0, 0), 0, 0, 0}};
// 11 jobs of the transpose-read kernel: an operand stream that two Dense blocks share is read once —
//   Dense_5 = [previous layer | position encoding] rows against dY_5 (10 k-tiles), Dense_10 = [bottleneck | view encoding] rows against
//   dY_9 (9 k-tiles), Dense_9 and the sigma head Dense_8 against the trunk output (8 + 1 n-tiles; the sigma bias comes from the rgb job)
static const WgradPlan kWgradPlanTr[12] = {
    {SAVE_PE, 4, 0, 16, job1(0, 0, 0, 0, 2, 8, 1), 0, 0, 0},
    {SAVE_L1 + 0, 16, 16, 16, job1(1, 1, 0, 0, 8, 8, 1), 0, 0, 0},   {SAVE_L1 + 16, 16, 32, 16, job1(2, 1, 0, 0, 8, 8, 1), 0, 0, 0},
    {SAVE_L1 + 32, 16, 48, 16, job1(3, 1, 0, 0, 8, 8, 1), 0, 0, 0},  {SAVE_L1 + 48, 16, 64, 16, job1(4, 1, 0, 0, 8, 8, 1), 0, 0, 0},
    {SAVE_L1 + 64, 20, 80, 16, WgradJob{5, 1, 0, 0, 10, 8, 1, /*KTa*/ 8, /*xkind2*/ 0, /*row_off2*/ 256, /*NTa*/ 8, 0, 0, 0}, SAVE_PE, 0, 0},
    {SAVE_L1 + 80, 16, 96, 16, job1(6, 1, 0, 0, 8, 8, 1), 0, 0, 0},  {SAVE_L1 + 96, 16, 112, 16, job1(7, 1, 0, 0, 8, 8, 1), 0, 0, 0},
    {SAVE_L1 + 112, 16, 128, 16, WgradJob{9, 1, 0, 0, 8, 9, 1, 8, 0, 0, /*NTa*/ 8, /*dense2*/ 8, /*dkind2*/ 1, 0}, 0, DY_HEADS, 1},
    {SAVE_L1 + 128, 18, DY_L9, 8, WgradJob{10, 1, 0, 0, 9, 4, 1, /*KTa*/ 8, /*xkind2*/ 2, /*row_off2*/ 256, 4, 0, 0, 0}, SAVE_VIEW, 0, 0},
    {SAVE_RGBIN, 8, DY_HEADS, 1, job1(11, 1, 0, 2, 4, 1, 1, /*bias_sigma*/ 1), 0, 0, 0},
    {0, 0, 0, 0, job1(0, 0, 0, 0, 0, 0, 0), 0, 0, 0}};

// legacy = the bf16 body (MFMA transposition): shares by MFMA count, 4 rounds of workgroups; otherwise the transpose-read bodies, paced
// by HBM: shares by bytes streamed per row, 2 rounds (measured with RNERF_WGRAD_TRACE: all jobs' workgroups finish within 10 %)
// mult: rounds of one-per-CU workgroups the rows are cut into (0: the default of the body — 4 legacy, 2 transposing)
static size_t build_wgrad_table(int cus, WgradTable& t, bool legacy, int mult = 0) {
  const WgradPlan* plan = legacy ? kWgradPlan : kWgradPlanTr;
  int n = 0;
  double cost[16], total = 0;
  for (; plan[n].KSx != 0; ++n) {
    const WgradJob& jb = plan[n].job;
    cost[n] = legacy ? 2.0 * jb.KT * jb.NT + 2 * (jb.KT + jb.NT) + 16 : jb.KT + jb.NT + 1;
    total += cost[n];
  }
  t.n = n;
  static const int mult_env = RNERF_ENV("RNERF_WGRAD_MULT") ? atoi(RNERF_ENV("RNERF_WGRAD_MULT")) : 0;
  const int budget = (mult_env > 0 ? mult_env : (mult > 0 ? mult : (legacy ? 4 : 2))) * cus;
  size_t off = 0;
  int wg = 0;
  for (int i = 0; i < n; ++i) {
    const WgradPlan& p = plan[i];
    int share = (int)(budget * cost[i] / total + 0.5);
    if (share < 8) share = 8;
    t.qx[i] = p.qx; t.KSx[i] = p.KSx; t.qd[i] = p.qd; t.KSd[i] = p.KSd; t.job[i] = p.job;
    t.qx2[i] = p.qx2; t.qd2[i] = p.qd2; t.KSd2[i] = p.KSd2;
    t.wg0[i] = wg; wg += share;
    t.poff[i] = (long long)off; off += (size_t)share * p.job.KT * 32 * p.job.NT * 32;
    t.pboff[i] = (long long)off; off += (size_t)share * p.job.NT * 32;
  }
  t.wg0[n] = wg;
  return off;
}

static int device_cus() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return cus;
}

// tr_small: the transposing body's table for a FEW rows (<= kWgradSmallRows: both levels of a 128-ray shard of the reference's default
// batch): one round of workgroups instead of two — every workgroup writes its whole partial dW block whatever its share of the rows, and
// with < 100 rows per workgroup that fixed cost is the kernel (128 rays: 1.29 -> 1.23 ms per step; from 32 768 rows on — the coarse level of
// a 512-ray batch — two rounds are faster: 2.14 against 2.18 ms, tools/r04/wgrad_mult.sh).  Smaller than `tr` in workgroups and partials:
// the workspace size is unchanged.
constexpr long long kWgradSmallRows = 24576;
struct WgradTables { WgradTable legacy, tr, tr_small; size_t partial_floats; int max_wgs; };
static const WgradTables& wgrad_tables() {
  // built once, by whichever host thread comes first (a function-local static's initialisation is thread-safe); sized by the compute units
  // of that thread's current device — every device a process drives is assumed to be the same part
  static const WgradTables tables = [] {
    WgradTables w;
    const size_t a = build_wgrad_table(device_cus(), w.legacy, true), b = build_wgrad_table(device_cus(), w.tr, false);
    const size_t s = build_wgrad_table(device_cus(), w.tr_small, false, 1);
    w.partial_floats = a > b ? a : b;
    if (s > w.partial_floats) w.partial_floats = s;
    w.max_wgs = w.legacy.wg0[w.legacy.n] > w.tr.wg0[w.tr.n] ? w.legacy.wg0[w.legacy.n] : w.tr.wg0[w.tr.n];
    if (w.tr_small.wg0[w.tr_small.n] > w.max_wgs) w.max_wgs = w.tr_small.wg0[w.tr_small.n];
    return w;
  }();
  return tables;
}

extern "C" size_t rnerf_nerfmlp_wgrad_workspace_bytes(void) {
  const WgradTables& w = wgrad_tables();
  return w.partial_floats * sizeof(float) + (size_t)w.max_wgs * 2 * sizeof(long long);       // + the RNERF_WGRAD_TRACE slots
}

extern "C" int rnerf_nerfmlp_wgrad(int fwd_precision, int backward, const void* save, const void* dy, int64_t rows, float* grads, void* workspace,
                                   void* stream) {
  RNERF_CHECK_ARG(save && dy && grads && workspace, "rnerf_nerfmlp_wgrad: null pointer");
  RNERF_CHECK_ARG(train_combo_ok(fwd_precision, backward), "rnerf_nerfmlp_wgrad: forward precision must be f16x3 (f16 with a single-plane backward mode; bf16x3 with backward bf16)");
  RNERF_CHECK_ARG(bwd_ok(backward), "rnerf_nerfmlp_wgrad: unknown backward mode %d", backward);
  RNERF_CHECK_ARG(rows >= 1, "rnerf_nerfmlp_wgrad: rows must be >= 1");
  const WgradTables& w = wgrad_tables();
  static DeviceOnce ready;
  if (ready.need()) {
    RNERF_CHECK_HIP(hipFuncSetAttribute((const void*)nerfmlp_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    RNERF_CHECK_HIP(hipFuncSetAttribute((const void*)nerfmlp_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    RNERF_CHECK_HIP(hipFuncSetAttribute((const void*)nerfmlp_wgrad_tr_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, wgtr_lds_bytes<1>()));
    RNERF_CHECK_HIP(hipFuncSetAttribute((const void*)nerfmlp_wgrad_tr_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, wgtr_lds_bytes<2>()));
    RNERF_CHECK_HIP(hipFuncSetAttribute((const void*)nerfmlp_wgrad_tr8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wgtr8_lds_bytes()));
    ready.set();
  }
  const long long R = (rows + 255) / 256 * 256;
  hipStream_t st = (hipStream_t)stream;
  const float* out_scale = nullptr;
  // RNERF_WGRAD_TRACE=1 (profiling aid): per-workgroup start / end times behind the partials -> per-job spans on stderr (synchronises)
  static const bool tracing = RNERF_ENV("RNERF_WGRAD_TRACE") != nullptr;
  long long* trace = tracing ? (long long*)((char*)workspace + w.partial_floats * sizeof(float)) : nullptr;
  const WgradTable& tab = backward == RNERF_BWD_BF16 ? w.legacy : (rows <= kWgradSmallRows ? w.tr_small : w.tr);
  if (backward == RNERF_BWD_BF16) {
    const int n_chunks = (int)((rows + 127) / 128);
    if (fwd_precision == RNERF_PREC_BF16X3)      // the range-safe step: the saved plane is bf16 already
      hipLaunchKernelGGL(nerfmlp_wgrad_kernel<false>, dim3(tab.wg0[tab.n]), dim3(512), 131072, st, (const uint4*)save, (const uint4*)dy, R,
                         (long long)rows, n_chunks, (float*)workspace, tab);
    else
      hipLaunchKernelGGL(nerfmlp_wgrad_kernel<true>, dim3(tab.wg0[tab.n]), dim3(512), 131072, st, (const uint4*)save, (const uint4*)dy, R,
                         (long long)rows, n_chunks, (float*)workspace, tab);
  } else if (backward == RNERF_BWD_F16) {
    hipLaunchKernelGGL(nerfmlp_wgrad_tr_kernel<1>, dim3(tab.wg0[tab.n]), dim3(512), wgtr_lds_bytes<1>(), st, (const uint4*)save, (const uint4*)dy, R,
                       (float*)workspace, tab, trace);
    out_scale = (const float*)((const uint4*)dy + dy_plane_uint4(R, 1)) + R;
  } else if (backward == RNERF_BWD_F16X3_LO8) {
    hipLaunchKernelGGL(nerfmlp_wgrad_tr8_kernel, dim3(tab.wg0[tab.n]), dim3(512), wgtr8_lds_bytes(), st, (const uint4*)save, (const uint4*)dy, R,
                       (float*)workspace, tab, trace);
    out_scale = (const float*)((const uint4*)dy + dy_plane_uint4(R, 2)) + R;
  } else {
    hipLaunchKernelGGL(nerfmlp_wgrad_tr_kernel<2>, dim3(tab.wg0[tab.n]), dim3(512), wgtr_lds_bytes<2>(), st, (const uint4*)save, (const uint4*)dy, R,
                       (float*)workspace, tab, trace);
    out_scale = (const float*)((const uint4*)dy + dy_plane_uint4(R, 2)) + R;
  }
  int max_elems = 0;
  for (int j = 0; j < tab.n; ++j) max_elems = tab.job[j].KT * tab.job[j].NT * 1024 > max_elems ? tab.job[j].KT * tab.job[j].NT * 1024 : max_elems;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((max_elems + 255) / 256, tab.n), dim3(256), 0, st, (const float*)workspace, tab, grads, out_scale);
  RNERF_CHECK_LAUNCH();
  if (trace && backward != RNERF_BWD_BF16) {
    const int n_wg = tab.wg0[tab.n];
    std::vector<long long> h(2 * (size_t)n_wg);
    RNERF_CHECK_HIP(hipStreamSynchronize(st));
    RNERF_CHECK_HIP(hipMemcpy(h.data(), trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    long long t0 = h[0];
    for (int i = 0; i < n_wg; ++i) t0 = h[2 * i] < t0 ? h[2 * i] : t0;
    for (int j = 0; j < tab.n; ++j) {
      long long a = -1, b = 0; double busy = 0;
      for (int i = tab.wg0[j]; i < tab.wg0[j + 1]; ++i) { a = (a < 0 || h[2 * i] < a) ? h[2 * i] : a; b = h[2 * i + 1] > b ? h[2 * i + 1] : b; busy += (double)(h[2 * i + 1] - h[2 * i]); }
      fprintf(stderr, "[wgrad trace] job %2d (KT %d NT %d) wgs %4d  first start %9lld  last end %9lld  mean wg ticks (100 MHz) %7.0f\n", j, tab.job[j].KT, tab.job[j].NT,
              tab.wg0[j + 1] - tab.wg0[j], a - t0, b - t0, busy / (tab.wg0[j + 1] - tab.wg0[j]));
    }
  }
  return RNERF_OK;
}

// RNERF_BKGD_EXACT=1: the exact-fp32 background-MLP kernels (v_mfma_f32_32x32x2_f32) instead of the f16 hi + lo ones (read once)
static bool bkgd_exact() {
  static const bool v = [] { const char* e = RNERF_ENV("RNERF_BKGD_EXACT"); return e && e[0] == '1'; }();
  return v;
}

extern "C" int rnerf_bkgd_forward(const float* params, const float* dirs, int32_t dir_stride, int64_t n, double rgb_padding,
                                  float* out_rgb, void* stream) {
  RNERF_CHECK_ARG(params && dirs && out_rgb, "rnerf_bkgd_forward: null pointer");
  RNERF_CHECK_ARG(dir_stride >= 3, "rnerf_bkgd_forward: dir_stride must be >= 3");
  RNERF_CHECK_ARG(n >= 1, "rnerf_bkgd_forward: n must be >= 1");
  if (bkgd_exact())
    hipLaunchKernelGGL(bkgd_fwd_kernel<false>, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, (hipStream_t)stream, params, dirs, dir_stride,
                       (long long)n, (float)(1 + 2 * rgb_padding), (float)rgb_padding, out_rgb, (float*)nullptr);
  else
    return launch_bkgd16_fwd(false, params, dirs, dir_stride, (long long)n, (float)(1 + 2 * rgb_padding), (float)rgb_padding, out_rgb, nullptr, (hipStream_t)stream);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

extern "C" size_t rnerf_bkgd_save_bytes(int64_t n) { return bkgd_save_floats(n) * sizeof(float); }
extern "C" size_t rnerf_bkgd_dy_bytes(int64_t n) { return bkgd_dy_floats(n) * sizeof(float); }

extern "C" int rnerf_bkgd_forward_train(const float* params, const float* dirs, int32_t dir_stride, int64_t n, double rgb_padding,
                                        float* out_rgb, void* save, void* stream) {
  RNERF_CHECK_ARG(params && dirs && out_rgb && save, "rnerf_bkgd_forward_train: null pointer");
  RNERF_CHECK_ARG(dir_stride >= 3 && n >= 1, "rnerf_bkgd_forward_train: need dir_stride >= 3 and n >= 1");
  if (bkgd_exact())
    hipLaunchKernelGGL(bkgd_fwd_kernel<true>, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, (hipStream_t)stream, params, dirs, dir_stride,
                       (long long)n, (float)(1 + 2 * rgb_padding), (float)rgb_padding, out_rgb, (float*)save);
  else
    return launch_bkgd16_fwd(true, params, dirs, dir_stride, (long long)n, (float)(1 + 2 * rgb_padding), (float)rgb_padding, out_rgb, (float*)save, (hipStream_t)stream);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

extern "C" int rnerf_bkgd_backward_dgrad(const float* params, const void* save, const float* d_out, int64_t n, double rgb_padding, void* dy,
                                         float* d_dirs, void* stream) {
  RNERF_CHECK_ARG(params && save && d_out && dy, "rnerf_bkgd_backward_dgrad: null pointer");
  RNERF_CHECK_ARG(n >= 1, "rnerf_bkgd_backward_dgrad: n must be >= 1");
  hipLaunchKernelGGL(bkgd_dgrad_kernel, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, (hipStream_t)stream, params, (const float*)save, d_out, (long long)n,
                     (float)(1 + 2 * rgb_padding), (float)rgb_padding, (float*)dy, (float4*)d_dirs);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

extern "C" int rnerf_bkgd_backward_wgrad(const void* save, void* dy, int64_t n, float* grads, int coresident, void* stream) {
  RNERF_CHECK_ARG(save && dy && grads, "rnerf_bkgd_backward_wgrad: null pointer");
  RNERF_CHECK_ARG(n >= 1, "rnerf_bkgd_backward_wgrad: n must be >= 1");
  hipStream_t st = (hipStream_t)stream;
  const float* sv = (const float*)save;
  float* dyf = (float*)dy;
  const unsigned chunks = (unsigned)((n + 255) / 256);
  float* partial = dyf + (size_t)n * 5 * 128;
  if (coresident)
    hipLaunchKernelGGL(bkgd_wgrad_co_kernel<0>, dim3(chunks, 18, 2), dim3(64), 0, st, sv, (const float*)dyf, (long long)n, partial);
  else
    hipLaunchKernelGGL(bkgd_wgrad_kernel<0>, dim3(((chunks + 7) / 8) * 8 * SmallNet<0>::UNITS), dim3(256), 0, st, sv, (const float*)dyf, (long long)n, partial);
  hipLaunchKernelGGL(bkgd_wgrad_reduce_kernel<RNERF_BKGDMLP_PARAMS>, dim3((RNERF_BKGDMLP_PARAMS + 255) / 256), dim3(256), 0, st, (const float*)partial, (int)chunks,
                     grads);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

extern "C" int rnerf_bkgd_backward(const float* params, const void* save, const float* d_out, int64_t n, double rgb_padding, void* dy,
                                   float* grads, float* d_dirs, void* stream) {
  RNERF_CHECK_ARG(params && save && d_out && dy && grads, "rnerf_bkgd_backward: null pointer");
  int rc = rnerf_bkgd_backward_dgrad(params, save, d_out, n, rgb_padding, dy, d_dirs, stream);
  if (rc != RNERF_OK) return rc;
  return rnerf_bkgd_backward_wgrad(save, dy, n, grads, 0, stream);
}

extern "C" int rnerf_so3_query(const float* table, const rnerf_grid* g, const float* so3_params, const float* window10, const float* pts,
                               const float* condition, int64_t n, float* out4, float* pred_grad, void* stream) {
  RNERF_CHECK_ARG(table && g && so3_params && window10 && pts && out4 && pred_grad, "rnerf_so3_query: null pointer");
  RNERF_CHECK_ARG(n >= 1, "rnerf_so3_query: n must be >= 1");
  GridParams gp;
  RNERF_CHECK_ARG(make_grid_params(g, &gp), "rnerf_so3_query: bad grid");
  So3Window w;
  for (int i = 0; i < 10; ++i) w.w[i] = window10[i];
  hipLaunchKernelGGL(so3_query_kernel, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, (hipStream_t)stream, (const float4*)table, gp, so3_params, w,
                     pts, condition, (long long)n, (float4*)out4, pred_grad);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

extern "C" size_t rnerf_so3_packed_bytes(void) { return (size_t)kSo3Blocks * 128 * sizeof(uint4); }

extern "C" int rnerf_march_all(const float* table, const rnerf_grid* g, const float* so3_params, void* so3_packed, const float* window10, const float* origins,
                               const float* viewdirs, int32_t B, double near, double far, int32_t num_nodes, float* path_pd, float* path_dr,
                               float* path_ior, const int32_t* ray_order, void* stream) {
  RNERF_CHECK_ARG(table && g && so3_params && so3_packed && window10 && origins && viewdirs && path_pd && path_dr, "rnerf_march_all: null pointer");
  RNERF_CHECK_ARG(B > 0 && num_nodes >= 2, "rnerf_march_all: need B > 0 and num_nodes >= 2");
  GridParams gp;
  RNERF_CHECK_ARG(make_grid_params(g, &gp), "rnerf_march_all: bad grid");
  So3Window w;
  for (int i = 0; i < 10; ++i) w.w[i] = window10[i];
  RNERF_CHECK_ARG(grid_fits_u32(gp), "rnerf_march_all: grid too large for 32-bit byte offsets (needs a table < 4 GiB)");
  const float stepf = (float)((far - near) / (num_nodes - 1));  // models.py:122
  hipLaunchKernelGGL(so3_pack16_kernel, dim3((kSo3Blocks * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, so3_params, (uint4*)so3_packed);
  hipLaunchKernelGGL(march_all_kernel, dim3((unsigned)((B + 15) / 16)), dim3(256), 0, (hipStream_t)stream, (const float4*)table, gp, so3_params,
                     (const uint4*)so3_packed, w, origins, viewdirs, B, (float)near, stepf, num_nodes, (float4*)path_pd, (float4*)path_dr, (float4*)path_ior,
                     (float4*)nullptr, (int*)nullptr, 0, (int2*)nullptr, (float4*)nullptr, (float4*)nullptr, (int*)nullptr, (const int*)ray_order);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

#include "ior_train_api.inc"
