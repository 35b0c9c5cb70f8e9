// RNERF_PREC_F32: the exact-fp32 NerfMLP forward — the on-device ARBITER of the MFMA engines (csrc/mlp.hip), not a product precision.
// Reference: rnerf/model_utils.py:30-90 (NerfMLP.__call__, computed in fp32 by XLA), :187-214 (pos_enc); call sites rnerf/models.py:257,
// 289-308.
//
// Every Dense is  y[n] = fl( fl(sum_k x[k] W[k][n]) + b[n] )  with the sum as ONE sequential chain of v_fma_f32 over k = 0 .. in-1 in the
// order of the kernel's rows (for the concat layers: the 256 previous outputs, then the encoding — rnerf/model_utils.py:68-69,82-83):
// fully defined fp32 semantics, no split operands, no weight scale, no matrix-core rounding behaviour to reason about.  What it is for:
// split-f16 error (f16x3 against this: 2^-22 class), the oracle's own summation order (numpy sgemm against this: 2^-24 sqrt(K) class) and
// real bugs separate cleanly on the device.  Speed is whatever falls out (one row per lane, weights through the scalar cache): ~25 G MAC/s
// per CU, fine for the parity tests' few thousand rows and two orders of magnitude away from a product mode.
//
// "Packed" stream of this precision = the flat fp32 parameter buffer itself (RNERF_NERFMLP_PARAMS floats, rnerf_nerfmlp_pack copies it).
#include "nerfmlp_layout.h"

namespace rnerf {

constexpr int F32_ROWS = 64;            // rows per workgroup (one per lane); 4 waves share them, each owning a quarter of a layer's outputs
constexpr int F32_XSTRIDE = 320;        // features of the widest input (319) rounded up

// out[n][row] = act(sum_k in[k][row] * W[k][n] + b[n]) for this wave's quarter of the N outputs; in / out: LDS, [feature][row]
template <int K, int N, bool RELU>
__device__ __forceinline__ void dense_f32(const float* __restrict__ W, const float* __restrict__ bias, const float* in, float* out, int row, int wave) {
  constexpr int NQ = N / 4;              // outputs per wave
  constexpr int NB = NQ < 8 ? NQ : 8;    // outputs per pass over k
  static_assert(NQ % NB == 0, "quarter must be a multiple of the block");
  for (int n0 = wave * NQ; n0 < (wave + 1) * NQ; n0 += NB) {
    float acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = 0.f;
    const float* __restrict__ w = W + n0;            // wave-uniform address: the weights travel through the scalar cache
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
      const float x = in[k * F32_ROWS + row];
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = fmaf(x, w[(size_t)k * N + j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float y = fadd(acc[j], bias[n0 + j]);
      out[(n0 + j) * F32_ROWS + row] = RELU ? (y < 0.f ? 0.f : y) : y;      // (not fmaxf: that would turn a NaN into 0 — the arbiter must not hide one)
    }
  }
}

// pos_enc(x, 0, L) into enc[f][row], f = 0 .. 3 + 6 L - 1: [x(3) | sin(2^d x_c) | sin(2^d x_c + pi/2)], degree-major (model_utils.py:211-214)
template <int L>
__device__ __forceinline__ void encode_f32(float3 v, float* enc, int row, int wave) {
  constexpr int NF = 3 + 6 * L;
  for (int f = wave; f < NF; f += 4) {
    float y;
    if (f < 3) y = f == 0 ? v.x : (f == 1 ? v.y : v.z);
    else {
      const int q = (f - 3) % (3 * L), d = q / 3, c = q % 3;
      const float phase = (f - 3) >= 3 * L ? 1.5707963705062866f : 0.0f;
      const float x = c == 0 ? v.x : (c == 1 ? v.y : v.z);
      y = pe_sin(fadd(fmul(x, (float)(1 << d)), phase));
    }
    enc[f * F32_ROWS + row] = y;
  }
}

__global__ void __launch_bounds__(256) nerfmlp_fwd_f32_kernel(const float* __restrict__ P, const float4* __restrict__ rows_pd, const float4* __restrict__ rows_dr,
                                                             const int* __restrict__ node_of_sample, int B, long long total_rows, float4* __restrict__ out_raw) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* bufA = sm;                                   // [320][64]: a layer's input (the 256 activations, then the concatenated encoding)
  float* bufB = sm + F32_XSTRIDE * F32_ROWS;          // [256][64]: its output
  float* head = bufB + 256 * F32_ROWS;                // [4][64]: raw r, g, b, sigma
  const int row = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  long long r = (long long)blockIdx.x * F32_ROWS + row;
  const bool ok = r < total_rows;
  if (!ok) r = total_rows - 1;
  size_t rec = (size_t)r;
  if (node_of_sample) { const long long s = r / B; rec = (size_t)node_of_sample[s] * B + (size_t)(r - s * B); }
  const float4 pd = rows_pd[rec], dr = rows_dr[rec];
  auto kern = [&](int d) { return P + nerf_koff(d); };
  auto bias = [&](int d) { return P + nerf_boff(d); };

  encode_f32<10>(make_float3(pd.x, pd.y, pd.z), bufA, row, wave);                      // rnerf/models.py:257
  __syncthreads();
  dense_f32<63, 256, true>(kern(0), bias(0), bufA, bufB, row, wave);                   // Dense_0
  __syncthreads();
  dense_f32<256, 256, true>(kern(1), bias(1), bufB, bufA, row, wave); __syncthreads();
  dense_f32<256, 256, true>(kern(2), bias(2), bufA, bufB, row, wave); __syncthreads();
  dense_f32<256, 256, true>(kern(3), bias(3), bufB, bufA, row, wave); __syncthreads();
  dense_f32<256, 256, true>(kern(4), bias(4), bufA, bufB, row, wave); __syncthreads(); // output of the layer with i = 4: skip concat after it
  // x = concatenate([x, inputs]) (model_utils.py:68-69): bufA <- [bufB (256) | pos_enc (63)]
  for (int f = wave; f < 256; f += 4) bufA[f * F32_ROWS + row] = bufB[f * F32_ROWS + row];
  encode_f32<10>(make_float3(pd.x, pd.y, pd.z), bufA + 256 * F32_ROWS, row, wave);
  __syncthreads();
  dense_f32<319, 256, true>(kern(5), bias(5), bufA, bufB, row, wave); __syncthreads();
  dense_f32<256, 256, true>(kern(6), bias(6), bufB, bufA, row, wave); __syncthreads();
  dense_f32<256, 256, true>(kern(7), bias(7), bufA, bufB, row, wave); __syncthreads(); // trunk output in bufB
  dense_f32<256, 256, false>(kern(9), bias(9), bufB, bufA, row, wave);                 // bottleneck (no activation, model_utils.py:75)
  if (wave == 3) {                                                                      // sigma head: Dense_8 on the trunk output (:70)
    float s = 0.f;
    const float* __restrict__ w = kern(8);
    for (int k = 0; k < 256; ++k) s = fmaf(bufB[k * F32_ROWS + row], w[k], s);
    head[3 * F32_ROWS + row] = fadd(s, bias(8)[0]);
  }
  encode_f32<4>(make_float3(dr.x, dr.y, dr.z), bufA + 256 * F32_ROWS, row, wave);      // [bottleneck | pos_enc(dir, 0, 4)] (:82-83)
  __syncthreads();
  dense_f32<283, 128, true>(kern(10), bias(10), bufA, bufB, row, wave);                // view layer
  __syncthreads();
  if (wave < 3) {                                                                       // rgb head: Dense_11 [128][3]
    float s = 0.f;
    const float* __restrict__ w = kern(11);
    for (int k = 0; k < 128; ++k) s = fmaf(bufB[k * F32_ROWS + row], w[3 * k + wave], s);
    head[wave * F32_ROWS + row] = fadd(s, bias(11)[wave]);
  }
  __syncthreads();
  if (wave == 0 && ok) out_raw[r] = make_float4(head[row], head[F32_ROWS + row], head[2 * F32_ROWS + row], head[3 * F32_ROWS + row]);
}

int launch_fwd_f32(const void* packed, const float* rows_pd, const float* rows_dr, const int32_t* node_of_sample, int32_t B, long long total_rows,
                   float* out_raw, hipStream_t st) {
  const size_t lds = (size_t)(F32_XSTRIDE + 256 + 4) * F32_ROWS * sizeof(float);      // 145 KiB
  static DeviceOnce attr_set;
  if (attr_set.need()) {
    RNERF_CHECK_HIP(hipFuncSetAttribute((const void*)nerfmlp_fwd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.set();
  }
  const unsigned grid = (unsigned)((total_rows + F32_ROWS - 1) / F32_ROWS);
  hipLaunchKernelGGL(nerfmlp_fwd_f32_kernel, dim3(grid), dim3(256), lds, st, (const float*)packed, (const float4*)rows_pd, (const float4*)rows_dr, node_of_sample, B,
                     total_rows, (float4*)out_raw);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

}  // namespace rnerf
