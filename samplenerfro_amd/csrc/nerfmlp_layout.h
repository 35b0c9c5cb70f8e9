// Shared by csrc/mlp.hip (the MFMA engines) and csrc/mlp_f32.hip (the exact-fp32 arbiter): the flat parameter layout of a NerfMLP and the
// sine of the positional encoding.
#pragma once
#include "common.h"

namespace rnerf {

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

// sin for the positional encoding: 3-constant Cody-Waite reduction by pi/2 (exact products through fma; |a| < ~2^15) and the
// Cephes sinf/cosf minimax polynomials on [-pi/4, pi/4] (~1 ulp).  ~20 VALU ops instead of the ~100 of the generic ocml
// sinf with its Payne-Hanek path.  The argument itself is formed exactly like the reference: fl(fl(x * 2^d) + fl(pi/2)).
// (cut into four dependent stages so that EncWork can issue them one per MFMA slot; pe_sin runs the same stages back to back)
struct PeSin {
  float a, k, r, z, sp;
  __device__ __forceinline__ void s0(float arg) { a = arg; k = rintf(a * 0.63661977236758134f); }
  __device__ __forceinline__ void s1() {
    r = fmaf(-k, 1.5707963705062866f, a);
    r = fmaf(-k, -4.3711388286737929e-08f, r);
    r = fmaf(-k, -1.7151245100059311e-15f, r);
    z = r * r;
  }
  __device__ __forceinline__ void s2() { sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r); }
  __device__ __forceinline__ float s3() const {
    const int q = (int)k;
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
    const float v = (q & 1) ? cp : sp;
    return (q & 2) ? -v : v;
  }
};
__device__ __forceinline__ float pe_sin(float a) {
  PeSin p;
  p.s0(a); p.s1(); p.s2();
  return p.s3();
}
__device__ __forceinline__ float pe_cos(float a) {      // the cosine from the same reduction and polynomials (the next quadrant's sine)
  PeSin p;
  p.s0(a); p.s1(); p.s2();
  const int q = (int)p.k;
  const float z = p.z;
  const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
  const float vc = (q & 1) ? p.sp : cp;
  return ((q + 1) & 2) ? -vc : vc;
}
// sin and cos of one argument from one range reduction (cos a = the next quadrant's sine)
__device__ __forceinline__ void pe_sincos(float a, float& sn, float& cs) {
  PeSin p;
  p.s0(a); p.s1(); p.s2();
  const int q = (int)p.k;
  const float z = p.z;
  const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
  const float vs = (q & 1) ? cp : p.sp, vc = (q & 1) ? p.sp : cp;
  sn = (q & 2) ? -vs : vs;
  cs = ((q + 1) & 2) ? -vc : vc;
}

}  // namespace rnerf
