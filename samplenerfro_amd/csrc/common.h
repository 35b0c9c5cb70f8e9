// Shared helpers for librnerf.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/rnerf.h"

// Experiment / test switches.  The product library reads NO environment variable (include/rnerf.h: no global state, nothing a stray
// RNERF_* variable on a bench box could change): RNERF_ENV folds to a null pointer and every switch to its default at compile time.
// librnerf_experiments.so (build.py, -DRNERF_EXPERIMENTS) is the same source with the switches live, for tools/ and for the tests that
// compare two launch shapes bit for bit (tests/test_gpu_half_tiles.py, tests/test_gpu_composite_lanes.py).
#ifdef RNERF_EXPERIMENTS
#include <stdlib.h>
#define RNERF_ENV(name) getenv(name)
#else
#define RNERF_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace rnerf {

void set_error(const char* fmt, ...);

// "Once per device" for the lazy kernel attributes (hipFuncSetAttribute is per device; a process may drive several).  Racing host threads
// at worst repeat an idempotent call: the flag is only ever set after the call succeeded.
struct DeviceOnce {
  volatile unsigned char done[64] = {0};
  int dev() const { int d = 0; return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) ? d : 0; }
  bool need() const { return !done[dev()]; }
  void set() { done[dev()] = 1; }
};

#define RNERF_CHECK_ARG(cond, ...)                 \
  do {                                             \
    if (!(cond)) {                                 \
      ::rnerf::set_error(__VA_ARGS__);             \
      return RNERF_ERR_ARG;                        \
    }                                              \
  } while (0)

#define RNERF_CHECK_HIP(expr)                                                          \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      ::rnerf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RNERF_ERR_HIP;                                                            \
    }                                                                                  \
  } while (0)

#define RNERF_CHECK_LAUNCH() RNERF_CHECK_HIP(hipGetLastError())

// float32 view of the grid geometry; every value is rounded from the double the reference computes
// (rnerf/ior_utils.py:140-144: ndelta = (nmax-nmin)/(ndim-1) in Python doubles, then weak-typed to f32).
struct GridParams {
  int dx, dy, dz;
  float nminx, nminy, nminz;
  float ndx, ndy, ndz;        // f32(ndelta)
  float tdx, tdy, tdz;        // f32(2*ndelta)
  // table addressing, both layouts (enum rnerf_table_layout) in one form: the BYTE offset of entry (x, y, z) is the sum over the axes of
  // (i >> 1) * sa[axis] + (i & 1) * sb[axis].  REFERENCE order: sb = the axis stride, sa = 2 sb; BRICKS: sa = the brick stride along
  // the axis, sb = 64 / 32 / 16 (the entry's place inside its 2x2x2 brick).
  int layout;
  unsigned sa[3], sb[3];
  unsigned long long table_bytes;
};
__host__ __device__ __forceinline__ size_t table_offset(const GridParams& g, int x, int y, int z) {      // in bytes
  return (size_t)(x >> 1) * g.sa[0] + (size_t)(x & 1) * g.sb[0] + (size_t)(y >> 1) * g.sa[1] + (size_t)(y & 1) * g.sb[1] +
         (size_t)(z >> 1) * g.sa[2] + (size_t)(z & 1) * g.sb[2];
}

inline bool make_grid_params(const rnerf_grid* g, GridParams* p) {
  if (!g) return false;
  for (int i = 0; i < 3; ++i)
    if (g->dims[i] < 2) return false;
  double nd[3];
  for (int i = 0; i < 3; ++i) nd[i] = (g->nmax[i] - g->nmin[i]) / (g->dims[i] - 1.0);
  p->dx = g->dims[0]; p->dy = g->dims[1]; p->dz = g->dims[2];
  p->nminx = (float)g->nmin[0]; p->nminy = (float)g->nmin[1]; p->nminz = (float)g->nmin[2];
  p->ndx = (float)nd[0]; p->ndy = (float)nd[1]; p->ndz = (float)nd[2];
  p->tdx = (float)(2 * nd[0]); p->tdy = (float)(2 * nd[1]); p->tdz = (float)(2 * nd[2]);
  p->layout = g->layout;
  if (g->layout == RNERF_TABLE_REFERENCE) {
    const unsigned long long sy = (unsigned long long)g->dims[2] * 16ull, sx = sy * (unsigned long long)g->dims[1];
    if (2 * sx > 0xFFFFFFFFull) return false;
    p->sb[0] = (unsigned)sx; p->sb[1] = (unsigned)sy; p->sb[2] = 16u;
    for (int i = 0; i < 3; ++i) p->sa[i] = 2u * p->sb[i];
    p->table_bytes = sx * (unsigned long long)g->dims[0];
  } else if (g->layout == RNERF_TABLE_BRICKS) {
    const unsigned long long by = (g->dims[1] + 1) / 2, bz = (g->dims[2] + 1) / 2, bx = (g->dims[0] + 1) / 2;
    if (by * bz * 128ull > 0xFFFFFFFFull) return false;
    p->sa[0] = (unsigned)(by * bz * 128ull); p->sa[1] = (unsigned)(bz * 128ull); p->sa[2] = 128u;
    p->sb[0] = 64u; p->sb[1] = 32u; p->sb[2] = 16u;
    p->table_bytes = bx * by * bz * 128ull;
  } else {
    return false;
  }
  return true;
}
// the marching kernels address the table with 32-bit byte offsets formed by 24-bit multiplies: table < 4 GiB, every factor < 2^24
// (the so3 / adjoint kernels multiply by sa in either layout)
inline bool grid_fits_u32(const GridParams& p) {
  return p.table_bytes < 4294967296ull && p.sa[0] < 16777216u && p.sa[1] < 16777216u && p.dx < 16777216 && p.dy < 16777216 && p.dz < 16777216;
}
// march_kernel: the REFERENCE layout multiplies the index by sb (the plain row strides), only BRICKS by sa — a non-cubic reference-order
// grid with dims[1] * dims[2] * 16 in [2^23, 2^24) is fine here although the so3 march rejects it
inline bool grid_fits_march(const GridParams& p) {
  const unsigned* s = p.layout == RNERF_TABLE_REFERENCE ? p.sb : p.sa;
  return p.table_bytes < 4294967296ull && s[0] < 16777216u && s[1] < 16777216u && p.dx < 16777216 && p.dy < 16777216 && p.dz < 16777216;
}

// Individually rounded fp32 ops: the march / lookup / resample kernels must not contract a*b+c
// (bit-exact integer indices against the oracle).  The library is also built with -ffp-contract=off.
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }
// sqrtf (not __fsqrt_rn, which lowers to the bare 1-ulp v_sqrt_f32) gives the correctly rounded IEEE result
__device__ __forceinline__ float fsqrt(float a) { return sqrtf(a); }

// x / d for a launch-constant divisor d: RN_f32(double(x) * RN_f64(1/d)) equals the correctly rounded f32 quotient
// (the f64 product is within 2^-52 of x/d, while a f32/f32 quotient is never closer than 2^-49 to a rounding tie and
// never exactly on one), at 3 instructions instead of the ~11 of the IEEE f32 division sequence.
__device__ __forceinline__ float div_const(float x, double rcp_d) { return (float)((double)x * rcp_d); }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// a*(1-t) + b*t with the reference's op order (rnerf/ior_utils.py:214-222)
__device__ __forceinline__ float4 lerp4(const float4 a, const float4 b, float omt, float t) {
  float4 r;
  r.x = fadd(fmul(a.x, omt), fmul(b.x, t));
  r.y = fadd(fmul(a.y, omt), fmul(b.y, t));
  r.z = fadd(fmul(a.z, omt), fmul(b.z, t));
  r.w = fadd(fmul(a.w, omt), fmul(b.w, t));
  return r;
}

// VoxMLP._linear3 (rnerf/ior_utils.py:188-223), split in two so a caller can put independent work between issuing the
// 8 corner loads and consuming them.  idx6 (nullable) receives the clamped x0,x1,y0,y1,z0,z1.
struct TriCell {
  float4 d000, d100, d001, d101, d010, d110, d011, d111;
  float xd, yd, zd;
};

struct GridRcp { double x, y, z; };      // RN_f64(1 / f32(ndelta)) per axis, for div_const
__device__ __forceinline__ GridRcp grid_rcp(const GridParams& g) { return GridRcp{1.0 / (double)g.ndx, 1.0 / (double)g.ndy, 1.0 / (double)g.ndz}; }

// OFF32: 32-bit byte offsets from the uniform table base (the launcher checks that the table is < 4 GiB): scalar base + per-lane
// offset addressing instead of 64-bit pointer arithmetic per corner.
template <bool RCP = false, bool OFF32 = false>
__device__ __forceinline__ void trilinear_load(const float4* __restrict__ tab, const GridParams& g, float px, float py,
                                               float pz, int* idx6, TriCell& c, const GridRcp* rcp = nullptr) {
  float x, y, z;
  if constexpr (RCP) {      // the same correctly rounded quotients at 3 instructions each instead of ~11
    x = div_const(fsub(px, g.nminx), rcp->x); y = div_const(fsub(py, g.nminy), rcp->y); z = div_const(fsub(pz, g.nminz), rcp->z);
  } else {
    x = fdiv(fsub(px, g.nminx), g.ndx); y = fdiv(fsub(py, g.nminy), g.ndy); z = fdiv(fsub(pz, g.nminz), g.ndz);
  }
  const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
  int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
  // (x - x0) / (x1 - x0): the divisor is exactly 1.0f, so the quotient is the (rounded) difference.
  c.xd = fsub(x, (float)x0); c.yd = fsub(y, (float)y0); c.zd = fsub(z, (float)z0);
  x0 = clampi(x0, 0, g.dx - 1); x1 = clampi(x1, 0, g.dx - 1);
  y0 = clampi(y0, 0, g.dy - 1); y1 = clampi(y1, 0, g.dy - 1);
  z0 = clampi(z0, 0, g.dz - 1); z1 = clampi(z1, 0, g.dz - 1);
  if (idx6) { idx6[0] = x0; idx6[1] = x1; idx6[2] = y0; idx6[3] = y1; idx6[4] = z0; idx6[5] = z1; }
  // byte offsets per axis (both table layouts: GridParams::sa / sb)
  if constexpr (OFF32) {
    const char* __restrict__ tb = (const char*)tab;
    const unsigned ax0 = (unsigned)(x0 >> 1) * g.sa[0] + (unsigned)(x0 & 1) * g.sb[0], ax1 = (unsigned)(x1 >> 1) * g.sa[0] + (unsigned)(x1 & 1) * g.sb[0];
    const unsigned ay0 = (unsigned)(y0 >> 1) * g.sa[1] + (unsigned)(y0 & 1) * g.sb[1], ay1 = (unsigned)(y1 >> 1) * g.sa[1] + (unsigned)(y1 & 1) * g.sb[1];
    const unsigned az0 = (unsigned)(z0 >> 1) * g.sa[2] + (unsigned)(z0 & 1) * g.sb[2], az1 = (unsigned)(z1 >> 1) * g.sa[2] + (unsigned)(z1 & 1) * g.sb[2];
#ifdef RNERF_TRILINEAR_NOLOAD      /* profiling ablation: the address arithmetic without the 8 gathers */
    {
      const float f = __uint_as_float(0x3f800000u + ((ax0 + ay0 + az0 + ax1 + ay1 + az1) & 16u) / 16u);
      const float4 v = make_float4(f, 0.f, 0.f, 0.f);
      c.d000 = v; c.d100 = v; c.d001 = v; c.d101 = v; c.d010 = v; c.d110 = v; c.d011 = v; c.d111 = v;
      return;
    }
#endif
    const unsigned b00 = ax0 + ay0, b10 = ax1 + ay0, b01 = ax0 + ay1, b11 = ax1 + ay1;
    c.d000 = *(const float4*)(tb + (b00 + az0)); c.d100 = *(const float4*)(tb + (b10 + az0));
    c.d001 = *(const float4*)(tb + (b00 + az1)); c.d101 = *(const float4*)(tb + (b10 + az1));
    c.d010 = *(const float4*)(tb + (b01 + az0)); c.d110 = *(const float4*)(tb + (b11 + az0));
    c.d011 = *(const float4*)(tb + (b01 + az1)); c.d111 = *(const float4*)(tb + (b11 + az1));
    return;
  }
  const char* __restrict__ tb = (const char*)tab;
  const size_t ax0 = (size_t)(x0 >> 1) * g.sa[0] + (size_t)(x0 & 1) * g.sb[0], ax1 = (size_t)(x1 >> 1) * g.sa[0] + (size_t)(x1 & 1) * g.sb[0];
  const size_t ay0 = (size_t)(y0 >> 1) * g.sa[1] + (size_t)(y0 & 1) * g.sb[1], ay1 = (size_t)(y1 >> 1) * g.sa[1] + (size_t)(y1 & 1) * g.sb[1];
  const size_t az0 = (size_t)(z0 >> 1) * g.sa[2] + (size_t)(z0 & 1) * g.sb[2], az1 = (size_t)(z1 >> 1) * g.sa[2] + (size_t)(z1 & 1) * g.sb[2];
  c.d000 = *(const float4*)(tb + (ax0 + ay0 + az0)); c.d100 = *(const float4*)(tb + (ax1 + ay0 + az0));
  c.d001 = *(const float4*)(tb + (ax0 + ay0 + az1)); c.d101 = *(const float4*)(tb + (ax1 + ay0 + az1));
  c.d010 = *(const float4*)(tb + (ax0 + ay1 + az0)); c.d110 = *(const float4*)(tb + (ax1 + ay1 + az0));
  c.d011 = *(const float4*)(tb + (ax0 + ay1 + az1)); c.d111 = *(const float4*)(tb + (ax1 + ay1 + az1));
}

__device__ __forceinline__ float4 trilinear_finish(const TriCell& c) {
  const float oxd = fsub(1.0f, c.xd), oyd = fsub(1.0f, c.yd), ozd = fsub(1.0f, c.zd);
  const float4 c00 = lerp4(c.d000, c.d100, oxd, c.xd);
  const float4 c01 = lerp4(c.d001, c.d101, oxd, c.xd);
  const float4 c10 = lerp4(c.d010, c.d110, oxd, c.xd);
  const float4 c11 = lerp4(c.d011, c.d111, oxd, c.xd);
  const float4 c0 = lerp4(c00, c10, oyd, c.yd);
  const float4 c1 = lerp4(c01, c11, oyd, c.yd);
  return lerp4(c0, c1, ozd, c.zd);
}

__device__ __forceinline__ float4 trilinear(const float4* __restrict__ tab, const GridParams& g, float px, float py,
                                            float pz, int* idx6) {
  TriCell c;
  trilinear_load(tab, g, px, py, pz, idx6, c);
  return trilinear_finish(c);
}

// ---- quad helpers: 4 consecutive lanes cooperate on one ray (march, compositing) -------------------------------------------
template <int SRC>
__device__ __forceinline__ int quad_bcast_i(int v) {   // value of lane SRC of this lane's quad (DPP quad_perm, no LDS)
  return __builtin_amdgcn_update_dpp(0, v, SRC * 0x55, 0xF, 0xF, true);
}
template <int SRC>
__device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, quad_bcast_i<SRC>(__builtin_bit_cast(int, v)));
}
// value of the NEXT lane of the quad (lane 3 keeps its own)
__device__ __forceinline__ float quad_next(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 1 | (2 << 2) | (3 << 4) | (3 << 6), 0xF, 0xF, true));
}
// value of the PREVIOUS lane of the quad (lane 0 keeps its own)
__device__ __forceinline__ float quad_prev(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0 | (0 << 2) | (1 << 4) | (2 << 6), 0xF, 0xF, true));
}

// (a0*a0 + a1*a1) + a2*a2 over the three coordinate lanes, in the reference's summation order
__device__ __forceinline__ float quad_sumsq3(float a) {
  const float a2 = fmul(a, a);
  return fadd(fadd(quad_bcast<0>(a2), quad_bcast<1>(a2)), quad_bcast<2>(a2));
}

// The marching waves (march.hip, march_all_kernel in mlp.hip) are bound by instruction issue, so its integer glue is written as the instructions the ISA has for it (hipcc emits 3-4 for
// each): clamp to [0, hi] = the median of (v, 0, hi); floor + convert in one; a*(1-t) + b*t with both products in one packed multiply
// (v_pk_mul_f32 rounds each half like v_mul_f32: the same individually rounded ops in the same order).
__device__ __forceinline__ int clamp0(int v, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(hi));
  return r;
}
__device__ __forceinline__ int floor_to_int(float x) {
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lerp_pk(float a, float b, f32x2_t w) {       // w = (1 - t, t)
  const f32x2_t v = {a, b};
  const f32x2_t m = v * w;
  float r;      // as the instruction: left to itself hipcc pairs the sums of two lerps into v_pk_add_f32 and pays for it in register moves
  asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(m.x), "v"(m.y));
  return r;
}

}  // namespace rnerf
