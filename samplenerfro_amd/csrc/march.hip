// Eikonal march (E1/E2/E3).  Reference: rnerf/eikonal_utils.py:29-49 (OneEikonalStep), :100-124 (PathSampler.__call__),
// rnerf/math_utils.py:6-12 (safe_l2_normalize), step size rnerf/models.py:121-122.
//
// The recurrence over nodes is serial per ray (each step's gather address depends on the previous step's result), rays
// are independent, so the kernel is bound by the LATENCY of one step, not by bandwidth: a quad of 4 lanes works on one
// ray to shorten that chain.  Node records are written sample-major ([node][ray]): each store of a wave is one
// contiguous 256-byte segment.  All arithmetic is individually rounded fp32 in the reference's op order, so
// positions and voxel indices are bit-identical to the fp32 oracle (no transcendental is involved).
#include "common.h"

#include <stdlib.h>

namespace rnerf {

struct MarchParams {
  int dx, dy, dz;
  float nmin[3];
  double rcp_nd[3];   // RN_f64(1 / f32(ndelta))
  unsigned sa[3], sb[3];   // table addressing per axis (GridParams::sa / sb): byte offset = (i >> 1) * sa + (i & 1) * sb
};

// Eikonal march, 4 lanes per ray.  Lane q = lane&3 of a quad owns coordinate q of the ray state (q = 0,1,2) and
// component (q+1)&3 of the float4 table entries (lane 0..2: dn/dx, dn/dy, dn/dz; lane 3: n), so the three coordinate
// divisions, the 7 four-component lerps and the state update of one step run in parallel across the quad and the
// per-step dependent instruction chain is ~2.5x shorter than with one lane per ray.  Every value is produced by the same
// individually rounded fp32 ops, in the same order, as rnerf/eikonal_utils.py:29-49 + rnerf/ior_utils.py:188-223.
//
// Two waves per 16 rays (round 3).  The kernel is bound by the instruction issue of ONE wave per step (~4 cycles per instruction), and
// a third of a step's instructions produced the node RECORD — the safe-normalised direction (a correctly rounded sqrt + an IEEE
// division), the travelled distance (another sqrt), the two stores and their addresses — none of which the recurrence needs.  Wave 0
// (the marcher) now only advances the state and leaves (p, d) of every node in an LDS ring; wave 1 (the recorder), on another SIMD of
// the CU, turns them into records a chunk of kMarchChunk nodes later.  One s_barrier per chunk and wave: the recorder reads chunk c
// between barriers c and c + 1, the marcher overwrites that half of the ring only after barrier c + 1.  Same values, same bits.
// With the record gone a step takes 0.26 us, and two steps of lead no longer cover an HBM round trip on a 512^3 table (2.1 GB): the
// corners are gathered kMarchAhead steps ahead, into kMarchAhead + 1 register sets that rotate through an unrolled trip.
#ifndef RNERF_MARCH_AHEAD
#define RNERF_MARCH_AHEAD 2
#endif
constexpr int kMarchAhead = RNERF_MARCH_AHEAD;
constexpr int kMarchSets = kMarchAhead + 1;
constexpr int kMarchChunk = kMarchSets <= 4 ? 2 * kMarchSets : kMarchSets;      // nodes per ring half (a multiple of the register rotation); 2 x C x 512 B of LDS
static_assert(2 * kMarchChunk * 512 <= 12 * 1024, "the ring must fit beside the weight-gradient kernel's LDS (148 KiB)");

__device__ __forceinline__ void march_chunk_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// BRICKS: the table is in 2x2x2-brick order (rnerf_table_layout): two more integer instructions per gathered index, ~40 % fewer new cache
// lines per step on a table that does not fit the caches.
template <bool WANT_IOR, bool WANT_VOX, bool BRICKS>
__global__ void __launch_bounds__(128) march_kernel(const float* __restrict__ table, MarchParams g,
                                                    const float* __restrict__ origins, const float* __restrict__ viewdirs,
                                                    int B, float near, float step, int num_nodes,
                                                    float* __restrict__ path_pd, float* __restrict__ path_dr,
                                                    float* __restrict__ path_ior, int* __restrict__ vox, int rays_per_wg) {
  constexpr int C = kMarchChunk;
  __shared__ float2 ring[2][C][64];               // (p, d) of the lane's coordinate, per node
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // rays_per_wg = 16 (product), 8 or 4 (experiment, RNERF_MARCH_RPW): with fewer than 16 the upper quads of the wave replay the lower ones
  // (same values to the same addresses, no divergence).
  const int q = lane & 3;
  int r = blockIdx.x * rays_per_wg + ((lane >> 2) & (rays_per_wg - 1));
  if (r >= B) r = B - 1;   // surplus quads replay the last ray (same values to the same addresses, no divergence)
  const int qc = q < 3 ? q : 0;
  const int nchunks = (num_nodes + C - 1) / C;    // the marcher always runs whole chunks (surplus nodes: clamped gathers, nothing stored)
  const size_t node_stride = 4 * (size_t)B;

  if (wave != 0) {
    // ---- the recorder: node k = state before step k (eikonal_utils.py:112-114), direction safe-normalised (math_utils.py:6-12),
    //      distance = near + the lengths of the steps so far (eikonal_utils.py:46)
    float* __restrict__ out_pd = path_pd + 4 * (size_t)r + q;          // advanced by one node (B records) per node
    float* __restrict__ out_dr = path_dr + 4 * (size_t)r + q;
    float rt = near, p_prev = 0.f;
    for (int c = 0; c < nchunks; ++c) {
      march_chunk_barrier();                                           // chunk c is in the ring
      const float2* __restrict__ src = &ring[c & 1][0][lane];
#pragma unroll
      for (int s = 0; s < C; ++s) {
        const int k = c * C + s;
        if (k >= num_nodes) break;
        const float2 v = src[s * 64];
        const float p = v.x, d = v.y;
        if (k > 0) rt = fadd(rt, fsqrt(quad_sumsq3(fsub(p_prev, p))));
        p_prev = p;
        const float nrm = fsqrt(fmaxf(quad_sumsq3(d), 1e-6f));
#ifdef RNERF_MARCH_NT
        __builtin_nontemporal_store(q < 3 ? p : rt, out_pd);
        __builtin_nontemporal_store(q < 3 ? fdiv(d, nrm) : 0.f, out_dr);
#else
        *out_pd = q < 3 ? p : rt;
        *out_dr = q < 3 ? fdiv(d, nrm) : 0.f;
#endif
        out_pd += node_stride; out_dr += node_stride;
      }
    }
    return;
  }

  // ---- the marcher
  const float nmin_q = g.nmin[qc];
  const double rcp_q = g.rcp_nd[qc];
  const int dim_q = qc == 0 ? g.dx : (qc == 1 ? g.dy : g.dz);
  const unsigned comp = (q + 1) & 3;
  // byte offsets into the table fit 32 bits (checked by the launcher): one v_mad_u32_u24 + adds per corner, SGPR base
  const char* __restrict__ tabc = (const char*)table;   // uniform base; this lane's component goes into the 32-bit offset
  const unsigned cofs = comp * 4u;
  const unsigned stride_q = g.sb[qc];                                // byte stride of this lane's axis (reference order)
  const unsigned sa_q = g.sa[qc], sb_q = g.sb[qc];                   // brick stride / place inside the brick (BRICKS)
  const int hi_q = dim_q - 1;
  float d = q < 3 ? viewdirs[3 * r + qc] : 0.f;
  float p = q < 3 ? fadd(origins[3 * r + qc], fmul(near, d)) : 0.f;   // eikonal_utils.py:104-106

  // The gathers of steps k+1 and k+2 are issued EARLY, from predicted cells (grid coordinate extrapolated linearly from the
  // last two nodes), so that their latency overlaps the arithmetic of the steps before instead of adding to the per-step
  // dependent chain.  When the real cell of a step is known it is compared with the prediction (integer indices): equal ->
  // the values in flight are exactly the ones the reference gathers; different (ray within ~1 ulp of a cell face, a few per
  // million steps) -> they are gathered again.  Results stay bit-identical to the un-speculated march.
  struct Corners { float c[8]; int i0, i1; };     // 000 100 001 101 010 110 011 111 (xyz) + the indices they were gathered with
  Corners cs[kMarchSets];
  auto gather = [&](int i0, int i1, Corners& o) {
    o.i0 = i0; o.i1 = i1;
    // every lane scales its own axis, the quad exchanges byte offsets (the broadcasts fold into the adds as DPP operands)
    unsigned m0, m1;
    if constexpr (BRICKS) {
      m0 = __umul24((unsigned)i0 >> 1, sa_q) + __umul24((unsigned)i0 & 1u, sb_q);
      m1 = __umul24((unsigned)i1 >> 1, sa_q) + __umul24((unsigned)i1 & 1u, sb_q);
    } else {
      m0 = __umul24((unsigned)i0, stride_q); m1 = __umul24((unsigned)i1, stride_q);
    }
    const unsigned y0 = quad_bcast_i<1>(m0), y1 = quad_bcast_i<1>(m1);
    const unsigned z0 = quad_bcast_i<2>(m0) + cofs, z1 = quad_bcast_i<2>(m1) + cofs;
    const unsigned b00 = quad_bcast_i<0>(m0) + y0, b10 = quad_bcast_i<0>(m1) + y0, b01 = quad_bcast_i<0>(m0) + y1, b11 = quad_bcast_i<0>(m1) + y1;
#if defined(RNERF_MARCH_ABL) && (RNERF_MARCH_ABL & 1)   /* profiling ablation: no gathers */
    for (int i_ = 0; i_ < 8; ++i_) o.c[i_] = __uint_as_float(0x3f800000u + ((b00 + z0 + b11 + z1) & 1u));
    return;
#endif
    o.c[0] = *(const float*)(tabc + (b00 + z0)); o.c[1] = *(const float*)(tabc + (b10 + z0));
    o.c[2] = *(const float*)(tabc + (b00 + z1)); o.c[3] = *(const float*)(tabc + (b10 + z1));
    o.c[4] = *(const float*)(tabc + (b01 + z0)); o.c[5] = *(const float*)(tabc + (b11 + z0));
    o.c[6] = *(const float*)(tabc + (b01 + z1)); o.c[7] = *(const float*)(tabc + (b11 + z1));
  };
  auto predict = [&](float xp, Corners& o) {
    const int j = floor_to_int(xp);
    gather(clamp0(j, hi_q), clamp0(j + 1, hi_q), o);
  };
  float x_prev;
  {
    const float x = div_const(fsub(p, nmin_q), rcp_q);
    x_prev = div_const(fsub(fsub(p, fmul(step, d)), nmin_q), rcp_q);   // as if a vacuum step had led here
    const float dx0 = fsub(x, x_prev);
#pragma unroll
    for (int j = 0; j < kMarchAhead; ++j) predict(fadd(x, fmul((float)j, dx0)), cs[j]);
  }
  float* __restrict__ out_ior = WANT_IOR ? path_ior + 4 * (size_t)r + comp : nullptr;
  // one step; cn = corners of this step, nx = where the corners of step k + kMarchAhead are gathered to, slot = this node's ring entry
  auto one_step = [&](int k, Corners& cn, Corners& nx, float2* slot) {
    *slot = make_float2(p, d);                     // the node record is the recorder's business
    // ---- VoxMLP._linear3 addressing (ior_utils.py:188-211): one coordinate per lane
#if defined(RNERF_MARCH_ABL) && (RNERF_MARCH_ABL & 4)   /* profiling ablation: f32 multiply instead of the f64 product */
    const float x = fmul(fsub(p, nmin_q), (float)rcp_q);
#else
    const float x = div_const(fsub(p, nmin_q), rcp_q);
#endif
    const float fx = floorf(x);
    const int i = (int)fx;
    const float t = fsub(x, fx);                   // (x - x0) / (x1 - x0), divisor exactly 1
    const int i0 = clamp0(i, hi_q), i1 = clamp0(i + 1, hi_q);
#if !(defined(RNERF_MARCH_ABL) && (RNERF_MARCH_ABL & 8))   /* profiling ablation: no misprediction check */
    if (__builtin_amdgcn_ballot_w64(i0 != cn.i0 || i1 != cn.i1) != 0) gather(i0, i1, cn);     // mispredicted somewhere in the wave
#endif
    // ---- speculative gather for step k + kMarchAhead
    const float dx = fsub(x, x_prev);
    predict(kMarchAhead == 2 ? fadd(x, fadd(dx, dx)) : fadd(x, fmul((float)kMarchAhead, dx)), nx);
    x_prev = x;
    const float xd = quad_bcast<0>(t), yd = quad_bcast<1>(t), zd = quad_bcast<2>(t);
    if (WANT_VOX && q < 3 && k < num_nodes) { const size_t o = (size_t)k * B + r; vox[6 * o + 2 * q] = i0; vox[6 * o + 2 * q + 1] = i1; }
    // ---- 7 lerps a*(1-t) + b*t (ior_utils.py:214-222)
    const f32x2_t wx = {fsub(1.0f, xd), xd}, wy = {fsub(1.0f, yd), yd}, wz = {fsub(1.0f, zd), zd};
    const float c00 = lerp_pk(cn.c[0], cn.c[1], wx);
    const float c01 = lerp_pk(cn.c[2], cn.c[3], wx);
    const float c10 = lerp_pk(cn.c[4], cn.c[5], wx);
    const float c11 = lerp_pk(cn.c[6], cn.c[7], wx);
    const float c0 = lerp_pk(c00, c10, wy);
    const float c1 = lerp_pk(c01, c11, wy);
    const float c = lerp_pk(c0, c1, wz);   // lanes 0..2: grad component q, lane 3: n
    if (WANT_IOR && k < num_nodes) { *out_ior = c; out_ior += node_stride; }
    // ---- OneEikonalStep (eikonal_utils.py:41-45)
    const float n = quad_bcast<3>(c);
#if defined(RNERF_MARCH_ABL) && (RNERF_MARCH_ABL & 2)   /* profiling ablation: no IEEE division on the chain */
    const float s = fmul(step, n);
#else
    const float s = fdiv(step, n);
#endif
    p = fadd(p, fmul(s, d));
    d = fadd(d, fmul(step, c));
  };
  static_assert(C % kMarchSets == 0, "a chunk is a whole number of register rotations");
  for (int c = 0, k = 0; c < nchunks; ++c, k += C) {      // the corner register sets rotate instead of being copied
    float2* slot = &ring[c & 1][0][lane];
#pragma unroll
    for (int u = 0; u < C; ++u) one_step(k + u, cs[u % kMarchSets], cs[(u + kMarchAhead) % kMarchSets], slot + 64 * u);
    march_chunk_barrier();
  }
}

}  // namespace rnerf

using namespace rnerf;

extern "C" int rnerf_march(const float* table, const rnerf_grid* g, const float* origins, const float* viewdirs,
                           int32_t B, double near, double far, int32_t num_nodes, float* path_pd, float* path_dr,
                           float* path_ior, int32_t* vox, void* stream) {
  RNERF_CHECK_ARG(table && g && origins && viewdirs && path_pd && path_dr, "rnerf_march: null pointer");
  RNERF_CHECK_ARG(B > 0 && num_nodes >= 2, "rnerf_march: need B > 0 and num_nodes >= 2");
  RNERF_CHECK_ARG((((uintptr_t)table | (uintptr_t)path_pd | (uintptr_t)path_dr | (uintptr_t)path_ior) & 15) == 0,
                  "rnerf_march: table/path buffers must be 16-byte aligned");
  GridParams gp;
  RNERF_CHECK_ARG(make_grid_params(g, &gp), "rnerf_march: bad grid");
  RNERF_CHECK_ARG(grid_fits_march(gp), "rnerf_march: grid too large for 32-bit byte offsets (needs a table < 4 GiB and 24-bit row strides)");
  MarchParams p;
  p.dx = gp.dx; p.dy = gp.dy; p.dz = gp.dz;
  p.nmin[0] = gp.nminx; p.nmin[1] = gp.nminy; p.nmin[2] = gp.nminz;
  p.rcp_nd[0] = 1.0 / (double)gp.ndx; p.rcp_nd[1] = 1.0 / (double)gp.ndy; p.rcp_nd[2] = 1.0 / (double)gp.ndz;
  for (int i = 0; i < 3; ++i) { p.sa[i] = gp.sa[i]; p.sb[i] = gp.sb[i]; }
  const float stepf = (float)((far - near) / (num_nodes - 1));  // models.py:122, Python double -> f32
  const float nearf = (float)near;
  // rays per workgroup (marcher + recorder wave): 16 = every quad of the wave.  Spreading a 4096-ray batch over twice / four times the
  // workgroups (8 / 4 rays each, two / four marching waves per CU) was measured SLOWER (round 4, profiles/r04/march_experiments.txt:
  // 0.35 -> 0.45 ms at 64^3, 0.53 -> 0.57 at 512^3): the waves of a CU share its vector-memory path, which is what a step waits on.
  int rpw = 16;
  if (const char* e = RNERF_ENV("RNERF_MARCH_RPW")) rpw = atoi(e);       // experiment switch (tools/r04/march_rpw.sh)
  RNERF_CHECK_ARG(rpw == 16 || rpw == 8 || rpw == 4, "rnerf_march: RNERF_MARCH_RPW must be 16, 8 or 4");
  const dim3 block(128), grid((B + rpw - 1) / rpw);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH2(I, V, K)                                                                                           \
  hipLaunchKernelGGL((march_kernel<I, V, K>), grid, block, 0, st, table, p, origins, viewdirs, B, nearf, stepf, num_nodes, \
                     path_pd, path_dr, path_ior, vox, rpw)
#define LAUNCH(I, V) do { if (gp.layout == RNERF_TABLE_BRICKS) LAUNCH2(I, V, true); else LAUNCH2(I, V, false); } while (0)
  if (path_ior && vox) LAUNCH(true, true);
  else if (path_ior) LAUNCH(true, false);
  else if (vox) LAUNCH(false, true);
  else LAUNCH(false, false);
#undef LAUNCH
#undef LAUNCH2
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}
