// Eikonal march (E1/E2/E3).  Reference: rnerf/eikonal_utils.py:29-49 (OneEikonalStep), :100-124 (PathSampler.__call__),
// rnerf/math_utils.py:6-12 (safe_l2_normalize), step size rnerf/models.py:121-122.
//
// One lane per ray; the recurrence over nodes is serial per ray (each step's gather address depends on the previous
// step's result), rays are independent.  Node records are written sample-major ([node][ray]) so every store of a
// wave is one contiguous 1 KiB segment.  All arithmetic is individually rounded fp32 in the reference's op order, so
// positions and voxel indices are bit-identical to the fp32 oracle (no transcendental is involved).
#include "common.h"

namespace rnerf {

template <bool WANT_IOR, bool WANT_VOX>
__global__ void __launch_bounds__(64) march_kernel(const float4* __restrict__ table, GridParams g,
                                                   const float* __restrict__ origins, const float* __restrict__ viewdirs,
                                                   int B, float near, float step, int num_nodes,
                                                   float4* __restrict__ path_pd, float4* __restrict__ path_dr,
                                                   float4* __restrict__ path_ior, int* __restrict__ vox) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B) return;
  const float ox = origins[3 * r], oy = origins[3 * r + 1], oz = origins[3 * r + 2];
  float dx = viewdirs[3 * r], dy = viewdirs[3 * r + 1], dz = viewdirs[3 * r + 2];
  // eikonal_utils.py:104-106
  float px = fadd(ox, fmul(near, dx)), py = fadd(oy, fmul(near, dy)), pz = fadd(oz, fmul(near, dz));
  float rt = near;
  for (int k = 0; k < num_nodes; ++k) {
    const size_t o = (size_t)k * B + r;
    // node k = state before step k (eikonal_utils.py:112-114); direction safe-normalised (math_utils.py:6-12)
    const float nrm = fsqrt(fmaxf(fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz)), 1e-6f));
    path_pd[o] = make_float4(px, py, pz, rt);
    path_dr[o] = make_float4(fdiv(dx, nrm), fdiv(dy, nrm), fdiv(dz, nrm), 0.f);
    int id[6];
    const float4 c = trilinear(table, g, px, py, pz, WANT_VOX ? id : nullptr);
    if (WANT_IOR) path_ior[o] = c;
    if (WANT_VOX)
      for (int q = 0; q < 6; ++q) vox[6 * o + q] = id[q];
    // eikonal_utils.py:41-45
    const float s = fdiv(step, c.x);
    const float nx = fadd(px, fmul(s, dx)), ny = fadd(py, fmul(s, dy)), nz = fadd(pz, fmul(s, dz));
    dx = fadd(dx, fmul(step, c.y)); dy = fadd(dy, fmul(step, c.z)); dz = fadd(dz, fmul(step, c.w));
    const float ex = fsub(px, nx), ey = fsub(py, ny), ez = fsub(pz, nz);
    rt = fadd(rt, fsqrt(fadd(fadd(fmul(ex, ex), fmul(ey, ey)), fmul(ez, ez))));
    px = nx; py = ny; pz = nz;
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
  GridParams p;
  RNERF_CHECK_ARG(make_grid_params(g, &p), "rnerf_march: bad grid");
  const float stepf = (float)((far - near) / (num_nodes - 1));  // models.py:122, Python double -> f32
  const float nearf = (float)near;
  const dim3 block(64), grid((B + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(I, V)                                                                                               \
  hipLaunchKernelGGL((march_kernel<I, V>), grid, block, 0, st, (const float4*)table, p, origins, viewdirs, B, nearf, \
                     stepf, num_nodes, (float4*)path_pd, (float4*)path_dr, (float4*)path_ior, vox)
  if (path_ior && vox) LAUNCH(true, true);
  else if (path_ior) LAUNCH(true, false);
  else if (vox) LAUNCH(false, true);
  else LAUNCH(false, false);
#undef LAUNCH
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}
