/* TEST INFRASTRUCTURE — a second, independently written restatement, in plain scalar C, of the part of the path whose integer outputs carry
 * the bit-exact contract: the (n, grad n) table, the trilinear lookup with clamp-to-edge and the eikonal march.  The numpy oracle
 * (oracle/ref_np.py) is a vectorised reading of the same reference lines; this file was written from the reference, one ray and one voxel
 * at a time, and tests/test_oracle_c_ref.py holds the two to each other bit for bit (positions, directions, distances, looked-up values
 * and the six clamped voxel indices of every step).  Two readings agreeing narrows the room for a slip in either; it does not pin the
 * reference itself (no JAX here: DESIGN.md section 1).
 *
 * What it restates (paths under the reference checkout):
 *   rnerf/ior_utils.py:140-144   ndelta = (nmax - nmin) / (ndim - 1.)            Python doubles
 *   rnerf/ior_utils.py:161-172   data = concat(grid, central differences of the edge-padded grid / (2 ndelta))
 *   rnerf/ior_utils.py:188-223   _linear3: coordinates (p - nmin) / ndelta, floor, weights BEFORE the clamp, clamp to edge, 7 lerps
 *                                 a * (1 - t) + b * t, flat index ndim1 * ndim2 * x + ndim2 * y + z
 *   rnerf/eikonal_utils.py:29-49 OneEikonalStep (stage radiance*): next_rp = rp + step / n * rd, next_rd = rd + step * grad,
 *                                 next_rt = rt + |rp - next_rp|
 *   rnerf/eikonal_utils.py:100-124 PathSampler.__call__: node k = the state BEFORE step k, directions safe-l2-normalised
 *   rnerf/math_utils.py:6-12     safe_l2_normalize: x / sqrt(max(sum x^2, 1e-6))
 *   rnerf/models.py:121-122      step_size = (far - near) / (num_samples - 1)   a Python double, met as float32
 * Arithmetic: every operation is one IEEE float32 operation (the reference's jnp float32 with weakly typed Python scalars: a Python
 * double constant is rounded to float32 before it meets float32 data); sums of three squares are associated (x^2 + y^2) + z^2 like the
 * numpy oracle and the HIP kernels (XLA leaves the order of a reduction unspecified).  Build: gcc -O2 -ffp-contract=off (no FMA, no fast math).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void ndelta_of(const int32_t ndim[3], const double nmin[3], const double nmax[3], double nd[3]) {
  for (int a = 0; a < 3; ++a) nd[a] = (nmax[a] - nmin[a]) / (ndim[a] - 1.0);
}

/* grid: float[G0*G1*G2], x slowest.  table: float[G0*G1*G2][4] = (n, dn/dx, dn/dy, dn/dz). */
void rnerf_ref_build_table(const float* grid, const int32_t ndim[3], const double nmin[3], const double nmax[3], float* table) {
  double nd[3];
  ndelta_of(ndim, nmin, nmax, nd);
  const float two_dx = (float)(2 * nd[0]), two_dy = (float)(2 * nd[1]), two_dz = (float)(2 * nd[2]);
  const int G0 = ndim[0], G1 = ndim[1], G2 = ndim[2];
  for (int x = 0; x < G0; ++x)
    for (int y = 0; y < G1; ++y)
      for (int z = 0; z < G2; ++z) {
        /* "edge" padding by one voxel: the neighbour beyond a face is the face voxel itself */
        const int xm = x > 0 ? x - 1 : 0, xp = x < G0 - 1 ? x + 1 : G0 - 1;
        const int ym = y > 0 ? y - 1 : 0, yp = y < G1 - 1 ? y + 1 : G1 - 1;
        const int zm = z > 0 ? z - 1 : 0, zp = z < G2 - 1 ? z + 1 : G2 - 1;
#define AT(i, j, k) grid[((size_t)(i) * G1 + (j)) * G2 + (k)]
        float* t = table + 4 * (((size_t)x * G1 + y) * G2 + z);
        t[0] = AT(x, y, z);
        t[1] = (AT(xp, y, z) - AT(xm, y, z)) / two_dx;
        t[2] = (AT(x, yp, z) - AT(x, ym, z)) / two_dy;
        t[3] = (AT(x, y, zp) - AT(x, y, zm)) / two_dz;
#undef AT
      }
}

/* one lookup: p[3] -> out[4]; idx (nullable) = clamped x0, x1, y0, y1, z0, z1 */
static void linear3_one(const float* table, const int32_t ndim[3], const float nminf[3], const float ndf[3], const float p[3], float out[4], int32_t* idx) {
  int i0[3], i1[3];
  float w[3];
  for (int a = 0; a < 3; ++a) {
    const float c = (p[a] - nminf[a]) / ndf[a];
    const float f = floorf(c);
    const int lo = (int)f, hi = lo + 1;
    w[a] = (c - (float)lo) / (float)(hi - lo);          /* the weight uses the UNclamped cell (ior_utils.py:201-203) */
    i0[a] = clampi(lo, 0, ndim[a] - 1);
    i1[a] = clampi(hi, 0, ndim[a] - 1);
  }
  if (idx) { idx[0] = i0[0]; idx[1] = i1[0]; idx[2] = i0[1]; idx[3] = i1[1]; idx[4] = i0[2]; idx[5] = i1[2]; }
  const size_t s1 = (size_t)ndim[1] * ndim[2], s2 = (size_t)ndim[2];
  const float xd = w[0], yd = w[1], zd = w[2];
  const float omx = 1.0f - xd, omy = 1.0f - yd, omz = 1.0f - zd;
  for (int ch = 0; ch < 4; ++ch) {
#define D(X, Y, Z) table[4 * (s1 * (size_t)(X) + s2 * (size_t)(Y) + (size_t)(Z)) + ch]
    const float c00 = D(i0[0], i0[1], i0[2]) * omx + D(i1[0], i0[1], i0[2]) * xd;
    const float c01 = D(i0[0], i0[1], i1[2]) * omx + D(i1[0], i0[1], i1[2]) * xd;
    const float c10 = D(i0[0], i1[1], i0[2]) * omx + D(i1[0], i1[1], i0[2]) * xd;
    const float c11 = D(i0[0], i1[1], i1[2]) * omx + D(i1[0], i1[1], i1[2]) * xd;
#undef D
    const float c0 = c00 * omy + c10 * yd;
    const float c1 = c01 * omy + c11 * yd;
    out[ch] = c0 * omz + c1 * zd;
  }
}

void rnerf_ref_linear3(const float* table, const int32_t ndim[3], const double nmin[3], const double nmax[3], const float* pts, int64_t n,
                       float* out4, int32_t* idx6) {
  double nd[3];
  ndelta_of(ndim, nmin, nmax, nd);
  const float nminf[3] = {(float)nmin[0], (float)nmin[1], (float)nmin[2]}, ndf[3] = {(float)nd[0], (float)nd[1], (float)nd[2]};
  for (int64_t i = 0; i < n; ++i) linear3_one(table, ndim, nminf, ndf, pts + 3 * i, out4 + 4 * i, idx6 ? idx6 + 6 * i : NULL);
}

static float sumsq3(const float v[3]) { return (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]; }

/* origins, viewdirs: float[B][3].  Outputs, ray-major like the reference's [batch, sample, feature]: pos, dir (normalised), grad: float[B][N][3];
 * dist, ior: float[B][N]; vox (nullable): int32[B][N][6]. */
void rnerf_ref_path_sampler(const float* table, const int32_t ndim[3], const double nmin[3], const double nmax[3], const float* origins,
                            const float* viewdirs, int32_t B, double near, double far, int32_t N, float* pos, float* dir, float* dist, float* ior,
                            float* grad, int32_t* vox) {
  double nd[3];
  ndelta_of(ndim, nmin, nmax, nd);
  const float nminf[3] = {(float)nmin[0], (float)nmin[1], (float)nmin[2]}, ndf[3] = {(float)nd[0], (float)nd[1], (float)nd[2]};
  const float step = (float)((far - near) / (N - 1));
  const float nearf = (float)near;
  for (int32_t b = 0; b < B; ++b) {
    float rp[3], rd[3], rt = nearf;
    for (int a = 0; a < 3; ++a) { rd[a] = viewdirs[3 * b + a]; rp[a] = origins[3 * b + a] + nearf * rd[a]; }
    for (int32_t k = 0; k < N; ++k) {
      const size_t o = (size_t)b * N + k;
      float nrm = sumsq3(rd);
      nrm = sqrtf(nrm > 1e-6f ? nrm : 1e-6f);
      for (int a = 0; a < 3; ++a) { pos[3 * o + a] = rp[a]; dir[3 * o + a] = rd[a] / nrm; }
      dist[o] = rt;
      float c[4];
      linear3_one(table, ndim, nminf, ndf, rp, c, vox ? vox + 6 * o : NULL);
      ior[o] = c[0];
      for (int a = 0; a < 3; ++a) grad[3 * o + a] = c[1 + a];
      const float s = step / c[0];
      float nrp[3], dl[3];
      for (int a = 0; a < 3; ++a) { nrp[a] = rp[a] + s * rd[a]; dl[a] = rp[a] - nrp[a]; }
      for (int a = 0; a < 3; ++a) rd[a] = rd[a] + step * c[1 + a];
      rt = rt + sqrtf(sumsq3(dl));
      for (int a = 0; a < 3; ++a) rp[a] = nrp[a];
    }
  }
}

/* ---- S1 / S2: hierarchical resampling along the bent path --------------------------------------------------------------------------------
 *   rnerf/model_utils.py:312-374  sorted_piecewise_constant_pdf (the uniform draws u are an input: the key chain is restated elsewhere)
 *   rnerf/model_utils.py:377-435  sample_pdf: z = sort(concat(z_vals[:, jitter], fine)), node idx = max(searchsorted(z_vals, z, "left") - 1, 0),
 *                                 pos = path_pos[idx] + path_dir[idx] * (z - z_vals[idx]), dir = path_dir[idx]
 * The reference finds the interval of a draw with a dense [bins, draws] mask and max / min reductions; this reading walks the (sorted)
 * cdf instead: the last k with u >= cdf[k] starts the interval — the same interval by the reference's own comment ("takes advantage of the
 * fact that x is sorted").  Sums that XLA may associate freely (the weight sum) are sequential, as in the numpy oracle and the HIP kernels.
 * bins: float[B][nb + 1], weights: float[B][nb], u: float[B][F], z_vals / path_pos / path_dir over the full path of N nodes, jitter: int32[S].
 * Outputs: z [B][S + F], pos / dir [B][S + F][3], idx int32 [B][S + F].  scratch: float[2 * (nb + 1)] per call (caller-provided). */
static void sort_floats(float* a, int n) {       /* insertion sort: stable, exact, n <= a few hundred */
  for (int i = 1; i < n; ++i) {
    const float v = a[i];
    int j = i - 1;
    while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; }
    a[j + 1] = v;
  }
}

void rnerf_ref_sample_pdf(const float* u, const float* bins, const float* weights, int32_t B, int32_t nb, int32_t F, const float* z_vals,
                          const float* path_pos, const float* path_dir, int32_t N, const int32_t* jitter, int32_t S, float* z_out, float* pos_out,
                          float* dir_out, int32_t* idx_out, float* scratch) {
  float* w = scratch;                 /* nb padded weights, then reused for the pdf */
  float* cdf = scratch + nb + 1;      /* nb + 1 entries: 0, min(1, cumsum(pdf[:-1])), 1 */
  const float eps = 1e-5f;
  for (int32_t b = 0; b < B; ++b) {
    const float* bn = bins + (size_t)b * (nb + 1);
    float sum = 0.0f;
    for (int k = 0; k < nb; ++k) sum = sum + weights[(size_t)b * nb + k];
    const float pad = (eps - sum) > 0.0f ? (eps - sum) : 0.0f;                  /* max(0, eps - weight_sum) */
    const float add = pad / (float)nb;
    const float tot = sum + pad;
    cdf[0] = 0.0f;
    float run = 0.0f;
    for (int k = 0; k < nb; ++k) {
      w[k] = (weights[(size_t)b * nb + k] + add) / tot;                        /* pdf */
      if (k < nb - 1) { run = run + w[k]; cdf[k + 1] = run < 1.0f ? run : 1.0f; }
    }
    cdf[nb] = 1.0f;
    float* z = z_out + (size_t)b * (S + F);
    for (int s = 0; s < S; ++s) z[s] = z_vals[(size_t)b * N + jitter[s]];
    for (int f = 0; f < F; ++f) {
      const float uu = u[(size_t)b * F + f];
      /* mask[k] = (u >= cdf[k]); x0 = max over k of where(mask, x[k], x[0]); x1 = min over k of where(!mask, x[k], x[last]) */
      float b0 = bn[0], b1 = bn[nb], c0 = cdf[0], c1 = cdf[nb];
      for (int k = 0; k <= nb; ++k) {
        if (uu >= cdf[k]) { if (bn[k] > b0) b0 = bn[k]; if (cdf[k] > c0) c0 = cdf[k]; }
        else { if (bn[k] < b1) b1 = bn[k]; if (cdf[k] < c1) c1 = cdf[k]; }
      }
      float t = (uu - c0) / (c1 - c0);
      if (t != t) t = 0.0f;                                                    /* nan_to_num(., 0) */
      t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);                             /* clip (also folds +-inf) */
      z[S + f] = b0 + t * (b1 - b0);
    }
    sort_floats(z, S + F);
    const float* zv = z_vals + (size_t)b * N;
    for (int i = 0; i < S + F; ++i) {
      int lo = 0, hi = N;                                                      /* searchsorted(zv, z[i], side="left"): first j with zv[j] >= z[i] */
      while (lo < hi) { const int mid = (lo + hi) / 2; if (zv[mid] < z[i]) lo = mid + 1; else hi = mid; }
      const int id = lo > 0 ? lo - 1 : 0;
      const size_t o = (size_t)b * N + id, q = (size_t)b * (S + F) + i;
      idx_out[q] = id;
      const float dz = z[i] - zv[id];
      for (int a = 0; a < 3; ++a) { const float d = path_dir[3 * o + a]; dir_out[3 * q + a] = d; pos_out[3 * q + a] = path_pos[3 * o + a] + d * dz; }
    }
  }
}
