/*
 * rnerf.h — C ABI of librnerf.so, the MI355X (gfx950) implementation of the SampleNeRFRO
 * volumetric-rendering hot path.
 *
 * The reference (alexkeroro86/SampleNeRFRO) has no FFI/plugin layer: the path sits behind
 * Python callables.  Each entry point below names the reference function
 * (file:line under the reference checkout) whose arithmetic it replaces; the Python host
 * (samplenerfro_amd/) re-creates the reference call surface (NerfModel.__call__, render_image,
 * train_step) on top of these.  INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless named h_*;
 *   - nothing is allocated per call; scratch comes from the caller (`*_workspace_bytes`);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return value: 0 = ok, <0 = error (message via rnerf_last_error(), thread local);
 *   - "sample-major" layout: an array indexed [s][b] stores sample/node s of ray b at s*B + b,
 *     so that one-lane-per-ray kernels read and write coalesced.
 *   - a "row record" is two float4 arrays: pd = (pos.x, pos.y, pos.z, dist) and
 *     dr = (dir.x, dir.y, dir.z, 0) with dir already safe-l2-normalised.
 */
#ifndef RNERF_H_
#define RNERF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this header: entry points AND struct layouts.  2 (round 5): rnerf_grid carries `layout` (so every later field of rnerf_model
 * moved), rnerf_train_cfg carries grads_stream / aux2_stream, rnerf_adam_cfg carries use_lr_override (lr_override alone is ignored).  A
 * consumer compares rnerf_version() with the RNERF_VERSION it was compiled against before the first call (the Python binding does:
 * samplenerfro_amd/_lib.py load()).
 * 3 (round 6): everything that moved after the first version-2 header — rnerf_composite_backward takes `int mask_mode`, rnerf_adam_cfg carries
 * skip_nonfinite, RNERF_ADAM_SCRATCH_FLOATS is 3076 (was 2052), every f16-based packed NerfMLP buffer ends with a bf16x3 range-safe
 * stream (rnerf_nerfmlp_packed_bytes grew) — and this round's additions: enum rnerf_backward gains F16X3_LO8, rnerf_sample_batch. */
#define RNERF_VERSION 3

enum rnerf_status {
  RNERF_OK = 0,
  RNERF_ERR_ARG = -1,   /* bad argument (null pointer, size, alignment) */
  RNERF_ERR_HIP = -2,   /* a HIP runtime call failed */
  RNERF_ERR_UNSUPPORTED = -3
};

/* MLP arithmetic. F32 (rnerf_nerfmlp_pack / rnerf_nerfmlp_forward only) = every Dense as ONE sequential chain of v_fma_f32 per output, then + bias: exact
 * fp32 semantics with a defined summation order — the on-device arbiter of the MFMA precisions (csrc/mlp_f32.hip; ~100 x slower, for
 * parity tests: it separates split-f16 error, the oracle's summation order and real bugs on the device); F16X3 / BF16X3 = hi/lo split of
 * both operands, 3 MFMAs per tile (error ~2^-21 / ~2^-16 per product); F16 / BF16 = single MFMA;
 * F16X2 (forward / inference only) = exact hi+lo f16 weights x activations rounded to f16, 2 MFMAs per tile: the f16x3 operand stream,
 * 2/3 of its matrix work; measured end-to-end |dRGB| vs f16x3 3e-5 .. 5e-5 on the bench workload — inside the 1e-4
 * contract for weights of ordinary size but without the 2x margin f16x3 keeps everywhere, hence opt-in.
 * F16F8 (forward / inference only) = f16 main term + the two cross terms of the hi/lo split on v_mfma_f32_32x32x16_fp8_fp8 (e4m3): the
 * instruction count and operand bytes of F16X3 at lower power (the engine is power-limited: ~8 % faster); measured end-to-end |dRGB| vs the
 * oracle 2e-6.  Its operand stream scales the weights by 2^14: a NerfMLP weight of magnitude >= 3.99 raises a flag in rnerf_nerfmlp_pack, and
 * every rnerf_nerfmlp_forward launch then steps aside for the F16X3 launch queued behind it (the F16F8 packed buffer carries both streams):
 * a per-launch fallback decided on the device, F16X3's bits, no host round trip.  Own packed stream: pack with the precision you run.
 * Round 6: a row with an operand of magnitude >= 448 (e4m3's largest value: fp8(x) of the W_lo cross term would be clamped) is given up like
 * a row out of f16's range and recomputed by the range-safe second pass.  Accuracy depends on the weights: 2e-6 |dRGB| on glorot-initialised
 * networks, 2e-4 with hidden kernels x 1.5 and N(0, 0.3) biases (outside the 1e-4 contract): an opt-in precision, not a default.
 * Range of the f16-based modes.  The forward watches the largest f16 operand it forms per row; a weight of magnitude >= 256 (2^8-scaled
 * streams) or a hidden activation above f16's 65504 makes the first pass give the row up (NaN), never return a plausible wrong colour.
 * EVALUATION (rnerf_nerfmlp_forward with F16X3 / F16X2 / F16F8 / F16, hence rnerf_forward): every launch is followed on the device by a range-safe
 * second pass in BF16X3 (fp32's exponent range; its stream sits behind the others in the packed buffer) that recomputes exactly the rows
 * the first pass gave up on and touches no other row — the reference's fp32 nn.Dense (rnerf/model_utils.py:58-89) is finite there, and so is
 * this; without such a row the pass reads the outputs once (~5 us) and the first pass's bits stand.  The
 * TRAINING forward (rnerf_nerfmlp_forward_train) has no per-row second pass — the saved f16 operands of the backward cannot represent such a
 * row: it stays NaN, reaches the non-finite-gradient count of rnerf_adam_update (scratch[3]), the update is skipped
 * (rnerf_adam_cfg.skip_nonfinite) and the STEP is re-run in BF16X3 + backward BF16, the range-safe training arithmetic. */
enum rnerf_precision {
  RNERF_PREC_F32 = 0,
  RNERF_PREC_F16X3 = 1,
  RNERF_PREC_BF16X3 = 2,
  RNERF_PREC_F16 = 3,
  RNERF_PREC_BF16 = 4,
  RNERF_PREC_F16X2 = 5,
  RNERF_PREC_F16F8 = 6
};

/* Arithmetic of the NerfMLP backward (dgrad + wgrad).  The reference differentiates in fp32 (train.py:164).
 *   BF16  : gradients rounded to bf16 (8-bit significand), saved activations rounded to bf16 in the wgrad: 1 MFMA per product.
 *   F16   : every row's gradient chain is normalised by a power of two m_row ~ max |d raw[row]| (the chain is linear in d raw[row], so
 *           the normalised values fit f16's range whatever the loss scale); f16 operands (11-bit significand, the class of the TF32
 *           tensor-core arithmetic XLA uses for fp32 matmuls on the authors' Ampere GPU): 1 MFMA per product, same HBM traffic as BF16.
 *   F16X3 : as F16 with hi + lo parts of the saved activations and of every gradient (22 bits), 3 MFMAs per product: fp32-grade
 *           (<= 1e-5 of the largest gradient entry against float64); twice the saved bytes.
 *   F16X3_LO8 (round 6): F16X3 with the lo plane of every saved activation and of every gradient STORED as one e4m3 byte per value
 *           (scaled by a fixed power of two, clamped instead of overflowing) and decoded back to f16 inside the wgrad: the same three f16
 *           MFMAs per product with the exact f16 hi parts in both cross terms, three quarters of the bytes of every operand stream of
 *           the step (hi 2 B + lo 1 B per value).  The dgrad chain itself runs on the full hi + lo f16 parts (registers); only what the
 *           wgrad reads is 8-bit.  An 11 + 4 bit significand per operand: measured gradient error vs float64 in DESIGN.md §3.3.
 *           Buffer sizes are those of F16X3 (the lo planes use half of their region). */
enum rnerf_backward {
  RNERF_BWD_BF16 = 0,
  RNERF_BWD_F16 = 1,
  RNERF_BWD_F16X3 = 2,   /* hi + lo f16 planes, 3 MFMAs per product */
  RNERF_BWD_F16X2 = 2,   /* the name of versions <= 1 (two planes); same value */
  RNERF_BWD_F16X3_LO8 = 4   /* hi f16 plane + lo e4m3 plane (3 is an internal dgrad variant) */
};

/* Voxel grid geometry: reference VoxMLP.ndim/nmin/nmax (rnerf/ior_utils.py:124-144).  Doubles, because the
 * reference derives ndelta = (nmax-nmin)/(ndim-1) in Python doubles before it meets float32 data. */
typedef struct rnerf_grid {
  int32_t dims[3];
  double nmin[3];
  double nmax[3];
  int32_t layout;   /* enum rnerf_table_layout: the memory order of the float4 table built for / read with this grid */
} rnerf_grid;

/* Memory order of the (n, grad n) table.  REFERENCE = flat index x*Gy*Gz + y*Gz + z, x slowest (rnerf/ior_utils.py:161,214 — the order the
 * reference indexes with).  BRICKS = 2x2x2 bricks of 128 bytes, bricks x-major: entry (x, y, z) lives at
 *   (((x>>1)*By + (y>>1))*Bz + (z>>1))*8 + (x&1)*4 + (y&1)*2 + (z&1),   B* = ceil(G* / 2)
 * so the 8 corners of a trilinear lookup in an even-aligned cell are ONE 128-byte line and a bent ray meets ~40 % fewer new lines per step
 * (SURVEY 8f N2 "Morton / brick relayout"; measured in profiles/r04/march_time.txt).  Values and indices are the reference's either way:
 * only addresses differ.  Every entry point that takes (table, grid) reads the table in grid->layout; rnerf_grid_build_table writes it so. */
enum rnerf_table_layout {
  RNERF_TABLE_REFERENCE = 0,
  RNERF_TABLE_BRICKS = 1
};
/* floats a table of this grid / layout occupies (REFERENCE: 4 G^3; BRICKS: 32 per brick, odd dimensions padded to whole bricks) */
size_t rnerf_grid_table_floats(const rnerf_grid* g);

/* Sizes of the flat fp32 parameter buffers (flax creation order Dense_0.. ; per layer kernel[in][out]
 * row-major followed by bias[out]).  NerfMLP: rnerf/model_utils.py:30-90, MLP: :93-140. */
#define RNERF_NERFMLP_PARAMS 595844
#define RNERF_BKGDMLP_PARAMS 56963
#define RNERF_SO3MLP_PARAMS 65411   /* so3_mlp: 60->128->128->128(+60)->128->3 (rnerf/ior_utils.py:148-152) */

const char* rnerf_last_error(void);
int rnerf_version(void);
/* Number of compute units of the current device (for host-side sizing); <0 on error. */
int rnerf_device_cus(void);

/* ---- G1: Gaussian prefilter of the IoR grid.  Replaces ior_utils.conv3d_normal (rnerf/ior_utils.py:327-363).
 * src/dst/tmp: float[dims0*dims1*dims2], x slowest.  Separable evaluation of the same normalised kernel. */
int rnerf_grid_prefilter(const float* src, float* dst, float* tmp, const int32_t dims[3], int ksize, double ksigma,
                         void* stream);

/* ---- G2: n + central-difference gradient table.  Replaces VoxMLP.setup/_compute_grad
 * (rnerf/ior_utils.py:139-172).  table: float4[G^3] = (n, dn/dx, dn/dy, dn/dz). */
int rnerf_grid_build_table(const float* grid, float* table, const rnerf_grid* g, void* stream);

/* ---- G3: trilinear lookup with clamp-to-edge.  Replaces VoxMLP._linear3 (rnerf/ior_utils.py:188-223).
 * pts: float[n][3]; out: float[n][4]; idx (nullable): int32[n][6] = clamped x0,x1,y0,y1,z0,z1 (debug tap). */
int rnerf_grid_query(const float* table, const rnerf_grid* g, const float* pts, int64_t n, float* out, int32_t* idx,
                     void* stream);

/* ---- E1/E2/E3: eikonal march.  Replaces PathSampler.__call__ + OneEikonalStep.__call__
 * (rnerf/eikonal_utils.py:29-49,100-124) with stage="radiance*" and math_utils.safe_l2_normalize
 * (rnerf/math_utils.py:6-12).  num_nodes = N_c*P, step = (far-near)/(num_nodes-1) (rnerf/models.py:121-122).
 * origins, viewdirs: float[B][3].  path_pd, path_dr: float4[num_nodes][B] node records (node k = state before
 * step k).  path_ior (nullable): float4[num_nodes][B] = (n, grad n) at node k.  vox (nullable): int32
 * [num_nodes][B][6] clamped voxel indices (debug tap for the bit-exact test). */
int rnerf_march(const float* table, const rnerf_grid* g, const float* origins, const float* viewdirs, int32_t B,
                double near, double far, int32_t num_nodes, float* path_pd, float* path_dr, float* path_ior,
                int32_t* vox, void* stream);

/* ---- N1 weights: pack a flat fp32 NerfMLP parameter buffer into the MFMA operand stream of `precision`. */
size_t rnerf_nerfmlp_packed_bytes(int precision);
int rnerf_nerfmlp_pack(const float* params, int precision, void* packed, void* stream);

/* ---- P1 + N1: positional encoding + NerfMLP over sample rows.  Replaces model_utils.pos_enc
 * (rnerf/model_utils.py:187-214) and NerfMLP.__call__ (:30-90) as called at rnerf/models.py:257,289,305,394,426,441.
 * Rows are sample-major: row = s*B + b.  If node_of_sample != NULL (int32[S], device) the record of row (s,b) is
 * read at node_of_sample[s]*B + b (coarse pass reading the path record through the jitter); otherwise at s*B + b.
 * packed: the buffer written by rnerf_nerfmlp_pack (weight stream + fp32 biases and sigma/rgb heads).
 * out_raw: float4[S*B] = (raw_r, raw_g, raw_b, raw_sigma) before activation.
 * max_workgroups: cap of the persistent grid (0 = one workgroup per CU).  The kernel owns a whole CU (512 VGPRs, 160 KiB LDS); a
 * caller that marches the NEXT ray batch on another stream leaves a few CUs free for it with this argument (no library state). */
int rnerf_nerfmlp_forward(const void* packed, int precision, const float* rows_pd, const float* rows_dr,
                          const int32_t* node_of_sample, int32_t S, int32_t B, float* out_raw, int32_t max_workgroups, void* stream);

/* ---- P1 + N2: background MLP on one direction per ray.  Replaces bkgd_mlp(viewdirs_enc[:, -1:]) +
 * rgb activation (rnerf/models.py:303,336-337) and NerfModel.forward_envmap (:181-191).
 * dirs: float[n][dir_stride] (dir_stride 3 or 4), out_rgb: float[n][3] after sigmoid*(1+2p)-p. */
int rnerf_bkgd_forward(const float* params, const float* dirs, int32_t dir_stride, int64_t n, double rgb_padding,
                       float* out_rgb, void* stream);

/* ---- V1: activations + alpha compositing.  Replaces rgb/sigma activation (rnerf/models.py:334-338) and
 * model_utils.volumetric_rendering (rnerf/model_utils.py:247-309).  Row addressing as rnerf_nerfmlp_forward.
 * bkgd: float[B][3].  Outputs: rgb float[B][3], dist float[B], acc float[B], trans float[B],
 * trans_bkgd float[B][3]; weights (nullable) float[S][B]; alpha (nullable) float[S][B].
 * mask_mode / bbox (host double[6] = min xyz, max xyz): the bd_cut_dist masks of rnerf/models.py:479-524 — 0 none,
 * 1: density_delta *= mask_bbox (1 up to the last sample inside the box), 2: density_delta *= 1 - mask_bbox;
 * 3: use_mask_bbox (rnerf/models.py:261-271,398-408): density_delta *= 1[this sample is inside the box]. */
int rnerf_composite(const float* raw, const float* rows_pd, const float* rows_dr, const int32_t* node_of_sample,
                    int32_t S, int32_t B, const float* bkgd, int white_bkgd, double rgb_padding, double sigma_bias,
                    float* rgb, float* dist, float* acc, float* trans, float* trans_bkgd, float* weights,
                    float* alpha, int mask_mode, const double* bbox, void* stream);

/* ---- S1 + S2: PDF resampling along the bent path.  Replaces sorted_piecewise_constant_pdf and sample_pdf
 * (rnerf/model_utils.py:312-435) as called at rnerf/models.py:371-384.
 * jitter: int32[S] coarse node indices; weights: float[S][B] coarse weights; u: the uniform draws,
 * float[num_fine][B] if u_per_ray else float[num_fine] shared by all rays (randomized=False: linspace(0, 1-eps32, F),
 * rnerf/model_utils.py:355-356).  u must be non-decreasing along the sample axis (true for both reference branches).
 * Outputs (S+num_fine rows per ray, sample-major): rows_pd/rows_dr records, node_idx (nullable) int32[S+F][B]
 * the searchsorted node index (bit-exact contract).  scratch: float[S+num_fine][B] (the merged depths). */
int rnerf_resample(const float* path_pd, const float* path_dr, int32_t num_nodes, int32_t B, const int32_t* jitter,
                   int32_t S, const float* weights, const float* u, int32_t u_per_ray, int32_t num_fine, float* rows_pd,
                   float* rows_dr, int32_t* node_idx, float* scratch, void* stream);

/* ---- S1 (randomized=True): the stratified uniform draws of sorted_piecewise_constant_pdf, rnerf/model_utils.py:345-354:
 * u = min(arange(F)/F + jax.random.uniform(key, [B,F], maxval=1/F-eps), 1-eps) with jax's threefry2x32 bit stream.
 * key: HOST uint32[2]; u: device float[num_fine][B] (the layout rnerf_resample takes with u_per_ray = 1). */
int rnerf_stratified_u(const uint32_t* key, int32_t B, int32_t num_fine, float* u, void* stream);

/* ---- the voxeliser.  Replaces voxelize_mesh.py:54-106 (pysdf point-in-mesh in a Python loop over G^3 voxels):
 * out[i][j][k] = mean over the K^3 sub-samples c + linspace(-1,1,K)^3 * pitch of (inside ? ior_inside : ior_outside), x slowest.
 * verts: device double[V][3]; faces: device int32[F][3]; bin_start int32[num_bins^2 + 1] / bin_tris: CSR lists of the triangles
 * overlapping each cell of a num_bins x num_bins grid over the xy plane (cell index bx * num_bins + by), built by the host;
 * bin_origin_size: HOST double[4] = (x0, y0, cell size x, cell size y).  Inside = odd number of surface crossings along +z
 * (top-left rule on shared edges, fp64).  count: device int32[G^3] scratch; overflow: device int32[1], non-zero if a column met
 * more than 96 crossings (result then invalid). */
int rnerf_voxelize(const double* verts, const int32_t* faces, const int32_t* bin_start, const int32_t* bin_tris, int32_t num_bins,
                   const double* bin_origin_size, const rnerf_grid* g, int32_t num_samples, double ior_inside, double ior_outside,
                   int32_t* count, float* out, int32_t* overflow, void* stream);

/* Robust containment for meshes that are NOT watertight (pysdf casts one parity ray per point in a randomly rotated frame,
 * sdf/src/sdf.cpp:156-168,270-322; a single ray through a hole misclassifies the point): three passes of the same column test along
 * +z, +x, +y, then a per-sample majority.
 *   rnerf_voxelize_samples : as rnerf_voxelize, but writes the inside flag of every sample: inside uint8[GK][GK][GK], GK = G * K, indexed
 *                            [x sample][y sample][z sample] of the coordinates it was GIVEN.  The caller runs it three times with the mesh
 *                            and the grid axes rotated: (x, y, z), (y, z, x), (z, x, y) (samplenerfro_amd/voxelize.py: robust=True).
 *   rnerf_voxelize_majority: in_z / in_x / in_y = the outputs of those three passes; count int32[G^3] = samples with >= 2 of 3 votes. */
int rnerf_voxelize_samples(const double* verts, const int32_t* faces, const int32_t* bin_start, const int32_t* bin_tris, int32_t num_bins,
                           const double* bin_origin_size, const rnerf_grid* g, int32_t num_samples, uint8_t* inside, int32_t* overflow, void* stream);
int rnerf_voxelize_majority(const uint8_t* in_z, const uint8_t* in_x, const uint8_t* in_y, int32_t num_voxels, int32_t num_samples,
                            double ior_inside, double ior_outside, int32_t* count, float* out, void* stream);

/* ---- G4 + P2: VoxMLP.__call__ (rnerf/ior_utils.py:269-312, shipped gin: annealed, use_residual, use_direct_output):
 * (n, grad n) by trilinear lookup and pred_grad = grad n rotated (Rodrigues) by the axis-angle so3_mlp(annealed_pos_enc(x)).
 * so3_params: device float[RNERF_SO3MLP_PARAMS] (flax order); window10: HOST float[10] = cosine_easing_window(0, 9, 10,
 * annealed_alpha * 10) (rnerf/model_utils.py:218-233); pts: device float[n][3]; out4: float4[n] = (n, grad n); pred_grad: float[n][3].
 * condition (nullable, device float[n][3]): rotate this vector instead of the looked-up gradient = VoxMLP.wrapper_grad_mlp
 * (rnerf/ior_utils.py:225-267), the building block of E4 compute_normal_loss_and_smooth (rnerf/eikonal_utils.py:84-98). */
int rnerf_so3_query(const float* table, const rnerf_grid* g, const float* so3_params, const float* window10, const float* pts,
                    const float* condition, int64_t n, float* out4, float* pred_grad, void* stream);

/* ---- E1/E2 with stage "all*": the march with grad = where(|grad n| > 1e-3, pred_grad, grad n) (rnerf/eikonal_utils.py:34-39).
 * Outputs as rnerf_march (path_ior nullable).  so3_packed: device scratch of rnerf_so3_packed_bytes() bytes — the call packs the so3
 * parameters into the f16 hi + lo A-operand stream of the in-march MLP (3 x v_mfma_f32_16x16x32_f16 per tile, fp32 accumulate).
 * ray_order (nullable): int32[B], a permutation of the rays: workgroup i marches rays ray_order[16 i .. 16 i + 15] (a group of 16
 * evaluates the MLP whenever ANY of its rays is inside the boundary shell, so rays with similar shell intervals should share a group).
 * Every record is written at the ray's own index: the order changes no output.
 * Grid size limit of every marching entry point: the table is addressed with 32-bit byte offsets and 24-bit strides —
 * dims[0] * dims[1] * dims[2] * 16 B < 4 GiB in all of them; rnerf_march_all / rnerf_march_all_train / rnerf_march_adjoint (and
 * rnerf_march on a BRICKS table) need the stride of TWO x-planes below 2^24: dims[1] * dims[2] * 32 < 2^24 (reference order;
 * ceil(dims[1]/2) * ceil(dims[2]/2) * 128 < 2^24 for bricks); rnerf_march on a REFERENCE-order table needs only dims[1] * dims[2] * 16 < 2^24.
 * Cubic grids up to 645^3 pass every check (every shipped config: 128^3 .. 512^3); larger grids are rejected with RNERF_ERR_ARG, never
 * mis-addressed. */
size_t rnerf_so3_packed_bytes(void);
int rnerf_march_all(const float* table, const rnerf_grid* g, const float* so3_params, void* so3_packed, const float* window10, const float* origins,
                    const float* viewdirs, int32_t B, double near, double far, int32_t num_nodes, float* path_pd, float* path_dr,
                    float* path_ior, const int32_t* ray_order, void* stream);

/* ---- pinhole ray generation on the device.  Replaces Dataset._generate_rays (rnerf/datasets.py:216-242, Blender
 * model: opencv = 0, fx = fy = focal, cx = W/2, cy = H/2) and the OpenCV variant (:486-518: opencv = 1, fx, fy, cx, cy from cam_mat),
 * for rows [row0, row0+rows) of one view.  camtoworld: HOST float[3][4] (row-major, rotation | translation).
 * origins / directions (nullable) / viewdirs: device float[rows][W][3]. */
int rnerf_generate_rays(const float* camtoworld, int32_t opencv, double fx, double fy, double cx, double cy, double pixel_center,
                        int32_t W, int32_t row0, int32_t rows, float* origins, float* directions, float* viewdirs, void* stream);

/* ---- the gather half of Dataset._next_train (rnerf/datasets.py:151-176: `batch_pixels = self.images[...][ray_indices]`,
 * `batch_rays = namedtuple_map(lambda r: r[...][ray_indices], self.rays)`; :178-197 for the env-map patch) on the device.  The training
 * views stay resident: camtoworlds DEVICE float[n_img][3][4], images DEVICE float[n_img][H][W][channels] (nullable together with
 * `pixels`: rays only, the env-map patch).  ray_indices: DEVICE int64[B], flat `image * H * W + row * W + column` — "all_images" batching
 * indexes the concatenation of all views exactly like that (datasets.py:131-136), "single_image" adds image_index * H * W on the host.  The
 * rays of the drawn pixels are generated on the fly by the arithmetic of rnerf_generate_rays (same camera arguments: bit-identical to
 * indexing the reference's ray arrays); no ray array is ever stored.  bad_count: DEVICE int32, incremented for every index outside
 * [0, n_img * H * W) (such a ray reads pixel 0; the caller zeroes and checks the counter).  origins / directions (nullable) / viewdirs:
 * float[B][3]; pixels: float[B][channels].  The index DRAW stays the caller's (the reference draws with numpy's global generator,
 * datasets.py:154-169: a host sequence this library does not restate). */
int rnerf_sample_batch(const float* camtoworlds, int32_t n_img, int32_t opencv, double fx, double fy, double cx, double cy, double pixel_center,
                       int32_t W, int32_t H, const float* images, int32_t channels, const int64_t* ray_indices, int32_t B, float* origins,
                       float* directions, float* viewdirs, float* pixels, int32_t* bad_count, void* stream);

/* ---- mip-style integrated positional encoding along the curved ray.  Replaces mip.cast_rays(t_vals, ray_pos_c, ray_dir_c,
 * rays.radii, "cone", near) + mip.integrated_pos_enc(samples, min_deg, max_deg) (rnerf/mip.py:26-57,60-91,116-175) as the commented call sites
 * of the reference use them (rnerf/models.py:249-254,386-391): the S samples of a ray (row addressing as rnerf_nerfmlp_forward; depths in
 * pd.w) are the axes of conical frusta between consecutive depths (the last one 1e-3 long); each becomes a diagonal Gaussian whose mean is
 * accumulated along the bent path.  radii: float[B] (Rays.radii).  Outputs (each nullable, sample-major): out_mean4 float4[S][B] =
 * (mean xyz, t_mean), out_cov4 float4[S][B] = (diagonal covariance xyz, t_var), out_enc float[S][B][6 (max_deg - min_deg)] =
 * exp(-var/2) sin(.) of [y, y + pi/2], y = mean * 2^deg degree-major (no identity features). */
int rnerf_integrated_pos_enc(const float* rows_pd, const float* rows_dr, const int32_t* node_of_sample, int32_t S, int32_t B, const float* radii,
                             double near, int32_t min_deg, int32_t max_deg, float* out_mean4, float* out_cov4, float* out_enc, void* stream);

/* ---- T1 (loss): the reductions of train_step.loss_fn (train.py:89-92,105) for stage "radiance*".
 * rgb_c (nullable, N_f == 0), rgb_f: float[B][3]; trans_f: float[B]; trans_bkgd_f, pixels: float[B][3].
 * sums: float[4] (device) = { sum (rgb_f-pix)^2, sum (rgb_c-pix)^2, sum mask*|trans_bkgd_f-pix|, sum mask },
 * mask = trans_f > 0.5.  loss = sums0/(3B) + sums1/(3B) + bg_weight*1[alpha>0]*sums2/(sums3+1) + ... */
int rnerf_loss_reduce(const float* rgb_c, const float* rgb_f, const float* trans_f, const float* trans_bkgd_f,
                      const float* pixels, int32_t B, float* sums, void* stream);

/* ---- T1 (scalar tail of loss_fn).  rnerf_env_smooth_backward: the env-map smoothness term of train.py:127-130 on the patch
 * rgb_env float[ps][ps][3]: d_out float[ps*ps][3] = grad_scale * d mean(0.5 dv^2 + 0.5 dh^2) / d rgb_env; loss_sum (device,
 * rnerf_env_smooth_sum_floats(ps) floats, need not be cleared) receives the un-normalised sum as one partial per workgroup behind 4 reserved
 * floats — rnerf_train_stats (env_loss_sum = this buffer) adds them in index order: no float atomics, the same bits from run to run (round 4;
 * the same holds for rnerf_theta_sumsq, one workgroup with a fixed summation order).  rnerf_train_stats: utils.Stats scalars (train.py:147-162) into stats8 (device float[8], zeroed by the
 * caller): [0] loss, [1] loss_c, [2] loss_bg = bg_scale * sums2 / (sums3 + 1) with bg_scale = bg_weight * 1[annealed_alpha > 0], [3] loss_bg_smooth, [4] weight_l2 = (sum theta^2 + frozen_sq) / n_all, [6] psnr,
 * [7] psnr_c; sums = rnerf_loss_reduce's output. */
size_t rnerf_env_smooth_sum_floats(int32_t ps);
int rnerf_env_smooth_backward(const float* rgb_env, int32_t ps, double grad_scale, float* d_out, float* loss_sum, void* stream);
int rnerf_train_stats(const float* sums, int32_t B, int32_t two_levels, double bg_scale, const float* env_loss_sum, int32_t ps, double env_on,
                      const float* theta, int64_t n_theta, double frozen_sq, int64_t n_all, float* stats8, void* stream);
/* rnerf_train_stats in two parts (the sum of squares of theta does not depend on the step's data and may run anywhere before the scalars):
 * rnerf_theta_sumsq writes sum theta^2 to stats8[5]; rnerf_train_stats with theta == NULL then only forms the scalars. */
int rnerf_theta_sumsq(const float* theta, int64_t n_theta, float* stats8, void* stream);

/* ---- T1 (backward of V1 + activations): d loss / d raw of one level, replacing jax.value_and_grad through
 * volumetric_rendering and the rgb/sigma activations (rnerf/model_utils.py:247-309, rnerf/models.py:334-338; train.py:164).
 * rgb: this level's comp_rgb float[B][3]; mse_scale = 2/(3B); bg_scale = bg_weight*1[annealed_alpha>0] for the level
 * that carries loss_bg (the last one), 0 otherwise (then trans/trans_bkgd/sums may be NULL).
 * d_raw: float4[S][B]; d_bkgd: float[B][3] gradient w.r.t. the activated background colour (accumulated if
 * accumulate_bkgd != 0: both levels composite over the coarse pass's bkgd, rnerf/models.py:468-476).
 * white_bkgd: comp_rgb carried the + (1 - acc) term of rnerf/model_utils.py:307-308.
 * mask_mode / bbox (host double[6] = min xyz, max xyz; NULL with mask_mode 0): 1 = the level's trans / trans_bkgd are the bd_cut_dist pair of
 * rnerf/models.py:479-524 (trans = mask_mode-1 transmittance, trans_bkgd = trans * mask_mode-2 colour over bkgd); 3 = the level was rendered
 * with use_mask_bbox (rnerf/models.py:261-271,398-408: density_delta *= 1[sample inside the box]), the gradient carries the same mask. */
int rnerf_composite_backward(const float* raw, const float* rows_pd, const float* rows_dr, const int32_t* node_of_sample,
                             int32_t S, int32_t B, const float* bkgd, double rgb_padding, double sigma_bias, const float* rgb,
                             const float* pixels, const float* trans, const float* trans_bkgd, const float* sums,
                             double mse_scale, double bg_scale, float* d_raw, float* d_bkgd, int accumulate_bkgd, int white_bkgd,
                             int mask_mode, const double* bbox, void* stream);

/* ---- T1 (backward of P1+N1): gradient of the NerfMLP parameters, replacing jax.value_and_grad through
 * NerfMLP.__call__ (train.py:164; rnerf/model_utils.py:30-90).  `backward` is an enum rnerf_backward, the same in all calls of a step.
 *   rnerf_nerfmlp_forward_train : as rnerf_nerfmlp_forward — precision F16X3 (the fp32-grade default), F16 with backward F16 / BF16 (the
 *       single-pass training arithmetic: one MFMA per product, a labelled bench leg), or BF16X3 with backward BF16 (the RANGE-SAFE training
 *       arithmetic: fp32's exponent range in the forward, the saved operands and the gradients; 16-bit forward products, 8-bit gradients —
 *       what a step whose f16-based run met a row outside f16's range is re-run in); the dgrad / wgrad calls of the step take the same
 *       precision; bf16 / f16x2 / f16f8 are inference precisions — and keeps
 *       the 16-bit operands of every layer in `save` (rnerf_nerfmlp_save_bytes(S*B, backward) bytes; F16X3 backward: hi and lo parts);
 *   rnerf_nerfmlp_pack_bwd : transposed weight stream of the dgrad chain (rnerf_nerfmlp_bwd_packed_bytes() bytes);
 *   rnerf_nerfmlp_dgrad : d_raw float4[rows] (d loss / d raw rgb, sigma) -> dy (rnerf_nerfmlp_dy_bytes(rows, backward) bytes), the
 *       gradients w.r.t. every layer's pre-activation output (F16 modes: normalised per row, + the row scales);
 *   rnerf_nerfmlp_wgrad : (save, dy) -> grads float[RNERF_NERFMLP_PARAMS] in the flat parameter order (every entry is
 *       overwritten); workspace: rnerf_nerfmlp_wgrad_workspace_bytes() bytes. */
size_t rnerf_nerfmlp_save_bytes(int64_t rows, int backward);
size_t rnerf_nerfmlp_dy_bytes(int64_t rows, int backward);
size_t rnerf_nerfmlp_bwd_packed_bytes(void);
size_t rnerf_nerfmlp_wgrad_workspace_bytes(void);
int rnerf_nerfmlp_forward_train(const void* packed, int precision, const float* rows_pd, const float* rows_dr,
                                const int32_t* node_of_sample, int32_t S, int32_t B, float* out_raw, void* save, int backward,
                                int32_t max_workgroups /* as rnerf_nerfmlp_forward: 0 = one workgroup per CU */, void* stream);
int rnerf_nerfmlp_pack_bwd(const float* params, int backward, void* packed_bwd, void* stream);
int rnerf_nerfmlp_dgrad(const void* packed_bwd, const void* packed_fwd, int fwd_precision, int backward, const void* save, const float* d_raw,
                        int64_t rows, void* dy, void* stream);
int rnerf_nerfmlp_wgrad(int fwd_precision, int backward, const void* save, const void* dy, int64_t rows, float* grads, void* workspace,
                        void* stream);

/* ---- T1 (backward of P1+N2): the background MLP, exact fp32 on the matrix cores like its forward.
 * rnerf_bkgd_forward_train = rnerf_bkgd_forward + `save` (rnerf_bkgd_save_bytes(n) bytes).
 * rnerf_bkgd_backward: d_out float[n][3] (d loss / d activated bkgd colour) -> ACCUMULATES into grads
 * float[RNERF_BKGDMLP_PARAMS] (the caller zeroes it once per step: the per-ray bkgd and the env-map patch, train.py:127-130,
 * both add to it); dy: scratch of rnerf_bkgd_dy_bytes(n) bytes.  d_dirs (nullable): float4[n], d loss / d direction through pos_enc(dir)
 * (stage "all*": the direction of the last coarse sample is a function of the so3 parameters). */
size_t rnerf_bkgd_save_bytes(int64_t n);
size_t rnerf_bkgd_dy_bytes(int64_t n);
int rnerf_bkgd_forward_train(const float* params, const float* dirs, int32_t dir_stride, int64_t n, double rgb_padding,
                             float* out_rgb, void* save, void* stream);
int rnerf_bkgd_backward(const float* params, const void* save, const float* d_out, int64_t n, double rgb_padding, void* dy,
                        float* grads, float* d_dirs, void* stream);
/* The two halves of rnerf_bkgd_backward, for a host that overlaps them with other work: _dgrad fills dy (and d_dirs); _wgrad accumulates
 * grads from (save, dy).  coresident != 0 selects a wgrad kernel of at most 80 registers per lane and no LDS, which fits beside the
 * NerfMLP wgrad's waves on every CU (rnerf_train_cfg.coresident_bkgd_wgrad). */
int rnerf_bkgd_backward_dgrad(const float* params, const void* save, const float* d_out, int64_t n, double rgb_padding, void* dy, float* d_dirs,
                              void* stream);
int rnerf_bkgd_backward_wgrad(const void* save, void* dy, int64_t n, float* grads, int coresident, void* stream);

/* ---- training of stage "all*" — jax.value_and_grad (train.py:164) through the N-step eikonal recurrence
 * (rnerf/eikonal_utils.py:29-49,100-124) and through so3_mlp + the Rodrigues rotation (rnerf/ior_utils.py:269-312), with path_sampler
 * trainable (train.py:302-310).  The state of the recurrence is 6 numbers per ray, so its adjoint is a reverse scan with 3x3 Jacobians;
 * everything that involves the so3 MLP runs as parallel batches over the (ray, node) PAIRS at which pred_grad was selected:
 *   rnerf_march_all_train  : rnerf_march_all + the record the backward needs: path_rdn float4[N][B] = (raw direction, n) per node, and
 *                            the compacted pairs (pair_count device int32, zeroed by the call; pair_id int32[cap][2] = (ray, node);
 *                            pair_x / pair_g float4[cap] = position / looked-up gradient; pair_of_node int32[N][B], -1 = none);
 *   rnerf_so3_forward_train: so3_mlp on pts4 float4[n] -> save (rnerf_so3_save_bytes(n): fp32 [enc n x 60][X1..X4 n x 128][raw float4[n], at float offset
 *                            572 n], then the layers' ReLU sign bits uint32[4][n][4], which is all the dgrad chain reads of the activations);
 *   rnerf_so3_backward     : cotangents d_raw4 float4[nb] -> dx4 float4[nb] (nullable, d / d point) and, if grads != NULL (nb == n_save),
 *                            grads float[RNERF_SO3MLP_PARAMS] += ...; row i uses the saved activations of row i % n_save, so the three
 *                            unit cotangents of every point run as one batch of 3 n_save rows (the rows of J = d raw / d x);
 *                            dy (rnerf_so3_dy_bytes(nb)) is scratch for the parameter gradient: written only when grads != NULL;
 *   rnerf_so3_pair_jacobian: per pair A = d grad / d position (P J + (d pred / d g) G, G = d g / d x of the trilinear interpolant) and
 *                            P = d pred / d raw, float[np][12] each (9 used, row-major);
 *   rnerf_march_adjoint    : the reverse scan; a_pos / a_dir float4[S][B] = d loss / d (position, normalised direction) of the coarse
 *                            samples, sample_of_node int32[N] (-1 = the node is no sample) -> v4 float4[np], the cotangent of raw per pair;
 *   rnerf_nerfmlp_input_grad: d loss / d (position, direction) of every row of a NerfMLP level from the dgrad's dy buffer (F16 modes):
 *                            through pos_enc into Dense_0, the skip concat of Dense_5 and the view layer Dense_10. */
int rnerf_march_all_train(const float* table, const rnerf_grid* g, const float* so3_params, void* so3_packed, const float* window10, const float* origins,
                          const float* viewdirs, int32_t B, double near, double far, int32_t num_nodes, float* path_pd, float* path_dr,
                          float* path_rdn, int32_t* pair_count, int32_t pair_cap, int32_t* pair_id, float* pair_x, float* pair_g,
                          int32_t* pair_of_node, const int32_t* ray_order, void* stream);
size_t rnerf_so3_save_bytes(int64_t n);
size_t rnerf_so3_dy_bytes(int64_t nb);
int rnerf_so3_forward_train(const float* so3_params, const float* window10, const float* pts4, int64_t n, void* save, void* stream);
int rnerf_so3_backward(const float* so3_params, const float* window10, const float* pts4, const void* save, int64_t n_save, const float* d_raw4,
                       int64_t nb, void* dy, float* dx4, float* grads, void* stream);
int rnerf_so3_pair_jacobian(const float* table, const rnerf_grid* g, const float* pair_x, const float* pair_g, const float* raw4, const float* J4,
                            int64_t np, float* A12, float* P12, void* stream);
int rnerf_march_adjoint(const float* table, const rnerf_grid* g, const float* path_pd, const float* path_rdn, const int32_t* pair_of_node,
                        const float* A12, const float* P12, const float* a_pos, const float* a_dir, const int32_t* sample_of_node, int32_t B,
                        double near, double far, int32_t num_nodes, float* v4, void* stream);
int rnerf_nerfmlp_input_grad(const float* params, int backward, const void* dy, const float* rows_pd, const float* rows_dr,
                             const int32_t* node_of_sample, int32_t S, int32_t B, float* d_pos4, float* d_dir4, void* stream);

/* ====================================================================================================================================
 * Whole-path entry points ("rnerf_forward / rnerf_backward + a workspace-size query").  They sequence the stage launchers
 * above exactly as NerfModel.__call__ (rnerf/models.py:220-535) and train_step (train.py:58-183) do, on ONE stream, from caller-owned
 * device buffers, with no host round trip in between — so a host in any language drives the path with one call per ray batch, and a
 * whole step can be captured into one launch graph (rnerf_graph_*).  Every random number of the path comes from DEVICE-resident
 * jax.random keys (rnerf_rng_*), which is what makes the step capturable.
 * ==================================================================================================================================== */

/* What NerfModel closes over (attributes rnerf/models.py:42-90, setup :91-137).  Plain data; pointers are device pointers. */
typedef struct rnerf_model {
  const float* table;          /* float4[G^3] written by rnerf_grid_build_table (VoxMLP.setup, rnerf/ior_utils.py:139-172) */
  rnerf_grid grid;
  double near, far;            /* rnerf/models.py:122 */
  int32_t num_coarse;          /* N_c  (num_coarse_samples) */
  int32_t num_fine;            /* N_f  (num_fine_samples; 0 = single level, rnerf/models.py:368) */
  int32_t num_path;            /* P    (num_path_samples): N = N_c * P eikonal nodes */
  int32_t precision;           /* enum rnerf_precision of packed_coarse / packed_fine */
  int32_t white_bkgd;          /* rnerf/model_utils.py:307-308 */
  int32_t bd_cut;              /* 1: the fine level's trans / trans_rgb_bkgd are the bd_cut_dist pair (rnerf/models.py:479-524);
                                * 2: use_mask_bbox (rnerf/models.py:261-271,398-408): both levels keep density only at samples inside bd_cut_bbox
                                *    (the reference's box is the grid's nmin / nmax) */
  double rgb_padding, sigma_bias;   /* rnerf/models.py:78-79 */
  double bd_cut_bbox[6];       /* min xyz, max xyz (rnerf/models.py:485-497) */
  const void* packed_coarse;   /* rnerf_nerfmlp_pack of coarse_mlp */
  const void* packed_fine;     /* rnerf_nerfmlp_pack of fine_mlp (NULL when num_fine == 0) */
  const float* bkgd_params;    /* float[RNERF_BKGDMLP_PARAMS] */
} rnerf_model;

/* Level outputs, struct-of-arrays in one buffer of 9*B floats (the 5-tuple of rnerf/models.py:359-361,532-535):
 * [0,3B) comp_rgb[B][3] | [3B,4B) distance[B] | [4B,5B) acc[B] | [5B,6B) trans[B] | [6B,9B) trans_rgb_bkgd[B][3]. */
#define RNERF_LEVEL_FLOATS 9

/* ---- device-resident jax.random keys (threefry2x32; pinned by tests/test_prng.py against values JAX publishes).
 * rnerf_rng_split3     : train_step's `rng, key_0, key_1 = random.split(rng, 3)` (train.py:74): rng_state uint32[2] is advanced in place,
 *                        keys4 uint32[4] receives (key_0, key_1).
 * rnerf_rng_forward    : the key chain of NerfModel.__call__: `key, rng_0 = split(rng_0)`; jitter = arange(0, N, P) (+ randint(key, [N_c],
 *                        0, P) when use_random_choice — also in eval, rnerf/models.py:240-242); `key, rng_1 = split(rng_1)` -> key_u, the key of
 *                        the stratified draws (rnerf/model_utils.py:345-354).  keys4 = (rng_0, rng_1); jitter int32[N_c]; key_u uint32[2].
 * rnerf_stratified_u_dev: rnerf_stratified_u with the key read from device memory. */
int rnerf_rng_split3(uint32_t* rng_state, uint32_t* keys4, void* stream);
int rnerf_rng_forward(const uint32_t* keys4, int32_t num_coarse, int32_t num_path, int32_t use_random_choice, int32_t* jitter, uint32_t* key_u,
                      void* stream);
int rnerf_stratified_u_dev(const uint32_t* key_dev, int32_t B, int32_t num_fine, float* u, void* stream);

/* ---- NerfModel.__call__ (rnerf/models.py:220-535; callers train.py:247, eval.py:97) for one batch of B rays: march -> background MLP on the
 * last coarse direction -> PE + coarse NerfMLP -> compositing [-> resampling -> PE + fine NerfMLP -> compositing (-> bd_cut pair)].
 * origins, viewdirs: float[B][3].  jitter: device int32[N_c] (rnerf_rng_forward, or any strictly increasing node indices).
 * u_fine (num_fine > 0): device float[N_f] shared by all rays (u_per_ray = 0; randomized=False: linspace(0, 1-eps, N_f)) or float[N_f][B].
 * out_coarse / out_fine: float[RNERF_LEVEL_FLOATS * B] (out_fine unused when num_fine == 0).
 * path_pd / path_dr (nullable, both or none): a path record float4[N][B] marched earlier for these rays (rnerf_march) — the march is
 * skipped; otherwise it runs here into the workspace.  workspace: rnerf_forward_workspace_bytes(m, B) bytes, 256-byte aligned.
 * max_workgroups: as rnerf_nerfmlp_forward. */
size_t rnerf_forward_workspace_bytes(const rnerf_model* m, int32_t B);
int rnerf_forward(const rnerf_model* m, const float* origins, const float* viewdirs, int32_t B, const int32_t* jitter, const float* u_fine,
                  int32_t u_per_ray, const float* path_pd, const float* path_dr, float* out_coarse, float* out_fine, void* workspace,
                  int32_t max_workgroups, void* stream);

/* ---- train_step.loss_fn + jax.value_and_grad (train.py:75-164) for the radiance stages: the training forward of every level, the loss
 * reductions, and the backward kernels down to the flat gradient.  theta: the flat fp32 parameter buffer [coarse_mlp | fine_mlp (if
 * num_fine > 0) | bkgd_mlp] (RNERF_NERFMLP_PARAMS, RNERF_NERFMLP_PARAMS, RNERF_BKGDMLP_PARAMS floats; m->packed_* / m->bkgd_params are
 * ignored: the operand streams are packed from theta inside the call).  grads: float[n_theta + 8]: d loss / d theta of THIS rank's rays
 * (before the mean over ranks, train.py:166; without the weight-decay term, which rnerf_adam_update adds) followed by the 8 stats scalars
 * of rnerf_train_stats.  pixels float[B][3]; env_dirs float[ps*ps][3] (bg_smooth_weight > 0).
 * keys4 (device uint32[4] = key_0, key_1 of rnerf_rng_split3) drives the jitter and the stratified draws; jitter_override / u_override
 * (nullable) inject them instead (parity tests).  path_pd / path_dr as in rnerf_forward. */
typedef struct rnerf_train_cfg {
  int32_t backward;            /* enum rnerf_backward */
  int32_t randomized;          /* FLAGS.randomized (rnerf/utils.py:131): stratified fine draws */
  int32_t use_random_choice;   /* rnerf/models.py:89 */
  int32_t bg_patch_size;       /* ps (0: no env-map smoothness term) */
  double bg_weight, bg_smooth_weight, annealed_alpha;     /* train.py:90-92,127-132 */
  double frozen_sq;            /* sum of squares / count of the variables outside theta (the frozen path_sampler): weight_l2, train.py:147-153 */
  int64_t frozen_count;
  void* aux_stream;            /* nullable: a second stream.  Everything of a step that depends on the parameters only — packing the operand streams of
                                  both directions, zeroing the gradient buffer, sum theta^2 — runs there beside the key kernels, the march and the
                                  background-MLP forward, and is joined inside the call before the first NerfMLP kernel */
  int32_t coresident_bkgd_wgrad;  /* experiment (needs aux_stream): the background MLP's weight gradient as a co-resident kernel beside the NerfMLP wgrad */
  void* grads_stream;          /* nullable: a stream the call orders behind the LAST NerfMLP wgrad (rnerf_fork at that point).  grads[0 .. NerfMLP
                                  segments) are final there: a caller with more than one rank starts their all-reduce (jax.lax.pmean, train.py:166 —
                                  95 % of the bytes) on this stream as soon as the call returns, beside the background-MLP backward and the loss tail
                                  still queued on `stream`, and joins before rnerf_adam_update */
  void* aux2_stream;           /* nullable (needs aux_stream): a third stream for hierarchical models at small batches, where the two levels' backward
                                  passes run side by side: the background MLP's backward (it needs d loss / d background only, final once both
                                  compositing backwards have run) goes there, beside both NerfMLP chains instead of in front of the coarse one */
} rnerf_train_cfg;
/* The march of the NEXT batch (it reads neither the parameters nor anything of this step): when `next` is given, its rays are marched on
 * next->side_stream, forked from `stream` right behind the last NerfMLP wgrad, so that the latency-bound march runs beside the small
 * kernels of the step's tail.  The caller joins (rnerf_join(stream, side_stream)) before it reads next->path_* — inside a captured graph:
 * before rnerf_graph_end. */
typedef struct rnerf_prefetch {
  const float* origins;        /* float[B][3] of the next batch */
  const float* viewdirs;
  float* path_pd;              /* float4[N][B] each: where the next batch's path record goes */
  float* path_dr;
  void* side_stream;
  int32_t beside_wgrad;        /* != 0: fork BEFORE the last wgrad instead of behind it (the march co-resident with the wgrad's waves) */
} rnerf_prefetch;
size_t rnerf_train_workspace_bytes(const rnerf_model* m, const rnerf_train_cfg* c, int32_t B);
int rnerf_train_forward_backward(const rnerf_model* m, const rnerf_train_cfg* c, const float* theta, const float* origins, const float* viewdirs,
                                 const float* pixels, const float* env_dirs, int32_t B, const uint32_t* keys4, const int32_t* jitter_override,
                                 const float* u_override, int32_t u_per_ray, const float* path_pd, const float* path_dr, float* grads,
                                 void* workspace, int32_t max_workgroups, const rnerf_prefetch* next, void* stream);

/* ---- train.py:169-183 + optax.adam behind multi_transform (:312-317) on the flat buffers: weight-decay gradient 2 wd theta / n_all,
 * value clip, global-norm clip (over theta's gradient and the frozen variables' weight-decay gradient, frozen_params nullable),
 * Adam with bias correction and the reference's learning-rate schedule (rnerf/utils.py:490-528) evaluated on the device from the
 * device-resident step counter, which is incremented.  scratch: device float[RNERF_ADAM_SCRATCH_FLOATS] (no initial contents needed); after
 * the call scratch[3] holds the number of non-finite (inf / NaN) gradient entries the update met (0 in a healthy step: a forward row outside
 * f16's range, or a gradient chain beyond the 2^10 of headroom the f16 backward modes normalise every row to, makes that visible), counted
 * BEFORE the value clip (which would otherwise turn a NaN into +-grad_max_val).  skip_nonfinite != 0: an update that met such an entry
 * leaves theta, mu and nu untouched (the step counter still advances) — the reference's fp32 step would have been finite on that batch, so
 * the caller re-runs it in the range-safe arithmetic (precision BF16X3 + backward BF16; samplenerfro_amd.train does — by default two steps
 * later, from a pinned copy of scratch[3], without a per-step synchronisation) instead of writing NaN into every parameter. */
#define RNERF_ADAM_SCRATCH_FLOATS 3076
typedef struct rnerf_adam_cfg {
  double lr_init, lr_final, lr_delay_mult;
  int64_t max_steps, lr_delay_steps;
  double b1, b2, eps;
  double weight_decay_mult, grad_max_val, grad_max_norm;
  int64_t n_all;               /* number of variables weight_l2 averages over (theta + frozen) */
  double lr_override;          /* the learning rate of this update when use_lr_override != 0 (a replaced schedule, tests); 0.0 is honoured */
  int32_t use_lr_override;     /* 0: the reference's schedule (rnerf/utils.py:490-528) from the device-resident step counter */
  int32_t skip_nonfinite;      /* != 0: no update when a gradient entry is inf / NaN (see above) */
} rnerf_adam_cfg;
int rnerf_adam_update(const rnerf_adam_cfg* c, float* theta, float* mu, float* nu, float* grads, int64_t n_theta, const float* frozen_params,
                      int64_t n_frozen, int32_t* step_counter, float* scratch, void* stream);

/* ---- one launch graph per step (hipGraph): begin capture on `stream`, issue any sequence of the calls above on it (and on streams forked
 * from it through rnerf_fork / rnerf_join), end -> an executable graph that replays the whole sequence with one launch. */
int rnerf_graph_begin(void* stream);
int rnerf_graph_end(void* stream, void** graph_exec);
int rnerf_graph_launch(void* graph_exec, void* stream);
int rnerf_graph_destroy(void* graph_exec);
/* side-stream helpers usable inside and outside capture: rnerf_fork makes `side` wait for everything issued on `main` so far;
 * rnerf_join makes `main` wait for everything issued on `side` so far (event pair owned by the library, per (main, side) call). */
int rnerf_fork(void* main_stream, void* side_stream);
int rnerf_join(void* main_stream, void* side_stream);

#ifdef __cplusplus
}
#endif
#endif /* RNERF_H_ */
