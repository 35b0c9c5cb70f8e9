// Whole-path entry points of librnerf.so: NerfModel.__call__ and train_step as ONE call per ray batch (include/rnerf.h, last section).
//
// The stage launchers (march / background MLP / PE + NerfMLP / compositing / resampling and their backward counterparts) stay what
// they are; this file owns what the Python host used to own between them — the order of the stages, the buffers they hand to each
// other (carved out of one caller-owned workspace), the jax.random key chain (on the device, so that nothing of a step is decided on the
// host), the loss tail and the optimiser update — and wraps hipGraph capture so that a whole step replays as one launch.
#include "common.h"

#include <math.h>
#include <mutex>
#include <utility>
#include <vector>

namespace rnerf {
// Compute units of the CURRENT device, cached per device ordinal (a process may drive several devices; one cached count would mis-size the
// side-by-side decisions — and the workspace carve that follows them — on the second).
static int current_device_cus() {
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cache[dev] == 0) {
    int cus = 0;
    cache[dev] = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus : 256;
  }
  return cache[dev];
}

// Ordering events (timing disabled) are pooled per host thread and device: a step orders its streams ~10 times, and creating / destroying an
// event each time costs two driver calls apiece.  An event goes back to the pool as soon as the wait on it is ENQUEUED: hipStreamWaitEvent
// captures the record it waits for at call time, a later re-record of the same event does not move it.  Events recorded inside a stream
// capture belong to the graph (tl_capture_events below) and never return.  The pool is not torn down at thread / process exit on purpose
// (the runtime may be gone by then); it holds at most kPoolCap events per device.
namespace {
constexpr size_t kPoolCap = 32;
struct EventPool { std::vector<hipEvent_t> free_[64]; };
thread_local EventPool tl_pool;
inline hipError_t pool_get(hipEvent_t* e) {
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !tl_pool.free_[dev].empty()) {
    *e = tl_pool.free_[dev].back();
    tl_pool.free_[dev].pop_back();
    return hipSuccess;
  }
  return hipEventCreateWithFlags(e, hipEventDisableTiming);
}
inline void pool_put(hipEvent_t e) {
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && tl_pool.free_[dev].size() < kPoolCap) tl_pool.free_[dev].push_back(e);
  else (void)hipEventDestroy(e);
}
}  // namespace

// csrc/mlp.hip: the operand-stream pack without its own memsets (the step zeroes every stream's range flags in one launch)
int nerfmlp_step_zero(int precision, void* const* packed, int count, int backward, void* const* dy, const int64_t* dy_rows, int dy_count, hipStream_t st);
int nerfmlp_pack_impl(const float* params, int precision, void* packed, bool zero_flags, hipStream_t st, bool with_safe);
int nerfmlp_dgrad_impl(const void* packed_bwd, const void* packed_fwd, int fwd_precision, int backward, const void* save, const float* d_raw, int64_t rows,
                       void* dy, bool zero_ref, bool allow_half, hipStream_t st);


// ---- threefry2x32-20 (Random123), the block cipher behind jax.random (samplenerfro_amd/prng.py; KAT in tests/test_prng.py) ---------
__device__ __forceinline__ unsigned rotl_u32(unsigned x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ void threefry(unsigned k0, unsigned k1, unsigned& x0, unsigned& x1) {
  const unsigned ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  x0 += ks[0];
  x1 += ks[1];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { x0 += x1; x1 = rotl_u32(x1, rot[i & 1][q]); x1 ^= x0; }
    x0 += ks[(i + 1) % 3];
    x1 += ks[(i + 2) % 3] + (unsigned)(i + 1);
  }
}
// element e of jax's _threefry_2x32(key, arange(size)): the counters are split in two halves (odd sizes padded with one zero),
// x0 = first half, x1 = second half, outputs concatenated
__device__ __forceinline__ unsigned random_bits_at(unsigned k0, unsigned k1, int e, int size) {
  const int half = (size + 1) / 2;
  const int j = e < half ? e : e - half;
  unsigned x0 = (unsigned)j, x1 = (j + half < size) ? (unsigned)(j + half) : 0u;
  threefry(k0, k1, x0, x1);
  return e < half ? x0 : x1;
}
// jax.random.split(key, num)[i] = (bits[2i], bits[2i+1]) of random_bits(key, 2 num)
__device__ __forceinline__ void split_at(unsigned k0, unsigned k1, int num, int i, unsigned& o0, unsigned& o1) {
  o0 = random_bits_at(k0, k1, 2 * i, 2 * num);
  o1 = random_bits_at(k0, k1, 2 * i + 1, 2 * num);
}

// rng, key_0, key_1 = random.split(rng, 3)  (train.py:74)
__global__ void rng_split3_kernel(unsigned* __restrict__ state, unsigned* __restrict__ keys4) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned k0 = state[0], k1 = state[1];
  unsigned o[6];
#pragma unroll
  for (int i = 0; i < 3; ++i) split_at(k0, k1, 3, i, o[2 * i], o[2 * i + 1]);
  state[0] = o[0]; state[1] = o[1];
  keys4[0] = o[2]; keys4[1] = o[3]; keys4[2] = o[4]; keys4[3] = o[5];
}

// key, rng_0 = split(rng_0); jitter = arange(0, N, P) + randint(key, [N_c], 0, P); key, rng_1 = split(rng_1) -> key_u
// (rnerf/models.py:232,240-242,371).  randint = jax 0.2.22's two-draw range reduction (prng.randint).
__global__ void __launch_bounds__(256) rng_forward_kernel(const unsigned* __restrict__ keys4, int Nc, int P, int use_random_choice,
                                                          int* __restrict__ jitter, unsigned* __restrict__ key_u) {
  unsigned a0, a1;
  split_at(keys4[0], keys4[1], 2, 0, a0, a1);                        // `key` of the first split
  unsigned h0, h1, l0, l1;
  split_at(a0, a1, 2, 0, h0, h1);                                    // k1, k2 = split(key) inside randint
  split_at(a0, a1, 2, 1, l0, l1);
  const unsigned span = (unsigned)(P > 1 ? P : 1);
  unsigned mult = 65536u % span;
  mult = (unsigned)(((unsigned long long)mult * mult) & 0xFFFFFFFFull) % span;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < Nc; e += gridDim.x * blockDim.x) {
    int j = e * P;
    if (use_random_choice) {
      const unsigned hi = random_bits_at(h0, h1, e, Nc), lo = random_bits_at(l0, l1, e, Nc);
      unsigned off = (unsigned)(((unsigned long long)(hi % span) * mult) & 0xFFFFFFFFull) + (lo % span);
      off %= span;
      j += (int)off;
    }
    jitter[e] = j;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && key_u) {
    unsigned u0, u1;
    split_at(keys4[2], keys4[3], 2, 0, u0, u1);
    key_u[0] = u0; key_u[1] = u1;
  }
}

// the stratified draws with the key in device memory (render.hip: stratified_u_kernel takes it by value)
__global__ void __launch_bounds__(256) stratified_u_dev_kernel(const unsigned* __restrict__ key, int B, int F, float s, float maxval, float one_m_eps,
                                                               float* __restrict__ u) {
  const long long size = (long long)B * F, half = (size + 1) / 2;
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= half) return;
  unsigned x0 = (unsigned)j, x1 = (j + half < size) ? (unsigned)(j + half) : 0u;
  threefry(key[0], key[1], x0, x1);
  auto put = [&](long long e, unsigned bits) {
    const int b = (int)(e / F), f = (int)(e % F);
    const float fl = fsub(__uint_as_float((bits >> 9) | 0x3F800000u), 1.0f);
    const float r = fmaxf(0.0f, fadd(fmul(fl, maxval), 0.0f));
    u[(size_t)f * B + b] = fminf(fadd(fmul((float)f, s), r), one_m_eps);
  };
  put(j, x0);
  if (j + half < size) put(j + half, x1);
}

// u = linspace(0, 1 - eps32, F) as numpy builds it (float64 arange * step, last element = stop), then float32 (model_utils.py:355-356)
__global__ void linspace_u_kernel(int F, double stop, float* __restrict__ u) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F) return;
  const double step = F > 1 ? stop / (double)(F - 1) : 0.0;
  u[i] = (F > 1 && i == F - 1) ? (float)stop : (float)((double)i * step);
}

// rows [0, B): (normalised) direction of the LAST coarse sample (rnerf/models.py:303), read through the device-resident jitter;
// rows [B, B + M): the env-map patch directions (train.py:127-130).  dst: float4 rows.
__global__ void __launch_bounds__(256) bkgd_dirs_kernel(const float4* __restrict__ path_dr, const int* __restrict__ jitter, int Nc, int B,
                                                        const float* __restrict__ env, int M, float4* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) {
    const float4 d = path_dr[(size_t)jitter[Nc - 1] * B + i];
    dst[i] = make_float4(d.x, d.y, d.z, 0.f);
  } else if (i < B + M) {
    const int e = i - B;
    dst[i] = make_float4(env[3 * e], env[3 * e + 1], env[3 * e + 2], 0.f);
  }
}

// (experiment, -DRNERF_TAIL_DELAY_US=<n>, default 0 = not launched)
// One wave that does nothing for `ticks` of the 100 MHz real-time counter.  Issued on the tail stream ahead of the co-resident kernels:
// streams express "after X completed", not "after X has STARTED", so without it the small kernels race the NerfMLP wgrad for the CUs at
// the fork, and wherever their waves land first the wgrad's 8-wave workgroup (2 x 216 registers per SIMD, 148 KiB of LDS) has to wait
// for them to drain.  A few tens of microseconds later the wgrad's workgroups are resident everywhere and the co-resident waves only
// take what it leaves free.
__global__ void spin_kernel(long long ticks) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// trans_rgb_bkgd = trans * rgb_behind (rnerf/models.py:520-524)
__global__ void __launch_bounds__(256) bd_cut_mul_kernel(const float* __restrict__ trans, const float* __restrict__ behind, int B,
                                                         float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * B) out[i] = fmul(trans[i / 3], behind[i]);
}

// ---- optimiser ------------------------------------------------------------------------------------------------------------------
constexpr int ADAM_BLOCKS = 1024;
// g <- clip_value(g + wd2 * theta); partial sums of g^2 per block (deterministic two-level reduction, no float atomics) and, per block, the
// number of non-finite entries BEFORE the value clip (fminf / fmaxf would turn a NaN into +-max_val: a silent, wrong, finite gradient)
__global__ void __launch_bounds__(256) adam_prep_kernel(const float* __restrict__ theta, float* __restrict__ g, long long n, float wd2, float max_val,
                                                        int want_norm, float* __restrict__ partial, float* __restrict__ bad_partial) {
  float s = 0.f;
  int bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = g[i];
    if (wd2 != 0.f) v = v + theta[i] * wd2;
    const bool nf = !(fabsf(v) <= 3.402823466e38f);      // inf or NaN: e.g. a row whose f16 gradient chain overflowed
    bad += nf;
    if (max_val > 0.f && !nf) v = fminf(fmaxf(v, -max_val), max_val);
    g[i] = v;
    s += v * v;
  }
  __shared__ float red[4];
  __shared__ int redb[4];
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); bad += __shfl_down(bad, o); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; redb[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (want_norm) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    bad_partial[blockIdx.x] = (float)((redb[0] + redb[1]) + (redb[2] + redb[3]));
  }
}
// the frozen variables' gradient is their weight-decay term (jax.grad returns it although their optimiser label is "zero")
__global__ void __launch_bounds__(256) adam_frozen_sq_kernel(const float* __restrict__ frozen, long long n, float wd2, float max_val,
                                                             float* __restrict__ partial) {
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = frozen[i] * wd2;
    if (max_val > 0.f) v = fminf(fmaxf(v, -max_val), max_val);
    s += v * v;
  }
  __shared__ float red[4];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
struct AdamSched { double lr_init, lr_final, lr_delay_mult, max_steps, lr_delay_steps, b1, b2, max_norm, lr_override; int use_override, skip_nonfinite; };
// scal[0] = -lr / (1 - b1^t), scal[1] = 1 / (1 - b2^t) (or -1: this update is skipped), scal[2] = norm-clip multiplier, scal[3] = the
// non-finite entries adam_prep_kernel counted (bad_partial; nullptr: adam_apply_kernel counts); the step counter is incremented.
// learning_rate_decay: rnerf/utils.py:490-528 in float64 like the host version.
__global__ void __launch_bounds__(256) adam_scalars_kernel(AdamSched c, int* __restrict__ step, const float* __restrict__ partial, int n_partial,
                                                           const float* __restrict__ bad_partial, float* __restrict__ scal) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n_partial; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  const double sumsq = red[0];
  __syncthreads();
  double nb = 0.0;
  if (bad_partial) for (int i = threadIdx.x; i < ADAM_BLOCKS; i += 256) nb += (double)bad_partial[i];
  red[threadIdx.x] = nb;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x != 0) return;
  const int count = *step;
  const double t = (double)count + 1.0;
  double lr;
  if (c.use_override) {          // an explicit flag, not "lr > 0": a schedule that returns exactly 0.0 (warm-up, count 0) is a zero step
    lr = c.lr_override;
  } else {
    double delay = 1.0;
    if (c.lr_delay_steps > 0) {
      const double x = fmin(fmax((double)count / c.lr_delay_steps, 0.0), 1.0);
      delay = c.lr_delay_mult + (1.0 - c.lr_delay_mult) * sin(0.5 * 3.141592653589793 * x);
    }
    const double start = fmin(fmax((double)count, 0.0), 1.0);
    const double tt = fmin(fmax(fmax((double)count, 0.0) / c.max_steps, 0.0), 1.0);
    lr = start * delay * exp(log(c.lr_init) * (1.0 - tt) + log(c.lr_final) * tt);
  }
  const double bad = red[0];
  scal[0] = (float)(-lr / (1.0 - pow(c.b1, t)));
  scal[1] = (c.skip_nonfinite && bad > 0.0) ? -1.0f : (float)(1.0 / (1.0 - pow(c.b2, t)));      // -1: adam_apply leaves theta, mu, nu alone
  float mult = 1.0f;
  if (c.max_norm > 0) {
    const float norm = sqrtf((float)sumsq);
    mult = fminf((float)c.max_norm / (1e-7f + norm), 1.0f);             // train.py:174-180
  }
  scal[2] = mult;
  scal[3] = (float)bad;   // (no count from adam_prep: 0, and adam_apply counts the non-finite gradient entries of this update here)
  *step = count + 1;      // a skipped update still counts: the schedule and the host's step number go on
}
// optax.scale_by_adam + scale_by_schedule: mu, nu, theta updated in place
__global__ void __launch_bounds__(256) adam_apply_kernel(float* __restrict__ theta, float* __restrict__ mu, float* __restrict__ nu,
                                                         const float* __restrict__ g, long long n, float b1, float b2, float eps,
                                                         float* __restrict__ scal, int count_bad) {
  const float a = scal[0], c2 = scal[1], mult = scal[2];
  if (c2 < 0.f) return;                                // rnerf_adam_cfg.skip_nonfinite and a non-finite gradient entry: no update
  int bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * mult;
    bad += !(fabsf(gi) <= 3.402823466e38f);            // inf or NaN: e.g. a row whose f16 gradient chain overflowed
    const float m = mu[i] * b1 + gi * (1.0f - b1);
    const float v = nu[i] * b2 + (gi * gi) * (1.0f - b2);
    mu[i] = m; nu[i] = v;
    theta[i] = theta[i] + a * (m / (sqrtf(v * c2) + eps));
  }
  if (count_bad && __builtin_amdgcn_ballot_w64(bad != 0) != 0) {    // (never taken on finite gradients)
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(&scal[3], (float)bad);
  }
}

// ---- workspace carving ------------------------------------------------------------------------------------------------------------
struct Carver {
  char* base; size_t off;
  explicit Carver(void* b) : base((char*)b), off(0) {}
  template <typename T> T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = (T*)(base ? base + off : nullptr);
    off += count * sizeof(T);
    return p;
  }
  void* bytes(size_t n) { return (void*)take<char>(n); }
};

struct FwdBuffers {
  float *path_pd, *path_dr, *bk_dirs, *bkgd, *raw_c, *weights, *u, *rows_pd, *rows_dr, *scratch, *raw_f, *tmp_level, *tmp_level2;
};
static size_t carve_forward(const rnerf_model* m, int32_t B, bool own_path, void* ws, FwdBuffers* f) {
  Carver c(ws);
  const size_t Nc = m->num_coarse, Nf = m->num_fine, N = Nc * (size_t)m->num_path, S = Nc + Nf;
  f->path_pd = own_path ? c.take<float>(N * B * 4) : nullptr;
  f->path_dr = own_path ? c.take<float>(N * B * 4) : nullptr;
  f->bk_dirs = c.take<float>((size_t)B * 4);
  f->bkgd = c.take<float>((size_t)B * 3);
  f->raw_c = c.take<float>(Nc * B * 4);
  f->weights = c.take<float>(Nc * B);
  f->u = f->rows_pd = f->rows_dr = f->scratch = f->raw_f = f->tmp_level = f->tmp_level2 = nullptr;
  if (Nf > 0) {
    f->u = c.take<float>(Nf * (size_t)B);
    f->rows_pd = c.take<float>(S * B * 4);
    f->rows_dr = c.take<float>(S * B * 4);
    f->scratch = c.take<float>(S * B);
    f->raw_f = c.take<float>(S * B * 4);
    if (m->bd_cut) { f->tmp_level = c.take<float>((size_t)RNERF_LEVEL_FLOATS * B); f->tmp_level2 = c.take<float>((size_t)RNERF_LEVEL_FLOATS * B); }
  }
  return (c.off + 255) & ~(size_t)255;
}

static int check_model(const rnerf_model* m, int32_t B, const char* who) {
  RNERF_CHECK_ARG(m, "%s: null model", who);
  RNERF_CHECK_ARG(m->table, "%s: model.table is null", who);
  RNERF_CHECK_ARG(B >= 1, "%s: B must be >= 1", who);
  RNERF_CHECK_ARG(m->num_coarse >= 3 && m->num_fine >= 0 && m->num_path >= 1, "%s: need num_coarse >= 3, num_fine >= 0, num_path >= 1", who);
  RNERF_CHECK_ARG((long long)m->num_coarse * m->num_path >= 2, "%s: need at least two eikonal nodes", who);
  return RNERF_OK;
}

// rnerf_model.bd_cut: 0 = none, 1 = the bd_cut_dist pair on the fine level (rnerf/models.py:479-524), 2 = use_mask_bbox (:261-271,398-408):
// density only at samples inside rnerf_model.bd_cut_bbox (the host puts the grid's nmin / nmax there), in BOTH levels' renderings
static inline int level_mask_mode(const rnerf_model* m) { return m->bd_cut == 2 ? 3 : 0; }
static inline const double* level_mask_box(const rnerf_model* m) { return m->bd_cut == 2 ? m->bd_cut_bbox : nullptr; }
struct Level { float *rgb, *dist, *acc, *trans, *tb; };
static Level level_of(float* out, int32_t B) { return Level{out, out + 3 * (size_t)B, out + 4 * (size_t)B, out + 5 * (size_t)B, out + 6 * (size_t)B}; }

// the bd_cut_dist pair of rnerf/models.py:479-524 on the fine level: trans <- mask-mode-1 transmittance, tb <- trans * (mask-mode-2 colour)
static int bd_cut_pair(const rnerf_model* m, const float* raw_f, const float* rows_pd, const float* rows_dr, int32_t S, int32_t B, const float* bkgd,
                       Level out, float* tmp1, float* tmp2, void* stream) {
  Level a = level_of(tmp1, B), b = level_of(tmp2, B);
  int rc = rnerf_composite(raw_f, rows_pd, rows_dr, nullptr, S, B, nullptr, m->white_bkgd, m->rgb_padding, m->sigma_bias, a.rgb, a.dist, a.acc, out.trans,
                           a.tb, nullptr, nullptr, 1, m->bd_cut_bbox, stream);
  if (rc) return rc;
  rc = rnerf_composite(raw_f, rows_pd, rows_dr, nullptr, S, B, bkgd, m->white_bkgd, m->rgb_padding, m->sigma_bias, b.rgb, b.dist, b.acc, b.trans, b.tb, nullptr,
                       nullptr, 2, m->bd_cut_bbox, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(bd_cut_mul_kernel, dim3((3 * B + 255) / 256), dim3(256), 0, (hipStream_t)stream, out.trans, b.rgb, B, out.tb);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

static int make_u(const rnerf_model* m, int32_t B, int randomized, const uint32_t* key_u_dev, float* u, int32_t* per_ray, void* stream) {
  const double eps = 1.1920928955078125e-07;
  if (randomized) {
    *per_ray = 1;
    return rnerf_stratified_u_dev(key_u_dev, B, m->num_fine, u, stream);
  }
  *per_ray = 0;
  hipLaunchKernelGGL(linspace_u_kernel, dim3((m->num_fine + 255) / 256), dim3(256), 0, (hipStream_t)stream, m->num_fine, 1.0 - eps, u);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

}  // namespace rnerf

using namespace rnerf;

#define RNERF_TRY(expr)        \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != RNERF_OK) return rc_; \
  } while (0)

extern "C" int rnerf_rng_split3(uint32_t* rng_state, uint32_t* keys4, void* stream) {
  RNERF_CHECK_ARG(rng_state && keys4, "rnerf_rng_split3: null pointer");
  hipLaunchKernelGGL(rng_split3_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rng_state, keys4);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

extern "C" int rnerf_rng_forward(const uint32_t* keys4, int32_t num_coarse, int32_t num_path, int32_t use_random_choice, int32_t* jitter, uint32_t* key_u,
                                 void* stream) {
  RNERF_CHECK_ARG(keys4 && jitter, "rnerf_rng_forward: null pointer");
  RNERF_CHECK_ARG(num_coarse >= 1 && num_path >= 1, "rnerf_rng_forward: need num_coarse >= 1 and num_path >= 1");
  hipLaunchKernelGGL(rng_forward_kernel, dim3((num_coarse + 255) / 256), dim3(256), 0, (hipStream_t)stream, keys4, num_coarse, num_path, use_random_choice,
                     jitter, key_u);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

extern "C" int rnerf_stratified_u_dev(const uint32_t* key_dev, int32_t B, int32_t num_fine, float* u, void* stream) {
  RNERF_CHECK_ARG(key_dev && u, "rnerf_stratified_u_dev: null pointer");
  RNERF_CHECK_ARG(B >= 1 && num_fine >= 1, "rnerf_stratified_u_dev: need B >= 1 and num_fine >= 1");
  const double eps = 1.1920928955078125e-07, s = 1.0 / num_fine;
  const long long half = ((long long)B * num_fine + 1) / 2;
  hipLaunchKernelGGL(stratified_u_dev_kernel, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, (hipStream_t)stream, key_dev, B, num_fine, (float)s,
                     (float)(s - eps), (float)(1.0 - eps), u);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

extern "C" size_t rnerf_forward_workspace_bytes(const rnerf_model* m, int32_t B) {
  if (!m || B < 1) return 0;
  FwdBuffers f;
  return carve_forward(m, B, true, nullptr, &f);
}

extern "C" int rnerf_forward(const rnerf_model* m, const float* origins, const float* viewdirs, int32_t B, const int32_t* jitter, const float* u_fine,
                             int32_t u_per_ray, const float* path_pd, const float* path_dr, float* out_coarse, float* out_fine, void* workspace,
                             int32_t max_workgroups, void* stream) {
  RNERF_TRY(check_model(m, B, "rnerf_forward"));
  RNERF_CHECK_ARG(jitter && out_coarse && workspace, "rnerf_forward: null pointer");
  RNERF_CHECK_ARG((path_pd == nullptr) == (path_dr == nullptr), "rnerf_forward: give both path_pd and path_dr or neither");
  RNERF_CHECK_ARG(path_pd || (origins && viewdirs), "rnerf_forward: origins / viewdirs are null and no marched path was given");
  RNERF_CHECK_ARG(m->packed_coarse && m->bkgd_params, "rnerf_forward: model.packed_coarse / bkgd_params are null");
  RNERF_CHECK_ARG(m->num_fine == 0 || (m->packed_fine && out_fine && u_fine), "rnerf_forward: num_fine > 0 needs packed_fine, out_fine and u_fine");
  RNERF_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "rnerf_forward: workspace must be 256-byte aligned");
  const int32_t Nc = m->num_coarse, Nf = m->num_fine, N = Nc * m->num_path, S = Nc + Nf;
  FwdBuffers f;
  carve_forward(m, B, path_pd == nullptr, workspace, &f);
  if (!path_pd) {
    RNERF_TRY(rnerf_march(m->table, &m->grid, origins, viewdirs, B, m->near, m->far, N, f.path_pd, f.path_dr, nullptr, nullptr, stream));
    path_pd = f.path_pd; path_dr = f.path_dr;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bkgd_dirs_kernel, dim3((B + 255) / 256), dim3(256), 0, st, (const float4*)path_dr, jitter, Nc, B, (const float*)nullptr, 0, (float4*)f.bk_dirs);
  RNERF_CHECK_LAUNCH();
  RNERF_TRY(rnerf_bkgd_forward(m->bkgd_params, f.bk_dirs, 4, B, m->rgb_padding, f.bkgd, stream));
  RNERF_TRY(rnerf_nerfmlp_forward(m->packed_coarse, m->precision, path_pd, path_dr, jitter, Nc, B, f.raw_c, max_workgroups, stream));
  Level c = level_of(out_coarse, B);
  RNERF_TRY(rnerf_composite(f.raw_c, path_pd, path_dr, jitter, Nc, B, f.bkgd, m->white_bkgd, m->rgb_padding, m->sigma_bias, c.rgb, c.dist, c.acc, c.trans, c.tb,
                            Nf > 0 ? f.weights : nullptr, nullptr, level_mask_mode(m), level_mask_box(m), stream));
  if (Nf > 0) {
    RNERF_TRY(rnerf_resample(path_pd, path_dr, N, B, jitter, Nc, f.weights, u_fine, u_per_ray, Nf, f.rows_pd, f.rows_dr, nullptr, f.scratch, stream));
    RNERF_TRY(rnerf_nerfmlp_forward(m->packed_fine, m->precision, f.rows_pd, f.rows_dr, nullptr, S, B, f.raw_f, max_workgroups, stream));
    Level o = level_of(out_fine, B);
    RNERF_TRY(rnerf_composite(f.raw_f, f.rows_pd, f.rows_dr, nullptr, S, B, f.bkgd, m->white_bkgd, m->rgb_padding, m->sigma_bias, o.rgb, o.dist, o.acc, o.trans,
                              o.tb, nullptr, nullptr, level_mask_mode(m), level_mask_box(m), stream));
    if (m->bd_cut == 1) RNERF_TRY(bd_cut_pair(m, f.raw_f, f.rows_pd, f.rows_dr, S, B, f.bkgd, o, f.tmp_level, f.tmp_level2, stream));
  }
  return RNERF_OK;
}

// ---- training --------------------------------------------------------------------------------------------------------------------
namespace rnerf {
#ifndef RNERF_TAIL_DELAY_US
#define RNERF_TAIL_DELAY_US 0
#endif
constexpr long long kTailDelayTicks = 100LL * RNERF_TAIL_DELAY_US;      // s_memrealtime counts at 100 MHz
static inline bool co_requested(const rnerf_train_cfg* c) { return c->aux_stream && c->coresident_bkgd_wgrad; }
// hierarchical models with an aux stream: the two levels' backward passes side by side when together they are at most two rounds of row
// tiles (see rnerf_train_forward_backward); also decides whether the workspace carries the coarse level's own dY / d raw / wgrad scratch
static bool levels_side_by_side(const rnerf_model* m, const rnerf_train_cfg* c, int32_t B) {
  if (m->num_fine <= 0 || !c->aux_stream || co_requested(c)) return false;
  const int cus = current_device_cus();
  const long long tiles_both = ((long long)m->num_coarse * B + 255) / 256 + ((long long)(m->num_coarse + m->num_fine) * B + 255) / 256;
  return tiles_both <= 2LL * cus;
}
struct TrainBuffers {
  FwdBuffers f;
  int32_t* jitter; uint32_t* key_u;
  void *packed_c, *packed_f, *packed_bwd, *packed_bwd_f, *save_c, *save_f, *save_bk, *dy, *dy_bk, *wgrad_ws;
  void *dy_c, *wgrad_ws_c;           // hierarchical models with an aux stream: the coarse level's backward runs beside the fine level's
  float *out_all, *level_c, *level_f, *sums, *d_all, *d_raw, *env_sum, *d_raw_c;
};
static size_t carve_train(const rnerf_model* m, const rnerf_train_cfg* c, int32_t B, bool own_path, void* ws, TrainBuffers* t) {
  const size_t Nc = m->num_coarse, Nf = m->num_fine, S = Nc + Nf;
  const size_t M = c->bg_smooth_weight > 0 ? (size_t)c->bg_patch_size * c->bg_patch_size : 0;
  // the forward part: as carve_forward, with the background rows widened to B + M
  Carver k(ws);
  const size_t N = Nc * (size_t)m->num_path;
  FwdBuffers& f = t->f;
  f.path_pd = own_path ? k.take<float>(N * B * 4) : nullptr;
  f.path_dr = own_path ? k.take<float>(N * B * 4) : nullptr;
  f.bk_dirs = k.take<float>((B + M) * 4);
  t->out_all = k.take<float>((B + M) * 3);           // rows [0,B): bkgd of the rays, [B,B+M): rgb of the env patch
  f.bkgd = t->out_all;
  f.raw_c = k.take<float>(Nc * B * 4);
  f.weights = k.take<float>(Nc * B);
  f.u = f.rows_pd = f.rows_dr = f.scratch = f.raw_f = f.tmp_level = f.tmp_level2 = nullptr;
  if (Nf > 0) {
    f.u = k.take<float>(Nf * (size_t)B);
    f.rows_pd = k.take<float>(S * B * 4);
    f.rows_dr = k.take<float>(S * B * 4);
    f.scratch = k.take<float>(S * B);
    f.raw_f = k.take<float>(S * B * 4);
    if (m->bd_cut) { f.tmp_level = k.take<float>((size_t)RNERF_LEVEL_FLOATS * B); f.tmp_level2 = k.take<float>((size_t)RNERF_LEVEL_FLOATS * B); }
  }
  t->jitter = k.take<int32_t>(Nc);
  t->key_u = k.take<uint32_t>(4);
  t->packed_c = k.bytes(rnerf_nerfmlp_packed_bytes(RNERF_PREC_F16X3));
  t->packed_f = Nf > 0 ? k.bytes(rnerf_nerfmlp_packed_bytes(RNERF_PREC_F16X3)) : nullptr;
  t->packed_bwd = k.bytes(rnerf_nerfmlp_bwd_packed_bytes());
  t->packed_bwd_f = Nf > 0 ? k.bytes(rnerf_nerfmlp_bwd_packed_bytes()) : nullptr;
  t->save_c = k.bytes(rnerf_nerfmlp_save_bytes((int64_t)Nc * B, c->backward));
  t->save_f = Nf > 0 ? k.bytes(rnerf_nerfmlp_save_bytes((int64_t)S * B, c->backward)) : nullptr;
  t->save_bk = k.bytes(rnerf_bkgd_save_bytes((int64_t)(B + M)));
  t->dy = k.bytes(rnerf_nerfmlp_dy_bytes((int64_t)S * B, c->backward));
  t->dy_bk = k.bytes(rnerf_bkgd_dy_bytes((int64_t)(B + M)));
  t->wgrad_ws = k.bytes(rnerf_nerfmlp_wgrad_workspace_bytes());
  t->level_c = k.take<float>((size_t)RNERF_LEVEL_FLOATS * B);
  t->level_f = Nf > 0 ? k.take<float>((size_t)RNERF_LEVEL_FLOATS * B) : nullptr;
  t->sums = k.take<float>(4);
  t->env_sum = k.take<float>(rnerf_env_smooth_sum_floats(c->bg_smooth_weight > 0 ? c->bg_patch_size : 0));
  t->d_all = k.take<float>((B + M) * 3);
  t->d_raw = k.take<float>(S * B * 4);
  t->dy_c = t->wgrad_ws_c = nullptr; t->d_raw_c = nullptr;
  if (levels_side_by_side(m, c, B)) {      // own buffers for the concurrent coarse backward (see rnerf_train_forward_backward)
    t->dy_c = k.bytes(rnerf_nerfmlp_dy_bytes((int64_t)Nc * B, c->backward));
    t->wgrad_ws_c = k.bytes(rnerf_nerfmlp_wgrad_workspace_bytes());
    t->d_raw_c = k.take<float>(Nc * B * 4);
  }
  return (k.off + 255) & ~(size_t)255;
}
}  // namespace rnerf

extern "C" size_t rnerf_train_workspace_bytes(const rnerf_model* m, const rnerf_train_cfg* c, int32_t B) {
  if (!m || !c || B < 1) return 0;
  TrainBuffers t;
  return carve_train(m, c, B, true, nullptr, &t);
}

static int mark_point(hipStream_t first, hipEvent_t* out);
static int wait_point(hipStream_t then, hipEvent_t e);

extern "C" int rnerf_train_forward_backward(const rnerf_model* m, const rnerf_train_cfg* c, const float* theta, const float* origins, const float* viewdirs,
                                            const float* pixels, const float* env_dirs, int32_t B, const uint32_t* keys4, const int32_t* jitter_override,
                                            const float* u_override, int32_t u_per_ray, const float* path_pd, const float* path_dr, float* grads,
                                            void* workspace, int32_t max_workgroups, const rnerf_prefetch* next, void* stream) {
  RNERF_TRY(check_model(m, B, "rnerf_train_forward_backward"));
  RNERF_CHECK_ARG(c && theta && pixels && grads && workspace, "rnerf_train_forward_backward: null pointer");
  RNERF_CHECK_ARG(keys4 || (jitter_override && (m->num_fine == 0 || u_override || !c->randomized)),
                  "rnerf_train_forward_backward: give keys4, or the jitter (and the stratified draws) explicitly");
  RNERF_CHECK_ARG((path_pd == nullptr) == (path_dr == nullptr), "rnerf_train_forward_backward: give both path_pd and path_dr or neither");
  RNERF_CHECK_ARG(path_pd || (origins && viewdirs), "rnerf_train_forward_backward: origins / viewdirs are null and no marched path was given");
  RNERF_CHECK_ARG(m->precision == RNERF_PREC_F16X3 || (m->precision == RNERF_PREC_F16 && c->backward != RNERF_BWD_F16X3 && c->backward != RNERF_BWD_F16X3_LO8) ||
                      (m->precision == RNERF_PREC_BF16X3 && c->backward == RNERF_BWD_BF16),
                  "rnerf_train_forward_backward: training is built on the f16x3 forward (f16: with backward f16 / bf16 only; bf16x3 + backward bf16: the range-safe step)");
  RNERF_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "rnerf_train_forward_backward: workspace must be 256-byte aligned");
  const bool smooth = c->bg_smooth_weight > 0;
  RNERF_CHECK_ARG(!smooth || (env_dirs && c->bg_patch_size >= 2), "rnerf_train_forward_backward: bg_smooth_weight > 0 needs env_dirs and bg_patch_size >= 2");
  const int32_t Nc = m->num_coarse, Nf = m->num_fine, N = Nc * m->num_path, S = Nc + Nf;
  const int32_t ps = smooth ? c->bg_patch_size : 0, M = ps * ps;
  const int bwd = c->backward, prec = m->precision;
  hipStream_t st = (hipStream_t)stream;
  TrainBuffers t;
  carve_train(m, c, B, path_pd == nullptr, workspace, &t);
  FwdBuffers& f = t.f;
  // flat parameter layout [coarse | fine | bkgd] (train.TrainState)
  const float* th_c = theta;
  const float* th_f = Nf > 0 ? theta + RNERF_NERFMLP_PARAMS : nullptr;
  const float* th_b = theta + (size_t)RNERF_NERFMLP_PARAMS * (Nf > 0 ? 2 : 1);
  const int64_t n_theta = (int64_t)RNERF_NERFMLP_PARAMS * (Nf > 0 ? 2 : 1) + RNERF_BKGDMLP_PARAMS;
  float* g_c = grads;
  float* g_f = Nf > 0 ? grads + RNERF_NERFMLP_PARAMS : nullptr;
  float* g_b = grads + (size_t)RNERF_NERFMLP_PARAMS * (Nf > 0 ? 2 : 1);
  float* stats8 = grads + n_theta;

  // ---- everything that depends on the parameters only — the operand streams of both directions, the zeroing of the gradient buffer,
  //      sum theta^2 for weight_l2 — goes to cfg->aux_stream, beside the key kernels, the march and the background-MLP forward below
  //      (~70 us of small launches off the critical path).  Three joins, each for what its consumer needs and no more: the coarse forward
  //      waits for the COARSE operand stream only (one flag-zeroing launch + one pack kernel: ~20 us, shorter than the main stream's
  //      keys + background forward beside it); the fine level's stream, the zeroing of the gradients, the backward's streams and the sum
  //      follow on the aux stream while the coarse forward runs and are joined before the fine forward / the backward.
  //      (Round 4: before, the coarse forward waited for five launches — three 16-byte memsets among them — and started at +88 us.)
  void* aux = c->aux_stream;
  bool pre_zeroed = false;      // the head of the step has cleared the accumulator words of its kernels (aux stream only)
  struct Mark { hipEvent_t e = nullptr; ~Mark() { if (e) pool_put(e); } } fine_packed;      // (an early error return must not leak it)
  if (aux) {
    RNERF_TRY(rnerf_fork(stream, aux));
    // one launch zeroes what the step's kernels need cleared: the streams' range flags, the row-scale reference of every dgrad that is the
    // FIRST writer of its dY buffer (without an aux-stream coarse level the coarse dgrad reuses the fine level's buffer: it clears its own).
    // (The env-map smoothness sum and sum theta^2 need no clearing: fixed-order reductions that assign their result.)
    void* streams[2] = {t.packed_c, t.packed_f};
    void* dys[2] = {t.dy, t.dy_c};
    const int64_t dy_rows[2] = {(int64_t)(Nf > 0 ? S : Nc) * B, (int64_t)Nc * B};
    RNERF_TRY(nerfmlp_step_zero(prec, streams, Nf > 0 ? 2 : 1, bwd, dys, dy_rows, t.dy_c ? 2 : 1, (hipStream_t)aux));
    pre_zeroed = true;
    RNERF_TRY(nerfmlp_pack_impl(th_c, prec, t.packed_c, false, (hipStream_t)aux, false));
  }
  // ---- forward (models.forward with ctx) ----
  const int32_t* jitter = jitter_override;
  if (keys4) {
    RNERF_TRY(rnerf_rng_forward(keys4, Nc, m->num_path, c->use_random_choice, t.jitter, t.key_u, stream));
    if (!jitter_override) jitter = t.jitter;
  }
  if (!path_pd) {
    RNERF_TRY(rnerf_march(m->table, &m->grid, origins, viewdirs, B, m->near, m->far, N, f.path_pd, f.path_dr, nullptr, nullptr, stream));
    path_pd = f.path_pd; path_dr = f.path_dr;
  }
  hipLaunchKernelGGL(bkgd_dirs_kernel, dim3((B + M + 255) / 256), dim3(256), 0, st, (const float4*)path_dr, jitter, Nc, B, env_dirs, M, (float4*)f.bk_dirs);
  RNERF_CHECK_LAUNCH();
  RNERF_TRY(rnerf_bkgd_forward_train(th_b, f.bk_dirs, 4, (int64_t)B + M, m->rgb_padding, t.out_all, t.save_bk, stream));
  const float* bkgd = t.out_all;
  const float* rgb_env = t.out_all + (size_t)3 * B;
  if (aux) {
    RNERF_TRY(rnerf_join(stream, aux));                    // the aux stream's tail HERE is the coarse pack: everything below follows it
    if (Nf > 0) {
      RNERF_TRY(nerfmlp_pack_impl(th_f, prec, t.packed_f, false, (hipStream_t)aux, false));
      RNERF_TRY(mark_point((hipStream_t)aux, &fine_packed.e));
    }
    RNERF_CHECK_HIP(hipMemsetAsync(grads, 0, (size_t)(n_theta + 8) * sizeof(float), (hipStream_t)aux));
    RNERF_TRY(rnerf_nerfmlp_pack_bwd(th_c, bwd, t.packed_bwd, aux));
    if (Nf > 0) RNERF_TRY(rnerf_nerfmlp_pack_bwd(th_f, bwd, t.packed_bwd_f, aux));
    RNERF_TRY(rnerf_theta_sumsq(theta, n_theta, stats8, aux));
  } else {
    RNERF_TRY(rnerf_nerfmlp_pack(th_c, prec, t.packed_c, stream));
  }
  RNERF_TRY(rnerf_nerfmlp_forward_train(t.packed_c, prec, path_pd, path_dr, jitter, Nc, B, f.raw_c, t.save_c, bwd, max_workgroups, stream));
  Level lc = level_of(t.level_c, B);
  RNERF_TRY(rnerf_composite(f.raw_c, path_pd, path_dr, jitter, Nc, B, bkgd, m->white_bkgd, m->rgb_padding, m->sigma_bias, lc.rgb, lc.dist, lc.acc, lc.trans, lc.tb,
                            Nf > 0 ? f.weights : nullptr, nullptr, level_mask_mode(m), level_mask_box(m), stream));
  Level lf = lc;
  if (Nf > 0) {
    const float* u = u_override;
    int32_t per_ray = u_per_ray;
    if (!u) { RNERF_TRY(make_u(m, B, c->randomized, t.key_u, f.u, &per_ray, stream)); u = f.u; }
    RNERF_TRY(rnerf_resample(path_pd, path_dr, N, B, jitter, Nc, f.weights, u, per_ray, Nf, f.rows_pd, f.rows_dr, nullptr, f.scratch, stream));
    if (!aux) RNERF_TRY(rnerf_nerfmlp_pack(th_f, prec, t.packed_f, stream));
    else { hipEvent_t e = fine_packed.e; fine_packed.e = nullptr; RNERF_TRY(wait_point(st, e)); }      // the fine level's operand stream (packed beside the coarse forward) and no more
    RNERF_TRY(rnerf_nerfmlp_forward_train(t.packed_f, prec, f.rows_pd, f.rows_dr, nullptr, S, B, f.raw_f, t.save_f, bwd, max_workgroups, stream));
    lf = level_of(t.level_f, B);
    RNERF_TRY(rnerf_composite(f.raw_f, f.rows_pd, f.rows_dr, nullptr, S, B, bkgd, m->white_bkgd, m->rgb_padding, m->sigma_bias, lf.rgb, lf.dist, lf.acc, lf.trans,
                              lf.tb, nullptr, nullptr, level_mask_mode(m), level_mask_box(m), stream));
    if (m->bd_cut == 1) RNERF_TRY(bd_cut_pair(m, f.raw_f, f.rows_pd, f.rows_dr, S, B, bkgd, lf, f.tmp_level, f.tmp_level2, stream));
  }
  // ---- loss reductions (train.py:89-92,105) ----
  RNERF_TRY(rnerf_loss_reduce(Nf > 0 ? lc.rgb : nullptr, lf.rgb, lf.trans, lf.tb, pixels, B, t.sums, stream));
  const double bg_on = (c->bg_weight > 0 && c->annealed_alpha > 0) ? 1.0 : 0.0;
  const double mse_scale = 2.0 / (3.0 * B);
  if (!aux) RNERF_CHECK_HIP(hipMemsetAsync(grads, 0, (size_t)(n_theta + 8) * sizeof(float), st));
  // The next batch's march on the side stream, forked from `stream` at the call.  With beside_wgrad it is issued right before the LARGEST
  // wgrad of the step (the fine level's when there is one): the wgrad keeps 64 registers free on every SIMD (RNERF_WGRAD_VGPRS), so the
  // march's waves are co-resident with it and the whole march hides behind that HBM-paced kernel.
  auto march_next = [&]() -> int {
    RNERF_CHECK_ARG(next->origins && next->viewdirs && next->path_pd && next->path_dr && next->side_stream, "rnerf_train_forward_backward: incomplete rnerf_prefetch");
    RNERF_TRY(rnerf_fork(stream, next->side_stream));
    return rnerf_march(m->table, &m->grid, next->origins, next->viewdirs, B, m->near, m->far, N, next->path_pd, next->path_dr, nullptr, nullptr,
                       next->side_stream);
  };
  // ---- backward, last level first ----
  if (aux) RNERF_TRY(rnerf_join(stream, aux));
  float* d_first = t.d_all;                      // rows [0,B): d loss / d bkgd of the rays; rows [B,B+M): the env-map patch
  // Hierarchical models with an aux stream (round 4): the two levels' backward passes are independent once both compositing backwards have
  // run (the coarse one accumulates d bkgd on top of the fine one's), so the coarse level's dgrad + wgrad go to the aux stream, with buffers
  // of their own, BESIDE the fine level's.  Each NerfMLP kernel owns whole CUs, so the two never share one — but whenever one level's
  // persistent grid leaves CUs idle (all of a small batch: 512 rays x (64 + 128) samples are 128 + 384 row tiles on 256 CUs) the other
  // level's workgroups take them: 2 rounds of tiles instead of 1 + 2 at that size.  Only for batches whose two levels together fit two
  // rounds: beyond that the kernels' static tile striding is delayed on the CUs the other level took first and the step gets SLOWER
  // (profiles/r04/levels_side_by_side.txt: 512 rays 2.28 -> 2.18 ms, 128 rays 1.50 -> 1.33; 1024 rays 3.36 -> 3.45, 4096 rays 11.9 -> 12.4).
  const bool split_levels = levels_side_by_side(m, c, B);
  static const bool bk_early_env = RNERF_ENV("RNERF_BKGD_BWD_EARLY") ? atoi(RNERF_ENV("RNERF_BKGD_BWD_EARLY")) != 0 : true;
  const bool bk_early = split_levels && bk_early_env && !(aux && c->coresident_bkgd_wgrad);
  void* bk2 = nullptr;      // the third stream, when the background backward went there
  float* d_raw_c = split_levels ? t.d_raw_c : t.d_raw;
  void* dy_c = split_levels ? t.dy_c : t.dy;
  if (Nf > 0) {
    RNERF_TRY(rnerf_composite_backward(f.raw_f, f.rows_pd, f.rows_dr, nullptr, S, B, bkgd, m->rgb_padding, m->sigma_bias, lf.rgb, pixels, lf.trans, lf.tb, t.sums,
                                       mse_scale, c->bg_weight * bg_on, t.d_raw, d_first, 0, m->white_bkgd, m->bd_cut == 1 ? 1 : level_mask_mode(m), m->bd_cut ? m->bd_cut_bbox : nullptr, stream));
    if (split_levels) {
      RNERF_TRY(rnerf_composite_backward(f.raw_c, path_pd, path_dr, jitter, Nc, B, bkgd, m->rgb_padding, m->sigma_bias, lc.rgb, pixels, nullptr, nullptr, nullptr,
                                         mse_scale, 0.0, d_raw_c, d_first, 1, m->white_bkgd, level_mask_mode(m), level_mask_box(m), stream));
      RNERF_TRY(rnerf_fork(stream, aux));
      if (bk_early) {      // d loss / d background is final (both compositing backwards have run): its whole backward goes beside the NerfMLP
        // chains — on the third stream when there is one (nothing waits behind it), else in front of the coarse level's on the aux stream.
        // (Measured, tools/r04/env_ab.sh RNERF_NO_AUX2_STREAM: 256 rays 1.48 -> 1.39 ms, 512 rays the same, 128 rays 1.09 -> 1.11: with a
        // quarter of the chip's tiles the coarse chain is not what the step waits for, and the third stream only adds its fork / join.)
        const long long tiles_both = ((long long)Nc * B + 255) / 256 + ((long long)S * B + 255) / 256;
        bk2 = (c->aux2_stream && tiles_both > current_device_cus() / 2) ? c->aux2_stream : nullptr;
        void* bk = bk2 ? bk2 : aux;
        if (bk2) RNERF_TRY(rnerf_fork(stream, bk));
        const double env_on_ = c->annealed_alpha > 0 ? 1.0 : 0.0;
        if (smooth) RNERF_TRY(rnerf_env_smooth_backward(rgb_env, ps, c->bg_smooth_weight * env_on_, t.d_all + (size_t)3 * B, t.env_sum, bk));
        RNERF_TRY(rnerf_bkgd_backward(th_b, t.save_bk, t.d_all, (int64_t)B + M, m->rgb_padding, t.dy_bk, g_b, nullptr, bk));
      }
      RNERF_TRY(nerfmlp_dgrad_impl(t.packed_bwd, t.packed_c, prec, bwd, t.save_c, d_raw_c, (int64_t)Nc * B, dy_c, !pre_zeroed, false, (hipStream_t)aux));
      RNERF_TRY(rnerf_nerfmlp_wgrad(prec, bwd, t.save_c, dy_c, (int64_t)Nc * B, g_c, t.wgrad_ws_c, aux));
    }
    if (!aux) RNERF_TRY(rnerf_nerfmlp_pack_bwd(th_f, bwd, t.packed_bwd_f, stream));
    RNERF_TRY(nerfmlp_dgrad_impl(t.packed_bwd_f, t.packed_f, prec, bwd, t.save_f, t.d_raw, (int64_t)S * B, t.dy, !pre_zeroed, !split_levels, st));
    if (next && next->beside_wgrad) RNERF_TRY(march_next());
    RNERF_TRY(rnerf_nerfmlp_wgrad(prec, bwd, t.save_f, t.dy, (int64_t)S * B, g_f, t.wgrad_ws, stream));
    if (split_levels) {
      RNERF_TRY(rnerf_join(stream, aux));
      if (bk2) RNERF_TRY(rnerf_join(stream, bk2));
    }
    else
      RNERF_TRY(rnerf_composite_backward(f.raw_c, path_pd, path_dr, jitter, Nc, B, bkgd, m->rgb_padding, m->sigma_bias, lc.rgb, pixels, nullptr, nullptr, nullptr,
                                         mse_scale, 0.0, t.d_raw, d_first, 1, m->white_bkgd, level_mask_mode(m), level_mask_box(m), stream));
  } else {
    RNERF_TRY(rnerf_composite_backward(f.raw_c, path_pd, path_dr, jitter, Nc, B, bkgd, m->rgb_padding, m->sigma_bias, lf.rgb, pixels, lf.trans, lf.tb, t.sums,
                                       mse_scale, c->bg_weight * bg_on, t.d_raw, d_first, 0, m->white_bkgd, level_mask_mode(m), level_mask_box(m), stream));
  }
  // (experiment, cfg->coresident_bkgd_wgrad: the background MLP's weight gradient as a co-resident kernel on the aux stream beside the
  //  NerfMLP wgrad — its dgrad chain then runs here, ahead of the NerfMLP dgrad; measured neutral at bench size)
  const double env_on = c->annealed_alpha > 0 ? 1.0 : 0.0;
  const bool co = aux && c->coresident_bkgd_wgrad;
  if (co) {
    if (smooth) RNERF_TRY(rnerf_env_smooth_backward(rgb_env, ps, c->bg_smooth_weight * env_on, t.d_all + (size_t)3 * B, t.env_sum, stream));
    RNERF_TRY(rnerf_bkgd_backward_dgrad(th_b, t.save_bk, t.d_all, (int64_t)B + M, m->rgb_padding, t.dy_bk, nullptr, stream));
  }
  if (!aux) RNERF_TRY(rnerf_nerfmlp_pack_bwd(th_c, bwd, t.packed_bwd, stream));
  if (!split_levels)      // (a hierarchical model's coarse dgrad is the SECOND writer of t.dy here: it clears its own reference)
    RNERF_TRY(nerfmlp_dgrad_impl(t.packed_bwd, t.packed_c, prec, bwd, t.save_c, t.d_raw, (int64_t)Nc * B, t.dy, !(pre_zeroed && Nf == 0), true, st));
  if (co) {      // forked HERE, not earlier: the NerfMLP dgrad owns every CU whole, the wgrad below leaves room for these waves
    RNERF_TRY(rnerf_fork(stream, aux));
    if (kTailDelayTicks > 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)aux, (long long)kTailDelayTicks);
    RNERF_TRY(rnerf_bkgd_backward_wgrad(t.save_bk, t.dy_bk, (int64_t)B + M, g_b, 1, aux));
  }
  if (next && next->beside_wgrad && Nf == 0) RNERF_TRY(march_next());
  if (!split_levels) RNERF_TRY(rnerf_nerfmlp_wgrad(prec, bwd, t.save_c, t.dy, (int64_t)Nc * B, g_c, t.wgrad_ws, stream));
  if (c->grads_stream) RNERF_TRY(rnerf_fork(stream, c->grads_stream));      // the NerfMLP gradient segments are final: the caller's collective may start
  if (next && !next->beside_wgrad) RNERF_TRY(march_next());     // beside the tail below (background-MLP backward, loss glue) and the update
  if (co) {
    RNERF_TRY(rnerf_join(stream, aux));
  } else if (!bk_early) {
    if (smooth) RNERF_TRY(rnerf_env_smooth_backward(rgb_env, ps, c->bg_smooth_weight * env_on, t.d_all + (size_t)3 * B, t.env_sum, stream));
    RNERF_TRY(rnerf_bkgd_backward(th_b, t.save_bk, t.d_all, (int64_t)B + M, m->rgb_padding, t.dy_bk, g_b, nullptr, stream));
  }
  RNERF_TRY(rnerf_train_stats(t.sums, B, Nf > 0, c->bg_weight * bg_on, smooth ? t.env_sum : nullptr, ps, env_on, aux ? nullptr : theta, n_theta, c->frozen_sq,
                              n_theta + c->frozen_count, stats8, stream));
  return RNERF_OK;
}

extern "C" int rnerf_adam_update(const rnerf_adam_cfg* c, float* theta, float* mu, float* nu, float* grads, int64_t n_theta, const float* frozen_params,
                                 int64_t n_frozen, int32_t* step_counter, float* scratch, void* stream) {
  RNERF_CHECK_ARG(c && theta && mu && nu && grads && step_counter && scratch, "rnerf_adam_update: null pointer");
  RNERF_CHECK_ARG(n_theta >= 1 && c->n_all >= n_theta, "rnerf_adam_update: need n_theta >= 1 and n_all >= n_theta");
  hipStream_t st = (hipStream_t)stream;
  const double wd2 = c->weight_decay_mult > 0 ? 2.0 * c->weight_decay_mult / (double)c->n_all : 0.0;
  const int want_norm = c->grad_max_norm > 0;
  float* scal = scratch;                      // [0..3]
  float* partial = scratch + 4;               // [ADAM_BLOCKS] + [ADAM_BLOCKS] (frozen part)
  float* bad_partial = nullptr;               // [ADAM_BLOCKS] non-finite counts of adam_prep_kernel, when it runs
  int n_partial = 0;
  if (wd2 != 0.0 || c->grad_max_val > 0 || want_norm || c->skip_nonfinite) {
    bad_partial = scratch + 4 + 2 * ADAM_BLOCKS;
    hipLaunchKernelGGL(adam_prep_kernel, dim3(ADAM_BLOCKS), dim3(256), 0, st, (const float*)theta, grads, (long long)n_theta, (float)wd2, (float)c->grad_max_val,
                       want_norm, partial, bad_partial);
    if (want_norm) {
      n_partial = ADAM_BLOCKS;
      if (frozen_params && n_frozen > 0 && wd2 != 0.0) {
        hipLaunchKernelGGL(adam_frozen_sq_kernel, dim3(ADAM_BLOCKS), dim3(256), 0, st, frozen_params, (long long)n_frozen, (float)wd2, (float)c->grad_max_val,
                           partial + ADAM_BLOCKS);
        n_partial = 2 * ADAM_BLOCKS;
      }
    }
  }
  AdamSched s{c->lr_init, c->lr_final, c->lr_delay_mult, (double)c->max_steps, (double)c->lr_delay_steps, c->b1, c->b2, c->grad_max_norm, c->lr_override, c->use_lr_override != 0,
              c->skip_nonfinite != 0};
  hipLaunchKernelGGL(adam_scalars_kernel, dim3(1), dim3(256), 0, st, s, step_counter, (const float*)partial, n_partial, (const float*)bad_partial, scal);
  hipLaunchKernelGGL(adam_apply_kernel, dim3(ADAM_BLOCKS), dim3(256), 0, st, theta, mu, nu, (const float*)grads, (long long)n_theta, (float)c->b1, (float)c->b2,
                     (float)c->eps, scal, bad_partial == nullptr);
  RNERF_CHECK_LAUNCH();
  return RNERF_OK;
}

// ---- hipGraph --------------------------------------------------------------------------------------------------------------------
namespace {
// Events recorded inside a stream capture become graph edges.  They are collected per capturing thread (capture mode is thread local) and
// handed to the executable graph at rnerf_graph_end; rnerf_graph_destroy releases them with it.  Outside capture the event is released as
// soon as the wait is enqueued (HIP defers the destruction until the event has completed).
std::mutex g_ev_mu;
thread_local std::vector<hipEvent_t> tl_capture_events;
std::vector<std::pair<void*, std::vector<hipEvent_t>>> g_graph_events;
void drop_events(std::vector<hipEvent_t>& v) {
  for (hipEvent_t e : v) (void)hipEventDestroy(e);
  v.clear();
}
}  // namespace

extern "C" int rnerf_graph_begin(void* stream) {
  drop_events(tl_capture_events);      // leftovers of a capture that failed before rnerf_graph_end
  RNERF_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return RNERF_OK;
}

extern "C" int rnerf_graph_end(void* stream, void** graph_exec) {
  RNERF_CHECK_ARG(graph_exec, "rnerf_graph_end: null pointer");
  hipGraph_t g = nullptr;
  hipError_t err = hipStreamEndCapture((hipStream_t)stream, &g);
  if (err != hipSuccess) { drop_events(tl_capture_events); set_error("hipStreamEndCapture failed: %s", hipGetErrorString(err)); return RNERF_ERR_HIP; }
  hipGraphExec_t e = nullptr;
  err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (err != hipSuccess) { drop_events(tl_capture_events); set_error("hipGraphInstantiate failed: %s", hipGetErrorString(err)); return RNERF_ERR_HIP; }
  {
    std::lock_guard<std::mutex> l(g_ev_mu);
    g_graph_events.emplace_back((void*)e, std::move(tl_capture_events));
  }
  tl_capture_events.clear();
  *graph_exec = (void*)e;
  return RNERF_OK;
}

extern "C" int rnerf_graph_launch(void* graph_exec, void* stream) {
  RNERF_CHECK_ARG(graph_exec, "rnerf_graph_launch: null graph");
  RNERF_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return RNERF_OK;
}

extern "C" int rnerf_graph_destroy(void* graph_exec) {
  if (!graph_exec) return RNERF_OK;
  std::vector<hipEvent_t> ev;
  {
    std::lock_guard<std::mutex> l(g_ev_mu);
    for (size_t i = 0; i < g_graph_events.size(); ++i)
      if (g_graph_events[i].first == graph_exec) { ev = std::move(g_graph_events[i].second); g_graph_events.erase(g_graph_events.begin() + i); break; }
  }
  RNERF_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  drop_events(ev);
  return RNERF_OK;
}

static int order_after(hipStream_t first, hipStream_t then) {
  hipEvent_t e = nullptr;
  RNERF_CHECK_HIP(pool_get(&e));
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  hipError_t err = hipStreamIsCapturing(first, &cs);
  if (err == hipSuccess) err = hipEventRecord(e, first);
  if (err == hipSuccess) err = hipStreamWaitEvent(then, e, 0);
  if (err != hipSuccess) {      // no leak on the error path (an event in an unknown state is destroyed, not pooled)
    (void)hipEventDestroy(e);
    set_error("stream ordering failed: %s", hipGetErrorString(err));
    return RNERF_ERR_HIP;
  }
  if (cs == hipStreamCaptureStatusNone) pool_put(e);
  else tl_capture_events.push_back(e);
  return RNERF_OK;
}

// order_after in two halves, for a dependency whose consumer is enqueued long after its producer: mark_point(first) now, wait_point(then)
// where the consumer goes.  A mark that is never waited for must be passed to wait_point all the same (it owns the event).
static int mark_point(hipStream_t first, hipEvent_t* out) {
  hipEvent_t e = nullptr;
  RNERF_CHECK_HIP(pool_get(&e));
  const hipError_t err = hipEventRecord(e, first);
  if (err != hipSuccess) { (void)hipEventDestroy(e); set_error("stream ordering failed: %s", hipGetErrorString(err)); return RNERF_ERR_HIP; }
  *out = e;
  return RNERF_OK;
}
static int wait_point(hipStream_t then, hipEvent_t e) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  hipError_t err = hipStreamIsCapturing(then, &cs);
  if (err == hipSuccess) err = hipStreamWaitEvent(then, e, 0);
  if (err != hipSuccess) { (void)hipEventDestroy(e); set_error("stream ordering failed: %s", hipGetErrorString(err)); return RNERF_ERR_HIP; }
  if (cs == hipStreamCaptureStatusNone) pool_put(e);
  else tl_capture_events.push_back(e);
  return RNERF_OK;
}

extern "C" int rnerf_fork(void* main_stream, void* side_stream) { return order_after((hipStream_t)main_stream, (hipStream_t)side_stream); }
extern "C" int rnerf_join(void* main_stream, void* side_stream) { return order_after((hipStream_t)side_stream, (hipStream_t)main_stream); }
