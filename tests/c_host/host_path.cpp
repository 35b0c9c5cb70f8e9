// A host WITHOUT Python or torch: the SampleNeRFRO hot path through the C ABI of librnerf.so only (include/rnerf.h).
//
//   host_path <in.bin> <out.bin>
//
// in.bin (written by tests/test_gpu_c_host.py): int32 header {G, Nc, Nf, P, B, ps}, double {near, far, extent}, uint32 keys[4] (rng_0, rng_1),
// uint32 rng_train[2], then float arrays: grid[G^3], coarse[595844], fine[595844], bkgd[56963], origins[B*3], viewdirs[B*3],
// pixels[B*3], env_dirs[ps*ps*3].
// out.bin: out_coarse[9B], out_fine[9B] of rnerf_forward (eval, randomized = false), then grads[n_theta + 8] of one
// rnerf_train_forward_backward (randomized stratified draws from the device key chain), then theta[n_theta] after rnerf_adam_update.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/rnerf.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define RN_OK(x) do { int r_ = (x); if (r_ != RNERF_OK) { fprintf(stderr, "%s failed (%d): %s\n", #x, r_, rnerf_last_error()); return 3; } } while (0)

template <typename T> static T* dev_from(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc((void**)&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}
template <typename T> static T* dev_alloc(size_t n) { T* d = nullptr; if (hipMalloc((void**)&d, n * sizeof(T)) != hipSuccess) return nullptr; hipMemset(d, 0, n * sizeof(T)); return d; }

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: host_path in.bin out.bin\n"); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int32_t hdr[6]; double dd[3]; uint32_t keys[4], rng_train[2];
  if (fread(hdr, 4, 6, f) != 6 || fread(dd, 8, 3, f) != 3 || fread(keys, 4, 4, f) != 4 || fread(rng_train, 4, 2, f) != 2) return 1;
  const int G = hdr[0], Nc = hdr[1], Nf = hdr[2], P = hdr[3], B = hdr[4], ps = hdr[5];
  auto rd = [&](size_t n) { std::vector<float> v(n); if (fread(v.data(), 4, n, f) != n) { fprintf(stderr, "short read\n"); exit(1); } return v; };
  const size_t n_net = RNERF_NERFMLP_PARAMS, n_bk = RNERF_BKGDMLP_PARAMS, n_theta = 2 * n_net + n_bk;
  std::vector<float> grid = rd((size_t)G * G * G), coarse = rd(n_net), fine = rd(n_net), bkgd = rd(n_bk), o = rd((size_t)B * 3), v = rd((size_t)B * 3),
                     pix = rd((size_t)B * 3), env = rd((size_t)ps * ps * 3);
  fclose(f);

  hipStream_t s;
  HIP_OK(hipStreamCreate(&s));
  // ---- the model: table + packed weights (rnerf_model is plain data)
  rnerf_model m = {};
  m.grid.dims[0] = m.grid.dims[1] = m.grid.dims[2] = G;
  for (int i = 0; i < 3; ++i) { m.grid.nmin[i] = -dd[2]; m.grid.nmax[i] = dd[2]; }
  float* d_grid = dev_from(grid);
  float* d_table = dev_alloc<float>((size_t)G * G * G * 4);
  RN_OK(rnerf_grid_build_table(d_grid, d_table, &m.grid, s));
  m.table = d_table; m.near = dd[0]; m.far = dd[1]; m.num_coarse = Nc; m.num_fine = Nf; m.num_path = P;
  m.precision = RNERF_PREC_F16X3; m.rgb_padding = 0.001; m.sigma_bias = -1.0;
  // theta = [coarse | fine | bkgd]: the flat buffer the training entry points take
  std::vector<float> theta_h(n_theta);
  std::copy(coarse.begin(), coarse.end(), theta_h.begin());
  std::copy(fine.begin(), fine.end(), theta_h.begin() + n_net);
  std::copy(bkgd.begin(), bkgd.end(), theta_h.begin() + 2 * n_net);
  float* theta = dev_from(theta_h);
  void* packed_c = dev_alloc<char>(rnerf_nerfmlp_packed_bytes(m.precision));
  void* packed_f = dev_alloc<char>(rnerf_nerfmlp_packed_bytes(m.precision));
  RN_OK(rnerf_nerfmlp_pack(theta, m.precision, packed_c, s));
  RN_OK(rnerf_nerfmlp_pack(theta + n_net, m.precision, packed_f, s));
  m.packed_coarse = packed_c; m.packed_fine = packed_f; m.bkgd_params = theta + 2 * n_net;

  float *d_o = dev_from(o), *d_v = dev_from(v), *d_pix = dev_from(pix), *d_env = dev_from(env);
  std::vector<uint32_t> k4(keys, keys + 4), kt(rng_train, rng_train + 2);
  uint32_t* d_keys = dev_from(k4);
  uint32_t* d_rng = dev_from(kt);
  int32_t* d_jit = dev_alloc<int32_t>(Nc);
  uint32_t* d_key_u = dev_alloc<uint32_t>(2);
  // ---- eval forward: jitter from the device key chain, u = linspace(0, 1 - eps32, N_f) (randomized = false, model_utils.py:355-356)
  RN_OK(rnerf_rng_forward(d_keys, Nc, P, 1, d_jit, d_key_u, s));
  std::vector<float> u_h(Nf);
  const double stop = 1.0 - 1.1920928955078125e-07;
  for (int i = 0; i < Nf; ++i) u_h[i] = (Nf > 1 && i == Nf - 1) ? (float)stop : (float)((double)i * (stop / (Nf - 1)));
  float* d_u = dev_from(u_h);
  void* ws = dev_alloc<char>(rnerf_forward_workspace_bytes(&m, B));
  float *out_c = dev_alloc<float>((size_t)RNERF_LEVEL_FLOATS * B), *out_f = dev_alloc<float>((size_t)RNERF_LEVEL_FLOATS * B);
  RN_OK(rnerf_forward(&m, d_o, d_v, B, d_jit, d_u, 0, nullptr, nullptr, out_c, out_f, ws, 0, s));

  // ---- one optimisation step: key split, forward + backward, Adam (train.py:58-183)
  rnerf_train_cfg c = {};
  c.backward = RNERF_BWD_F16X2; c.randomized = 1; c.use_random_choice = 1; c.bg_patch_size = ps;
  c.bg_weight = 0.025; c.bg_smooth_weight = 1.0; c.annealed_alpha = 0.5;
  void* tws = dev_alloc<char>(rnerf_train_workspace_bytes(&m, &c, B));
  float* grads = dev_alloc<float>(n_theta + 8);
  uint32_t* d_keys4 = dev_alloc<uint32_t>(4);
  RN_OK(rnerf_rng_split3(d_rng, d_keys4, s));
  RN_OK(rnerf_train_forward_backward(&m, &c, theta, d_o, d_v, d_pix, d_env, B, d_keys4, nullptr, nullptr, 0, nullptr, nullptr, grads, tws, 0, nullptr, s));
  std::vector<float> g_h(n_theta + 8);
  HIP_OK(hipStreamSynchronize(s));
  HIP_OK(hipMemcpy(g_h.data(), grads, g_h.size() * 4, hipMemcpyDeviceToHost));
  rnerf_adam_cfg a = {};
  a.lr_init = 5e-4; a.lr_final = 5e-6; a.lr_delay_mult = 0.01; a.max_steps = 1000000; a.lr_delay_steps = 2500;
  a.b1 = 0.9; a.b2 = 0.999; a.eps = 1e-8; a.n_all = (int64_t)n_theta;
  a.skip_nonfinite = 1;      // (what samplenerfro_amd.train sets: an update with an inf / NaN gradient entry writes nothing)
  float *mu = dev_alloc<float>(n_theta), *nu = dev_alloc<float>(n_theta), *scratch = dev_alloc<float>(RNERF_ADAM_SCRATCH_FLOATS);
  int32_t* step = dev_alloc<int32_t>(1);
  RN_OK(rnerf_adam_update(&a, theta, mu, nu, grads, (int64_t)n_theta, nullptr, 0, step, scratch, s));
  HIP_OK(hipStreamSynchronize(s));

  std::vector<float> oc((size_t)RNERF_LEVEL_FLOATS * B), of((size_t)RNERF_LEVEL_FLOATS * B), th(n_theta);
  HIP_OK(hipMemcpy(oc.data(), out_c, oc.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(of.data(), out_f, of.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(th.data(), theta, th.size() * 4, hipMemcpyDeviceToHost));
  FILE* g = fopen(argv[2], "wb");
  if (!g) { perror(argv[2]); return 1; }
  fwrite(oc.data(), 4, oc.size(), g); fwrite(of.data(), 4, of.size(), g); fwrite(g_h.data(), 4, g_h.size(), g); fwrite(th.data(), 4, th.size(), g);
  fclose(g);
  printf("host_path ok: B=%d Nc=%d Nf=%d P=%d G=%d, loss %.6f\n", B, Nc, Nf, P, G, g_h[n_theta]);
  return 0;
}
