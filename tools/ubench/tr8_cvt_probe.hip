// Semantics probes for the 8-bit lo planes of the NerfMLP backward (gfx950):
//   1. ds_read_b64_tr_b8: which (source lane, byte) every destination (lane, byte) receives;
//   2. v_cvt_scalef32_pk_fp8_f16: direction of the scale operand, op_sel half, saturation without MODE.FP16_OVFL, f16 subnormal inputs;
//   3. v_cvt_scalef32_pk_f16_fp8: direction of the scale, which half op_sel picks.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/tr8_cvt_probe.hip -o /tmp/tr8_cvt_probe && /tmp/tr8_cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int int2v __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__global__ void probe_tr8(uint8_t* out_lane, uint8_t* out_elem, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 64];
  const int lane = threadIdx.x;
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = lane; i < 64 * 64; i += 64) lds[i] = 0xff;
    __syncthreads();
    uint8_t* mine = lds + lane * stride_bytes;
    for (int e = 0; e < 8; ++e) mine[e] = pass == 0 ? (uint8_t)lane : (uint8_t)e;
    __syncthreads();
    const int2v v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((int2v __attribute__((address_space(3)))*)mine);
    uint8_t* o = pass == 0 ? out_lane : out_elem;
    for (int e = 0; e < 8; ++e) o[8 * lane + e] = (uint8_t)((unsigned)v[e >> 2] >> (8 * (e & 3)));
    __syncthreads();
  }
}

__global__ void probe_cvt(const float* in, int n, float scale, int set_ovfl, uint32_t* enc_lo, uint32_t* enc_hi, float* dec) {
  if (set_ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  const int i = threadIdx.x;
  if (i >= n) return;
  const half2v src = {(_Float16)in[i], (_Float16)(-in[i])};
  short2v o = {0x1111, 0x2222};
  o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o, src, scale, false);
  enc_lo[i] = ((uint32_t)(uint16_t)o[1] << 16) | (uint16_t)o[0];
  short2v p = {0x1111, 0x2222};
  p = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p, src, scale, true);
  enc_hi[i] = ((uint32_t)(uint16_t)p[1] << 16) | (uint16_t)p[0];
  const uint32_t word = enc_lo[i];
  const half2v d0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(word, scale, false);
  const half2v d1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(enc_hi[i], scale, true);
  dec[4 * i] = (float)d0[0]; dec[4 * i + 1] = (float)d0[1]; dec[4 * i + 2] = (float)d1[0]; dec[4 * i + 3] = (float)d1[1];
}

int main() {
  uint8_t *dl, *de; hipMalloc(&dl, 512); hipMalloc(&de, 512);
  for (int stride : {8, 16, 32}) {
    probe_tr8<<<1, 64>>>(dl, de, stride);
    uint8_t hl[512], he[512]; hipMemcpy(hl, dl, 512, hipMemcpyDeviceToHost); hipMemcpy(he, de, 512, hipMemcpyDeviceToHost);
    printf("tr8 stride %d bytes: dest lane: (src lane.byte) x8\n", stride);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
      printf("%2d:", l);
      for (int e = 0; e < 8; ++e) {
        printf(" %2d.%d", hl[8 * l + e], he[8 * l + e]);
        const int g = l & ~15, i = l & 15;
        if (hl[8 * l + e] != g + 2 * e + (i >> 3) || he[8 * l + e] != (i & 7)) ok = 0;
      }
      printf((l & 1) == 1 ? "\n" : "   ");
    }
    printf("hypothesis out[i][j] = in[2j + (i>>3)][i&7] per 16-lane group: %s\n", ok ? "HOLDS" : "FAILS");
  }
  const float vals[] = {1.0f, 1.5f, 0.5f, 448.f, 449.f, 480.f, 1000.f, 60000.f, 0.015625f, 0.001953125f, 0.0009f, 3.0e-5f, 6.0e-6f, 5.9e-8f, 0.3f, 3.3f};
  const int n = sizeof(vals) / sizeof(float);
  float *din, *ddec; uint32_t *dlo, *dhi; hipMalloc(&din, sizeof(vals)); hipMalloc(&dlo, 4 * n); hipMalloc(&dhi, 4 * n); hipMalloc(&ddec, 16 * n);
  hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl)
    for (float scale : {1.0f, 4.0f, 0.25f, 1.0f / 4096.f}) {
      probe_cvt<<<1, 64>>>(din, n, scale, ovfl, dlo, dhi, ddec);
      uint32_t hlo[32], hhi[32]; float hd[128];
      hipMemcpy(hlo, dlo, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(hhi, dhi, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(hd, ddec, 16 * n, hipMemcpyDeviceToHost);
      printf("FP16_OVFL %d, scale operand %g: src = (x, -x) as f16 -> fp8 pair in the low (op_sel 0) / high (op_sel 1) half of 0x22221111; decoded with the same scale\n", ovfl, scale);
      for (int i = 0; i < n; ++i)
        printf("  %-12g lo-half 0x%08x  hi-half 0x%08x   decode(lo-half word, opsel 0) = (%g, %g)  decode(hi-half word, opsel 1) = (%g, %g)\n", vals[i], hlo[i], hhi[i],
               hd[4 * i], hd[4 * i + 1], hd[4 * i + 2], hd[4 * i + 3]);
    }
  return 0;
}
