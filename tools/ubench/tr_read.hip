// Semantics probe of ds_read_b64_tr_b16 (gfx950):  hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_read.hip -o build/ub/tr_read && build/ub/tr_read
// Every lane L points at its own 8 bytes holding the four 16-bit values 4L .. 4L+3 (so the value names (source lane, element));
// the output shows which (source lane, element) each destination (lane, element) receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 64; i += 64) lds[i] = 0xffff;
  __syncthreads();
  uint16_t* mine = (uint16_t*)((char*)lds + lane * stride_bytes);
  for (int e = 0; e < 4; ++e) mine[e] = (uint16_t)(4 * lane + e);
  __syncthreads();
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)mine);
  for (int e = 0; e < 4; ++e) out[4 * lane + e] = (uint16_t)v[e];
}
int main() {
  uint16_t* d; hipMalloc(&d, 256 * 2);
  for (int stride : {8, 32}) {
    probe<<<1, 64>>>(d, stride);
    uint16_t h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("stride %d bytes: dest lane: (src lane.elem) x4\n", stride);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
      printf("%2d:", l);
      for (int e = 0; e < 4; ++e) {
        printf(" %2d.%d", h[4 * l + e] >> 2, h[4 * l + e] & 3);
        const int g = l & ~15, i = l & 15;
        if (h[4 * l + e] != 4 * (g + 4 * e + (i >> 2)) + (i & 3)) ok = 0;
      }
      printf((l & 3) == 3 ? "\n" : "   ");
    }
    printf("hypothesis out[i][j] = in[4j + (i>>2)][i&3] per 16-lane group: %s\n", ok ? "HOLDS" : "FAILS");
  }
  return 0;
}
