// What ds_read_b64_tr_b16 returns on this GPU: every lane reads 8 bytes at lds + lane * 8 of an LDS image holding
// element index = position; prints, per lane, the four source positions it received.  (The fragment-major LDS image of
// the TN GEMM operand - gemm_conv.hip, pp_offsets<true> - is built on: lane l, element j <- position
// (l & 15) + 16 * j + 64 * (l >> 4).)   build: hipcc --offload-arch=gfx950 tools/tr_read_probe.hip -o tools/build/tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
  __syncthreads();
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d;
  short h[256];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      printf(" %3d", h[l * 4 + j]);
      bad += h[l * 4 + j] != (l & 15) + 16 * j + 64 * (l >> 4);
    }
    printf("\n");
  }
  printf("%s\n", bad ? "MAPPING DIFFERS from (l & 15) + 16 j + 64 (l >> 4)" : "mapping as assumed: (l & 15) + 16 j + 64 (l >> 4)");
  return 0;
}
