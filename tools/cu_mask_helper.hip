// CU-masked streams + "which CU did this workgroup run on" (tools/cu_partition_probe.py).
//   build: hipcc --offload-arch=gfx950 -shared -fPIC tools/cu_mask_helper.hip -o tools/build/libcu_mask_helper.so
#include <hip/hip_runtime.h>
#include <cstdint>
extern "C" {
// mask: n 32-bit words, bit i = CU i of the device in the driver's enumeration; returns the stream handle (0 on failure)
void* cum_stream_create(const uint32_t* mask, int nwords) {
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask) != hipSuccess) return nullptr;
  return (void*)s;
}
int cum_stream_destroy(void* s) { return hipStreamDestroy((hipStream_t)s) == hipSuccess ? 0 : -1; }
}
// out[block] = (xcc_id << 16) | (se_id << 8) | (sh_id << 7 ...) raw HW_ID in the low half: decoded on the host
__global__ void where_kernel(uint32_t* out, int spin) {
  if (threadIdx.x == 0) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // stay resident for a while so that the blocks spread over every CU the mask allows
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
}
extern "C" int cum_where(uint32_t* out_dev, int blocks, int threads, int lds_bytes, int spin, void* stream) {
  hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, out_dev, spin);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
