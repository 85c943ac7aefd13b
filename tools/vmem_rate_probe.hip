// How many vector-memory wave-instructions per microsecond does ONE CU of an MI355X retire, by instruction form, access
// pattern and waves per CU?  (round 5: every mainloop tried for the trunk's 64x64 conv tiles - LDS-DMA ring, register ring
// with two / three LDS stages, the old two-K-group kernel - ended at the same ~0.4 us per 16-KB K slab per CU, with and
// without real memory traffic; this probe measures the ceiling they share.)
//   build:  hipcc --offload-arch=gfx950 -O3 -o tools/build/vmem_rate_probe tools/vmem_rate_probe.hip
//   run:    tools/build/vmem_rate_probe
// Every wave issues ITER x 8 loads of 16 bytes per lane (1 KB per wave-instruction) with at most 8 in flight, from a
// footprint that stays in L2 (mode-dependent), and discards the data.  One workgroup per CU, W waves each.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: buffer_load_dwordx4 offen, rows of 128 B (8 lanes per row), row pitch `pitch` bytes, in range
// MODE 1: the same, every lane out of range (no memory access)
// MODE 2: global_load_dwordx4 (flat address), same pattern
// MODE 3: buffer_load_dwordx4 ... lds (LDS-DMA), same pattern
// MODE 4: buffer_load_dword (4 B per lane, one 128-B line per lane-row of 32... 256 B per instruction, contiguous)
template <int MODE>
__global__ __launch_bounds__(1024) void probe(const char* base, long bytes_per_wg, int pitch, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* wg = base + (long)blockIdx.x * bytes_per_wg;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, (unsigned)bytes_per_wg, 0x00020000);
  // a wave-instruction covers 8 rows x 128 B; instruction u of a wave starts at row (wave * 8 + u) * 8
  const unsigned lane_off = (unsigned)((lane >> 3) * pitch + (lane & 7) * 16);
  i32x4 acc = {0, 0, 0, 0};
  const unsigned span = (unsigned)bytes_per_wg;
  unsigned rowbase = (unsigned)(wave * 64 * pitch);
  for (int it = 0; it < iters; ++it) {
    i32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      unsigned off = rowbase + (unsigned)(u * 8 * pitch) + lane_off;
      if (off + 16 > span) off -= span / 2;  // wrap inside the footprint
      if (MODE == 0) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
      else if (MODE == 1) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(0xFFFFFFF0u), 0, 0);
      else if (MODE == 2) v[u] = *(const i32x4*)(wg + off);
      else if (MODE == 3) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (wave * 8 + u) * 1024), 16, (int)off, 0, 0, 0);
        v[u] = i32x4{0, 0, 0, 0};
      } else {
        const int x = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(rowbase + u * 256 + lane * 4) % (int)span, 0, 0);
        v[u] = i32x4{x, 0, 0, 0};
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rowbase += (unsigned)(blockDim.x / 64) * 64u * (unsigned)pitch;
    if (rowbase + 64u * pitch > span) rowbase = (unsigned)(wave * 64 * pitch) % (span / 2);
  }
  if (acc[0] == 0x12345678 && acc[1] == 0x1) sink[0] = acc[2] + acc[3];
}

template <int MODE>
static void run(const char* name, const char* buf, long bytes_per_wg, int pitch, int waves, int* sink) {
  const int iters = 2000;
  const int smem = MODE == 3 ? waves * 8 * 1024 : 0;
  if (smem > 48 * 1024) CHECK(hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(waves * 64), smem, 0, buf, bytes_per_wg, pitch, iters, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double instr = (double)iters * 8 * waves;  // per CU
  const double per_us = instr / (ms * 1e3);
  const double bpi = MODE == 4 ? 256.0 : 1024.0;
  printf("%-34s pitch %5d  waves/CU %2d  footprint/CU %4ld KB : %6.1f instr/us/CU  %6.1f GB/s/CU  (%5.1f cycles/instr at 2.4 GHz)  chip %5.2f TB/s\n",
         name, pitch, waves, bytes_per_wg / 1024, per_us, per_us * bpi / 1e3, 2400.0 / per_us, per_us * bpi * 256 / 1e6);
}

int main() {
  const long total = 256L * 4 * 1024 * 1024;
  char* buf;
  int* sink;
  CHECK(hipMalloc(&buf, total));
  CHECK(hipMemset(buf, 1, total));
  CHECK(hipMalloc(&sink, 64));
  for (int waves : {4, 8, 16}) {
    for (long kb : {64L, 1024L}) {
      for (int pitch : {128, 512, 4608}) {
        run<0>("buffer_load_dwordx4 offen", buf, kb * 1024, pitch, waves, sink);
        run<2>("global_load_dwordx4", buf, kb * 1024, pitch, waves, sink);
        run<3>("buffer_load_dwordx4 ... lds", buf, kb * 1024, pitch, waves, sink);
      }
      run<1>("buffer_load_dwordx4 out of range", buf, kb * 1024, 128, waves, sink);
      run<4>("buffer_load_dword contiguous", buf, kb * 1024, 128, waves, sink);
    }
  }
  return 0;
}
