// Register-ring kernels for the bf16 NHWC convolutions of the trunk at real image sizes (round 5).
//
// What they replace on the reference path: F.conv2d + FrozenBatchNorm2d + relu_ + the shortcut add of the WS-ResNet / VGG
// blocks (projects/WSL/wsl/modeling/backbone/resnet_ws.py:217-237 BottleneckBlock.forward, :672-678 dilated res4 / res5,
// projects/WSL/wsl/modeling/backbone/vgg.py:104-122; detectron2/layers/wrappers.py:94-99, batch_norm.py:45-65).
//
// Why another kernel family.  At 800x1216 the res3 / res4 / res5 layers are GEMMs of 3800-15200 rows x 128-2048 columns with
// K = 128 .. 4608: far too few 256x256 tiles for the GEMM kernels of gemm_conv.hip (15 tiles for the res4 3x3), and the
// register-staged 64x64 / 128x128 conv tiles there ran at 0.5-0.9 us per 128-byte K slab (259 TFLOP/s on the res4 3x3,
// profiles/r4_62_conv_800.txt).  Two measurements of this round say why and what to build instead:
//   * `__syncthreads()` is `s_waitcnt vmcnt(0) lgkmcnt(0)` + `s_barrier`: the old mainloop's barrier per slab drained its whole
//     ring of prefetch loads, so a slab cost one full memory latency whatever the ring depth.
//   * LDS-DMA (`buffer_load ... lds`) is NOT the tool for small tiles: a first version of this file staged the slabs with
//     it (ring of 3-4 LDS stages, one barrier per slab) and ran the res4 3x3 in 17.5 us - and in 17.8 us with EVERY load
//     sent out of range (no memory traffic at all), 13.8 us with the MFMAs removed as well
//     (profiles/r5_03_lds_dma_knockouts.txt): a CU retires one 1-KB LDS-DMA instruction per ~40-55 cycles (~20 B/clk, ~45 GB/s)
//     whatever it fetches.  The 256x256 GEMMs live with that (64 KB per slab against 32 MFMAs per wave); a 64x64 tile does not.
// So: plain 16-byte buffer loads into a statically indexed ring of D register sets (64 B/clk per CU), ds_write into a
// two-stage LDS image one slab ahead, ONE LDS-only barrier per slab (lgkmcnt(0) + s_barrier: the loads stay in flight
// across it; the compiler's counted vmcnt waits sit in front of the ds_writes that consume them), the im2col gather done by
// the per-lane buffer OFFSET - row = output pixel, each K slab = 64 input channels of one tap; taps in the zero padding and
// rows beyond M get an out-of-range offset (the hardware returns zeros): one add and one select per load on a per-row
// tap-validity mask computed once.
// Same LDS image (128-byte rows, slot ^= (row >> 1) & 7), 2 x 2 wave layout, MFMA (32x32x16 bf16) and k order per output
// element as conv_nhwc_kernel: bit-identical to the tiled kernel for every tile shape here.
#include "drn_common.h"
#include "conv_params.h"

#include <type_traits>
#include <utility>

namespace {

using drn_conv::ConvParams;

__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn, int GM) {
  const int group_sz = GM * tiles_n;
  const int g = id / group_sz, in_g = id - g * group_sz;
  const int first_m = g * GM;
  const int gm = tiles_m - first_m < GM ? tiles_m - first_m : GM;
  tm = first_m + in_g % gm;
  tn = in_g / gm;
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// LDS-only barrier: every LDS write / read of this wave is complete, global loads stay in flight
#define RING_BARRIER()                                      \
  do {                                                      \
    __builtin_amdgcn_sched_barrier(0);                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);                      \
    __builtin_amdgcn_s_barrier();                           \
    __builtin_amdgcn_sched_barrier(0);                      \
  } while (0)

// BM x BN output tile, 4 waves (2 x 2), two LDS stages of (BM + BN) x 128 B, register ring of D slabs.
template <int BM, int BN, int D>
__global__ __launch_bounds__(256) void conv_ring_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MI = BM / 64, NJ = BN / 64;
  constexpr int AP = BM / 32, BP = BN / 32;  // 16-byte loads per thread and slab
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  static_assert(BM * BN * 4 <= 2 * STAGE, "epilogue staging fits the two stages");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int HoWo = p.Ho * p.Wo, Mtot = p.Nb * HoWo;
  const int tiles_m = (Mtot + BM - 1) / BM, tiles_n = (p.Cout + BN - 1) / BN;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn, 4);
  const int bm = tm * BM, bn = tn * BN;

  // ---- per-thread sources: A rows i * 32 + tid / 8 (output pixels), B rows j * 32 + tid / 8 (output channels), the 16 bytes
  // of k-slot tid & 7; LDS destination = the row's swizzled slot (the same for every i / j: 32 rows keep (row >> 1) & 7)
  const int r0 = tid >> 3, slot = tid & 7;
  const int ntaps = p.KH * p.KW;
  int abase[AP];
  unsigned amask[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = bm + i * 32 + r0;
    unsigned mask = 0;
    int base = 0;
    if (m < Mtot) {
      const int nb = m / HoWo, rem = m - nb * HoWo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      base = ((nb * p.H + hi0) * p.W + wi0) * p.Cin * 2 + slot * 16;
      for (int t = 0, kh = 0, kw = 0; t < ntaps; ++t) {
        const int hi = hi0 + kh * p.dil, wi = wi0 + kw * p.dil;
        if (hi >= 0 && hi < p.H && wi >= 0 && wi < p.W) mask |= 1u << t;
        if (++kw == p.KW) { kw = 0; ++kh; }
      }
    }
    abase[i] = base;
    amask[i] = mask;
  }
  const unsigned ldw_b = (unsigned)(p.ldw * 2);
  const int vob0 = (int)(r0 * ldw_b + slot * 16), vob_step = (int)(32 * ldw_b);
  const __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (unsigned)((long)p.Nb * p.H * p.W * p.Cin * 2), 0x00020000);
  long brem = (long)p.Cout - bn;
  if (brem > BN) brem = BN;
  const __amdgpu_buffer_rsrc_t rb =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wt + (long)bn * ldw_b), 0, (unsigned)(brem * ldw_b), 0x00020000);
  const int wofs = swz(r0, slot);  // + 32 * 128 per further row group

  // ---- load state (wave-uniform): the loads go out in slab order, so tap / channel position advance by one slab per call
  const int spt = p.Cin >> 6;  // 128-byte slabs per tap
  const int n = ntaps * spt;
  int l_s = 0, l_cs = 0, l_tap = 0, l_kw = 0, l_kh = 0;
  auto load = [&](i32x4_t (&a)[AP], i32x4_t (&b)[BP]) {
    const int delta = ((l_kh * p.dil) * p.W + l_kw * p.dil) * p.Cin * 2 + l_cs * 128;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const unsigned off = ((amask[i] >> l_tap) & 1u) ? (unsigned)(abase[i] + delta) : OOB;
      a[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)off, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BP; ++j) b[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, vob0 + j * vob_step, l_s * 128, 0);
    ++l_s;
    if (++l_cs == spt) {
      l_cs = 0;
      ++l_tap;
      if (++l_kw == p.KW) { l_kw = 0; ++l_kh; }
    }
  };
  auto store = [&](char* stage, const i32x4_t (&a)[AP], const i32x4_t (&b)[BP]) {
#pragma unroll
    for (int i = 0; i < AP; ++i) *(i32x4_t*)(stage + wofs + i * (32 * 128)) = a[i];
#pragma unroll
    for (int j = 0; j < BP; ++j) *(i32x4_t*)(stage + A_BYTES + wofs + j * (32 * 128)) = b[j];
  };

  f32x16_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int l31 = lane & 31, hi = lane >> 5;

  i32x4_t rga[D][AP], rgb[D][BP];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < n) load(rga[d], rgb[d]);
  store(smem, rga[0], rgb[0]);
  RING_BARRIER();
  // one slab: fragments of slab i out of its stage, the loads of slab i + D into the ring slot slab i came from, slab i + 1
  // (fetched D - 1 iterations ago) from its registers into the other stage, the MFMAs, one barrier
  auto slab = [&](auto dtag, int i, auto fulltag) {
    constexpr int d = decltype(dtag)::value;
    constexpr bool FULL = decltype(fulltag)::value;
    const char* cur = smem + (i & 1) * STAGE;
    char* nxt = smem + ((i & 1) ^ 1) * STAGE;
    i32x4_t fa[4][MI], fb[4][NJ];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int sl = ks * 2 + hi;
#pragma unroll
      for (int ii = 0; ii < MI; ++ii) fa[ks][ii] = *(const i32x4_t*)(cur + swz(wm * (BM / 2) + ii * 32 + l31, sl));
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[ks][j] = *(const i32x4_t*)(cur + A_BYTES + swz(wn * (BN / 2) + j * 32 + l31, sl));
    }
    if (FULL || i + D < n) load(rga[d], rgb[d]);
    if (FULL || i + 1 < n) store(nxt, rga[(d + 1) % D], rgb[(d + 1) % D]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int ii = 0; ii < MI; ++ii)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[ks][ii]),
                                                               __builtin_bit_cast(bf16x8_t, fb[ks][j]), acc[ii][j], 0, 0, 0);
    RING_BARRIER();
  };
  int base = 0;
  for (; base + 2 * D <= n; base += D)  // every i here has i + D < n (and i + 1 < n): branch-free, counted waits
    static_for<D>([&](auto dtag) { slab(dtag, base + decltype(dtag)::value, std::true_type{}); });
  for (; base < n; base += D)
    static_for<D>([&](auto dtag) {
      const int i = base + decltype(dtag)::value;
      if (i < n) slab(dtag, i, std::false_type{});
    });

  // ---- epilogue: per-channel affine -> fp32 [BM][BN] through the (now free) stages -> 8 consecutive channels of a pixel per
  // lane: 16-byte residual loads (issued before the staging so their latency hides behind it) and 16-byte stores
  constexpr int LPR = BN / 8, RPP = 256 / LPR, NR = BM / RPP;  // lanes per row, rows per pass, rows per lane
  const int cl = tid % LPR, rl = tid / LPR;
  const int nn = bn + cl * 8;
  const bool col_ok = nn < p.Cout;
  i32x4_t rv[NR];
  if (p.residual && col_ok) {
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int m = bm + rl + q * RPP;
      rv[q] = *(const i32x4_t*)(p.residual + ((long)(m < Mtot ? m : Mtot - 1) * p.ldres + nn) * 2);
    }
  }
  float* tile = (float*)smem;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int nl = wn * (BN / 2) + j * 32 + l31, nc = bn + nl;
    const float sc = (nc < p.Cout && p.scale) ? p.scale[nc] : 1.f;
    const float bi = (nc < p.Cout && p.bias) ? p.bias[nc] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        tile[ml * BN + nl] = acc[i][j][r] * sc + bi;
      }
  }
  RING_BARRIER();
  if (!col_ok) return;
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int ml = rl + q * RPP, m = bm + ml;
    if (m >= Mtot) break;
    const f32x4_t lo = *(const f32x4_t*)(tile + ml * BN + cl * 8), hi4 = *(const f32x4_t*)(tile + ml * BN + cl * 8 + 4);
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    if (p.residual) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t w = (uint32_t)rv[q][e];
        v[2 * e] += __builtin_bit_cast(float, w << 16) * p.res_mult;
        v[2 * e + 1] += __builtin_bit_cast(float, w & 0xffff0000u) * p.res_mult;
      }
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    i32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (int)((uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16));
    *(i32x4_t*)(p.Y + ((long)m * p.ldy + nn) * 2) = o;
  }
}

template <int BM, int BN, int D>
int launch_ring(const ConvParams& p, hipStream_t st) {
  const long Mtot = (long)p.Nb * p.Ho * p.Wo;
  const int tiles = (int)((Mtot + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
  constexpr int smem = 2 * (BM + BN) * 128;
  auto k = conv_ring_kernel<BM, BN, D>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DRN_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(tiles), dim3(256), smem, st, p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int g_conv_ring = 1;  // drn_tune(DRN_TUNE_CONV_RING = 23): 0 = off, 1 = tile by cost model, 64 / 128 pin 64x64 / 128x128

}  // namespace

// (hidden: called by drn_conv2d_nhwc_q / drn_tune in gemm_conv.hip)
__attribute__((visibility("hidden"))) int drn_conv_ring_set(int v) {
  const int old = g_conv_ring;
  if (v == 0 || v == 1 || v == 64 || v == 128) g_conv_ring = v;
  return old;
}

// Runs the convolution on the register-ring kernels when it is in their class; DRN_ERR_UNSUPPORTED otherwise (the caller
// then takes the kernels of gemm_conv.hip).  `cus` = compute units of the device; `tiles64_one` = 64x64 tiles of ONE image of
// this layer: the class is decided on one image's geometry, so a layer takes the same kernel family - the same fp32 summation
// order - whether its image runs alone or in a batch (graphed trunk groups vs eager steps, 2 ranks vs 1).
__attribute__((visibility("hidden"))) int drn_conv_ring_try(const ConvParams& p, int dtype, int cus, long tiles64_one, hipStream_t st) {
  if (!g_conv_ring || dtype != DRN_BF16 || p.out_dt != DRN_BF16 || (p.residual && p.res_dt != DRN_BF16)) return DRN_ERR_UNSUPPORTED;
  if ((p.Cin & 63) || (p.Cout & 7) || p.KH * p.KW > 32 || (p.ldy & 7) || (p.residual && (p.ldres & 7))) return DRN_ERR_UNSUPPORTED;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (!al16(p.X) || !al16(p.Wt) || !al16(p.Y) || (p.residual && !al16(p.residual)) || (p.ldw * 2) % 16 != 0) return DRN_ERR_UNSUPPORTED;
  if ((long)p.Cout * p.ldw * 2 >= 0xFFFFFFF0L) return DRN_ERR_UNSUPPORTED;
  const int nslab = p.KH * p.KW * (p.Cin >> 6);
  // Where the ring kernel wins (tools/conv_bench.py at 800x1216, profiles/r5_04_*, r5_12_*): layers of >= 4 K slabs on more
  // than CUs / 4 and up to ~4 rounds of 64x64 tiles per image - the res3 / res4 1x1 and 3x3 layers of a real-size image.
  // Single-slab 1x1s and the huge res2 maps are bound by their output traffic (the 128-wide tiles of gemm_conv.hip move fewer
  // operand bytes there); maps of fewer than 1024 pixels (the 224x224 benchmark image) stay in the small-map kernels' class.
  int pick = g_conv_ring;
  if (pick == 1) {
    if (nslab < 4 || tiles64_one <= cus / 4 || tiles64_one > 4L * cus || (long)p.Ho * p.Wo < 1024) return DRN_ERR_UNSUPPORTED;
    pick = 64;
  }
  if (pick == 128) return launch_ring<128, 128, 3>(p, st);
  return launch_ring<64, 64, 4>(p, st);
}
