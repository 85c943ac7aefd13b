// MFMA GEMM (NT) and implicit-GEMM NHWC convolution for gfx950.
//
// One mainloop serves both and both compute dtypes:
//   * bf16 : v_mfma_f32_32x32x16_bf16, fp32 accumulate  (fast mode; BASELINE configs[1])
//   * fp32 : v_mfma_f32_32x32x2_f32 (exact fp32 fma chain; parity mode, losses within 1e-4)
// Tile BMxBN outputs, K consumed in 128-BYTE slabs per row (64 bf16 / 32 f32), so the LDS image
// and the global->LDS staging are byte-identical for both dtypes; only the MFMA issue differs.
// 256 threads = 4 waves (2x2), each wave owns (BM/2)x(BN/2) as 32x32 MFMA tiles.
// Staging: 16-B buffer loads (hardware bounds check => free zero fill of ragged M/N edges) into
// registers, issued one K-slab ahead, written to a 2-stage XOR-swizzled LDS ring after the MFMAs
// (issue-early / write-late), one barrier per slab. ds_read_b128 is conflict-free under the
// swizzle slot ^= (row>>1)&7 (rows are 128 B, a 256-B bank row holds two).
//
// Replaces on the reference path: F.linear of fc6/fc7 and the predictor Linears
// (projects/WSL/wsl/modeling/roi_heads/box_head.py:82-91, fast_rcnn.py:493-527, :1363-1387) and
// F.conv2d + FrozenBatchNorm2d + relu_ + residual add (detectron2/layers/wrappers.py:94-99,
// detectron2/layers/batch_norm.py:45-65, projects/WSL/wsl/modeling/backbone/resnet_ws.py:217-237).
#include "drn_common.h"
#include "conv_params.h"

#include <type_traits>
#include <utility>

namespace {

struct GemmParams {
  const char* A;
  const char* B;
  float* C;
  int M, N, K;
  long lda, ldb, ldc;  // elements
  int k_slabs_per_split;
  long c_split_stride;  // elements
  int accumulate;
  // fused optimizer step (drn_gemm_tn_sgd, SgdPipe): the previous tile's update rides in the next tile's mainloop
  float* sgd_w;
  float* sgd_mom;
  bf16_t* sgd_shadow;  // bf16 compute copy of W (same layout) or null
  const SgdSeg* sgd_seg;  // lr / wd of this tensor, read on the device (a replayed hipGraph follows the schedule)
  float sgd_momentum, sgd_grad_scale;
  int sgd_first_step;
  long sgd_ld;  // SGDP (drn_gemm_tn_sgd): row pitch of sgd_w / sgd_mom / sgd_shadow in elements (C = the bf16 gradient bucket, ldc)
  int c_bf16;  // C holds bf16 (gradient buckets that cross xGMI in bf16); splits == 1, no accumulate
  int nsplit;  // number of K-splits (the persistent kernel's grid is 1-D: it cannot read it from gridDim.y)
  int gm;      // tile rows per group of the XCD patch mapping (tile_coords)
  // TN operand (drn_gemm_tn): B is given as Bt [kb_rows][ldb] row-major - the contraction index is the ROW index - and the
  // 256x256 ping-pong kernels read it through transposing LDS reads (ds_read_b64_tr_b16); rows >= kb_rows read as zeros
  int kb_rows;
};

using drn_conv::ConvParams;

template <int DT>
__device__ __forceinline__ void mma_step(f32x16_t& acc, const i32x4_t& a, const i32x4_t& b) {
  if constexpr (DT == DRN_BF16) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                  acc, 0, 0, 0);
  } else if constexpr (DT == DRN_FP8) {
    // a 16-byte fragment holds 16 fp8 k-values: two K=16 steps of v_mfma_f32_32x32x16_fp8_fp8 (8 bytes per lane each;
    // lanes 0-31 carry k 0-7, lanes 32-63 k 8-15 of a step).  A and B use the same byte -> k assignment, so any
    // assignment is a permutation of the contraction index.
    typedef long i64x2_t __attribute__((ext_vector_type(2)));
    const i64x2_t al = __builtin_bit_cast(i64x2_t, a), bl = __builtin_bit_cast(i64x2_t, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(al[0], bl[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(al[1], bl[1], acc, 0, 0, 0);
  } else {
    const f32x4_t af = __builtin_bit_cast(f32x4_t, a), bf = __builtin_bit_cast(f32x4_t, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
  }
}

// fp8 at the fp8 RATE (round 3): v_mfma_scale_f32_32x32x64_f8f6f4 with both formats e4m3 and unit E8M0 scales (0x7f = 2^0),
// the only K = 64 fp8 MFMA on gfx950 - the non-scaled 32x32x16 form above runs at the bf16 rate (MI355X_MICROARCH.md,
// matrix-core table).  A lane's 32 operand bytes are two 16-byte fragments of consecutive k-steps; A and B use the same
// (lane half, byte) -> k assignment, so whatever the hardware's assignment is, it is a permutation of the contraction
// index, and fp8 x fp8 products are exact in fp32 (4-bit significands): only the fp32 summation order differs from the
// K = 16 form.
typedef int i32x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mma_step64_fp8(f32x16_t& acc, const i32x4_t& a0, const i32x4_t& a1, const i32x4_t& b0,
                                               const i32x4_t& b1) {
  const i32x8_t A = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  const i32x8_t B = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}
static int g_fp8_k64 = 1;  // drn_tune(DRN_TUNE_FP8_K64 = 13): 0 = the K = 16 non-scaled fp8 MFMA (A/B; bf16 rate)

__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

template <int ROWS>
__device__ __forceinline__ void lds_store_tile(char* lds, const i32x4_t (&r)[ROWS / 32], int tid) {
  const int slot = tid & 7, r0 = tid >> 3;
#pragma unroll
  for (int p = 0; p < ROWS / 32; ++p) *(i32x4_t*)(lds + swz(r0 + 32 * p, slot)) = r[p];
}

// Plain row-major operand [rows][ld] read through a bounds-checked buffer descriptor.
struct RowLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned ld_bytes;
  template <int ROWS>
  __device__ __forceinline__ void load(i32x4_t (&r)[ROWS / 32], int slab, int tid) const {
    const unsigned slot = tid & 7, r0 = tid >> 3;
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p)
      r[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (r0 + 32 * p) * ld_bytes + slot * 16, slab * 128, 0);
  }
};

__device__ __forceinline__ RowLoader make_row_loader(const char* base, long row0, int rows_total, int rows_tile,
                                                     long ld_bytes) {
  RowLoader l;
  long rem = (long)rows_total - row0;
  if (rem > rows_tile) rem = rows_tile;
  if (rem < 0) rem = 0;
  l.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + row0 * ld_bytes), 0, (unsigned)(rem * ld_bytes), 0x00020000);
  l.ld_bytes = (unsigned)ld_bytes;
  return l;
}

// im2col-on-the-fly operand of an NHWC convolution: row = output pixel, k = (kh, kw, ci).  Branch-free: every lane
// issues a bounds-checked buffer load, and lanes that fall into the zero padding, beyond the last tap or beyond M
// use an offset past the descriptor's range, for which the hardware returns zeros.  (Conditional global loads made
// the compiler drain each one before leaving its branch, so a slab cost a full memory latency whatever the prefetch
// depth: 0.85 us per slab on the res4 3x3 convs.)
template <int DT, int ROWS>
struct ConvLoader {
  static constexpr int ES = EsOf<DT>::value;
  static constexpr unsigned OOB = 0xFFFFFFF0u;
  __amdgpu_buffer_rsrc_t rsrc;  // whole input tensor (launcher guarantees < 4 GB - 16)
  int H, W, Cin, KW, dil, ntaps;
  int hi0[ROWS / 32], wi0[ROWS / 32];
  unsigned nbase[ROWS / 32];  // byte offset of image n, or OOB for rows beyond M
  // (tap, channel) of this lane's 16-byte chunk in slab `at`: the mainloop asks for consecutive slabs, so the position is
  // advanced by one slab (128 bytes of k) instead of being re-derived with two integer divisions per slab - ~80 VALU
  // instructions that sat on the critical path of the latency-bound small-map layers
  int at = -1, tap = 0, ci = 0, kh = 0, kw = 0;
  __device__ __forceinline__ void seek(int slab, int tid) {
    const int k = (slab * 128 + (tid & 7) * 16) / ES;
    tap = k / Cin; ci = k - tap * Cin;
    kh = tap / KW; kw = tap - kh * KW;
    at = slab;
  }
  template <int R>
  __device__ __forceinline__ void load(i32x4_t (&r)[R / 32], int slab, int tid) {
    static_assert(R == ROWS, "tile rows");
    if (slab != at) seek(slab, tid);
    else if (Cin * ES < 128) seek(slab, tid);  // several taps per slab (the 3-channel stem): no single-step advance
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
      const int hi = hi0[p] + kh * dil, wi = wi0[p] + kw * dil;
      const bool ok = nbase[p] != OOB && tap < ntaps && hi >= 0 && hi < H && wi >= 0 && wi < W;
      const unsigned off = ok ? nbase[p] + (unsigned)(((hi * W + wi) * Cin + ci) * ES) : OOB;
      r[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    }
    ci += 128 / ES;  // next slab (exact when Cin * ES >= 128: at most one tap boundary per slab; otherwise re-sought)
    if (ci >= Cin) {
      ci -= Cin;
      ++tap;
      if (++kw == KW) { kw = 0; ++kh; }
    }
    at = slab + 1;
  }
};

// Register-staged mainloop of the 64x64 / 128x128 kernels (GEMM and conv share it).  These tiles do little MFMA work
// per K-slab (one to four MFMAs per wave and k-step), so a slab's cost is the latency of its global loads unless they
// are issued far ahead: the staging registers form a statically indexed ring of DEPTH slabs - while slab i is
// multiplied out of LDS, slab i+DEPTH is being fetched and slab i+1 (fetched DEPTH-1 iterations ago) moves from its
// registers into the other LDS stage.
// STAGES = 1 (the 64x64 conv): ONE 16-KB LDS stage, refilled between two barriers, and a 3-deep register ring.  Measured
// (tools/coexist_bench.py and an LDS-size probe): a workgroup shares a CU with a 256x256 GEMM workgroup (128 KB LDS,
// 2 x 200 VGPRs per SIMD) only with < 32 KB of LDS and <= 112 registers per wave; the two-stage version is exactly 32 KB
// and 120 registers, so every conv launch of the frozen trunk used to wait for a GEMM workgroup to retire (a chain of ten
// res4 convs beside the fc6 GEMM: 475 us instead of 190).
#ifndef CONV_DEPTH1
#define CONV_DEPTH1 4  // register-ring depth of the single-LDS-stage 64x64 conv (98 VGPRs; 5 -> 114 > the 112 that co-reside with a GEMM workgroup)
#endif
// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Register-staged mainloop (global -> ring of DEPTH register sets -> LDS -> MFMA).  The slab loop runs in two parts:
// a STEADY STATE of whole groups of DEPTH slabs in which every slab issues the loads of slab i + DEPTH and stores the
// registers of slab i + 1 unconditionally, then the last < 2 * DEPTH slabs with the bounds checks.  With the checks inside
// the only loop (round 1 / first half of round 2) the compiler's wait-count pass lost track of the ring across the
// branches and put `s_waitcnt vmcnt(0)` in front of every LDS store - it waited for the loads issued in the SAME
// iteration, i.e. the ring prefetched one slab ahead whatever its depth (which is why ring depth 3 / 4 / 5 all measured
// ~0.55 us per slab).  Branch-free, the waits become counted (vmcnt = loads of the DEPTH - 1 younger slabs).
template <int DT, int BM, int BN, class ALoader, class BLoader, int STAGES = 2, bool K64 = true>
__device__ __forceinline__ void mainloop(f32x16_t (&acc)[BM / 64][BN / 64], char* smem, ALoader& la,
                                         const BLoader& lb, int s0, int s1, int tid = threadIdx.x) {
  // tid: 0..255 within the four waves that share this tile's LDS stage (conv_nhwc_k2_kernel runs two such groups)
  constexpr int MI = BM / 64, NJ = BN / 64;
  constexpr int A_BYTES = BM * 128, STAGE = STAGES == 2 ? (BM + BN) * 128 : 0;
  constexpr int DEPTH = STAGES == 1 ? CONV_DEPTH1 : (BM <= 64 ? 4 : 3);
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  i32x4_t ra[DEPTH][BM / 32], rb[DEPTH][BN / 32];
  const int n = s1 - s0;
  if (n <= 0) return;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < n) {
      la.template load<BM>(ra[d], s0 + d, tid);
      lb.template load<BN>(rb[d], s0 + d, tid);
    }
  lds_store_tile<BM>(smem, ra[0], tid);
  lds_store_tile<BN>(smem + A_BYTES, rb[0], tid);
  __syncthreads();
  auto slab = [&](auto dtag, int i, auto fulltag) {
    constexpr int d = decltype(dtag)::value;
    constexpr bool FULL = decltype(fulltag)::value;
    char* cur = smem + (i & 1) * STAGE;
    char* nxt = smem + ((i & 1) ^ 1) * STAGE;
    if (FULL || i + DEPTH < n) {  // ring slot d held slab i, which already sits in LDS
      la.template load<BM>(ra[d], s0 + i + DEPTH, tid);
      lb.template load<BN>(rb[d], s0 + i + DEPTH, tid);
    }
    if constexpr (DT == DRN_FP8 && K64) {  // two k-steps per MFMA (K = 64 scaled form: the fp8 rate)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        i32x4_t fa[2][MI], fb[2][NJ];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int slot = (k2 * 2 + h) * 2 + (lane >> 5);
#pragma unroll
          for (int ii = 0; ii < MI; ++ii)
            fa[h][ii] = *(const i32x4_t*)(cur + swz(wm * (BM / 2) + ii * 32 + (lane & 31), slot));
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            fb[h][j] = *(const i32x4_t*)(cur + A_BYTES + swz(wn * (BN / 2) + j * 32 + (lane & 31), slot));
        }
#pragma unroll
        for (int ii = 0; ii < MI; ++ii)
#pragma unroll
          for (int j = 0; j < NJ; ++j) mma_step64_fp8(acc[ii][j], fa[0][ii], fa[1][ii], fb[0][j], fb[1][j]);
      }
      // keep a slab's MFMAs in the slab (an empty volatile asm that "uses" the accumulators): left alone, the compiler
      // sinks the register-only K = 64 MFMAs out of their blocks to the end of the unrolled slab group and holds every
      // slab's fragments live until then (208 VGPRs for the 64x64 tile instead of ~120: no longer co-resident with a
      // 256x256 GEMM workgroup)
#pragma unroll
      for (int ii = 0; ii < MI; ++ii)
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(acc[ii][j]));
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      i32x4_t fa[MI], fb[NJ];
      const int slot = ks * 2 + (lane >> 5);
#pragma unroll
      for (int ii = 0; ii < MI; ++ii) fa[ii] = *(const i32x4_t*)(cur + swz(wm * (BM / 2) + ii * 32 + (lane & 31), slot));
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        fb[j] = *(const i32x4_t*)(cur + A_BYTES + swz(wn * (BN / 2) + j * 32 + (lane & 31), slot));
#pragma unroll
      for (int ii = 0; ii < MI; ++ii)
#pragma unroll
        for (int j = 0; j < NJ; ++j) mma_step<DT>(acc[ii][j], fa[ii], fb[j]);
    }
    }
    if (STAGES == 1) __syncthreads();  // every wave is done reading the only stage
    if (FULL || i + 1 < n) {
      lds_store_tile<BM>(nxt, ra[(d + 1) % DEPTH], tid);
      lds_store_tile<BN>(nxt + A_BYTES, rb[(d + 1) % DEPTH], tid);
    }
    __syncthreads();
  };
  int base = 0;
  for (; base + 2 * DEPTH <= n; base += DEPTH)  // every i here has i + DEPTH < n (and i + 1 < n)
    static_for<DEPTH>([&](auto dtag) { slab(dtag, base + decltype(dtag)::value, std::true_type{}); });
  for (; base < n; base += DEPTH)
    static_for<DEPTH>([&](auto dtag) {
      const int i = base + decltype(dtag)::value;
      if (i < n) slab(dtag, i, std::false_type{});
    });
}

// logical tile id -> (tm, tn), grouped so that a contiguous id range (one XCD's share) covers a
// compact 2-D patch of tiles and re-reads its operand panels from that XCD's L2.
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn, int GM = 4) {
  const int group_sz = GM * tiles_n;
  const int g = id / group_sz, in_g = id - g * group_sz;
  const int first_m = g * GM;
  const int gm = tiles_m - first_m < GM ? tiles_m - first_m : GM;
  tm = first_m + in_g % gm;
  tn = in_g / gm;
}

template <int DT, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = DT == DRN_BF16 ? 2 : 4;
  constexpr int MI = BM / 64, NJ = BN / 64;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  // (tile, split) are linearised together before the XCD remap: one XCD then owns a contiguous
  // run of tiles of ONE K-split (a 4 x tiles_n patch), so its L2 fetches each operand panel once.
  const int tiles = tiles_m * tiles_n;
  const int logical = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, tiles * gridDim.y);
  const int split = logical / tiles;
  int tm, tn;
  tile_coords(logical - split * tiles, tiles_m, tiles_n, tm, tn, p.gm);
  const int bm = tm * BM, bn = tn * BN;
  const int nslab = p.K * ES / 128;
  const int s0 = split * p.k_slabs_per_split;
  const int s1 = s0 + p.k_slabs_per_split < nslab ? s0 + p.k_slabs_per_split : nslab;
  const RowLoader la = make_row_loader(p.A, bm, p.M, BM, p.lda * ES);
  const RowLoader lb = make_row_loader(p.B, bn, p.N, BN, p.ldb * ES);
  f32x16_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  mainloop<DT, BM, BN>(acc, smem, la, lb, s0, s1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float* C = p.C + (long)split * p.c_split_stride;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = bn + wn * (BN / 2) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = bm + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < p.M && n < p.N) {
          if (p.c_bf16) {
            ((bf16_t*)p.C)[(long)m * p.ldc + n] = f32_to_bf16(acc[i][j][r]);
          } else {
            float* dst = C + (long)m * p.ldc + n;
            *dst = p.accumulate ? *dst + acc[i][j][r] : acc[i][j][r];
          }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// PING-PONG mainloop of the 256x256 kernels (round 3; bf16).  Same tile, same wave layout (2 x 4 waves, 128x64 outputs per
// wave as 4x2 MFMA 32x32 tiles), same MFMA and the same k order per output element as the pipelined mainloop it replaces
// (bit-identical results), another SCHEDULE.  What the round-3 PMC passes said about the old one
// (profiles/r3_00_pmc_fwd.json, _dw.json): the MFMA pipes are 64 % / 56 % busy at the sustained clock, every wave is
// parked at a wait / barrier 40 % of its cycles and stalled at issue another 40-44 %, the LDS array is active 24 % of the
// CU cycles and sees no bank conflicts - operand FEED, not LDS bandwidth.  In the old loop all eight waves run the same
// stream in lock-step: the eight LDS-DMA issues of a slab (60-185 cycles each once the phase also carries fragment
// reads) sit in ONE k-step group of every wave at the same time, and at the single barrier per slab both waves of a SIMD
// are parked together, so nobody feeds the matrix pipe.  Here the two waves that share a SIMD (wave w and w + 4: the two
// rows of the 2 x 4 layout) run HALF A PHASE APART: a K slab is four phases of
//     [fragment reads + 2 LDS-DMA issues]  barrier  [8 MFMAs = one 64x32 quadrant x K 64, s_setprio 1]  barrier
// and the lower wave row enters the loop one barrier late, so while one wave of a SIMD multiplies, its partner reads
// fragments and issues DMA - the pipe always has a wave in its matrix section (cdna_hip_programming.md 8-phase
// template, restated for 32x32x16 MFMAs and the D^T epilogue of this file).
// LDS: a stage (64 KB) is four half tiles of 16 KB, each the rows ONE phase starts to read: A0 = the first 64 rows of both
// wave rows' 128-row blocks, B0 = the first 32 rows of the four wave columns' 64-row blocks, B1, A1 the second halves.
// Phase p of slab t stages ONE half tile of slab t + 1 (A0, B0, B1, A1: the order they are first read), 16 DMA pieces of
// 1 KB spread over the slab instead of one burst; a counted vmcnt(4) in phases 4, 1 and 2 retires exactly the half tile the
// NEXT phase starts to read and leaves the two younger ones in flight across the barriers.
// Hazards (the lower wave row runs one barrier interval late): a half tile is read one phase after the wait that
// retires it (every wave's wait precedes a barrier the reader passes); a half tile of stage b^1 is re-staged at phase p
// of slab t when its last fragment read was phase <= p of slab t - 1 - six or more barriers earlier.
constexpr int PP_A0 = 0, PP_B0 = 16384, PP_B1 = 32768, PP_A1 = 49152, PP_STAGE = 65536;

// per-thread source offsets of its two 16-byte chunks of every half tile (index = half * 2 + piece): the LDS image of a
// half tile is row-major [128][128 B], piece `pc` of thread tid lands in LDS row pc * 64 + tid / 8, slot tid & 7, and
// fetches the k-slot pre-swizzled with that row (same involution as the fragment reads)
//
// TN = true (round 3, the fc6 weight gradient reading the pooled matrix A [R][C*49] itself instead of a materialised A^T):
// the B operand arrives as Bt [k][n] row-major and is consumed through ds_read_b64_tr_b16, which hands lane l the 4-element
// COLUMN (l & 15) of a [4 k][16 n] block whose rows the 16 lanes of its group address freely, 8 bytes each.
//   * Wave columns: wave wn owns the 32-column blocks wn of the tile's two 128-column halves (b0: n = wn * 32 .., b1:
//     n = 128 + wn * 32 ..; the NT form owns 64 adjacent columns), so that the half tiles B0 / B1 of the staging order are
//     whole 256-byte row segments of Bt.
//   * LDS image of a half tile: 16 pieces of 1 KB, piece (k-step ks, quarter kq) = Bt rows ks * 16 + 4 * kq + j (j = 0..3)
//     x 128 columns, row pitch 256 B, the 32-byte column units u of row j stored at u ^ 2j.  One LDS-DMA instruction
//     fills one piece lane-linearly from four FULL 256-byte row segments (the first version fetched 16 rows x 64 bytes
//     per instruction and ran the fc6 dW slab at 206 us against 178 us for the NT form).
//   * A fragment half (k = 8 * (g >> 1) + 4 * i + j for lane group g) is one ds_read_b64_tr_b16: lanes 0-31 read piece
//     kq = i, lanes 32-63 piece kq = 2 + i, each group its 4 rows x 32 bytes at unit (2 * wn + (g & 1)) ^ 2j - the 32 lanes
//     of an LDS cycle cover 8 different units = all 64 banks once.
template <bool TN = false>
__device__ __forceinline__ void pp_offsets(unsigned lda_b, unsigned ldb_b, int tid, unsigned (&voa)[4], unsigned (&vob)[4]) {
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const unsigned row = pc * 64 + (tid >> 3), ks = (tid & 7) ^ ((row >> 1) & 7);
      const unsigned ga = pc * 128 + h * 64 + (tid >> 3);                             // A row: wave row pc, half h
      voa[h * 2 + pc] = ga * lda_b + ks * 16;
      if constexpr (TN) {
        const unsigned L = tid & 63, wv = tid >> 6;          // piece f = pc * 8 + wave: (ks = f >> 2, kq = f & 3)
        const unsigned ks_f = pc * 2 + (wv >> 2), kq_f = wv & 3;
        const unsigned j = L >> 4, pos = L & 15, u = (pos >> 1) ^ (2 * j);  // LDS row j, 16-byte position pos of 16
        const unsigned k = ks_f * 16 + kq_f * 4 + j;
        vob[h * 2 + pc] = k * ldb_b + h * 256 + u * 32 + (pos & 1) * 16;
      } else {
        const unsigned gb = (2 * pc + (tid >> 8)) * 64 + h * 32 + ((tid >> 3) & 31);  // B row: wave column 2 pc + tid / 256
        vob[h * 2 + pc] = gb * ldb_b + ks * 16;
      }
    }
}

template <int WHICH>  // 0: A0, 1: B0, 2: B1, 3: A1 - one half tile = 2 DMA pieces per thread
__device__ __forceinline__ void pp_issue(char* stage, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                                         const unsigned (&voa)[4], const unsigned (&vob)[4], int wave, int slab,
                                         unsigned bstep = 128) {  // bytes a K slab advances in B: 128 (NT) or 64 rows (TN)
  constexpr int off = WHICH == 0 ? PP_A0 : WHICH == 1 ? PP_B0 : WHICH == 2 ? PP_B1 : PP_A1;
  constexpr bool isA = WHICH == 0 || WHICH == 3;
  constexpr int h = (WHICH == 2 || WHICH == 3) ? 1 : 0;
#pragma unroll
  for (int pc = 0; pc < 2; ++pc) {
    char* dst = stage + off + (pc * 64 + wave * 8) * 128;  // wave-uniform LDS base of this 1-KB piece
    __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? ra : rb, (__attribute__((address_space(3))) void*)dst, 16,
                                             isA ? voa[h * 2 + pc] : vob[h * 2 + pc], isA ? slab * 128 : slab * bstep, 0, 0);
  }
}

#define PP_BARRIER()                        \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    __builtin_amdgcn_s_barrier();           \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)

// prologue half: the four half tiles of slab `slab` into stage 0 (the persistent kernel issues it ahead of the previous
// tile's epilogue)
__device__ __forceinline__ void pp_issue_first(char* smem, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                                               const unsigned (&voa)[4], const unsigned (&vob)[4], int wave, int slab,
                                               unsigned bstep = 128) {
  pp_issue<0>(smem, ra, rb, voa, vob, wave, slab, bstep);
  pp_issue<1>(smem, ra, rb, voa, vob, wave, slab, bstep);
  pp_issue<2>(smem, ra, rb, voa, vob, wave, slab, bstep);
  pp_issue<3>(smem, ra, rb, voa, vob, wave, slab, bstep);
}

// ---- optimizer step of the PREVIOUS tile inside this tile's mainloop (round 4: drn_gemm_tn_sgd) ------------------------------
// The fc6 weight gradient leaves its GEMM as a bf16 tile of the gradient bucket (the persistent kernel's LDS-staged
// epilogue).  Instead of a second kernel that streams w / momentum / gradient / shadow behind it - HBM-bound phases with the
// matrix pipes idle, or a co-resident kernel competing for the same power budget in bursts - every workgroup applies the
// update of the tile it finished LAST while it multiplies the next one: a tile is 256 rows x 1 KB of fp32 weights, a K slab
// is 1/32 of the mainloop, so slab i carries chunk i = 8 rows (one per wave; a lane owns 4 consecutive parameters): three
// 16/16/8-byte loads (w, momentum, the bf16 gradient the workgroup itself wrote - visible after the vmcnt(0) + barrier at
// the head of the mainloop) issued in phase 1 of slab i, ~30 VALU instructions and three stores in phase 3 of slab i + 1,
// i.e. in the fragment-read phases, while the SIMD's other wave is in its MFMA phase.  HBM traffic becomes a steady
// stream under the MFMA work instead of a burst between launches, and the gradient is read back from L2, not from HBM.
// All of it is inline asm: the mainloop's LDS-DMA pipeline lives on hand-counted vmcnt waits, vector memory operations
// retire in order, and every wait below is the old count plus the optimizer operations issued behind the piece it guards:
//   phase 1 / 2: vmcnt(4) -> vmcnt(7)   (3 loads of this slab's phase 1)      phase 4: vmcnt(4) -> vmcnt(7) from slab 1 on
//   (3 stores of phase 3);  chunk i - 1's loads have 18 (slab 1: 15) younger operations when phase 3 of slab i needs them.
// Same arithmetic, element for element, as sgd_kernel on the bf16 bucket (bit-identical: tests).
struct SgdPipe {
  float* w; float* m; const bf16_t* g; bf16_t* s;  // bases (wave-uniform); in-place update
  unsigned off, step;    // byte offset of this lane's 16 B in chunk 0 of the tile (fp32 arrays), bytes per chunk (8 rows)
  unsigned goff, gstep;  // the same in the bf16 gradient bucket (8 B per lane); the shadow uses off / 2, step / 2
  float lr, wd, mom, gs;
  int first;
};
typedef float f32x2_t_ __attribute__((ext_vector_type(2)));
struct SgdRegs { f32x4_t w, m; f32x2_t_ g; };

// (plain, compiler-visible accesses: the compiler's own counted vmcnt wait in front of the first use is exact - it counts
// the LDS-DMA pieces as the vector memory operations they are.  A first version issued them as inline asm with hand-placed
// waits: the compiler, taking an asm output for valid at once, spilled the registers to scratch right behind the loads.)
__device__ __forceinline__ void sgdp_load(const SgdPipe& sp, SgdRegs& r, unsigned off, unsigned goff) {
  r.w = __builtin_nontemporal_load((const f32x4_t*)((const char*)sp.w + off));
  r.m = __builtin_nontemporal_load((const f32x4_t*)((const char*)sp.m + off));
  r.g = __builtin_nontemporal_load((const f32x2_t_*)((const char*)sp.g + goff));
}
__device__ __forceinline__ void sgdp_apply_store(const SgdPipe& sp, const SgdRegs& r, unsigned off) {
#pragma clang fp contract(off)
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const u32x2_t x = __builtin_bit_cast(u32x2_t, r.g);
  const f32x4_t gg = {__builtin_bit_cast(float, x.x << 16), __builtin_bit_cast(float, x.x & 0xffff0000u),
                      __builtin_bit_cast(float, x.y << 16), __builtin_bit_cast(float, x.y & 0xffff0000u)};
  f32x4_t nb, nw;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float d = gg[e] * sp.gs;
    if (sp.wd != 0.f) d = d + sp.wd * r.w[e];
    nb[e] = sp.first ? d : sp.mom * r.m[e] + d;
    nw[e] = r.w[e] - sp.lr * nb[e];
  }
  u32x2_t o;
  o.x = (uint32_t)f32_to_bf16(nw[0]) | ((uint32_t)f32_to_bf16(nw[1]) << 16);
  o.y = (uint32_t)f32_to_bf16(nw[2]) | ((uint32_t)f32_to_bf16(nw[3]) << 16);
  __builtin_nontemporal_store(nb, (f32x4_t*)((char*)sp.m + off));
  __builtin_nontemporal_store(nw, (f32x4_t*)((char*)sp.w + off));
  *(u32x2_t*)((char*)sp.s + (off >> 1)) = o;
}

// slabs [s0, s1), s0 < s1; slab s0 has been issued into stage 0 by pp_issue_first.  Every wave passes the same number of
// barriers (wave row 1 one extra in front, wave row 0 one extra behind).
// SGDP: the optimizer step of the previous tile rides along (above); VAR == 1.  s1 - s0 < 32: the chunks the loop has no
// slab for follow it, exposed.
template <int DT, int VAR, bool TN = false, bool SGDP = false>
__device__ __forceinline__ void pp_mainloop(f32x16_t (&acc)[4][2], char* smem, __amdgpu_buffer_rsrc_t ra,
                                            __amdgpu_buffer_rsrc_t rb, const unsigned (&voa)[4], const unsigned (&vob)[4],
                                            int s0, int s1, int lane, int wave, unsigned bstep = 128,
                                            const SgdPipe* spp = nullptr) {
  SgdRegs srA, srB;  // chunk i is loaded into A (even slabs) / B (odd slabs) and applied one slab later
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 1) & 7;
  // LDS byte addresses of this lane's fragment of k-step ks inside the A0 / B0 half tile of the CURRENT stage; the other
  // half tiles are immediate offsets, the other stage one XOR per slab (in place: no second register set)
  unsigned oa[4], ob[4];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    oa[ks] = lds0 + PP_A0 + (wm * 64 + l31) * 128 + (((ks * 2 + hi) ^ sw) << 4);
    const int tg = lane >> 4, tj = (lane >> 2) & 3, tc = lane & 3;  // TN: lane group, row of its [4][16] block, 8-byte piece
    ob[ks] = TN ? lds0 + PP_B0 + (ks * 4 + 2 * (tg >> 1)) * 1024 + tj * 256 + (((wn * 2 + (tg & 1)) ^ (2 * tj)) << 5) + tc * 8
                : lds0 + PP_B0 + (wn * 32 + l31) * 128 + (((ks * 2 + hi) ^ sw) << 4);
  }
  typedef __attribute__((address_space(3))) const i32x4_t* lds_v4;
  i32x4_t fa[2][4], fb[4];
  auto rdA = [&](int ks, int off) {
    fa[0][ks] = *(lds_v4)(uintptr_t)(oa[ks] + off);
    fa[1][ks] = *(lds_v4)(uintptr_t)(oa[ks] + off + 4096);
  };
  // TN: the transposing reads are INLINE ASM.  Through the builtin (__builtin_amdgcn_ds_read_tr16_b64_*) the compiler puts
  // `s_waitcnt vmcnt(0)` in front of every group of them - it orders the intrinsic behind ALL pending LDS-DMA, i.e. it
  // drains the half tiles that are meant to stay in flight across the phase (measured: the fc6 dW slab 194 us vs 178 us
  // for the NT form, SQ_WAIT_ANY 0.49 vs 0.36 of the wave cycles, same LDS array cycles and no bank conflicts -
  // profiles/r3_15_pmc_dw*.json).  The compiler therefore does not count them in lgkmcnt: a phase that reads only B
  // fragments waits for them itself (mm<true>: counted waits tied to the fragment registers), and phase 1 issues its B
  // reads FIRST - LDS returns in order, so the compiler's own waits for the A fragments behind them cover them.
  auto rdB = [&](int ks, int off) {
    if constexpr (TN) {
      typedef int i32x2_t __attribute__((ext_vector_type(2)));
      i32x2_t l2, h2;
      const unsigned addr = ob[ks] + off;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(l2) : "v"(addr));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(h2) : "v"(addr));
      fb[ks] = i32x4_t{l2[0], l2[1], h2[0], h2[1]};
    } else {
      fb[ks] = *(lds_v4)(uintptr_t)(ob[ks] + off);
    }
  };
  auto mm = [&](f32x16_t& c0, f32x16_t& c1, auto wait_b_tag) {
    constexpr bool WAIT_B = decltype(wait_b_tag)::value;  // TN, a phase whose only LDS reads are this phase's 8 tr reads
    if (VAR != 3) __builtin_amdgcn_s_setprio(1);
    // (the sched barriers keep each counted wait in front of ITS two MFMAs; left alone the scheduler hoists all four)
    if constexpr (TN && WAIT_B) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fb[0]));
    mma_step<DT>(c0, fb[0], fa[0][0]);  // swapped: D^T[n][m]
    mma_step<DT>(c1, fb[0], fa[1][0]);
    if constexpr (TN && WAIT_B) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fb[1])); }
    mma_step<DT>(c0, fb[1], fa[0][1]);
    mma_step<DT>(c1, fb[1], fa[1][1]);
    if constexpr (TN && WAIT_B) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[2])); }
    mma_step<DT>(c0, fb[2], fa[0][2]);
    mma_step<DT>(c1, fb[2], fa[1][2]);
    if constexpr (TN && WAIT_B) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[3])); }
    mma_step<DT>(c0, fb[3], fa[0][3]);
    mma_step<DT>(c1, fb[3], fa[1][3]);
    if (VAR != 3) __builtin_amdgcn_s_setprio(0);
  };
  constexpr std::false_type NOWAIT{};
  constexpr std::true_type WAITB{};
  // one 1-KB DMA piece: q = 0..7 in staging order A0.0 A0.1 B0.0 B0.1 B1.0 B1.1 A1.0 A1.1
  auto piece = [&](char* stage, auto qtag, int slab) {
    constexpr int q = decltype(qtag)::value;
    constexpr int which = q >> 1, pc = q & 1;
    constexpr int off = which == 0 ? PP_A0 : which == 1 ? PP_B0 : which == 2 ? PP_B1 : PP_A1;
    constexpr bool isA = which == 0 || which == 3;
    constexpr int h = (which == 2 || which == 3) ? 1 : 0;
    char* dst = stage + off + (pc * 64 + wave * 8) * 128;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? ra : rb, (__attribute__((address_space(3))) void*)dst, 16,
                                             isA ? voa[h * 2 + pc] : vob[h * 2 + pc], isA ? slab * 128 : slab * bstep, 0, 0);
  };
#define PP_PIECE(q) piece(nxt, std::integral_constant<int, q>{}, sn)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PP_BARRIER();
  if (wm == 1) PP_BARRIER();  // the lower wave row runs one barrier interval behind the upper one
  auto slab = [&](int s, SgdRegs& sr_ld, const SgdRegs& sr_use) __attribute__((always_inline)) {
    char* nxt = smem + (((s - s0) & 1) ^ 1) * PP_STAGE;
    const int sn = s + 1 < s1 ? s + 1 : s1 - 1;  // past the end: the last slab again, into a stage nobody reads
    // ---- phase 1: quadrant (a0, b0); reads in the order the MFMAs consume them (counted lgkmcnt waits)
    if constexpr (TN) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) rdB(ks, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) rdA(ks, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { rdB(ks, 0); rdA(ks, 0); }
    }
    if (VAR == 2) {  // rebalanced: the phase with 12 fragment reads issues no DMA (pieces 0 / 3 / 2 / 3 per phase)
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // B1 of this slab: A1's two pieces may stay in flight
    } else {
      PP_PIECE(0); PP_PIECE(1);
      if constexpr (SGDP) {
        const unsigned i_ = (unsigned)(s - s0);
        if (i_ < 32) {  // (32 chunks per tile; longer K loops - R > 2048 - carry nothing in their later slabs)
          sgdp_load(*spp, sr_ld, spp->off + i_ * spp->step, spp->goff + i_ * spp->gstep);
          asm volatile("s_waitcnt vmcnt(7)" ::: "memory");  // B1 of this slab (3 optimizer loads younger than the pieces)
        } else {
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
      } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // B1 of this slab has landed (read in phase 2)
      }
    }
    PP_BARRIER();
    mm(acc[0][0], acc[1][0], NOWAIT);
    PP_BARRIER();
    // ---- phase 2: (a0, b1)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rdB(ks, PP_B1 - PP_B0);
    if (VAR == 2) {
      PP_PIECE(0); PP_PIECE(1); PP_PIECE(2);
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // A1 of this slab (phase 3)
    } else {
      PP_PIECE(2); PP_PIECE(3);
      if constexpr (SGDP) {
        if ((unsigned)(s - s0) < 32) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");  // A1 of this slab; P0 P1 L L L P2 P3 may be in flight
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // A1 of this slab (phase 3)
    }
    PP_BARRIER();
    mm(acc[0][1], acc[1][1], WAITB);
    PP_BARRIER();
    // ---- phase 3: (a1, b1)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rdA(ks, PP_A1 - PP_A0);
    if (VAR == 2) { PP_PIECE(3); PP_PIECE(4); } else { PP_PIECE(4); PP_PIECE(5); }
    if constexpr (SGDP) {
      const unsigned i_ = (unsigned)(s - s0);
      // chunk i - 1: its loads went out in phase 1 of the previous slab (18 vector memory operations ago; 15 in slab 1)
      if (i_ >= 1 && i_ <= 32) sgdp_apply_store(*spp, sr_use, spp->off + (i_ - 1) * spp->step);
    }
    PP_BARRIER();
    mm(acc[2][1], acc[3][1], NOWAIT);
    PP_BARRIER();
    // ---- phase 4: (a1, b0); b0 is read again (4 reads in a phase that has none) rather than kept: 16 registers, which
    // decide whether a trunk conv workgroup still fits beside this kernel (DESIGN 'trunk beside the GEMMs')
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rdB(ks, 0);
    if (VAR == 2) { PP_PIECE(5); PP_PIECE(6); PP_PIECE(7); } else { PP_PIECE(6); PP_PIECE(7); }
    if constexpr (SGDP) {
      // A0 and B0 of the next slab; behind them P4 P5 [S S S] P6 P7 may be in flight (no stores yet in slab 0)
      if (s > s0 && s - s0 <= 32) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // A0 and B0 of the next slab (its phase 1)
    PP_BARRIER();
    mm(acc[2][0], acc[3][0], WAITB);
    PP_BARRIER();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { oa[ks] ^= PP_STAGE; ob[ks] ^= PP_STAGE; }
  };
  if constexpr (SGDP) {
    int s = s0;
    for (; s + 1 < s1; s += 2) {  // chunk i lives in A (even slabs) / B (odd slabs): static register sets
      slab(s, srA, srB);
      slab(s + 1, srB, srA);
    }
    if (s < s1) slab(s, srA, srB);  // (odd slab counts: R = 4000 is 63 slabs)
  } else {
    for (int s = s0; s < s1; ++s) slab(s, srA, srA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant tail fetches must land before LDS is reused
  if constexpr (SGDP) {
    // <= 32 slabs: the chunk loaded in phase 1 of the last slab is still due (set A after an even slab index, B after an odd
    // one).  FEWER than 32 slabs (fewer than 2048 proposals - real data): the loop carried one chunk per slab, chunks
    // S .. 31 of the previous tile are left - four in flight per trip here, exposed (the registers of the fragment pipeline
    // are dead at this point, the accumulators are not touched).
    const unsigned S = (unsigned)(s1 - s0);
    if (S <= 32) {
      if (S & 1) sgdp_apply_store(*spp, srA, spp->off + (S - 1) * spp->step);
      else sgdp_apply_store(*spp, srB, spp->off + (S - 1) * spp->step);
    }
    for (unsigned c = S; c < 32; c += 4) {
      SgdRegs q[4];
#pragma unroll
      for (unsigned j = 0; j < 4; ++j) {
        const unsigned cc = c + j < 32 ? c + j : 31;  // (clamped: a duplicate load, never applied)
        sgdp_load(*spp, q[j], spp->off + cc * spp->step, spp->goff + cc * spp->gstep);
      }
#pragma unroll
      for (unsigned j = 0; j < 4; ++j)
        if (c + j < 32) sgdp_apply_store(*spp, q[j], spp->off + (c + j) * spp->step);
    }
  }
  if (wm == 0) PP_BARRIER();
#undef PP_PIECE
}

// ------------------------------------------------------------------------------------------------
// 256x256 tile, 8 waves (2 x 4, 128x64 per wave = 4x2 MFMA 32x32 tiles), LDS-DMA staging.
//   * global -> LDS by `buffer_load_dwordx4 ... lds` (no staging VGPRs, hardware bounds check keeps the free zero
//     fill of ragged edges).  An LDS-DMA wave-instruction writes base + lane*16, i.e. 8 rows x 128 B, so the LDS
//     image is row-major and the XOR swizzle is applied to the per-lane SOURCE k-slot (same involution on the
//     ds_read side): the 8 lanes of a row still read one whole 128-B line.
//   * 2 LDS stages of 64 KB; the next slab's 8 loads per thread are issued before the current slab is consumed and
//     stay in flight across the barrier: counted `s_waitcnt vmcnt(8)` + raw s_barrier (a __syncthreads() would
//     drain them).
//   * operands are swapped in the MFMA (D^T = B.A^T) so a lane holds 4 consecutive output columns per register
//     quad: the epilogue issues 16-B stores (4x fewer store instructions; the fc6 dW output is 411 MB).
// Buffer descriptor of the TN operand's tile: Bt [kb_rows][ldb] from column bn on; rows beyond kb_rows (the K padding) and
// everything behind the matrix read as zeros.  (Columns beyond N inside a row run into the next row: finite values that
// only reach output columns >= N, which are never stored.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_tn_rsrc(const char* B, int bn, int kb_rows, long ldb_bytes, int es) {
  long bytes = (long)kb_rows * ldb_bytes - (long)bn * es;
  if (bytes < 0) bytes = 0;
  return __builtin_amdgcn_make_buffer_rsrc((void*)(B + (long)bn * es), 0, (unsigned)bytes, 0x00020000);
}

template <int DT, bool PIPE, int PP = 0, bool TN = false>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = DT == DRN_BF16 ? 2 : 4;
  constexpr int BM = 256, BN = 256, MI = 4, NJ = 2;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  // (tile, split) are linearised together before the XCD remap: one XCD then owns a contiguous
  // run of tiles of ONE K-split (a 4 x tiles_n patch), so its L2 fetches each operand panel once.
  const int tiles = tiles_m * tiles_n;
  const int logical = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, tiles * gridDim.y);
  const int split = logical / tiles;
  int tm, tn;
  tile_coords(logical - split * tiles, tiles_m, tiles_n, tm, tn, p.gm);
  const int bm = tm * BM, bn = tn * BN;
  const int nslab = p.K * ES / 128;
  const int s0 = split * p.k_slabs_per_split;
  const int s1 = s0 + p.k_slabs_per_split < nslab ? s0 + p.k_slabs_per_split : nslab;
  const RowLoader la = make_row_loader(p.A, bm, p.M, BM, p.lda * ES);
  const RowLoader lb = make_row_loader(p.B, bn, p.N, BN, p.ldb * ES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // per-thread source offsets of its 4 A-chunks and 4 B-chunks (row = i*64 + tid/8, k-slot pre-swizzled)
  unsigned voa[4], vob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned row = i * 64 + (tid >> 3), ks = (tid & 7) ^ ((row >> 1) & 7);
    voa[i] = row * la.ld_bytes + ks * 16;
    vob[i] = row * lb.ld_bytes + ks * 16;
  }
  auto issue = [&](char* stage, int slab) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      char* dst = stage + (i * 64 + wave * 8) * 128;  // wave-uniform LDS base of this 1-KB piece
      __builtin_amdgcn_raw_ptr_buffer_load_lds(la.rsrc, (__attribute__((address_space(3))) void*)dst, 16, voa[i], slab * 128, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(lb.rsrc, (__attribute__((address_space(3))) void*)(dst + A_BYTES), 16, vob[i], slab * 128, 0, 0);
    }
  };
  f32x16_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto load_frags = [&](const char* buf, int ks, i32x4_t (&fa)[MI], i32x4_t (&fb)[NJ]) {
    const int slot = ks * 2 + (lane >> 5);
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = *(const i32x4_t*)(buf + swz(wm * 128 + i * 32 + (lane & 31), slot));
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[j] = *(const i32x4_t*)(buf + A_BYTES + swz(wn * 64 + j * 32 + (lane & 31), slot));
  };
  auto mma_all = [&](const i32x4_t (&fa)[MI], const i32x4_t (&fb)[NJ]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) mma_step<DT>(acc[i][j], fb[j], fa[i]);  // swapped: D^T[n][m]
  };
  if (s0 < s1) {
    if constexpr (PP) {
      unsigned pva[4], pvb[4];
      pp_offsets<TN>(la.ld_bytes, lb.ld_bytes, tid, pva, pvb);
      const unsigned bstep = TN ? 64u * lb.ld_bytes : 128u;
      const __amdgpu_buffer_rsrc_t rbb = TN ? make_tn_rsrc(p.B, bn, p.kb_rows, p.ldb * ES, ES) : lb.rsrc;
      pp_issue_first(smem, la.rsrc, rbb, pva, pvb, wave, s0, bstep);
      pp_mainloop<DT, PP, TN>(acc, smem, la.rsrc, rbb, pva, pvb, s0, s1, lane, wave, bstep);
    } else if constexpr (PIPE) {
      // Software-pipelined schedule: fragments of k-step k+1 are read while the MFMAs of k-step k run (two register
      // sets), and ONE barrier per slab - placed after the slab's last fragment read and before its last MFMA block -
      // serves both as "slab s+1 has landed" and "everybody is done reading slab s"; the DMA of slab s+2 is issued
      // right behind it and has a whole slab of MFMAs to land.
      i32x4_t fa0[MI], fb0[NJ], fa1[MI], fb1[NJ];
      issue(smem, s0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s0 + 1 < s1) issue(smem + STAGE, s0 + 1);
      load_frags(smem, 0, fa0, fb0);
      // each group below = 6 fragment reads for the NEXT k-step + 8 MFMAs of the current one; the scheduling groups
      // interleave them one read per MFMA so a wave's matrix instructions never queue behind its own LDS reads
      // (+5 % on the fc6 shapes).  The last slab is peeled so the steady-state body is one branch-free block, which
      // lets the 4th group also take the next slab's DMA issue (8 buffer_load..lds) under its MFMAs.
#define DRN_INTERLEAVE()                                                     \
  _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                           \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        \
  }                                                                          \
  __builtin_amdgcn_sched_group_barrier(0x008, 2, 0)
      int s = s0;
      for (; s + 1 < s1; ++s) {
        char* cur = smem + ((s - s0) & 1) * STAGE;
        char* nxt = smem + (((s - s0) & 1) ^ 1) * STAGE;
        load_frags(cur, 1, fa1, fb1);
        mma_all(fa0, fb0);
        DRN_INTERLEAVE();
        load_frags(cur, 2, fa0, fb0);
        mma_all(fa1, fb1);
        DRN_INTERLEAVE();
        load_frags(cur, 3, fa1, fb1);
        mma_all(fa0, fb0);
        DRN_INTERLEAVE();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // slab s+2 into the stage everybody just finished reading; past the end the last slab is fetched again into
        // a stage nobody will read (keeps this block branch-free)
        issue(cur, s + 2 < s1 ? s + 2 : s1 - 1);
        load_frags(nxt, 0, fa0, fb0);
        mma_all(fa1, fb1);
#pragma unroll
        for (int q_ = 0; q_ < 6; ++q_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      {  // last slab of this split: nothing left to fetch
        char* cur = smem + ((s - s0) & 1) * STAGE;
        load_frags(cur, 1, fa1, fb1);
        mma_all(fa0, fb0);
        DRN_INTERLEAVE();
        load_frags(cur, 2, fa0, fb0);
        mma_all(fa1, fb1);
        DRN_INTERLEAVE();
        load_frags(cur, 3, fa1, fb1);
        mma_all(fa0, fb0);
        DRN_INTERLEAVE();
        mma_all(fa1, fb1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant tail fetches must land before LDS is reused
      }
#undef DRN_INTERLEAVE
    } else {
      issue(smem, s0);
      for (int s = s0; s < s1; ++s) {
        char* cur = smem + ((s - s0) & 1) * STAGE;
        char* nxt = smem + (((s - s0) & 1) ^ 1) * STAGE;
        if (s + 1 < s1) {
          issue(nxt, s + 1);
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          i32x4_t fa[MI], fb[NJ];
          load_frags(cur, ks, fa, fb);
          mma_all(fa, fb);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  // D^T layout: lane -> m (A row) = lane&31, register r -> n = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* C = p.C + (long)split * p.c_split_stride;
  const bool vec_ok = (p.ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = bm + wm * 128 + i * 32 + (lane & 31);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = bn + (TN ? j * 128 + wn * 32 : wn * 64 + j * 32) + 8 * q + 4 * (lane >> 5);
        float* dst = C + (long)m * p.ldc + n;
        f32x4_t v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        if (p.c_bf16) {
          bf16_t* d16 = (bf16_t*)p.C + (long)m * p.ldc + n;
          if (vec_ok && n + 4 <= p.N) {
            uint2 o;
            o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
            o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
            *(uint2*)d16 = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) d16[e] = f32_to_bf16(v[e]);
          }
        } else if (vec_ok && n + 4 <= p.N) {
          if (p.accumulate) { const f32x4_t o = *(const f32x4_t*)dst; v += o; }
          *(f32x4_t*)dst = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) dst[e] = p.accumulate ? dst[e] + v[e] : v[e];
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// PERSISTENT form of the 256x256 kernel (same tile, same LDS-DMA pipeline, same scheduling groups): the grid is one
// workgroup per CU and every workgroup walks its share of the (tile, K-split) list.
//   * A one-tile workgroup gives its CU back when it retires; with an HBM-streaming kernel (the optimizer pass) on
//     another stream, that kernel's small, long-lived workgroups take the freed registers and the next 128-KB-LDS /
//     400-VGPR GEMM workgroup cannot be placed there any more: the GEMM loses whole CUs ("overlap" measured as time
//     slicing in round 1).  A persistent workgroup keeps its CU; the streaming kernel runs in what is left over
//     (112 VGPRs per SIMD beside two 200-VGPR waves).
//   * The first slab of the NEXT tile is issued (LDS-DMA into stage 0) before the current tile's epilogue stores, so the
//     epilogue and the prologue latency overlap instead of costing a workgroup launch per tile.
// Work order: workgroup (xcd = bid & 7, idx = bid >> 3) takes elements idx, idx + nwg/8, ... of its XCD's contiguous
// chunk of the logical (tile, split) list - the same elements in the same temporal order as the one-tile grid under the
// dispatcher's round-robin, so the L2 reuse pattern of the XCD patch mapping is unchanged.
struct GemmWork { int bm, bn, s0, s1, split; __amdgpu_buffer_rsrc_t ra, rb; };  // one (tile, K-split) work item

// PAIR (round 3): TWO independent GEMMs in one persistent launch - the first `pair_wg0` workgroups of every XCD walk
// problem 0, the others problem 1 (drn_gemm_nt_pair splits them by work).  For two launches that each leave CUs idle:
// the fc7 weight gradient (128 tiles: half the CUs in one round) and the fc7 dX (64 tiles x 4 K-splits of half the
// length), 46 + 41 us one after the other.  (Two concurrent launches on a forked stream do the same on paper and cost
// 250 us in the captured step: the graph executor starts the branch late.)
template <int DT, int PP = 0, bool TN = false, bool PAIR = false, bool SGDP = false>
__global__ __launch_bounds__(512) void gemm_nt256p_kernel(GemmParams p_in, GemmParams p2_in, int pair_wg0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool second = PAIR && (int)(blockIdx.x >> 3) >= pair_wg0;
  const GemmParams p = second ? p2_in : p_in;
  constexpr int ES = DT == DRN_BF16 ? 2 : 4;
  constexpr int BM = 256, BN = 256, MI = 4, NJ = 2;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int tiles = tiles_m * tiles_n;
  const int total = tiles * p.nsplit;
  const int nslab = p.K * ES / 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int bid = blockIdx.x;
  const int per = PAIR ? (second ? (int)(gridDim.x >> 3) - pair_wg0 : pair_wg0) : (int)(gridDim.x >> 3);
  const int xcd = bid & 7, q = total >> 3, r = total & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int len = q + (xcd < r ? 1 : 0);
  int j = PAIR ? (second ? (bid >> 3) - pair_wg0 : (bid >> 3)) : (bid >> 3);
  if (j >= len) return;
  const unsigned lda_b = (unsigned)(p.lda * ES), ldb_b = (unsigned)(p.ldb * ES);
  unsigned voa[4], vob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned row = i * 64 + (tid >> 3), ks = (tid & 7) ^ ((row >> 1) & 7);
    voa[i] = row * lda_b + ks * 16;
    vob[i] = row * ldb_b + ks * 16;
  }
  using Work = GemmWork;
  auto setup = [&](int logical, Work& w) {
    w.split = logical / tiles;
    int tm, tn;
    tile_coords(logical - w.split * tiles, tiles_m, tiles_n, tm, tn, p.gm);
    w.bm = tm * BM;
    w.bn = tn * BN;
    w.s0 = w.split * p.k_slabs_per_split;
    w.s1 = w.s0 + p.k_slabs_per_split < nslab ? w.s0 + p.k_slabs_per_split : nslab;
    w.ra = make_row_loader(p.A, w.bm, p.M, BM, p.lda * ES).rsrc;
    w.rb = TN ? make_tn_rsrc(p.B, w.bn, p.kb_rows, p.ldb * ES, ES) : make_row_loader(p.B, w.bn, p.N, BN, p.ldb * ES).rsrc;
  };
  const unsigned bstep = TN ? 64u * ldb_b : 128u;
  auto issue = [&](const Work& w, char* stage, int slab) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      char* dst = stage + (i * 64 + wave * 8) * 128;  // wave-uniform LDS base of this 1-KB piece
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w.ra, (__attribute__((address_space(3))) void*)dst, 16, voa[i], slab * 128, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w.rb, (__attribute__((address_space(3))) void*)(dst + A_BYTES), 16, vob[i], slab * 128, 0, 0);
    }
  };
  f32x16_t acc[MI][NJ];
  auto load_frags = [&](const char* buf, int ks, i32x4_t (&fa)[MI], i32x4_t (&fb)[NJ]) {
    const int slot = ks * 2 + (lane >> 5);
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = *(const i32x4_t*)(buf + swz(wm * 128 + i * 32 + (lane & 31), slot));
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) fb[jj] = *(const i32x4_t*)(buf + A_BYTES + swz(wn * 64 + jj * 32 + (lane & 31), slot));
  };
  auto mma_all = [&](const i32x4_t (&fa)[MI], const i32x4_t (&fb)[NJ]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) mma_step<DT>(acc[i][jj], fb[jj], fa[i]);  // swapped: D^T[n][m]
  };
  unsigned pva[4], pvb[4];
  if constexpr (PP) pp_offsets<TN>(lda_b, ldb_b, tid, pva, pvb);
  Work cur;
  setup(base + j, cur);
  if (cur.s0 < cur.s1) {
    if constexpr (PP) pp_issue_first(smem, cur.ra, cur.rb, pva, pvb, wave, cur.s0, bstep);
    else issue(cur, smem, cur.s0);
  }
  // SGDP: the optimizer step of the tile finished last rides in the next tile's mainloop (SgdPipe above)
  [[maybe_unused]] SgdPipe sp;
  [[maybe_unused]] bool have_prev = false;
  if constexpr (SGDP) {
    sp.w = p.sgd_w; sp.m = p.sgd_mom; sp.g = (const bf16_t*)p.C; sp.s = p.sgd_shadow;
    sp.step = (unsigned)(8 * p.sgd_ld * 4); sp.gstep = (unsigned)(8 * p.ldc * 2);
    sp.lr = p.sgd_seg->lr; sp.wd = p.sgd_seg->wd; sp.mom = p.sgd_momentum; sp.gs = p.sgd_grad_scale;
    sp.first = p.sgd_first_step;
    sp.off = sp.goff = 0;
  }
  for (;;) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) acc[i][jj][rr] = 0.f;
    const int s0 = cur.s0, s1 = cur.s1;
    if constexpr (PP) {
      if constexpr (SGDP) {
        if (have_prev) pp_mainloop<DT, PP, TN, true>(acc, smem, cur.ra, cur.rb, pva, pvb, s0, s1, lane, wave, bstep, &sp);
        else pp_mainloop<DT, PP, TN>(acc, smem, cur.ra, cur.rb, pva, pvb, s0, s1, lane, wave, bstep);
      } else if (s0 < s1) pp_mainloop<DT, PP, TN>(acc, smem, cur.ra, cur.rb, pva, pvb, s0, s1, lane, wave, bstep);
    } else if (s0 < s1) {
      i32x4_t fa0[MI], fb0[NJ], fa1[MI], fb1[NJ];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // slab s0 (issued ahead of the previous tile's epilogue) has landed
      __builtin_amdgcn_s_barrier();
      if (s0 + 1 < s1) issue(cur, smem + STAGE, s0 + 1);
      load_frags(smem, 0, fa0, fb0);
#define DRN_INTERLEAVE()                                                     \
  _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                           \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        \
  }                                                                          \
  __builtin_amdgcn_sched_group_barrier(0x008, 2, 0)
      int s = s0;
      for (; s + 1 < s1; ++s) {
        char* cs = smem + ((s - s0) & 1) * STAGE;
        char* nx = smem + (((s - s0) & 1) ^ 1) * STAGE;
        load_frags(cs, 1, fa1, fb1);
        mma_all(fa0, fb0);
        DRN_INTERLEAVE();
        load_frags(cs, 2, fa0, fb0);
        mma_all(fa1, fb1);
        DRN_INTERLEAVE();
        load_frags(cs, 3, fa1, fb1);
        mma_all(fa0, fb0);
        DRN_INTERLEAVE();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(cur, cs, s + 2 < s1 ? s + 2 : s1 - 1);
        load_frags(nx, 0, fa0, fb0);
        mma_all(fa1, fb1);
#pragma unroll
        for (int q_ = 0; q_ < 6; ++q_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      {  // last slab of this (tile, split)
        char* cs = smem + ((s - s0) & 1) * STAGE;
        load_frags(cs, 1, fa1, fb1);
        mma_all(fa0, fb0);
        DRN_INTERLEAVE();
        load_frags(cs, 2, fa0, fb0);
        mma_all(fa1, fb1);
        DRN_INTERLEAVE();
        load_frags(cs, 3, fa1, fb1);
        mma_all(fa0, fb0);
        DRN_INTERLEAVE();
        mma_all(fa1, fb1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant tail fetches must land before LDS is reused
      }
#undef DRN_INTERLEAVE
    }
    // ---- next work item: its first slab goes out before this tile's epilogue stores
    j += per;
    const bool more = j < len;
    Work nxt;
    if (more) {
      setup(base + j, nxt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave is done reading both stages
      if (nxt.s0 < nxt.s1) {
        if constexpr (PP) pp_issue_first(smem, nxt.ra, nxt.rb, pva, pvb, wave, nxt.s0, bstep);
        else issue(nxt, smem, nxt.s0);
      }
    }
    const int bm = cur.bm, bn = cur.bn;
    // D^T layout: lane -> m (A row) = lane&31, register r -> n = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (p.c_bf16 && (p.ldc & 7) == 0 && (((uintptr_t)p.C) & 15) == 0) {
      // bf16 output (the fc6 dW gradient bucket: one epilogue per 32 K-slabs, 128 KB per tile): the MFMA layout gives a
      // lane 4 consecutive columns of 32 DIFFERENT rows, i.e. 8-byte stores scattered over 32 lines per instruction.
      // The tile goes through the free LDS stage instead, one 128-row half at a time (64 KB), and leaves as 16-byte
      // pieces of whole 512-byte rows.  8-byte chunk c of row r sits at chunk c ^ ((r & 31) << 1): the 32 rows of a
      // store instruction spread over all banks, and a reader's 16-byte pair stays adjacent.
      // LDS accesses are inline asm: the compiler orders every ds access it can see behind ALL pending LDS-DMA
      // (s_waitcnt vmcnt(0)), i.e. behind the next tile's slab that was just issued into the OTHER stage
      const unsigned ep = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + STAGE);
      bf16_t* C16 = (bf16_t*)p.C;
      // (the ~40 LDS / global offsets below are functions of the thread id alone: without this the compiler hoists them
      // out of the tile loop and keeps them alive through the mainloop - 204 -> 229 VGPRs, which would cost the conv
      // workgroups their place beside this kernel, see DESIGN 'trunk beside the GEMMs')
      int lane_e = lane, tid_e = tid;
      asm volatile("" : "+v"(lane_e), "+v"(tid_e));
      if (!more) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int row = i * 32 + (lane_e & 31);
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
              for (int qq = 0; qq < 4; ++qq) {
                const int c = (TN ? jj * 32 + wn * 8 : wn * 16 + jj * 8) + 2 * qq + (lane_e >> 5);
                const unsigned long long o =
                    (unsigned long long)((uint32_t)f32_to_bf16(acc[i][jj][4 * qq]) | ((uint32_t)f32_to_bf16(acc[i][jj][4 * qq + 1]) << 16)) |
                    ((unsigned long long)((uint32_t)f32_to_bf16(acc[i][jj][4 * qq + 2]) | ((uint32_t)f32_to_bf16(acc[i][jj][4 * qq + 3]) << 16)) << 32);
                asm volatile("ds_write_b64 %0, %1" ::"v"(ep + row * 512 + ((c ^ ((row & 31) << 1)) << 3)), "v"(o) : "memory");
              }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (SGDP && (p.c_bf16 & 2)) {
          // (the fused dW + SGD launch: whole tiles only, and its 256 registers are spent anyway - four reads in flight per
          // trip instead of one read, one wait, one store eight times over)
#pragma unroll 1
          for (int it = 0; it < 8; it += 4) {
            i32x4_t v0, v1, v2, v3;
            const int r0_ = (it * 512 + tid_e) >> 5, pc_ = tid_e & 31;  // rows r0_, r0_ + 16, + 32, + 48: the same (row & 31) parity class shifts by 16
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                         : "v"(ep + (r0_) * 512 + ((pc_ ^ ((r0_) & 31)) << 4)), "v"(ep + (r0_ + 16) * 512 + ((pc_ ^ ((r0_ + 16) & 31)) << 4)),
                           "v"(ep + (r0_ + 32) * 512 + ((pc_ ^ ((r0_ + 32) & 31)) << 4)), "v"(ep + (r0_ + 48) * 512 + ((pc_ ^ ((r0_ + 48) & 31)) << 4))
                         : "memory");
            bf16_t* dst = C16 + (long)(bm + half * 128 + r0_) * p.ldc + bn + pc_ * 8;
            *(i32x4_t*)dst = v0;
            *(i32x4_t*)(dst + 16 * p.ldc) = v1;
            *(i32x4_t*)(dst + 32 * p.ldc) = v2;
            *(i32x4_t*)(dst + 48 * p.ldc) = v3;
          }
        } else
#pragma unroll 1
        for (int it = 0; it < 8; ++it) {
          const int idx = it * 512 + tid_e;
          const int row = idx >> 5, pc = idx & 31;
          const int m = bm + half * 128 + row, n = bn + pc * 8;
          i32x4_t v;
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ep + row * 512 + ((pc ^ (row & 31)) << 4)) : "memory");
          if (m < p.M) {
            bf16_t* dst = C16 + (long)m * p.ldc + n;
            if (n + 8 <= p.N) {
              *(i32x4_t*)dst = v;
            } else {
              const bf16_t* e8 = (const bf16_t*)&v;
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (n + e < p.N) dst[e] = e8[e];
            }
          }
        }
        if (half == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
      }
    } else {
      float* C = p.C + (long)cur.split * p.c_split_stride;
      const bool vec_ok = (p.ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = bm + wm * 128 + i * 32 + (lane & 31);
        if (m >= p.M) continue;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int n = bn + (TN ? jj * 128 + wn * 32 : wn * 64 + jj * 32) + 8 * qq + 4 * (lane >> 5);
            float* dst = C + (long)m * p.ldc + n;
            f32x4_t v = {acc[i][jj][4 * qq], acc[i][jj][4 * qq + 1], acc[i][jj][4 * qq + 2], acc[i][jj][4 * qq + 3]};
            if (p.c_bf16) {
              bf16_t* d16 = (bf16_t*)p.C + (long)m * p.ldc + n;
              if (vec_ok && n + 4 <= p.N) {
                uint2 o;
                o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                *(uint2*)d16 = o;
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (n + e < p.N) d16[e] = f32_to_bf16(v[e]);
              }
            } else if (vec_ok && n + 4 <= p.N) {
              if (p.accumulate) { const f32x4_t o = *(const f32x4_t*)dst; v += o; }
              *(f32x4_t*)dst = v;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < p.N) dst[e] = p.accumulate ? dst[e] + v[e] : v[e];
            }
          }
      }
    }
    if constexpr (SGDP) {  // this tile's update: in the next tile's mainloop, or in the drain below
      sp.off = (unsigned)(((long)(bm + wave) * p.sgd_ld + bn + lane * 4) * 4);
      sp.goff = (unsigned)(((long)(bm + wave) * p.ldc + bn + lane * 4) * 2);
      have_prev = true;
    }
    if (!more) break;
    cur = nxt;
  }
  if constexpr (SGDP) {
    // drain: the update of the workgroup's LAST tile, two chunks (rows) in flight per trip
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's gradient stores of this tile have reached L2
    for (unsigned c = 0; c < 32; c += 2) {
      SgdRegs q0, q1;
      sgdp_load(sp, q0, sp.off + c * sp.step, sp.goff + c * sp.gstep);
      sgdp_load(sp, q1, sp.off + (c + 1) * sp.step, sp.goff + (c + 1) * sp.gstep);
      sgdp_apply_store(sp, q0, sp.off + c * sp.step);
      sgdp_apply_store(sp, q1, sp.off + (c + 1) * sp.step);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolution (stride 1, no padding) of a LARGE map as a GEMM on the 256x256 ping-pong mainloop (round 5): the bf16
// bottleneck 1x1s of the shipped dilated-C5 trunk at a real image size are [15000 x Cin] . [Cout x Cin]^T with Cout = 1024 /
// 2048 - 236 / 472 tiles of 256 x 256 - and ran at 300-640 TFLOP/s on the 128x128 register-staged tile (profiles/r5_12_*).
// A = the NHWC input itself (row = pixel), B = the packed weights; the epilogue is the conv's: per-channel affine (folded
// FrozenBN), shortcut add, ReLU, bf16 store.  The accumulators leave through LDS as fp32, one 128-row half of the tile (128 KB
// = both operand stages, free behind the mainloop) at a time, 16-byte chunk c of row r at c ^ (r & 15): a lane then owns 4
// consecutive channels of one pixel, a wave instruction one whole 512-byte row - 8-byte shortcut loads and 8-byte stores of
// complete rows instead of the MFMA layout's 32 rows x 16 bytes per instruction.
// Same slab order, k-steps and MFMA per output element as the tiled conv kernels: bit-identical to them.
__global__ __launch_bounds__(512) void conv1x1_pp_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Mtot = p.Nb * p.Ho * p.Wo;
  const int tiles_m = (Mtot + 255) / 256, tiles_n = (p.Cout + 255) / 256;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn, 4);
  const int bm = tm * 256, bn = tn * 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const RowLoader la = make_row_loader(p.X, bm, Mtot, 256, (long)p.Cin * 2);
  const RowLoader lb = make_row_loader(p.Wt, bn, p.Cout, 256, p.ldw * 2);
  unsigned pva[4], pvb[4];
  pp_offsets<false>(la.ld_bytes, lb.ld_bytes, tid, pva, pvb);
  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nslab = p.Cin >> 6;
  pp_issue_first(smem, la.rsrc, lb.rsrc, pva, pvb, wave, 0);
  pp_mainloop<DRN_BF16, 1>(acc, smem, la.rsrc, lb.rsrc, pva, pvb, 0, nslab, lane, wave);
  // D^T layout: lane -> m (A row) = lane & 31, register r -> n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int l31 = lane & 31, hi = lane >> 5;
  float* tile = (float*)smem;  // [128][256] fp32, 16-byte chunks XOR-swizzled with the row
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const int nn = bn + lane * 4;  // this lane's 4 channels in the read-back phase
  const bool col_ok = nn < p.Cout;
  f32x4_t sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
  if (col_ok && p.scale) sc = *(const f32x4_t*)(p.scale + nn);
  if (col_ok && p.bias) bi = *(const f32x4_t*)(p.bias + nn);
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    __syncthreads();  // every wave is done with the operand stages / with the previous half
    // shortcut rows of this half, fetched ahead of the staging: wave w owns rows w, w + 8, ... of the half
    u32x2_t rv[16];
    if (p.residual && col_ok) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = bm + half * 128 + q * 8 + wave;
        rv[q] = *(const u32x2_t*)(p.residual + ((long)(m < Mtot ? m : Mtot - 1) * p.ldres + nn) * 2);
      }
    }
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 32 + l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = wn * 16 + j * 8 + 2 * q + hi;  // 16-byte chunk = 4 channels
            const f32x4_t v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *(f32x4_t*)(tile + row * 256 + ((c ^ (row & 15)) << 2)) = v;
          }
      }
    }
    __syncthreads();
    if (col_ok) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = q * 8 + wave, m = bm + half * 128 + row;
        if (m >= Mtot) break;
        const f32x4_t a = *(const f32x4_t*)(tile + row * 256 + ((lane ^ (row & 15)) << 2));
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a[e] * sc[e] + bi[e];
        if (p.residual) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            v[2 * e] += __builtin_bit_cast(float, rv[q][e] << 16) * p.res_mult;
            v[2 * e + 1] += __builtin_bit_cast(float, rv[q][e] & 0xffff0000u) * p.res_mult;
          }
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        u32x2_t o;
        o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        *(u32x2_t*)(p.Y + ((long)m * p.ldy + nn) * 2) = o;
      }
    }
  }
}

static int g_conv_pp = 1;  // drn_tune(DRN_TUNE_CONV_PP = 24): 0 = never run a 1x1 conv on the 256x256 ping-pong GEMM mainloop

// Epilogue of the tiled conv kernels: per-channel affine (+ residual, ReLU) and the store.  `tid` = 0..255 within the
// four waves that own the accumulators; `active` = false for waves that only take part in the barriers (the second
// K-group of conv_nhwc_k2_kernel, whose partial sums were already added in).
// 8 consecutive channels of one pixel, packed: bf16 = 16 bytes, fp8 = 8 bytes (the vector epilogues below)
__device__ __forceinline__ bool vec8_ok(int dt, const void* ptr, long ld) {
  return dt == DRN_BF16 ? ((ld & 7) == 0 && (((uintptr_t)ptr) & 15) == 0)
                        : dt == DRN_FP8 ? ((ld & 7) == 0 && (((uintptr_t)ptr) & 7) == 0) : false;
}
__device__ __forceinline__ i32x4_t load8(int dt, const char* base, long elem) {
  if (dt == DRN_BF16) return *(const i32x4_t*)(base + elem * 2);
  const uint2 v = *(const uint2*)(base + elem);
  return i32x4_t{(int)v.x, (int)v.y, 0, 0};
}
__device__ __forceinline__ void add8(int dt, const i32x4_t& rv, float mult, float (&v)[8]) {
  if (dt == DRN_BF16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t w = (uint32_t)rv[e];
      v[2 * e] += __builtin_bit_cast(float, w << 16) * mult;
      v[2 * e + 1] += __builtin_bit_cast(float, w & 0xffff0000u) * mult;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      v[4 * e] += __builtin_amdgcn_cvt_f32_fp8(rv[e], 0) * mult;
      v[4 * e + 1] += __builtin_amdgcn_cvt_f32_fp8(rv[e], 1) * mult;
      v[4 * e + 2] += __builtin_amdgcn_cvt_f32_fp8(rv[e], 2) * mult;
      v[4 * e + 3] += __builtin_amdgcn_cvt_f32_fp8(rv[e], 3) * mult;
    }
  }
}
__device__ __forceinline__ void store8(int dt, char* base, long elem, const float (&v)[8]) {
  if (dt == DRN_BF16) {
    i32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (int)((uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16));
    *(i32x4_t*)(base + elem * 2) = o;
  } else {  // saturating RNE like f32_to_fp8, two values per v_cvt_pk_fp8_f32
    float c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = fminf(fmaxf(v[e], -448.f), 448.f);
    uint2 o;
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
    o.x = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w, true);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], 0, false);
    o.y = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], w, true);
    *(uint2*)(base + elem) = o;
  }
}

template <int DT, int BM, int BN>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16_t (&acc)[BM / 64][BN / 64], char* smem, int bm,
                                              int bn, int Mtot, int tid, bool active) {
  constexpr int MI = BM / 64, NJ = BN / 64;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // bf16 output (+ bf16 residual): the affine result goes through LDS as fp32 [BM][BN] (exactly the mainloop's LDS, now
  // free) so that every lane then owns 8 consecutive channels of a pixel - 16-byte residual loads and 16-byte stores,
  // whole 128/256-byte row segments per instruction.  The MFMA layout gives a lane ONE channel of 16 * MI pixels: written
  // straight from the accumulators that is 2-byte accesses, which ran the write-heavy 1x1 convs of large maps at
  // ~1 TB/s (res2 conv3 at 200 x 304: 72 us for 70 MB; tools/conv_bench.py).  Same arithmetic, same rounding.
  // (round 3: fp8 outputs / residuals take the same path with 8-byte accesses - written from the accumulators an fp8
  // layer stores single BYTES)
  const bool vec_epi = (p.Cout & 7) == 0 && vec8_ok(p.out_dt, p.Y, p.ldy) &&
                       (!p.residual || vec8_ok(p.res_dt, p.residual, p.ldres));
  if (vec_epi) {
    float* tile = (float*)smem;
    __syncthreads();  // the last slab's LDS reads are done
    if (active) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int nl = wn * (BN / 2) + j * 32 + (lane & 31), n = bn + nl;
        const float sc = (n < p.Cout && p.scale) ? p.scale[n] : 1.f;
        const float bi = (n < p.Cout && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ml = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            tile[ml * BN + nl] = acc[i][j][r] * sc + bi;
          }
      }
    }
    __syncthreads();
    if (!active) return;
    constexpr int LPR = BN / 8, RPP = 256 / LPR, NR = BM / RPP;  // lanes per row, rows per pass, rows per lane
    const int cl = tid % LPR, rl = tid / LPR;
    const int n = bn + cl * 8;
    if (n >= p.Cout) return;
    i32x4_t rv[NR];
    if (p.residual) {
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const int m = bm + rl + q * RPP;
        rv[q] = load8(p.res_dt, p.residual, (long)(m < Mtot ? m : Mtot - 1) * p.ldres + n);
      }
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int ml = rl + q * RPP, m = bm + ml;
      if (m >= Mtot) break;
      const f32x4_t lo = *(const f32x4_t*)(tile + ml * BN + cl * 8), hi = *(const f32x4_t*)(tile + ml * BN + cl * 8 + 4);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (p.residual) add8(p.res_dt, rv[q], p.res_mult, v);
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      store8(p.out_dt, p.Y, (long)m * p.ldy + n, v);
    }
    return;
  }
  if (!active) return;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = bn + wn * (BN / 2) + j * 32 + (lane & 31);
    if (n >= p.Cout) continue;
    const float sc = p.scale ? p.scale[n] : 1.f;
    const float bi = p.bias ? p.bias[n] : 0.f;
    // every residual value of this lane's column is fetched BEFORE the first store: Y and the residual may alias as far
    // as the compiler knows, so a load placed behind a store stays there and each of the 16 * MI row steps would pay a
    // full memory latency (measured: 36 us for the K = 256 res4 conv3 at 50 x 76, most of it here)
    float rres[MI][16];
    if (p.residual) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = bm + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const long ri = (long)(m < Mtot ? m : Mtot - 1) * p.ldres + n;
          rres[i][r] = p.res_dt == DRN_BF16 ? bf16_to_f32(((const bf16_t*)p.residual)[ri])
                       : p.res_dt == DRN_FP8 ? fp8_to_f32(((const uint8_t*)p.residual)[ri])
                                             : ((const float*)p.residual)[ri];
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = bm + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < Mtot) {
          float v = acc[i][j][r] * sc + bi;
          if (p.residual) v += rres[i][r] * p.res_mult;
          if (p.relu) v = fmaxf(v, 0.f);
          const long yi = (long)m * p.ldy + n;
          if (p.out_dt == DRN_BF16) ((bf16_t*)p.Y)[yi] = f32_to_bf16(v);
          else if (p.out_dt == DRN_FP8) ((uint8_t*)p.Y)[yi] = f32_to_fp8(v);
          else ((float*)p.Y)[yi] = v;
        }
      }
  }
}

template <int DT, int BM, int BN, bool K64 = true>
__global__ __launch_bounds__(256, (DT == DRN_FP8 && K64 && BM == 128) ? 2 : 1) void conv_nhwc_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = EsOf<DT>::value;
  constexpr int MI = BM / 64, NJ = BN / 64;
  const int Mtot = p.Nb * p.Ho * p.Wo;
  const int tiles_m = (Mtot + BM - 1) / BM, tiles_n = (p.Cout + BN - 1) / BN;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn);
  const int bm = tm * BM, bn = tn * BN;
  ConvLoader<DT, BM> la;
  la.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (unsigned)((long)p.Nb * p.H * p.W * p.Cin * ES), 0x00020000);
  la.H = p.H; la.W = p.W; la.Cin = p.Cin; la.KW = p.KW; la.dil = p.dil; la.ntaps = p.KH * p.KW;
  const int r0 = threadIdx.x >> 3;
#pragma unroll
  for (int q = 0; q < BM / 32; ++q) {
    const int m = bm + r0 + 32 * q;
    if (m < Mtot) {
      const int n = m / (p.Ho * p.Wo), rem = m - n * p.Ho * p.Wo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      la.hi0[q] = ho * p.stride - p.pad;
      la.wi0[q] = wo * p.stride - p.pad;
      la.nbase[q] = (unsigned)((long)n * p.H * p.W * p.Cin * ES);
    } else {
      la.hi0[q] = 0; la.wi0[q] = 0; la.nbase[q] = ConvLoader<DT, BM>::OOB;
    }
  }
  const RowLoader lb = make_row_loader(p.Wt, bn, p.Cout, BN, p.ldw * ES);
  f32x16_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nslab = (p.Ktot * ES + 127) / 128;
  mainloop<DT, BM, BN, decltype(la), decltype(lb), (BM == 64 && BN == 64) ? 1 : 2, K64>(acc, smem, la, lb, 0, nslab);
  conv_epilogue<DT, BM, BN>(p, acc, smem, bm, bn, Mtot, threadIdx.x, true);
}

// 3x3 / stride 1 / 64 -> 64 channels on LARGE maps (stem.conv2/3 and the res2 3x3s of a real-size image: 60k-240k pixels).
// The im2col tiling fetches every input pixel nine times - 128-row tiles x 9 slabs x 16 KB + the weights again per tile:
// 420 MB through L2 for the 31-MB stem layer at 800x1216, which bounds it at ~52 us = 8 TB/s of L2 traffic.  Here a
// workgroup owns an 8 x 32 block of output pixels of one image: the 10 x 34 x 64-channel input patch (43.5 KB) and ALL
// weights (9 taps x 64 x 64 = 73.7 KB) are brought into LDS ONCE (zero padding = out-of-range buffer offsets), then the
// whole K loop - 9 taps x 4 k-steps x (2 x 2) MFMAs per wave - runs out of LDS with no barrier and no global load: the
// A fragment of tap (dy, dx) is the patch row of pixel (y + dy, x + dx), 16 bytes per lane, swizzled like every other
// 128-byte-row LDS image here.  Same k order and MFMA as the tiled kernels (tap-major, channels ascending).
constexpr int P3_TH = 8, P3_TW = 32, P3_PH = P3_TH + 2, P3_PW = P3_TW + 2;
constexpr int P3_PATCH = P3_PH * P3_PW * 128, P3_WTS = 9 * 64 * 128;
constexpr int P3_NCH = P3_PH * P3_PW * 8, P3_NIT = (P3_NCH + 255) / 256;  // 16-byte chunks of a patch, per-thread share
// PERSISTENT: one workgroup per CU keeps the weights in LDS and walks its XCD's share of the pixel blocks; the patch of
// the NEXT block is fetched into registers while the current one is multiplied, so per block only the LDS store of the
// patch, the MFMA phase and the epilogue (two 128-pixel halves staged as fp32 in the dead patch area) remain.
// PW = true (round 4, the tail of a res2 bottleneck as ONE kernel - resnet_ws.py:217-237 conv2 -> conv3 + shortcut): the
// 3x3's output tile [256 pixels x 64 channels] (affine, ReLU, rounded to bf16 exactly as the 3x3 alone would store it) goes
// to the dead patch area in A-fragment layout instead of to memory, and each wave multiplies its 64 pixels with the 1x1's
// weights [256 x 64] (32 KB more of LDS, resident like the 3x3's) - four chunks of 64 output channels, each through the
// same staged epilogue (affine, residual, ReLU, 16-byte stores).  Same MFMA and k order as the two kernels it replaces.
constexpr int P3_PW_WTS = 256 * 128;
// LDS-only barrier: __syncthreads() also waits (vmcnt) for every global store and prefetch load in flight - the block loop
// below used to drain its output stores and the next block's patch fetch at each of its barriers
#define LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
template <bool RES, bool PW = false, bool POOL = false>
__global__ __launch_bounds__(256) void conv3x3_c64_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  char* wts = smem + P3_PATCH;
  [[maybe_unused]] char* w3s = smem + P3_PATCH + P3_WTS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (p.Wo + P3_TW - 1) / P3_TW, tiles_y = (p.Ho + P3_TH - 1) / P3_TH;
  const int per_img = tiles_y * tiles_x, total = p.Nb * per_img;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (unsigned)((long)p.Nb * p.H * p.W * 128), 0x00020000);
  // this thread's 16-byte chunks of a patch: position inside the patch and byte offset relative to the block's first output
  // pixel, once per workgroup (26 VALU instructions per load otherwise - divisions, 64-bit products - 10 % of a block's cycles
  // with one wave per SIMD); the buffer offsets are 32-bit (the launcher checks the tensor is < 4 GB)
  int pyx[P3_NIT], poff[P3_NIT];
#pragma unroll
  for (int it = 0; it < P3_NIT; ++it) {
    const int c = it * 256 + tid, pix = c >> 3, ch = c & 7;
    const int py = pix / P3_PW, px = pix - py * P3_PW;
    pyx[it] = c < P3_NCH ? (py | (px << 8)) : -1;
    poff[it] = ((py - 1) * p.W + (px - 1)) * 128 + ch * 16;
  }
  auto fetch = [&](int t, i32x4_t (&v)[P3_NIT]) {  // block t's patch -> registers (zero padding = out-of-range offsets)
    const int n = t / per_img, tt = t - n * per_img;
    const int y0 = (tt / tiles_x) * P3_TH, x0 = (tt % tiles_x) * P3_TW;
    const int base = ((n * p.H + y0) * p.W + x0) * 128;
#pragma unroll
    for (int it = 0; it < P3_NIT; ++it) {
      const int iy = y0 - 1 + (pyx[it] & 0xff), ix = x0 - 1 + ((pyx[it] >> 8) & 0xff);
      const bool ok = pyx[it] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const unsigned off = ok ? (unsigned)(base + poff[it]) : 0xFFFFFFF0u;
      v[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    }
  };
  int k = blockIdx.x;
  if (k >= total) return;
  i32x4_t pv[P3_NIT];
  fetch(xcd_remap(k, total), pv);
  // weights: [64 cout][ldw] with k = tap * 64 + ci -> LDS [tap][cout][128 B], once per workgroup
  {
    const char* wg = p.Wt;
    const long ldw_b = p.ldw * 2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      i32x4_t v[9];
#pragma unroll
      for (int it = 0; it < 9; ++it) {
        const int c = (half * 9 + it) * 256 + tid;  // 0 .. 4607
        const int tap = c >> 9, co = (c >> 3) & 63, ch = c & 7;
        v[it] = *(const i32x4_t*)(wg + co * ldw_b + tap * 128 + ch * 16);
      }
#pragma unroll
      for (int it = 0; it < 9; ++it) {
        const int c = (half * 9 + it) * 256 + tid;
        const int tap = c >> 9, co = (c >> 3) & 63, ch = c & 7;
        *(i32x4_t*)(wts + tap * (64 * 128) + swz(co, ch)) = v[it];
      }
    }
  }
  if constexpr (PW) {  // the 1x1's weights: [256 cout][pw_ldw] (64 input channels = one 128-byte row) -> LDS [cout][128 B]
    const long ldw3_b = p.pw_ldw * 2;
    i32x4_t v[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c = it * 256 + tid;  // 0 .. 2047
      v[it] = *(const i32x4_t*)(p.pw_w + (long)(c >> 3) * ldw3_b + (c & 7) * 16);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c = it * 256 + tid;
      *(i32x4_t*)(w3s + swz(c >> 3, c & 7)) = v[it];
    }
  }
  const int lx = lane & 31, lh = lane >> 5;
  [[maybe_unused]] float sc3[4][2], bi3[4][2];
  if constexpr (PW) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        sc3[cb][j] = p.pw_scale ? p.pw_scale[cb * 64 + j * 32 + lx] : 1.f;
        bi3[cb][j] = p.pw_bias ? p.pw_bias[cb * 64 + j * 32 + lx] : 0.f;
      }
  }
  // this lane's two output channels' affine, once per workgroup (inside the block loop each half of every epilogue paid
  // a global-load latency for them: one wave per SIMD hides nothing - 16 of 38 us on the stem layer at 800x1216)
  float sc2[2], bi2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sc2[j] = p.scale ? p.scale[j * 32 + lx] : 1.f;
    bi2[j] = p.bias ? p.bias[j * 32 + lx] : 0.f;
  }
  for (; k < total; k += gridDim.x) {
    const int t = xcd_remap(k, total);
    const int n = t / per_img, tt = t - n * per_img;
    const int y0 = (tt / tiles_x) * P3_TH, x0 = (tt % tiles_x) * P3_TW;
#pragma unroll
    for (int it = 0; it < P3_NIT; ++it) {
      const int c = it * 256 + tid;
      if (c < P3_NCH) *(i32x4_t*)(patch + swz(c >> 3, c & 7)) = pv[it];
    }
    LDS_BARRIER();
    if (k + (int)gridDim.x < total) fetch(xcd_remap(k + gridDim.x, total), pv);  // lands under the MFMA phase
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // 36 k-steps (9 taps x 4), software-pipelined by hand: one wave per SIMD has nobody to hide its LDS latency behind,
    // so the fragments of step s + 1 are requested before the four MFMAs of step s are issued
    i32x4_t fa[2][2], fb[2][2];
    auto frags = [&](int s_, i32x4_t (&a)[2], i32x4_t (&b)[2]) {
      const int tap = s_ >> 2, ks = s_ & 3, dy = tap / 3, dx = tap - dy * 3, slot = ks * 2 + lh;
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *(const i32x4_t*)(patch + swz((wave * 2 + i + dy) * P3_PW + lx + dx, slot));
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *(const i32x4_t*)(wts + tap * (64 * 128) + swz(j * 32 + lx, slot));
    };
    frags(0, fa[0], fb[0]);
#pragma unroll
    for (int s_ = 0; s_ < 36; ++s_) {
      if (s_ + 1 < 36) frags(s_ + 1, fa[(s_ + 1) & 1], fb[(s_ + 1) & 1]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma_step<DRN_BF16>(acc[i][j], fa[s_ & 1][i], fb[s_ & 1][j]);
      // one k-step's reads, then one k-step's MFMAs: keep the compiler from sinking the reads to their first use
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    // epilogue: affine -> fp32 [128 pixels][64 channels] in the (dead) patch area, one half of the block at a time -> 8
    // channels per lane: residual, ReLU, 16-byte stores
    // WAVE-PRIVATE and barrier-free (cycle counters on the first version - fp32 halves of the whole block staged by all
    // four waves, two barriers per half - put 38 % of a block's time in the epilogue, one wave per SIMD hiding nothing): a wave
    // stages one of its two 32-pixel output rows at a time as fp32 [32 pixels][64 channels] in its own 8.5 KB (row pitch 68
    // floats: the two half-waves, four pixels apart, land on different banks), reads it back as 32-byte runs - LDS operations
    // of one wave execute in order, so its own writes and reads need no barrier - adds the residual, ReLU, and stores 16-byte
    // pieces of its own pixels' 128-byte rows.  The four waves run their epilogues independently.
    constexpr int EP_PITCH = 68;
    float* wtile = (float*)(patch + wave * (32 * EP_PITCH * 4));
    auto epilogue = [&](f32x16_t (&ac)[2][2], const float (&sc_)[2], const float (&bi_)[2], int ch0, int relu_) {
    [[maybe_unused]] int keep[4][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y = y0 + wave * 2 + i;
      i32x4_t rv[4];
      if constexpr (RES) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int idx = it * 64 + lane, ml = idx >> 3, cg = idx & 7;
          const int yc = min(y, p.Ho - 1), xc = min(x0 + ml, p.Wo - 1);
          rv[it] = *(const i32x4_t*)((const bf16_t*)p.residual + (((long)n * p.Ho + yc) * p.Wo + xc) * p.ldres + ch0 + cg * 8);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          wtile[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP_PITCH + j * 32 + lx] = ac[i][j][r] * sc_[j] + bi_[j];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * 64 + lane, ml = idx >> 3, cg = idx & 7;
        const int x = x0 + ml;
        const long m = ((long)n * p.Ho + y) * p.Wo + x;
        const f32x4_t lo = *(const f32x4_t*)(wtile + ml * EP_PITCH + cg * 8), hi = *(const f32x4_t*)(wtile + ml * EP_PITCH + cg * 8 + 4);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        i32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (RES) {
            const uint32_t w = (uint32_t)rv[it][e];
            v[2 * e] += __builtin_bit_cast(float, w << 16) * p.res_mult;
            v[2 * e + 1] += __builtin_bit_cast(float, w & 0xffff0000u) * p.res_mult;
          }
          if (relu_) { v[2 * e] = fmaxf(v[2 * e], 0.f); v[2 * e + 1] = fmaxf(v[2 * e + 1], 0.f); }
          o[e] = (int)((uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16));
        }
        if constexpr (!POOL) {
          if (y < p.Ho && x < p.Wo) *(i32x4_t*)((bf16_t*)p.Y + m * p.ldy + ch0 + cg * 8) = o;
        } else {
          // 2 x 2 / stride-2 maximum of the ReLU'd, bf16-rounded outputs (non-negative: their bit patterns order like
          // integers): the wave's two output rows are one pooled row - vertical partner = the same thread's piece of row
          // i = 0, horizontal partner (pixel ml ^ 1) = lane ^ 8
          if (i == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) keep[it][e] = o[e];
          } else {
            typedef short s16x2p_t __attribute__((ext_vector_type(2)));
            auto pkmax = [](int a, int b) {
              return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(s16x2p_t, a), __builtin_bit_cast(s16x2p_t, b)));
            };
            int pe[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int mx = pkmax(keep[it][e], o[e]);
              pe[e] = pkmax(mx, __shfl_xor(mx, 8, 64));
            }
            const i32x4_t po = {pe[0], pe[1], pe[2], pe[3]};
            const int Hp = (p.Ho - 2) / 2 + 1, Wp = (p.Wo - 2) / 2 + 1;
            const int yp = (y0 >> 1) + wave, xp = (x0 + ml) >> 1;
            if ((ml & 1) == 0 && yp < Hp && xp < Wp)
              *(i32x4_t*)((bf16_t*)p.Y + (((long)n * Hp + yp) * Wp + xp) * p.ldy + ch0 + cg * 8) = po;
          }
        }
      }
    }
    };
    if constexpr (!PW) {
      LDS_BARRIER();  // every wave is done reading the patch: the staging rows reuse its space
      epilogue(acc, sc2, bi2, 0, p.relu);
    } else {
      // the 3x3's output -> bf16 A fragments: rows = this wave's 64 pixels (wave * 64 + i * 32 + pixel), 128 B = 64 channels
      LDS_BARRIER();  // every wave is done reading the patch
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ml = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, nl = j * 32 + lx;
            float v = acc[i][j][r] * sc2[j] + bi2[j];
            if (p.relu) v = fmaxf(v, 0.f);
            *(bf16_t*)(patch + swz(ml, nl >> 3) + (nl & 7) * 2) = f32_to_bf16(v);
          }
      i32x4_t ya[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ya[i][ks] = *(const i32x4_t*)(patch + swz(wave * 64 + i * 32 + lx, ks * 2 + lh));
      LDS_BARRIER();  // every wave has its fragments: the staging rows below overlap the other waves' y2 rows
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {  // (unrolled: sc3 / bi3 are register arrays)
        f32x16_t a3[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) a3[i][j][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          i32x4_t b3[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) b3[j] = *(const i32x4_t*)(w3s + swz(cb * 64 + j * 32 + lx, ks * 2 + lh));
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma_step<DRN_BF16>(a3[i][j], ya[i][ks], b3[j]);
        }
        epilogue(a3, sc3[cb], bi3[cb], cb * 64, p.pw_relu);
      }
    }
    LDS_BARRIER();  // tile reads done before the next block's patch overwrites the area
  }
}

#undef LDS_BARRIER
// Mid-size layers (the res3 / res4 convs of a real-size image: a few hundred 64x64 tiles, 8 .. 72 K-slabs): one 64x64 tile
// per CU on four waves is latency-bound (~0.5 us per slab), the 32x32 wave-K-split kernel moves 2x the operand bytes per
// flop and runs into the L2 bandwidth (res4 3x3 on 3800 pixels: 952 workgroups x 36 slabs x 8 KB = 274 MB in 16 us).
// Here TWO groups of four waves share one 64x64 tile: group g multiplies the K-slabs [g * n/2, (g + 1) * n/2) out of
// its own 16-KB LDS stage with its own register ring - twice the loads in flight and twice the MFMA issue per tile, the
// operand bytes of the 64x64 tiling - and group 1's partial tile is added to group 0's through LDS (fixed order) in
// front of the common epilogue.  Needs an even slab count; chosen on ONE image's geometry like the other variants.
template <int DT, bool K64 = true>
__global__ __launch_bounds__(512) void conv_nhwc_k2_kernel(ConvParams p) {  // (forcing 128 VGPRs for two workgroups per CU: 8 B of scratch, res4 layers 3 % slower, not kept)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x 16 KB
  constexpr int ES = EsOf<DT>::value;
  const int Mtot = p.Nb * p.Ho * p.Wo;
  const int tiles_m = (Mtot + 63) / 64, tiles_n = (p.Cout + 63) / 64;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn);
  const int bm = tm * 64, bn = tn * 64;
  const int tid = threadIdx.x & 255, g = threadIdx.x >> 8;
  ConvLoader<DT, 64> la;
  la.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (unsigned)((long)p.Nb * p.H * p.W * p.Cin * ES), 0x00020000);
  la.H = p.H; la.W = p.W; la.Cin = p.Cin; la.KW = p.KW; la.dil = p.dil; la.ntaps = p.KH * p.KW;
  const int r0 = tid >> 3;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int m = bm + r0 + 32 * q;
    if (m < Mtot) {
      const int n = m / (p.Ho * p.Wo), rem = m - n * p.Ho * p.Wo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      la.hi0[q] = ho * p.stride - p.pad;
      la.wi0[q] = wo * p.stride - p.pad;
      la.nbase[q] = (unsigned)((long)n * p.H * p.W * p.Cin * ES);
    } else {
      la.hi0[q] = 0; la.wi0[q] = 0; la.nbase[q] = ConvLoader<DT, 64>::OOB;
    }
  }
  const RowLoader lb = make_row_loader(p.Wt, bn, p.Cout, 64, p.ldw * ES);
  f32x16_t acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  const int half = ((p.Ktot * ES + 127) / 128) >> 1;  // launcher: the slab count is even
  mainloop<DT, 64, 64, decltype(la), decltype(lb), 1, K64>(acc, smem + g * (128 * 128), la, lb, g * half, (g + 1) * half,
                                                           tid);
  // group 1's partial tile -> LDS (its own stage: nobody reads it any more), added by group 0 lane for lane
  const int lane = tid & 63, wave = tid >> 6;
  float* part = (float*)(smem + 128 * 128) + wave * 1024 + lane;
  if (g == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r * 64] = acc[0][0][r];
  }
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += part[r * 64];
  }
  conv_epilogue<DT, 64, 64>(p, acc, smem, bm, bn, Mtot, tid, g == 0);
}

// Small-map convolution: 32 x 32 output tile per workgroup, the K range of every 128-byte slab split over the FOUR WAVES
// (wave w multiplies k-step w: one MFMA and two LDS fragment reads per wave and slab), their accumulators added in wave
// order through LDS at the end.  For the res3 / res4 layers of a 224 x 224 image (196 .. 784 pixels) the 64 x 64 kernel
// runs on 16 .. 26 workgroups and its per-slab chain - 8 fragment reads, 4 dependent MFMAs, two barriers - costs ~0.55 us
// whatever the prefetch depth (measured: ring depth 3 / 4 / 5 and division-free addressing all within 5 %), so a 36-slab
// 3 x 3 conv took 23 us on 196 pixels.  Here the chain per slab is a quarter of that and there are 4x the workgroups.
// (Splitting K over workgroups instead - partial tiles in HBM, last arriver reduces - was built and measured slower
// than no split: the cross-XCD coherence of the partials costs more than the split saves; HISTORY.md section 5.)
template <int DT, bool K64 = true>
__global__ __launch_bounds__(256) void conv_nhwc_ks_kernel(ConvParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 32 * 32 * 4];  // one 8-KB operand stage, then 4 partial tiles
  constexpr int ES = EsOf<DT>::value;
  constexpr int DEPTH = 4;
  const int Mtot = p.Nb * p.Ho * p.Wo;
  const int tiles_m = (Mtot + 31) / 32, tiles_n = (p.Cout + 31) / 32;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn);
  const int bm = tm * 32, bn = tn * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ConvLoader<DT, 32> la;
  la.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (unsigned)((long)p.Nb * p.H * p.W * p.Cin * ES), 0x00020000);
  la.H = p.H; la.W = p.W; la.Cin = p.Cin; la.KW = p.KW; la.dil = p.dil; la.ntaps = p.KH * p.KW;
  {
    const int m = bm + (tid >> 3);
    if (m < Mtot) {
      const int n = m / (p.Ho * p.Wo), rem = m - n * p.Ho * p.Wo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      la.hi0[0] = ho * p.stride - p.pad;
      la.wi0[0] = wo * p.stride - p.pad;
      la.nbase[0] = (unsigned)((long)n * p.H * p.W * p.Cin * ES);
    } else {
      la.hi0[0] = 0; la.wi0[0] = 0; la.nbase[0] = ConvLoader<DT, 32>::OOB;
    }
  }
  const RowLoader lb = make_row_loader(p.Wt, bn, p.Cout, 32, p.ldw * ES);
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int n = (p.Ktot * ES + 127) / 128;
  i32x4_t ra[DEPTH][1], rb[DEPTH][1];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < n) {
      la.template load<32>(ra[d], d, tid);
      lb.template load<32>(rb[d], d, tid);
    }
  lds_store_tile<32>(smem, ra[0], tid);
  lds_store_tile<32>(smem + 32 * 128, rb[0], tid);
  __syncthreads();
  const int slot = wave * 2 + (lane >> 5);
  auto slab = [&](auto dtag, int i, auto fulltag) {  // steady state / bounds-checked tail: see mainloop()
    constexpr int d = decltype(dtag)::value;
    constexpr bool FULL = decltype(fulltag)::value;
    if (FULL || i + DEPTH < n) {
      la.template load<32>(ra[d], i + DEPTH, tid);
      lb.template load<32>(rb[d], i + DEPTH, tid);
    }
    if constexpr (DT == DRN_FP8 && K64) {
      // K = 64 scaled MFMA: waves 0 / 1 multiply the two halves of the slab (k-steps 2w, 2w + 1), waves 2 / 3 add zeros
      if (wave < 2) {
        const int s2 = wave * 4 + (lane >> 5);
        const i32x4_t fa0 = *(const i32x4_t*)(smem + swz(lane & 31, s2)), fa1 = *(const i32x4_t*)(smem + swz(lane & 31, s2 + 2));
        const i32x4_t fb0 = *(const i32x4_t*)(smem + 32 * 128 + swz(lane & 31, s2));
        const i32x4_t fb1 = *(const i32x4_t*)(smem + 32 * 128 + swz(lane & 31, s2 + 2));
        mma_step64_fp8(acc, fa0, fa1, fb0, fb1);
      }
    } else {
      const i32x4_t fa = *(const i32x4_t*)(smem + swz(lane & 31, slot));
      const i32x4_t fb = *(const i32x4_t*)(smem + 32 * 128 + swz(lane & 31, slot));
      mma_step<DT>(acc, fa, fb);
    }
    __syncthreads();
    if (FULL || i + 1 < n) {
      lds_store_tile<32>(smem, ra[(d + 1) % DEPTH], tid);
      lds_store_tile<32>(smem + 32 * 128, rb[(d + 1) % DEPTH], tid);
    }
    __syncthreads();
  };
  int base = 0;
  for (; base + 2 * DEPTH <= n; base += DEPTH)
    static_for<DEPTH>([&](auto dtag) { slab(dtag, base + decltype(dtag)::value, std::true_type{}); });
  for (; base < n; base += DEPTH)
    static_for<DEPTH>([&](auto dtag) {
      const int i = base + decltype(dtag)::value;
      if (i < n) slab(dtag, i, std::false_type{});
    });
  // the four k-step partials, as [wave][row][col] fp32
  float* part = (float*)smem;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ml = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    part[(wave * 32 + ml) * 32 + (lane & 31)] = acc[r];
  }
  __syncthreads();
  const bool vec_epi = (p.Cout & 7) == 0 && vec8_ok(p.out_dt, p.Y, p.ldy) &&
                       (!p.residual || vec8_ok(p.res_dt, p.residual, p.ldres));
  if (vec_epi) {  // 128 lanes: one pixel row x 8 channels each (see conv_nhwc_kernel)
    if (tid >= 128) return;
    const int ml = tid >> 2, cg = tid & 3, m = bm + ml, nn = bn + cg * 8;
    if (m >= Mtot || nn >= p.Cout) return;
    i32x4_t rv = {0, 0, 0, 0};
    if (p.residual) rv = load8(p.res_dt, p.residual, (long)m * p.ldres + nn);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int o = ml * 32 + cg * 8 + e;
      const float sum = ((part[o] + part[1024 + o]) + part[2048 + o]) + part[3072 + o];
      v[e] = sum * (p.scale ? p.scale[nn + e] : 1.f) + (p.bias ? p.bias[nn + e] : 0.f);
    }
    if (p.residual) add8(p.res_dt, rv, p.res_mult, v);
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    store8(p.out_dt, p.Y, (long)m * p.ldy + nn, v);
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int o = tid + 256 * q, ml = o >> 5, nl = o & 31, m = bm + ml, nn = bn + nl;
    if (m >= Mtot || nn >= p.Cout) continue;
    const float sum = ((part[o] + part[1024 + o]) + part[2048 + o]) + part[3072 + o];
    float v = sum * (p.scale ? p.scale[nn] : 1.f) + (p.bias ? p.bias[nn] : 0.f);
    if (p.residual) {
      const long ri = (long)m * p.ldres + nn;
      const float rres = p.res_dt == DRN_BF16 ? bf16_to_f32(((const bf16_t*)p.residual)[ri])
                         : p.res_dt == DRN_FP8 ? fp8_to_f32(((const uint8_t*)p.residual)[ri])
                                               : ((const float*)p.residual)[ri];
      v += rres * p.res_mult;
    }
    if (p.relu) v = fmaxf(v, 0.f);
    const long yi = (long)m * p.ldy + nn;
    if (p.out_dt == DRN_BF16) ((bf16_t*)p.Y)[yi] = f32_to_bf16(v);
    else if (p.out_dt == DRN_FP8) ((uint8_t*)p.Y)[yi] = f32_to_fp8(v);
    else ((float*)p.Y)[yi] = v;
  }
}

template <int DT, int BM, int BN>
int launch_gemm(const GemmParams& p, int splits, hipStream_t st) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  constexpr int smem = 2 * (BM + BN) * 128;
  auto k = gemm_nt_kernel<DT, BM, BN>;
  static bool attr = false;
  if (!attr && smem > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DRN_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(tiles, splits), dim3(256), smem, st, p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

template <int DT, bool PIPE, int PP = 0, bool TN = false>
int launch_gemm256(const GemmParams& p, int splits, hipStream_t st) {
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  constexpr int smem = 2 * 512 * 128;
  auto k = gemm_nt256_kernel<DT, PIPE, PP, TN>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DRN_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(tiles, splits), dim3(512), smem, st, p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

static int g_pingpong = 1;  // drn_tune(DRN_TUNE_GEMM_PINGPONG = 12): bf16 256x256 GEMMs run the ping-pong mainloop
static int g_tail_split = 1;  // drn_tune(DRN_TUNE_GEMM_TAIL_SPLIT): peel a nearly empty last round off persistent launches
static int g_persistent = 1;  // drn_tune(DRN_TUNE_GEMM_PERSISTENT): 256x256 GEMMs with more work items than CUs loop

static int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int DT, int PP = 0, bool TN = false>
int launch_gemm256p(const GemmParams& p, int nwg, hipStream_t st) {
  constexpr int smem = 2 * 512 * 128;
  auto k = gemm_nt256p_kernel<DT, PP, TN>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DRN_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(nwg), dim3(512), smem, st, p, p, 0);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

static int launch_gemm256p_tn_sgdp(const GemmParams& p, int nwg, hipStream_t st) {
  constexpr int smem = 2 * 512 * 128;
  auto k = gemm_nt256p_kernel<DRN_BF16, 1, true, false, true>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DRN_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(nwg), dim3(512), smem, st, p, p, 0);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

static int launch_gemm256p_pair(const GemmParams& p0, const GemmParams& p1, int nwg, int wg0, hipStream_t st) {
  constexpr int smem = 2 * 512 * 128;
  auto k = gemm_nt256p_kernel<DRN_BF16, 1, false, true>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DRN_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(nwg), dim3(512), smem, st, p0, p1, wg0);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// Tile rows per group of the XCD patch mapping (tile_coords).  One XCD (32 CUs) works on 32 consecutive tiles of its
// chunk at a time: GM rows x 32/GM columns.  4-row groups were tuned on the fc6 forward in round 1 (PMC: 1214 -> 612 MB).
// For the fc6 dW (8 x 196 tiles) 8-row groups - every B panel streamed once per XCD instead of twice - were measured in
// round 2 and are WORSE: FETCH_SIZE 734 MB vs 646 MB (profiles/r2_04_pmc_fc6_dw_gm8.json vs r2_03_..._gm4.json): the
// 8-MB A operand (dP1^T) does not stay in the 4-MB L2 between rounds and is re-read 6 times per XCD (mostly from the
// Infinity Cache - the counter sits on the fabric side of L2, so it counts those too); step rate unchanged.  4 stays.
static int g_group_rows = 0;  // drn_tune(DRN_TUNE_GEMM_GROUP_ROWS): 0 = default (4)
static int gemm256_group_rows(int M, int N, int splits) {
  (void)M; (void)N; (void)splits;
  return g_group_rows > 0 ? g_group_rows : 4;
}

// number of workgroups of the persistent launch, or 0 when the one-tile grid should be used
static int g_sgdp_ep4 = 1;  // drn_tune(DRN_TUNE_SGDP_EPILOGUE = 20): the fused dW + SGD launch's tile epilogue reads LDS four pieces at a time
static int g_nwg = 0;  // drn_tune(DRN_TUNE_GEMM_NWG = 18): resident workgroups of persistent launches (0 = one per CU); for launches
                       // on a CU-masked stream (the GEMM on a subset of the CUs, an HBM-bound kernel on the others)
static int persistent_grid(long total) {
  if (!g_persistent) return 0;
  const int nwg = g_nwg > 0 ? g_nwg : (cu_count() / 8) * 8;
  return (nwg >= 8 && total > nwg) ? nwg : 0;
}

static int g_conv_ksplit = 1;  // drn_tune(DRN_TUNE_CONV_KSPLIT): 0 = never use the 32x32 wave-K-split kernel
static int g_conv_patch = 1;  // drn_tune(DRN_TUNE_CONV_PATCH): 0 = never use conv3x3_c64_kernel; > 1 = minimum pixels per image
static long g_conv_patch_min = 32768;
static int g_conv_k2_tiles = -1;  // drn_tune(DRN_TUNE_CONV_K2_TILES): largest 64x64-tile count of ONE image for the two-K-group kernel (-1 = 2 x CUs, 0 = off)
static int g_conv_ks_tiles = 0;  // drn_tune(DRN_TUNE_CONV_KS_TILES): largest 64x64-tile count of ONE image that still takes it (0 = CUs / 4)

template <int DT, bool K64 = true>
int launch_conv_ks(const ConvParams& p, hipStream_t st) {
  const int Mtot = p.Nb * p.Ho * p.Wo;
  const int tiles = ((Mtot + 31) / 32) * ((p.Cout + 31) / 32);
  hipLaunchKernelGGL((conv_nhwc_ks_kernel<DT, K64>), dim3(tiles), dim3(256), 0, st, p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

static int launch_conv3x3_c64(const ConvParams& p, hipStream_t st) {
  const int tiles = p.Nb * ((p.Ho + P3_TH - 1) / P3_TH) * ((p.Wo + P3_TW - 1) / P3_TW);
  constexpr int smem = P3_PATCH + P3_WTS;  // 117 KB: one workgroup per CU
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)conv3x3_c64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv3x3_c64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DRN_ERR_LAUNCH;
    attr = true;
  }
  const int nwg = tiles < cu_count() ? tiles : cu_count();  // persistent: one workgroup per CU walks the pixel blocks
  if (p.residual) hipLaunchKernelGGL(conv3x3_c64_kernel<true>, dim3(nwg), dim3(256), smem, st, p);
  else hipLaunchKernelGGL(conv3x3_c64_kernel<false>, dim3(nwg), dim3(256), smem, st, p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

template <bool RES, bool PW, bool POOL>
static int launch_c64_variant(const ConvParams& p, int nwg, int smem, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)conv3x3_c64_kernel<RES, PW, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DRN_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((conv3x3_c64_kernel<RES, PW, POOL>), dim3(nwg), dim3(256), smem, st, p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}
// the fused forms (drn_conv3x3_pw_nhwc): [3x3] [+ 1x1 to 256 channels] [+ 2x2 max pool]
static int launch_conv3x3_c64_pw(const ConvParams& p, hipStream_t st) {
  const int tiles = p.Nb * ((p.Ho + P3_TH - 1) / P3_TH) * ((p.Wo + P3_TW - 1) / P3_TW);
  const bool pw = p.pw_w != nullptr, res = p.residual != nullptr, pool = p.pool != 0;
  const int smem = P3_PATCH + P3_WTS + (pw ? P3_PW_WTS : 0);  // 117 / 146.5 KB: one workgroup per CU
  const int nwg = tiles < cu_count() ? tiles : cu_count();
#define C64_CASE(R_, W_, P_) if (res == R_ && pw == W_ && pool == P_) return launch_c64_variant<R_, W_, P_>(p, nwg, smem, st)
  C64_CASE(false, true, false); C64_CASE(true, true, false); C64_CASE(false, true, true); C64_CASE(true, true, true);
  C64_CASE(false, false, true); C64_CASE(true, false, true);
#undef C64_CASE
  return DRN_ERR_UNSUPPORTED;  // (no 1x1 and no pool: drn_conv2d_nhwc's own class)
}

template <int DT, bool K64 = true>
int launch_conv_k2(const ConvParams& p, hipStream_t st) {
  const int Mtot = p.Nb * p.Ho * p.Wo;
  const int tiles = ((Mtot + 63) / 64) * ((p.Cout + 63) / 64);
  hipLaunchKernelGGL((conv_nhwc_k2_kernel<DT, K64>), dim3(tiles), dim3(512), 2 * 128 * 128, st, p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

static int launch_conv1x1_pp(const ConvParams& p, hipStream_t st) {
  const long Mtot = (long)p.Nb * p.Ho * p.Wo;
  const int tiles = (int)((Mtot + 255) / 256) * ((p.Cout + 255) / 256);
  constexpr int smem = 2 * 512 * 128;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)conv1x1_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DRN_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(conv1x1_pp_kernel, dim3(tiles), dim3(512), smem, st, p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

template <int DT, int BM, int BN, bool K64 = true>
int launch_conv(const ConvParams& p, hipStream_t st) {
  const int Mtot = p.Nb * p.Ho * p.Wo;
  const int tiles = ((Mtot + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
  constexpr int smem = ((BM == 64 && BN == 64) ? 1 : 2) * (BM + BN) * 128;
  auto k = conv_nhwc_kernel<DT, BM, BN, K64>;
  static bool attr = false;
  if (!attr && smem > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DRN_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(tiles), dim3(256), smem, st, p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

}  // namespace

// Tail balancing of a persistent 256x256 launch (see drn_gemm_nt): number of leading output columns that form an exact
// number of rounds of `nwg` resident workgroups; N when nothing is peeled.
static long tail_split_main_cols(int M, int N, int splits, int nwg) {
  const int tm = (M + 255) / 256, tn = (N + 255) / 256;
  const long wg256 = (long)tm * tn * splits, rem = wg256 % nwg;
  if (!g_tail_split || splits != 1 || rem == 0 || rem * 8 > (long)nwg * 3) return N;
  int a = nwg, b = tm;
  while (b) { const int t = a % b; a = b; b = t; }
  const int step = nwg / a;                   // tile columns per exact multiple of nwg tiles
  const int cols_main = (tn / step) * step;
  if (cols_main > 0 && (long)(tn - cols_main) * tm == rem) return (long)cols_main * 256;
  return N;
}

static int g_force_tile = 0;  // 0 = heuristic; 64 / 128 / 256 pin the tile (tuning + tests)

// conv_ring.hip: the register-ring kernels (bf16, Cin % 64 == 0); DRN_ERR_UNSUPPORTED outside their class
__attribute__((visibility("hidden"))) int drn_conv_ring_try(const ConvParams& p, int dtype, int cus, long tiles64_one, hipStream_t st);
__attribute__((visibility("hidden"))) int drn_conv_ring_set(int v);
// pp8.hip: the eight-wave 128x128 kernel (bf16, Cin % 64 == 0); DRN_ERR_UNSUPPORTED outside its class
__attribute__((visibility("hidden"))) int drn_pp8_conv_try(const ConvParams& p, int dtype, int cus, hipStream_t st);
__attribute__((visibility("hidden"))) int drn_pp8_set(int knob, int v);

extern "C" {

// tuning/test hook: pin the GEMM tile (0 restores the heuristic). Returns the previous value.
int drn_gemm_set_tile(int tile) {
  const int old = g_force_tile;
  if (tile == 0 || tile == 64 || tile == 128 || tile == 255 || tile == 256) g_force_tile = tile;
  return old;
}

// tuning knobs (A/B measurements and tests; defaults are the measured best).  Returns the previous value or -1.
__attribute__((visibility("hidden"))) int drn_sgd_set_grid(int blocks_x);  // head.hip
__attribute__((visibility("hidden"))) int drn_roi_set_map64(int on);        // pool.hip
__attribute__((visibility("hidden"))) int drn_roi_set_chunks(int cpb);      // pool.hip
__attribute__((visibility("hidden"))) int drn_roi_set_prefetch(int on);     // pool.hip
__attribute__((visibility("hidden"))) int drn_roi_set_map64_a(int on);      // pool.hip
__attribute__((visibility("hidden"))) int drn_roi_set_lds_kb(int kb);       // pool.hip
__attribute__((visibility("hidden"))) int drn_roi_set_lane(int on);         // pool.hip
__attribute__((visibility("hidden"))) int drn_roi_set_lane_reps(int reps);  // pool.hip
__attribute__((visibility("hidden"))) int drn_roi_set_st(int on);             // pool.hip
__attribute__((visibility("hidden"))) int drn_msm_set_wave(int on);           // head.hip
int drn_tune(int knob, int value) {
  if (knob == 1) {  // DRN_TUNE_GEMM_PERSISTENT
    const int old = g_persistent;
    g_persistent = value != 0;
    return old;
  }
  if (knob == 2) return drn_sgd_set_grid(value);  // DRN_TUNE_SGD_GRID
  if (knob == 4) return drn_roi_set_map64(value);  // DRN_TUNE_ROI_MAP64
  if (knob == 10) return drn_roi_set_chunks(value);    // DRN_TUNE_ROI_CPB
  if (knob == 11) return drn_roi_set_prefetch(value);  // DRN_TUNE_ROI_PREFETCH
  if (knob == 14) return drn_roi_set_map64_a(value);   // DRN_TUNE_ROI_MAP64_A
  if (knob == 15) return drn_roi_set_lds_kb(value);    // DRN_TUNE_ROI_LDS_KB
  if (knob == 19) return drn_roi_set_lane(value);      // DRN_TUNE_ROI_LANE
  if (knob == 22) return drn_roi_set_lane_reps(value);  // DRN_TUNE_ROI_LANE_REPS
  if (knob == 31) return drn_roi_set_st(value);         // DRN_TUNE_ROI_ST
  if (knob == 32) return drn_msm_set_wave(value);       // DRN_TUNE_MSM_WAVE
  if (knob == 5) {  // DRN_TUNE_CONV_KSPLIT
    const int old = g_conv_ksplit;
    g_conv_ksplit = value != 0;
    return old;
  }
  if (knob == 9) {  // DRN_TUNE_CONV_PATCH
    const int old = g_conv_patch ? (int)g_conv_patch_min : 0;
    g_conv_patch = value != 0;
    if (value > 1) g_conv_patch_min = value;
    return old;
  }
  if (knob == 8) {  // DRN_TUNE_CONV_K2_TILES
    const int old = g_conv_k2_tiles;
    g_conv_k2_tiles = value;
    return old;
  }
  if (knob == 7) {  // DRN_TUNE_CONV_KS_TILES
    const int old = g_conv_ks_tiles;
    if (value >= 0) g_conv_ks_tiles = value;
    return old;
  }
  if (knob == 6) {  // DRN_TUNE_GEMM_TAIL_SPLIT
    const int old = g_tail_split;
    g_tail_split = value != 0;
    return old;
  }
  if (knob == 12) {  // DRN_TUNE_GEMM_PINGPONG
    const int old = g_pingpong;
    g_pingpong = value < 0 ? 0 : value;
    return old;
  }
  if (knob == 13) {  // DRN_TUNE_FP8_K64
    const int old = g_fp8_k64;
    g_fp8_k64 = value != 0;
    return old;
  }
  if (knob == 18) {  // DRN_TUNE_GEMM_NWG
    const int old = g_nwg;
    if (value >= 0 && value % 8 == 0 && value <= 4096) g_nwg = value;
    return old;
  }
  if (knob == 20) {  // DRN_TUNE_SGDP_EPILOGUE
    const int old = g_sgdp_ep4;
    g_sgdp_ep4 = value != 0;
    return old;
  }
  if (knob == 23) return drn_conv_ring_set(value);  // DRN_TUNE_CONV_RING
  if (knob >= 25 && knob <= 30) return drn_pp8_set(knob, value);  // DRN_TUNE_PP8, _STAGES, _VARIANT, _PROFILE, _WIDE, _WIDE_VARIANT
  if (knob == 24) {  // DRN_TUNE_CONV_PP
    const int old = g_conv_pp;
    if (value >= 0) g_conv_pp = value;
    return old;
  }
  if (knob == 3) {  // DRN_TUNE_GEMM_GROUP_ROWS
    const int old = g_group_rows;
    if (value >= 0 && value <= 64) g_group_rows = value;
    return old;
  }
  return -1;
}

// Columns [0, n0) of an [M, N] output that drn_gemm_nt keeps for its persistent 256x256 launch; the columns from n0 on
// are the ones it peels off into a small-tile launch first (n0 == N: no peel, or another kernel takes the shape).
long drn_gemm_nt_main_cols(int M, int N, int splits) {
  if (M <= 0 || N <= 0 || splits < 1) return N;
  const long wg256 = (long)((M + 255) / 256) * ((N + 255) / 256) * splits;
  if (!((g_force_tile == 256 || (g_force_tile == 0 && wg256 >= 192)))) return N;
  const int nwg = persistent_grid(wg256);
  return nwg ? tail_split_main_cols(M, N, splits, nwg) : N;
}

// C[split][M,N] (fp32) = A[M,K] * B[N,K]^T over this split's K range.  See include/drn_wsod.h.
int drn_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, int dtype,
                int c_dtype, int splits, long c_split_stride, int accumulate, void* stream) {
  if (!A || !B || !C || M < 0 || N < 0 || K < 0) return DRN_ERR_ARG;
  if (M == 0 || N == 0) return DRN_OK;
  const int es = drn_esize(dtype);
  if (dtype != DRN_F32 && dtype != DRN_BF16) return DRN_ERR_ARG;
  if ((K * es) % 128 != 0 || (lda * es) % 16 != 0 || (ldb * es) % 16 != 0 || lda < K || ldb < K) return DRN_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)B) & 15) return DRN_ERR_ARG;
  if (splits < 1) return DRN_ERR_ARG;
  if (splits > 1 && accumulate) return DRN_ERR_ARG;
  if (c_dtype != DRN_F32 && c_dtype != DRN_BF16) return DRN_ERR_ARG;
  if (c_dtype == DRN_BF16 && (splits != 1 || accumulate)) return DRN_ERR_ARG;
  const int nslab = K * es / 128;
  GemmParams p{(const char*)A, (const char*)B, (float*)C, M, N, K, lda, ldb, ldc, (nslab + splits - 1) / splits,
               c_split_stride, accumulate};
  p.c_bf16 = c_dtype == DRN_BF16;
  p.nsplit = splits;
  p.gm = 4;
  hipStream_t st = (hipStream_t)stream;
  // 256x256 LDS-DMA kernel when it can put >= ~3/4 of the 256 CUs to work (1 workgroup of 128 KB LDS per CU);
  // otherwise the 128x128 / 64x64 register-staged kernels (more, smaller workgroups)
  const long wg256 = (long)((M + 255) / 256) * ((N + 255) / 256) * splits;
  const int force = g_force_tile;
  if (force == 255)  // the non-pipelined 256 kernel (kept for A/B comparison)
    return dtype == DRN_BF16 ? launch_gemm256<DRN_BF16, false>(p, splits, st) : launch_gemm256<DRN_F32, false>(p, splits, st);
  if ((force == 256 || (force == 0 && wg256 >= 192)) && (((uintptr_t)C) & 3) == 0) {
    p.gm = gemm256_group_rows(M, N, splits);
    if (const int nwg = persistent_grid(wg256)) {
      // Tail balancing.  The persistent kernel runs ceil(tiles / CUs) rounds: the fc6 dW row slab [1024 x 50176] is
      // 4 x 196 = 784 tiles = 3.06 rounds on 256 CUs, i.e. FOUR rounds with 240 CUs idle in the last one (measured:
      // 448 us for two slabs = 8 rounds of 56 us where 6.125 rounds of work exist).  When the last round is less
      // than 3/8 full, whole tile columns are peeled off so that the persistent launch is an exact number of rounds,
      // and the peeled columns (16 tiles of that slab) run first as their own launch of the small-tile kernel - many
      // short workgroups that fill every CU.  Same slab order and the same MFMA per output element in both kernels,
      // so the result is bit-identical to the unsplit launch (test_gemm_tail_split_bit_identical).
      const long n0 = tail_split_main_cols(M, N, splits, nwg);
      if (n0 < N) {
        GemmParams q = p;
        q.B = p.B + n0 * ldb * es;
        q.C = (float*)((char*)p.C + n0 * (p.c_bf16 ? 2 : 4));
        q.N = N - (int)n0;
        const bool small_t = (long)((M + 127) / 128) * ((q.N + 127) / 128) < 128;
        const int rc = dtype == DRN_BF16
                           ? (small_t ? launch_gemm<DRN_BF16, 64, 64>(q, 1, st) : launch_gemm<DRN_BF16, 128, 128>(q, 1, st))
                           : (small_t ? launch_gemm<DRN_F32, 64, 64>(q, 1, st) : launch_gemm<DRN_F32, 128, 128>(q, 1, st));
        if (rc != DRN_OK) return rc;
        p.N = (int)n0;
      }
      if (dtype == DRN_BF16 && g_pingpong == 1) return launch_gemm256p<DRN_BF16, 1>(p, nwg, st);
      if (dtype == DRN_BF16 && g_pingpong == 2) return launch_gemm256p<DRN_BF16, 2>(p, nwg, st);
      if (dtype == DRN_BF16 && g_pingpong == 3) return launch_gemm256p<DRN_BF16, 3>(p, nwg, st);
      return dtype == DRN_BF16 ? launch_gemm256p<DRN_BF16>(p, nwg, st) : launch_gemm256p<DRN_F32>(p, nwg, st);
    }
    if (dtype == DRN_BF16 && g_pingpong == 1) return launch_gemm256<DRN_BF16, true, 1>(p, splits, st);
    if (dtype == DRN_BF16 && g_pingpong == 2) return launch_gemm256<DRN_BF16, true, 2>(p, splits, st);
    if (dtype == DRN_BF16 && g_pingpong == 3) return launch_gemm256<DRN_BF16, true, 3>(p, splits, st);
    return dtype == DRN_BF16 ? launch_gemm256<DRN_BF16, true>(p, splits, st) : launch_gemm256<DRN_F32, true>(p, splits, st);
  }
  // 64x64 tiles (4x the workgroups) when 128x128 tiles would not even give every CU one workgroup: these launches are
  // latency-bound per K slab, not MFMA-bound (the joint peel of the fc6 dW, [2048 x 1024] x K 2000: 128 tiles of 128 took
  // 31 us; bit-identical either way)
  const bool small = force == 64 || (force == 0 && (long)((M + 127) / 128) * ((N + 127) / 128) * splits < cu_count());
  if (dtype == DRN_BF16) return small ? launch_gemm<DRN_BF16, 64, 64>(p, splits, st) : launch_gemm<DRN_BF16, 128, 128>(p, splits, st);
  return small ? launch_gemm<DRN_F32, 64, 64>(p, splits, st) : launch_gemm<DRN_F32, 128, 128>(p, splits, st);
}

// C[M,N] = A[M,K] * Bt[K,N]: the B operand given K-major ("TN"), read by the 256x256 ping-pong kernels through
// transposing LDS reads.  Same tile, slab order and MFMA per output element as drn_gemm_nt on a materialised transpose of
// Bt (bit-identical: test_gemm_tn_equals_nt_on_the_transpose).  bf16 only.  See include/drn_wsod.h.
int drn_gemm_tn(const void* A, const void* Bt, void* C, int M, int N, int K, int kb_rows, long lda, long ldb, long ldc,
                int c_dtype, int splits, long c_split_stride, int accumulate, void* stream) {
  if (!A || !Bt || !C || M < 0 || N < 0 || K < 0 || kb_rows < 0 || kb_rows > K) return DRN_ERR_ARG;
  if (M == 0 || N == 0) return DRN_OK;
  if ((K * 2) % 128 != 0 || (lda * 2) % 16 != 0 || (ldb * 2) % 16 != 0 || lda < K || ldb < N) return DRN_ERR_ARG;
  if ((((uintptr_t)A | (uintptr_t)Bt) & 15) || (((uintptr_t)C) & 3)) return DRN_ERR_ARG;
  if (splits < 1 || (splits > 1 && accumulate)) return DRN_ERR_ARG;
  if (c_dtype != DRN_F32 && c_dtype != DRN_BF16) return DRN_ERR_ARG;
  if (c_dtype == DRN_BF16 && (splits != 1 || accumulate)) return DRN_ERR_ARG;
  if ((long)K * ldb * 2 >= 0xFFFFFFF0L) return DRN_ERR_UNSUPPORTED;  // one 32-bit buffer offset spans Bt
  const int nslab = K * 2 / 128;
  GemmParams p{(const char*)A, (const char*)Bt, (float*)C, M, N, K, lda, ldb, ldc, (nslab + splits - 1) / splits,
               c_split_stride, accumulate};
  p.c_bf16 = c_dtype == DRN_BF16;
  p.nsplit = splits;
  p.gm = gemm256_group_rows(M, N, splits);
  p.kb_rows = kb_rows;
  hipStream_t st = (hipStream_t)stream;
  const long wg256 = (long)((M + 255) / 256) * ((N + 255) / 256) * splits;
  if (const int nwg = persistent_grid(wg256)) return launch_gemm256p<DRN_BF16, 1, true>(p, nwg, st);
  return launch_gemm256<DRN_BF16, true, 1, true>(p, splits, st);
}

// Two independent NT GEMMs (bf16 operands, fp32 outputs) in ONE persistent launch of the 256x256 ping-pong kernel: the
// workgroups of every XCD are divided between the problems in proportion to their work.  Same kernel arithmetic as two
// drn_gemm_nt calls (bit-identical).  Falls back to two calls when a problem is empty or the device has no 8-XCD grid.
int drn_gemm_nt_pair(const void* A0, const void* B0, void* C0, int M0, int N0, int K0, long lda0, long ldb0, long ldc0,
                     int splits0, long stride0, int accumulate0, const void* A1, const void* B1, void* C1, int M1, int N1,
                     int K1, long lda1, long ldb1, long ldc1, int splits1, long stride1, int accumulate1, void* stream) {
  auto two_calls = [&]() {
    const int rc = drn_gemm_nt(A0, B0, C0, M0, N0, K0, lda0, ldb0, ldc0, DRN_BF16, DRN_F32, splits0, stride0, accumulate0, stream);
    return rc != DRN_OK ? rc
                        : drn_gemm_nt(A1, B1, C1, M1, N1, K1, lda1, ldb1, ldc1, DRN_BF16, DRN_F32, splits1, stride1, accumulate1, stream);
  };
  if (M0 <= 0 || N0 <= 0 || M1 <= 0 || N1 <= 0 || !g_persistent || g_pingpong != 1 || g_force_tile == 64 || g_force_tile == 128)
    return two_calls();
  auto ok = [](const void* A, const void* B, void* C, int K, long lda, long ldb, int splits, int acc) {
    return A && B && C && K > 0 && (K * 2) % 128 == 0 && (lda * 2) % 16 == 0 && (ldb * 2) % 16 == 0 && lda >= K && ldb >= K &&
           !(((uintptr_t)A | (uintptr_t)B) & 15) && !(((uintptr_t)C) & 3) && splits >= 1 && !(splits > 1 && acc);
  };
  if (!ok(A0, B0, C0, K0, lda0, ldb0, splits0, accumulate0) || !ok(A1, B1, C1, K1, lda1, ldb1, splits1, accumulate1))
    return DRN_ERR_ARG;
  auto mk = [](const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, int splits, long stride,
               int acc) {
    const int nslab = K * 2 / 128;
    GemmParams p{(const char*)A, (const char*)B, (float*)C, M, N, K, lda, ldb, ldc, (nslab + splits - 1) / splits, stride, acc};
    p.nsplit = splits;
    p.gm = gemm256_group_rows(M, N, splits);
    return p;
  };
  const GemmParams p0 = mk(A0, B0, C0, M0, N0, K0, lda0, ldb0, ldc0, splits0, stride0, accumulate0);
  const GemmParams p1 = mk(A1, B1, C1, M1, N1, K1, lda1, ldb1, ldc1, splits1, stride1, accumulate1);
  const int nwg = (cu_count() / 8) * 8, per = nwg / 8;
  if (per < 2) return two_calls();
  const double w0 = (double)((M0 + 255) / 256) * ((N0 + 255) / 256) * splits0 * p0.k_slabs_per_split;
  const double w1 = (double)((M1 + 255) / 256) * ((N1 + 255) / 256) * splits1 * p1.k_slabs_per_split;
  int wg0 = (int)(per * w0 / (w0 + w1) + 0.5);
  wg0 = wg0 < 1 ? 1 : (wg0 > per - 1 ? per - 1 : wg0);
  return launch_gemm256p_pair(p0, p1, nwg, wg0, (hipStream_t)stream);
}

// G[M,N] = A[M,K] . Bt[K,N] (bf16, the TN form of drn_gemm_tn) into the bf16 gradient bucket AND W <- SGD(W, momentum_buf, G)
// in the same launch: every workgroup applies the update of the tile it finished last inside the next tile's mainloop
// (SgdPipe).  DRN_ERR_UNSUPPORTED outside its shape class (the caller then runs drn_gemm_tn + drn_sgd_step_block).
int drn_gemm_tn_sgd(const void* A, const void* Bt, void* grad_bucket, int M, int N, int K, int kb_rows, long lda, long ldb,
                    long ldc, float* weights, float* momentum_buf, void* shadow, long ld_w, const void* seg_dev,
                    float momentum, int first_step, float grad_scale, void* stream) {
  if (!A || !Bt || !grad_bucket || !weights || !momentum_buf || !seg_dev || M <= 0 || N <= 0 || K <= 0 || kb_rows < 0 ||
      kb_rows > K)
    return DRN_ERR_ARG;
  if (lda < K || ldb < N || ldc < N || ld_w < N) return DRN_ERR_ARG;
  // (alignment is part of the shape class, not an argument error: the caller's unfused pair takes such buffers - ADVICE r4)
  if ((lda * 2) % 16 != 0 || (ldb * 2) % 16 != 0 ||
      (((uintptr_t)A | (uintptr_t)Bt | (uintptr_t)grad_bucket | (uintptr_t)weights | (uintptr_t)momentum_buf | (uintptr_t)shadow) & 15))
    return DRN_ERR_UNSUPPORTED;
  // shape class of the pipelined update: whole K slabs (one 8-row chunk of the previous tile rides in each; with fewer than 32
  // slabs - fewer than 2048 proposals - the rest follows the mainloop, exposed), whole tiles, a bf16 shadow, 32-bit byte
  // offsets, enough tiles for the persistent grid, the ping-pong mainloop
  if (K < 128 || (K & 63) || (M & 255) || (N & 255) || !shadow || (ldc & 7) || (ld_w & 3) || g_pingpong != 1 ||
      (long)M * ld_w * 4 >= 0xFFFFFFF0L || (long)K * ldb * 2 >= 0xFFFFFFF0L)
    return DRN_ERR_UNSUPPORTED;
  const long tiles = (long)(M / 256) * (N / 256);
  const int nwg = persistent_grid(tiles);
  if (!nwg) return DRN_ERR_UNSUPPORTED;
  GemmParams p{(const char*)A, (const char*)Bt, (float*)grad_bucket, M, N, K, lda, ldb, ldc, K * 2 / 128, 0, 0,
               weights, momentum_buf, (bf16_t*)shadow, (const SgdSeg*)seg_dev, momentum, grad_scale, first_step};
  p.sgd_ld = ld_w;
  p.c_bf16 = g_sgdp_ep4 ? 3 : 1;  // (bit 1: four-at-a-time epilogue reads)
  p.nsplit = 1;
  p.gm = gemm256_group_rows(M, N, 1);
  p.kb_rows = kb_rows;
  return launch_gemm256p_tn_sgdp(p, nwg, (hipStream_t)stream);
}

// NHWC conv + per-channel affine (folded FrozenBN or bias) + optional residual + optional ReLU; `dtype` is the element
// type of x / w (fp32, bf16 or fp8 e4m3fn), y and the residual may be stored in another one (see include/drn_wsod.h).
// The tail of a 64-channel bottleneck on a large map as ONE launch (conv3x3_c64_kernel<.., PW, POOL>): the 3x3's output never
// goes to memory; w3 == NULL: no 1x1 stage (y has 64 channels, the residual - if any - too); pool: MaxPool2d(2, 2) in the
// epilogue (needs the last ReLU).  Same shape class as the LDS-resident-patch kernel takes on its own; anything else:
// DRN_ERR_UNSUPPORTED (the caller runs the separate launches).
int drn_conv3x3_pw_nhwc(const void* x, const void* w2, const float* scale2, const float* bias2, int relu2, const void* w3,
                        const float* scale3, const float* bias3, const void* residual, void* y, int Nb, int H, int W,
                        long ldw2, long ldw3, float res_mult, int relu3, int pool, void* stream) {
  if (!x || !w2 || !y || Nb <= 0 || H <= 0 || W <= 0) return DRN_ERR_ARG;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (!g_conv_patch || (long)H * W < g_conv_patch_min || (long)Nb * H * W * 128 >= 0xFFFFFFF0L ||
      ldw2 < 9 * 64 || (w3 && ldw3 < 64) || (ldw2 * 2) % 16 != 0 || (w3 && (ldw3 * 2) % 16 != 0) || !al16(x) || !al16(w2) ||
      (w3 && !al16(w3)) || !al16(y) || (residual && !al16(residual)) || (!w3 && !pool) ||
      (pool && (!(w3 ? relu3 : relu2) || H < 2 || W < 2)))
    return DRN_ERR_UNSUPPORTED;
  const int cy = w3 ? 256 : 64;
  ConvParams p{(const char*)x, (const char*)w2, (char*)y, scale2, bias2, (const char*)residual, Nb, H, W, 64, H, W,
               64, 3, 3, 1, 1, 1, relu2, 9 * 64, ldw2, cy, cy, DRN_BF16, DRN_BF16, res_mult, 0,
               (const char*)w3, ldw3, scale3, bias3, relu3, pool};
  return launch_conv3x3_c64_pw(p, (hipStream_t)stream);
}

int drn_conv2d_nhwc_q(const void* x, const void* w, void* y, const float* scale, const float* bias,
                      const void* residual, int Nb, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                      int pad, int dil, long ldw, long ldy, long ldres, int relu, int dtype, int out_dtype,
                      int res_dtype, float res_mult, void* stream) {
  if (!x || !w || !y) return DRN_ERR_ARG;
  auto known = [](int d) { return d == DRN_F32 || d == DRN_BF16 || d == DRN_FP8; };
  if (!known(dtype) || !known(out_dtype) || (residual && !known(res_dtype))) return DRN_ERR_ARG;
  const int es = drn_esize(dtype);
  if ((Cin * es) % 16 != 0 || (ldw * es) % 16 != 0) return DRN_ERR_ARG;  // 16-B chunks never straddle taps
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  if (Ho <= 0 || Wo <= 0 || Nb <= 0) return DRN_ERR_ARG;
  if ((long)Nb * H * W * Cin * es >= 0xFFFFFFF0L) return DRN_ERR_ARG;  // one buffer descriptor spans the input
  const int Ktot = KH * KW * Cin;
  if (ldw * es < ((Ktot * es + 127) / 128) * 128) return DRN_ERR_ARG;  // weight rows zero-padded to 128-B slabs
  ConvParams p{(const char*)x, (const char*)w, (char*)y, scale, bias, (const char*)residual, Nb, H, W, Cin, Ho, Wo,
               Cout, KH, KW, stride, pad, dil, relu, Ktot, ldw, ldy, ldres, out_dtype, res_dtype, res_mult, g_fp8_k64};
  hipStream_t st = (hipStream_t)stream;
  const long Mtot = (long)Nb * Ho * Wo;
  const bool small = ((Mtot + 127) / 128) * ((Cout + 127) / 128) < 128;
  // narrow outputs (the 64-channel stem / res2 layers at real image sizes): a 128x128 tile would run half empty
  const bool narrow = !small && Cout <= 64;
  // few 64x64 tiles and a long K loop: latency-bound, see conv_nhwc_ks_kernel
  const int nslab = (Ktot * es + 127) / 128;
  // (decided on ONE image's geometry: this kernel adds the K partials in another order than the tiled ones, and a layer
  // must round the same way whether its image runs alone or in a batch - graphed trunk pairs vs eager, 2 ranks vs 1)
  const long tiles64 = (((long)Ho * Wo + 63) / 64) * ((Cout + 63) / 64);
  // (round 2, tools/conv_bench.py at 800x1216: up to one 64x64 tile per CU the 36-slab res4 3x3 still gains, 23.1 ->
  // 20.5 us, while layers with few slabs lose - the 9-slab stem 3x3 8.6 -> 9.7 us at 224x224: deep K only)
  const long ks_max = g_conv_ks_tiles > 0 ? g_conv_ks_tiles : (nslab >= 32 ? cu_count() : cu_count() / 4);
  // LDS-resident patch + weights for the 64-channel 3x3 layers of large maps (conv3x3_c64_kernel; 117 KB of LDS, so only
  // where the trunk is not meant to share CUs with the heads' GEMMs: maps of >= 32k pixels)
  if (g_conv_patch && dtype == DRN_BF16 && out_dtype == DRN_BF16 && (!residual || res_dtype == DRN_BF16) && Cin == 64 &&
      Cout == 64 && KH == 3 && KW == 3 && stride == 1 && dil == 1 && pad == 1 && (long)Ho * Wo >= g_conv_patch_min &&
      (ldy & 7) == 0 && (((uintptr_t)y) & 15) == 0 && (!residual || ((ldres & 7) == 0 && (((uintptr_t)residual) & 15) == 0)) &&
      (ldw * 2) % 16 == 0 && (((uintptr_t)w) & 15) == 0)
    return launch_conv3x3_c64(p, st);
  // everything beyond the latency-bound small maps: the register-ring kernels (conv_ring.hip; bf16, Cin % 64 == 0).  Decided on
  // ONE image's geometry: the small-map kernels below add their K partials in another order
  // 1x1 / stride 1 convs to >= 256 channels of a large map: the GEMM ping-pong mainloop with the conv epilogue
  // (conv1x1_pp_kernel; the dilated-C5 trunk's 1x1s to 512 / 1024 / 2048 channels and the res2 1x1s to 256 channels at a real
  // image size).  Class measured (profiles/r5_17_conv1x1_pp_threshold_*.txt, r5_19_conv_800.txt): from ~3/4 of the CUs' worth
  // of 256x256 tiles per image on it wins at any K; at 100-191 tiles - half the chip holds a tile - only with a long K loop
  // (>= 16 slabs: 2048 -> 512 at 118 tiles 47 -> 44 us, but 128 -> 512 at 120 tiles 12 -> 15 us); at 59-60 tiles (res4 of the
  // C4 trunk, the DC5 trunk's 1x1s to 256 channels) the small tiles win; a 64-channel output wastes three quarters of the
  // tile.  Decided on ONE image's geometry; same bits as the tiled kernels either way.
  if (drn_pp8_set(25, -1) == 2) {  // (A/B pin: every layer in the eight-wave kernel's class takes it)
    const int rc = drn_pp8_conv_try(p, dtype, cu_count(), st);
    if (rc != DRN_ERR_UNSUPPORTED) return rc;
  }
  const bool pp_ok = g_conv_pp && dtype == DRN_BF16 && out_dtype == DRN_BF16 && (!residual || res_dtype == DRN_BF16) && KH == 1 &&
                     KW == 1 && stride == 1 && pad == 0 && (Cin & 63) == 0 && (Cout & 7) == 0 && (ldy & 3) == 0 &&
                     (!residual || (ldres & 3) == 0) &&
                     (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)residual) & 15) == 0 && (ldw * 2) % 16 == 0 &&
                     (long)Cout * ldw * 2 < 0xFFFFFFF0L;
  const long t256 = (((long)Ho * Wo + 255) / 256) * ((Cout + 255) / 256);
  // (1) >= 3/4 of the CUs' worth of 256x256 tiles: the ping-pong GEMM mainloop, whatever K
  if (pp_ok && (g_conv_pp > 1 ? t256 >= g_conv_pp : (Cout >= 256 && t256 >= 192))) return launch_conv1x1_pp(p, st);
  // (2) the eight-wave 128x128 / 256x128 kernel (pp8.hip; round 6): layers with too few 256x256 tiles for (1) - the 3x3s of
  // res4 / res5, the 1x1s to 256 / 512 channels of the dilated-C5 trunk, the C4 trunk's 1x1s to 1024 channels
  {
    const int rc = drn_pp8_conv_try(p, dtype, cu_count(), st);
    if (rc != DRN_ERR_UNSUPPORTED) return rc;
  }
  // (3) 100-191 tiles of 256x256 with a long K loop (round 5's class; reached when (2) is switched off)
  if (pp_ok && g_conv_pp == 1 && Cout >= 256 && t256 >= 100 && (Cin >> 6) >= 16) return launch_conv1x1_pp(p, st);
  {
    const int rc = drn_conv_ring_try(p, dtype, cu_count(), tiles64, st);
    if (rc != DRN_ERR_UNSUPPORTED) return rc;
  }
  // two K-groups per 64x64 tile (conv_nhwc_k2_kernel): mid-size layers - more 64x64 tiles than the wave-K-split kernel
  // takes, at most one per CU (the kernel keeps one 512-thread workgroup per CU) - with an even slab count >= 8
  const long k2_max = g_conv_k2_tiles >= 0 ? g_conv_k2_tiles : cu_count();
  if ((nslab & 1) == 0 && nslab >= 8 && tiles64 > cu_count() / 4 && tiles64 <= k2_max && Nb <= 64)
    return dtype == DRN_BF16 ? launch_conv_k2<DRN_BF16>(p, st)
           : dtype == DRN_FP8 ? (p.fp8_k64 ? launch_conv_k2<DRN_FP8>(p, st) : launch_conv_k2<DRN_FP8, false>(p, st))
                              : launch_conv_k2<DRN_F32>(p, st);
  if (g_conv_ksplit && tiles64 <= ks_max && nslab >= 8 && Nb <= 64)
    return dtype == DRN_BF16 ? launch_conv_ks<DRN_BF16>(p, st)
           : dtype == DRN_FP8 ? (p.fp8_k64 ? launch_conv_ks<DRN_FP8>(p, st) : launch_conv_ks<DRN_FP8, false>(p, st))
                              : launch_conv_ks<DRN_F32>(p, st);
  if (dtype == DRN_BF16)
    return small ? launch_conv<DRN_BF16, 64, 64>(p, st)
                 : narrow ? launch_conv<DRN_BF16, 128, 64>(p, st) : launch_conv<DRN_BF16, 128, 128>(p, st);
  if (dtype == DRN_FP8 && p.fp8_k64)
    return small ? launch_conv<DRN_FP8, 64, 64>(p, st)
                 : narrow ? launch_conv<DRN_FP8, 128, 64>(p, st) : launch_conv<DRN_FP8, 128, 128>(p, st);
  if (dtype == DRN_FP8)
    return small ? launch_conv<DRN_FP8, 64, 64, false>(p, st)
                 : narrow ? launch_conv<DRN_FP8, 128, 64, false>(p, st) : launch_conv<DRN_FP8, 128, 128, false>(p, st);
  return small ? launch_conv<DRN_F32, 64, 64>(p, st)
               : narrow ? launch_conv<DRN_F32, 128, 64>(p, st) : launch_conv<DRN_F32, 128, 128>(p, st);
}

int drn_conv2d_nhwc(const void* x, const void* w, void* y, const float* scale, const float* bias,
                    const void* residual, int Nb, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                    int pad, int dil, long ldw, long ldy, long ldres, int relu, int dtype, void* stream) {
  if (dtype != DRN_F32 && dtype != DRN_BF16) return DRN_ERR_ARG;
  return drn_conv2d_nhwc_q(x, w, y, scale, bias, residual, Nb, H, W, Cin, Cout, KH, KW, stride, pad, dil, ldw, ldy, ldres,
                           relu, dtype, dtype, dtype, 1.0f, stream);
}

}  // extern "C"
