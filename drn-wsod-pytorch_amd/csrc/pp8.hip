// 128x128-tile MFMA kernel for gfx950 with EIGHT waves: two waves per SIMD, half a phase apart (round 6).
//
// What it replaces on the reference path:
//   * conv mode: F.conv2d + FrozenBatchNorm2d + relu_ + the shortcut add of the WS-ResNet / VGG blocks at real image sizes
//     (projects/WSL/wsl/modeling/backbone/resnet_ws.py:217-237 BottleneckBlock.forward, :672-678 dilated res4 / res5;
//     projects/WSL/wsl/modeling/backbone/vgg.py:104-122; detectron2/layers/wrappers.py:94-99, batch_norm.py:45-65);
//   * fc mode: relu_(fc(x)) + F.dropout of DiscriminativeAdaptionNeck.forward
//     (projects/WSL/wsl/modeling/roi_heads/box_head.py:82-91) as ONE launch - no split-K partials, no second pass.
//
// Why.  The mid-size layers of the trunk (res3 / res4 / res5 of an 800x1216 image: 3800-15200 rows x 128-2048 columns) and
// fc7 ([2000 x 2048] . [4096 x 2048]^T) offer a few hundred 128x128 tiles: too few for the 256x256 ping-pong GEMM (one tile
// per 2-4 CUs), and on four-wave 64x64 / 128x128 tiles each SIMD has ONE wave that does its loads, its fragment reads and its
// MFMAs strictly in turn (HISTORY 11.1: MFMA pipes 7-32 % busy).  Here a 128x128 tile is worked by eight waves as 2 (M) x 4 (N)
// wave tiles of 64x32 - two MFMA 32x32 accumulators per wave - and a K slab (128 bytes per row) is ONE phase pair
//     [12 fragment reads + 4 LDS-DMA pieces + the wait that retires the next slab]  barrier  [8 MFMAs, s_setprio 1]  barrier
// The two waves of a SIMD (wave w and w + 4 = the two rows of the wave layout) run one barrier interval apart: while one
// multiplies, its partner reads fragments and issues the DMA of a later slab - the 256x256 kernel's schedule
// (gemm_conv.hip pp_mainloop) at a quarter of its tile.
// Staging: `buffer_load_dwordx4 ... lds` into a ring of NSTG stages of 32 KB (A rows [128][128 B] then B rows [128][128 B],
// 16-byte k-slots XOR-swizzled with (row >> 1) & 7 on the SOURCE side, same involution on the fragment reads).  The im2col
// gather is the per-lane source offset: row = output pixel, a K slab = 64 input channels of one tap; taps in the zero padding,
// rows beyond M and weight rows beyond N read out of range, i.e. as zeros.  A plain row-major operand (1x1 convs, fc) is the
// same thing with one tap.
// Hazards (NSTG = DIST + 2, the lagging wave row passes every barrier one instance later):
//   RAW  slab t + 1 is read in iteration t + 1; every wave's counted vmcnt for ITS pieces of slab t + 1 stands in front of
//        its mid barrier of iteration t, which is at the latest the barrier the leading row passes at the end of iteration t;
//   WAR  iteration t issues slab t + DIST into the stage that held slab t - 2, whose last MFMA (hence last fragment read) ended
//        before the lagging row's end barrier of iteration t - 2 - two barrier instances before the leading row's issue.
// Same LDS image, slab order, k-steps and MFMA per output element as conv_nhwc_kernel / conv_ring_kernel / gemm_nt256 (the
// operands are swapped in the MFMA - D^T - exactly as conv1x1_pp_kernel does): bit-identical to them.
#include "drn_common.h"
#include "conv_params.h"

#include <stdio.h>
#include <string.h>
#include <type_traits>

namespace {

using drn_conv::ConvParams;

typedef __attribute__((address_space(3))) void* lds_ptr;

struct Pp8Params {
  const char* A;      // im2col source: NHWC input (conv) or the row-major matrix (fc)
  unsigned a_bytes;   // size of the A operand (buffer descriptor range)
  unsigned pix_b;     // bytes per pixel = row pitch of A
  const char* B;      // weights [N][ldb] K-major
  unsigned ldb_b;     // bytes per weight row
  int Mtot, N;
  int HoWo, Wo, H, W, KH, KW, stride, pad, dil;
  int spt;            // 128-byte K slabs per tap (Cin / 64; fc: K / 64)
  // epilogue
  const float* scale;     // conv: per-channel multiplier or null
  const float* bias;      // per-channel addend or null
  const char* residual; long ldres; float res_mult;  // conv: bf16 shortcut [Mtot][ldres] or null
  int relu;
  char* Y; long ldy;      // bf16 [Mtot][ldy]
  char* YT; long ldyt;    // fc: bf16 transposed copy [N][ldyt] or null
  const float* mask;      // fc: explicit dropout multipliers [Mtot][N] or null
  unsigned long long seed; const unsigned long long* seed_dev; float drop_p;  // fc: counter-based dropout (drn_common.h)
};

__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn, int GM) {
  const int group_sz = GM * tiles_n;
  const int g = id / group_sz, in_g = id - g * group_sz;
  const int first_m = g * GM;
  const int gm = tiles_m - first_m < GM ? tiles_m - first_m : GM;
  tm = first_m + in_g % gm;
  tn = in_g / gm;
}

#define PP8_BARRIER()                    \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

template <int N_>
__device__ __forceinline__ void pp8_wait_vm() {
  static_assert(N_ >= 0 && N_ <= 12, "pieces in flight");
  if constexpr (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N_ == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N_ == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N_ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N_ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N_ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N_ == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else { static_assert(N_ == 12, "a count this file uses"); asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
}

constexpr int PP8_CONV = 0, PP8_FC = 1;

// VAR & 8 (drn_tune(DRN_TUNE_PP8_PROFILE)): shader-clock split of the mainloop for the waves of workgroup 0 -
// [wave][0] fragment-read + issue phase, [1] wait at the mid barrier, [2] MFMA phase, [3] wait at the end barrier, [4] prologue,
// [5] epilogue, [6] slabs
__device__ unsigned long long g_pp8_prof[8][8];

// position (16-byte chunk index 0..31) of logical chunk k (4 consecutive output columns) in row `row` of the fp32 epilogue tile
// [128][128]: pairs (2j, 2j + 1) sit 256 bytes apart at j ^ (row & 15) - the accumulator writes (8 lanes = 8 rows of one chunk)
// and the row reads (16 lanes = the 16 pairs of a row, first halves then second halves) are both free of bank conflicts, and a
// lane that reads positions p and p + 16 holds 8 consecutive columns
__device__ __forceinline__ int pp8_pos(int k, int row) { return ((k >> 1) ^ (row & 15)) | ((k & 1) << 4); }

// VAR & 3 (BM = 128) = where a slab's four DMA pieces are issued: 0 = all in the fragment-read phase, 1 = two there and two
// between the MFMAs, 2 = all between the MFMAs (needs DIST >= 2: a piece issued in an MFMA phase is waited for a whole slab
// later); VAR & 4 = no s_setprio around the MFMAs; VAR & 8 = profile build.
//
// BM = 256 ("wide": 256 x 128 tile, waves 4 (M) x 2 (N), wave tile 64 x 64 = four accumulators): a K slab is TWO phase pairs
//     L0 [12 reads: A blocks 0 / 1, B block 0]  M0 [8 MFMAs]  L1 [4 reads: B block 1; 3 DMA pieces; vmcnt]  M1 [8 MFMAs + 3 pieces]
// on a ring of THREE 48-KB stages: iteration t issues slab t + 2 in its second half (L1 / M1) into the stage that held slab
// t - 1, whose last MFMA ended two barrier instances earlier for either wave row, and waits for slab t + 1 in L1 - a whole
// slab (four barrier intervals) after it was issued, one barrier instance ahead of the first read of slab t + 1.
// Per MFMA it moves 3/4 of the DMA pieces and 2/3 of the fragment bytes of the 128 x 128 form (the LDS port -
// DMA landing + fragment reads - is what bounds these tiles).
template <int BM, int NSTG, int EPI, int VAR>
__global__ __launch_bounds__(512) void pp8_kernel(Pp8Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BN = 128, A_BYTES = BM * 128, STAGE = (BM + BN) * 128, DIST = NSTG - 2;
  constexpr int NJ = BM / 128, WN = 4 / NJ, AP = BM / 64;  // B blocks per wave, wave columns, A pieces per wave and slab
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr int NL = (VAR & 3) == 0 ? 4 : (VAR & 3) == 1 ? 2 : 0;  // BM = 128: pieces issued in the fragment-read phase
  constexpr bool PRIO = !(VAR & 4);
  constexpr bool PROF = (VAR & 8) != 0;
  [[maybe_unused]] unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
  if constexpr (PROF) t0 = __builtin_amdgcn_s_memtime();
  static_assert(BM == 128 || (BM == 256 && NSTG == 3), "tile shapes");
  static_assert(DIST >= 1 && DIST <= 3, "ring depth");
  static_assert(BM == 256 || NL == 4 || DIST >= 2, "pieces issued beside the MFMAs need two slabs of distance");
  static_assert(BM * BN * 4 <= NSTG * STAGE, "the fp32 epilogue tile fits the ring");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool lag = wave >= 4;  // waves w and w + 4 share a SIMD: the upper four run one barrier interval behind
  const int tiles_m = (p.Mtot + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn, 4);
  const int bm = tm * BM, bn = tn * BN;

  // ---- epilogue operands of the conv form, fetched NOW: in the read-back phase a lane owns the 8 consecutive channels
  // nn .. nn + 7 of rows rl + 32 q - their affine and their shortcut rows do not depend on the mainloop, and at the end of the
  // kernel each would cost a global-memory round trip with nothing to hide behind (16 + 4 NPASS registers)
  const int cl = tid & 15, rl = tid >> 4;
  const int nn = bn + cl * 8;
  const bool col_ok = nn < p.N;
  constexpr int NPASS = BM / 32;
  [[maybe_unused]] i32x4_t rv[NPASS];
  [[maybe_unused]] f32x4_t sc8[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}}, bi8[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if constexpr (EPI == PP8_CONV) {
    if (col_ok) {
      if (p.scale) { sc8[0] = *(const f32x4_t*)(p.scale + nn); sc8[1] = *(const f32x4_t*)(p.scale + nn + 4); }
      if (p.bias) { bi8[0] = *(const f32x4_t*)(p.bias + nn); bi8[1] = *(const f32x4_t*)(p.bias + nn + 4); }
      if (p.residual) {
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
          const int m = bm + rl + 32 * q;
          rv[q] = *(const i32x4_t*)(p.residual + ((long)(m < p.Mtot ? m : p.Mtot - 1) * p.ldres + nn) * 2);
        }
      }
    }
  }

  // fc form: the bias of this lane's accumulator columns (n = 8 q + 4 (lane >> 5) + e of each B block), fetched now for the same reason
  [[maybe_unused]] f32x4_t fcb[BM / 128][4];
  if constexpr (EPI == PP8_FC) {
#pragma unroll
    for (int j = 0; j < BM / 128; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nc = bn + (wave % (4 / (BM / 128))) * (32 * (BM / 128)) + j * 32 + 8 * q + 4 * (lane >> 5);
        const f32x4_t b4 = *(const f32x4_t*)((p.bias ? p.bias : (const float*)p.B) + (nc + 4 <= p.N ? nc : 0));  // (clamped: unconditional load)
        fcb[j][q] = (p.bias && nc < p.N) ? b4 : f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
  }

  // ---- DMA sources.  Wave w fills pieces w, w + 8, .. of A and of B: LDS rows 8 w + 64 q + lane / 8, physical slot lane & 7,
  // which holds the k-slot (lane & 7) ^ ((row >> 1) & 7) of that row (the XOR is the same for every q)
  const int r8 = lane >> 3;
  const int ks_src = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
  const int ntaps = p.KH * p.KW;
  const int n = ntaps * p.spt;
  constexpr int NPRO = BM == 256 ? 2 : DIST;  // slabs the prologue issues
  unsigned vob[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) vob[q] = (unsigned)(8 * wave + 64 * q + r8) * p.ldb_b + ks_src * 16;
  int brem = p.N - bn;
  if (brem > BN) brem = BN;
  const __amdgpu_buffer_rsrc_t rb =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (long)bn * p.ldb_b), 0, (unsigned)brem * p.ldb_b, 0x00020000);
  // the weight pieces of the prologue's slabs go out before the im2col arithmetic below: the first touch of global memory costs
  // a few thousand cycles either way, ~2500 cycles of address arithmetic fit under it
#pragma unroll
  for (int d = 0; d < NPRO; ++d)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(smem + d * STAGE + wave * 1024 + A_BYTES + q * 8192), 16, (int)vob[q],
                                               (d < n ? d : n - 1) * 128, 0, 0);
  int abase[AP];
  unsigned amask[AP];
#pragma unroll
  for (int q = 0; q < AP; ++q) {
    const int m = bm + 8 * wave + 64 * q + r8;
    unsigned mask = 0;
    int base = 0;
    if (m < p.Mtot) {
      const int nb = m / p.HoWo, rem = m - nb * p.HoWo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      base = ((nb * p.H + hi0) * p.W + wi0) * (int)p.pix_b + ks_src * 16;
      // tap (kh, kw) is inside the image iff its row is and its column is: KH + KW range checks instead of KH * KW
      unsigned cmask = 0;
      for (int kw = 0; kw < p.KW; ++kw) cmask |= (unsigned)((unsigned)(wi0 + kw * p.dil) < (unsigned)p.W) << kw;
      for (int kh = 0; kh < p.KH; ++kh)
        if ((unsigned)(hi0 + kh * p.dil) < (unsigned)p.H) mask |= cmask << (kh * p.KW);
    }
    abase[q] = base;
    amask[q] = mask;
  }
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);

  // ---- issue state (wave-uniform): slabs go out in order; past the last slab the last one is fetched again into a stage
  // nobody reads any more (keeps the loop free of branches and the vmcnt counts constant)
  int l_s = 0, l_cs = 0, l_tap = 0, l_kw = 0, l_kh = 0;
  auto piece = [&](int stg, auto qtag) {  // q < AP: A rows 8 w + 64 q ..; q = AP, AP + 1: B rows
    constexpr int q = decltype(qtag)::value;
    char* sa = smem + stg * STAGE + wave * 1024;
    if constexpr (q < AP) {
      const int delta = ((l_kh * p.dil) * p.W + l_kw * p.dil) * (int)p.pix_b + l_cs * 128;
      // (arithmetic select - one v_bfe_i32 + one v_bfi_b32: written as `cond ? a : b` the compiler branches around the add with
      // s_and_saveexec, which also splits the loop body into blocks and turns its counted LDS waits into lgkmcnt(0))
      const unsigned sel = 0u - ((amask[q] >> l_tap) & 1u);
      const unsigned off = ((unsigned)(abase[q] + delta) & sel) | (OOB & ~sel);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sa + q * 8192), 16, (int)off, 0, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sa + A_BYTES + (q - AP) * 8192), 16, (int)vob[q - AP], l_s * 128, 0, 0);
    }
  };
  auto advance = [&]() {
    if (l_s + 1 < n) {
      ++l_s;
      if (++l_cs == p.spt) {
        l_cs = 0;
        ++l_tap;
        if (++l_kw == p.KW) { l_kw = 0; ++l_kh; }
      }
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
  // ---- fragment addresses inside stage 0 (second A / B block = + 4096; other stages = + stage * STAGE)
  const int l31 = lane & 31, hi = lane >> 5, swr = (l31 >> 1) & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned oa[4], ob[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    oa[ks] = lds0 + (wm * 64 + l31) * 128 + (((ks * 2 + hi) ^ swr) << 4);
    ob[ks] = lds0 + A_BYTES + (wn * (32 * NJ) + l31) * 128 + (((ks * 2 + hi) ^ swr) << 4);
  }
  typedef __attribute__((address_space(3))) const i32x4_t* lds_v4;

  f32x16_t c[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;

  // operands swapped in the MFMA: D^T[n][m] - a lane holds 4 consecutive output columns per register quad
#define PP8_MM(J, ks)                                                                                                             \
  c[0][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[ks]), __builtin_bit_cast(bf16x8_t, fa0[ks]), c[0][J], 0, 0, 0); \
  c[1][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[ks]), __builtin_bit_cast(bf16x8_t, fa1[ks]), c[1][J], 0, 0, 0)
#define PP8_PIECE(Q) do { __builtin_amdgcn_sched_barrier(0); piece(is, Q{}); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP8_CLK(K) do { if constexpr (PROF) { t1 = __builtin_amdgcn_s_memtime(); tp[K] += t1 - t0; t0 = t1; } } while (0)

  // ---- prologue: the A pieces of its slabs (queue: B(0) .. B(NPRO - 1), A(0) .. A(NPRO - 1)); everything but the LAST slab's
  // A pieces has landed - slab 0 for the first reads, and the steady-state counts below hold from iteration 0 on
  static_assert(AP == 2 || AP == 4, "A pieces per wave");
#pragma unroll
  for (int d = 0; d < NPRO; ++d) {
    piece(d, I0{}); piece(d, I1{});
    if constexpr (AP == 4) { piece(d, I2{}); piece(d, I3{}); }
    advance();
  }
  pp8_wait_vm<NPRO == 1 ? 0 : AP>();
  PP8_BARRIER();
  if (lag) PP8_BARRIER();
  int rs = 0, is = BM == 256 ? 2 : DIST;
  PP8_CLK(4);
  if constexpr (BM == 128) {
    for (int t = 0; t < n; ++t) {
      const unsigned so = (unsigned)rs * STAGE;
      i32x4_t fa0[4], fa1[4], fb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {  // in the order the MFMAs consume them
        fb[ks] = *(lds_v4)(uintptr_t)(ob[ks] + so);
        fa0[ks] = *(lds_v4)(uintptr_t)(oa[ks] + so);
        fa1[ks] = *(lds_v4)(uintptr_t)(oa[ks] + so + 4096);
      }
      if constexpr (NL == 4) { piece(is, I0{}); piece(is, I1{}); piece(is, I2{}); piece(is, I3{}); }
      else if constexpr (NL == 2) { piece(is, I0{}); piece(is, I2{}); }
      // this wave's pieces of slab t + 1 have landed; the younger slabs (and this slab's first pieces) stay in flight
      pp8_wait_vm<NL == 4 ? (DIST - 1) * 4 : (DIST - 2) * 4 + NL>();
      PP8_CLK(0);
      PP8_BARRIER();
      PP8_CLK(1);
      if (PRIO) __builtin_amdgcn_s_setprio(1);
      PP8_MM(0, 0);
      if constexpr (NL == 0) PP8_PIECE(I0);
      PP8_MM(0, 1);
      if constexpr (NL == 2) PP8_PIECE(I1);
      if constexpr (NL == 0) PP8_PIECE(I1);
      PP8_MM(0, 2);
      if constexpr (NL == 2) PP8_PIECE(I3);
      if constexpr (NL == 0) PP8_PIECE(I2);
      PP8_MM(0, 3);
      if constexpr (NL == 0) PP8_PIECE(I3);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      advance();
      if constexpr (PROF) asm volatile("s_nop 0" : "+v"(c[0][0]), "+v"(c[1][0]));
      PP8_CLK(2);
      PP8_BARRIER();
      PP8_CLK(3);
      rs = rs + 1 == NSTG ? 0 : rs + 1;
      is = is + 1 == NSTG ? 0 : is + 1;
    }
  } else {
    for (int t = 0; t < n; ++t) {
      const unsigned so = (unsigned)rs * STAGE;
      i32x4_t fa0[4], fa1[4], fb[4];
      // ---- L0: both A blocks and B block 0 of slab t
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fb[ks] = *(lds_v4)(uintptr_t)(ob[ks] + so);
        fa0[ks] = *(lds_v4)(uintptr_t)(oa[ks] + so);
        fa1[ks] = *(lds_v4)(uintptr_t)(oa[ks] + so + 4096);
      }
      PP8_CLK(0);
      PP8_BARRIER();
      PP8_CLK(1);
      if (PRIO) __builtin_amdgcn_s_setprio(1);
      PP8_MM(0, 0); PP8_MM(0, 1); PP8_MM(0, 2); PP8_MM(0, 3);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      if constexpr (PROF) asm volatile("s_nop 0" : "+v"(c[0][0]), "+v"(c[1][0]));
      PP8_CLK(2);
      PP8_BARRIER();
      PP8_CLK(3);
      // ---- L1: B block 1; the first half of slab t + 2; slab t + 1 has landed (this wave's pieces)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fb[ks] = *(lds_v4)(uintptr_t)(ob[ks] + so + 4096);
      // (VAR & 1: all six pieces here - this phase has four fragment reads where L0 has twelve, and a piece issued between
      // MFMAs stalls the wave's matrix stream; VAR & 1 == 0: three here, three between the MFMAs of M1)
      if constexpr (VAR & 1) {
        piece(is, I0{}); piece(is, I1{}); piece(is, I4{}); piece(is, I2{}); piece(is, I3{}); piece(is, I5{});
        pp8_wait_vm<6>();
      } else {
        piece(is, I0{}); piece(is, I1{}); piece(is, I4{});
        pp8_wait_vm<3>();
      }
      PP8_CLK(0);
      PP8_BARRIER();
      PP8_CLK(1);
      if (PRIO) __builtin_amdgcn_s_setprio(1);
      PP8_MM(1, 0);
      if constexpr (!(VAR & 1)) PP8_PIECE(I2);
      PP8_MM(1, 1);
      if constexpr (!(VAR & 1)) PP8_PIECE(I3);
      PP8_MM(1, 2);
      if constexpr (!(VAR & 1)) PP8_PIECE(I5);
      PP8_MM(1, 3);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      advance();
      if constexpr (PROF) asm volatile("s_nop 0" : "+v"(c[0][1]), "+v"(c[1][1]));
      PP8_CLK(2);
      PP8_BARRIER();
      PP8_CLK(3);
      rs = rs + 1 == NSTG ? 0 : rs + 1;
      is = is + 1 == NSTG ? 0 : is + 1;
    }
  }
#undef PP8_MM
#undef PP8_PIECE
  pp8_wait_vm<0>();  // the redundant tail fetches must land before LDS is reused
  if (!lag) PP8_BARRIER();
  __syncthreads();
  if constexpr (PROF) t0 = __builtin_amdgcn_s_memtime();

  // ---- epilogue.  D^T layout: lane & 31 -> m, register quad q -> n = 8 q + 4 (lane >> 5) + e.  The tile goes through LDS as
  // fp32 [BM][128] (pp8_pos) so that a lane then owns 8 consecutive columns of one row: 16-byte residual loads and stores
  float* tile = (float*)smem;
  [[maybe_unused]] DrnDropRule drop{};
  // (descriptor of the transposed copy; a zero-size range when there is none)
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t ryt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(EPI == PP8_FC && p.YT ? p.YT : p.Y), 0, (EPI == PP8_FC && p.YT) ? (unsigned)((long)p.N * p.ldyt * 2) : 0u, 0x00020000);
  if constexpr (EPI == PP8_FC) drop = drn_drop_rule(p.seed + (p.seed_dev ? p.seed_dev[0] : 0ULL), p.drop_p);
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + i * 32 + l31, m = bm + row;
    const int nb0 = bn + wn * (32 * NJ) + j * 32;  // this wave's 32-column block: one 32-index dropout group per row (N % 32 == 0)
    // fc form, p == 0.5 and 32-aligned rows: ONE hash gives the keep bits of the lane's 16 elements of this block
    [[maybe_unused]] uint32_t kbits = 0xFFFFFFFFu;
    [[maybe_unused]] bool fast_bits = false;
    if constexpr (EPI == PP8_FC) {
      fast_bits = !p.mask && p.drop_p > 0.f && drop.half && (p.N & 31) == 0;
      if (fast_bits) kbits = drn_drop_bits32(drop, ((unsigned long long)m * p.N + nb0) >> 5);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = wn * (32 * NJ) + j * 32 + 8 * q + 4 * hi, nc = bn + nl;
      f32x4_t v;
      if constexpr (EPI == PP8_CONV) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c[i][j][4 * q + e];  // (the affine follows in the read-back phase: 8 channels per lane)
      } else {
        const f32x4_t bi = fcb[j][q];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = c[i][j][4 * q + e] + bi[e];
          if (p.relu) x = fmaxf(x, 0.f);
          v[e] = x;
        }
        if (fast_bits) {
          const uint32_t kb = kbits >> (8 * q + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= ((kb >> e) & 1u) ? drop.scale : 0.f;
        } else if (m < p.Mtot && nc < p.N) {
          const unsigned long long idx = (unsigned long long)m * p.N + nc;
          if (p.mask) {
            const f32x4_t mk = *(const f32x4_t*)(p.mask + idx);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= mk[e];
          } else if (p.drop_p > 0.f) {
            float dm[4];
            drn_drop_mult4(drop, idx, dm);  // (the rule of drn_common.h: the same mask as drn_bias_act_fwd draws)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= dm[e];
          }
        }
      }
      *(f32x4_t*)(tile + row * 128 + (pp8_pos(nl >> 2, row) << 2)) = v;
      if constexpr (EPI == PP8_FC) {
        if (p.YT) {
          // transposed copy [N][ldyt] straight from the accumulator layout (lane & 31 = row m, the quad's 4 registers = columns
          // nc .. nc + 3): neighbouring lanes exchange their bf16 values (one DPP move each), even lanes then hold the row PAIR
          // (m, m + 1) of columns nc, nc + 1 and odd lanes that of columns nc + 2, nc + 3 - 4-byte stores, the 16 lanes of a kind
          // write 64 contiguous bytes of one output row.  (Through the LDS tile - 8 scalar reads per 16-byte store, 32 output rows
          // x 32 bytes per store instruction - this copy cost more than the mainloop of a 32-slab GEMM saved.)
          // (packed: two conversions, two DPP moves, two selects and two byte permutes per four elements)
          const uint32_t p01 = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
          const uint32_t p23 = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
          const uint32_t o01 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p01, 0xB1, 0xF, 0xF, false);  // quad_perm [1, 0, 3, 2]: lane ^ 1
          const uint32_t o23 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p23, 0xB1, 0xF, 0xF, false);
          const bool odd = lane & 1;
          const int m0 = m - (odd ? 1 : 0), e0 = odd ? 2 : 0;
          const uint32_t lo_row = odd ? o23 : p01, hi_row = odd ? p23 : o01;  // values of rows m0 / m0 + 1 at this lane's two columns
          const uint32_t w0 = __builtin_amdgcn_perm(hi_row, lo_row, 0x05040100u);  // column e0:     [lo_row.lo16, hi_row.lo16]
          const uint32_t w1 = __builtin_amdgcn_perm(hi_row, lo_row, 0x07060302u);  // column e0 + 1: [lo_row.hi16, hi_row.hi16]
          if (nc < p.N && m0 < p.Mtot) {
            const unsigned off = (unsigned)(((long)(nc + e0) * p.ldyt + m0) * 2), ldb2 = (unsigned)(p.ldyt * 2);
            if (m0 + 1 < p.Mtot) {
              __builtin_amdgcn_raw_buffer_store_b32(w0, ryt, off, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b32(w1, ryt, off + ldb2, 0, 0);
            } else {
              __builtin_amdgcn_raw_buffer_store_b16((unsigned short)w0, ryt, off, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b16((unsigned short)w1, ryt, off + ldb2, 0, 0);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if (col_ok) {
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int row = rl + 32 * q, m = bm + row;
      if (m >= p.Mtot) break;
      const int pos = cl ^ (row & 15);
      const f32x4_t lo = *(const f32x4_t*)(tile + row * 128 + (pos << 2)), hi4 = *(const f32x4_t*)(tile + row * 128 + ((pos + 16) << 2));
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      if constexpr (EPI == PP8_CONV) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * sc8[e >> 2][e & 3] + bi8[e >> 2][e & 3];
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
      }
      i32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (int)((uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16));
      *(i32x4_t*)(p.Y + ((long)m * p.ldy + nn) * 2) = o;
    }
  }
  if constexpr (PROF) {
    t1 = __builtin_amdgcn_s_memtime();
    tp[5] = t1 - t0;
    if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) g_pp8_prof[wave][k] += tp[k];
      g_pp8_prof[wave][6] += (unsigned long long)n;
      g_pp8_prof[wave][7] += 1;
    }
  }
#undef PP8_CLK
}

template <int BM, int NSTG, int EPI, int VAR>
int launch_pp8(const Pp8Params& p, hipStream_t st) {
  const int tiles = ((p.Mtot + BM - 1) / BM) * ((p.N + 127) / 128);
  constexpr int smem = NSTG * (BM + 128) * 128;
  auto k = pp8_kernel<BM, NSTG, EPI, VAR>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return DRN_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(tiles), dim3(512), smem, st, p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int g_pp8 = 1;         // drn_tune(DRN_TUNE_PP8 = 25): 0 = off, 1 = default class, 2 = every layer in the kernel's class
int g_pp8_stages = 5;  // drn_tune(DRN_TUNE_PP8_STAGES = 26): LDS ring stages of the 128x128 form (3 / 4 / 5 = 1 / 2 / 3 slabs in flight)
int g_pp8_var = 1;     // drn_tune(DRN_TUNE_PP8_VARIANT = 27): schedule variant (pp8_kernel VAR), A/B knob
int g_pp8_wvar = 4;    // drn_tune(DRN_TUNE_PP8_WIDE_VARIANT = 30): VAR of the 256x128 form (1 = all DMA pieces in phase L1, 4 = no s_setprio, 8 = profile)
int g_pp8_wide = 1;    // drn_tune(DRN_TUNE_PP8_WIDE = 29): the 256x128 form: 0 = never, 1 = where it fills the chip, 2 = always

template <int EPI>
int launch_pp8_any(const Pp8Params& p, bool wide, hipStream_t st) {
  if (wide) {  // (variants of the 256x128 form: DRN_TUNE_PP8_WIDE_VARIANT)
    switch (g_pp8_wvar) {
      case 0: return launch_pp8<256, 3, EPI, 0>(p, st);
      case 1: return launch_pp8<256, 3, EPI, 1>(p, st);
      case 4: return launch_pp8<256, 3, EPI, 4>(p, st);
      case 8: return launch_pp8<256, 3, EPI, 8>(p, st);
      case 9: return launch_pp8<256, 3, EPI, 9>(p, st);
      case 13: return launch_pp8<256, 3, EPI, 13>(p, st);
      case 5: return launch_pp8<256, 3, EPI, 5>(p, st);
      default: return launch_pp8<256, 3, EPI, 4>(p, st);
    }
  }
#define PP8_CASE(S_, V_) if (g_pp8_stages == S_ && g_pp8_var == V_) return launch_pp8<128, S_, EPI, V_>(p, st)
  PP8_CASE(3, 0);
  PP8_CASE(4, 0); PP8_CASE(4, 1); PP8_CASE(4, 5);
  PP8_CASE(5, 0); PP8_CASE(5, 1); PP8_CASE(5, 2); PP8_CASE(5, 5);
  PP8_CASE(5, 8); PP8_CASE(5, 9); PP8_CASE(5, 10);  // (profile builds of the variants above)
#undef PP8_CASE
  return launch_pp8<128, 5, EPI, 1>(p, st);
}

// the 256x128 form where one image's layer gives it at least ~5/8 of the CUs' worth of tiles (its tiles are twice the work)
static bool pp8_wide_ok(long rows_one, int N, int cus) {
  if (g_pp8_wide == 0) return false;
  if (g_pp8_wide == 2) return true;
  const long t256 = ((rows_one + 255) / 256) * ((N + 127) / 128);
  return t256 * 8 >= 5L * cus;
}

}  // namespace

// (hidden: drn_tune in gemm_conv.hip)
__attribute__((visibility("hidden"))) int drn_pp8_set(int knob, int v) {
  if (knob == 25) {
    const int old = g_pp8;
    if (v >= 0 && v <= 2) g_pp8 = v;
    return old;
  }
  if (knob == 27) {
    const int old = g_pp8_var;
    if (v >= 0 && v <= 10 && (v & 3) != 3) g_pp8_var = v;
    return old;
  }
  if (knob == 28) {  // DRN_TUNE_PP8_PROFILE: print and clear the shader-clock split the profile variants (VAR & 8) accumulated
    unsigned long long h[8][8];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pp8_prof), sizeof(h)) != hipSuccess) return -1;
    for (int w = 0; w < 8; ++w) {
      const double nl = h[w][7] ? (double)h[w][7] : 1.0, ns = h[w][6] ? (double)h[w][6] : 1.0;
      fprintf(stderr, "pp8 profile wave %d: %llu launches, %.1f slabs each | per slab: read+issue %.0f  mid-barrier %.0f  mfma %.0f  "
                      "end-barrier %.0f cycles | per launch: prologue %.0f  epilogue %.0f\n",
              w, h[w][7], ns / nl, h[w][0] / ns, h[w][1] / ns, h[w][2] / ns, h[w][3] / ns, h[w][4] / nl, h[w][5] / nl);
    }
    memset(h, 0, sizeof(h));
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_pp8_prof), h, sizeof(h)) != hipSuccess) return -1;
    return 0;
  }
  if (knob == 30) {
    const int old = g_pp8_wvar;
    if (v == 0 || v == 1 || v == 4 || v == 5 || v == 8 || v == 9 || v == 13) g_pp8_wvar = v;
    return old;
  }
  if (knob == 29) {
    const int old = g_pp8_wide;
    if (v >= 0 && v <= 2) g_pp8_wide = v;
    return old;
  }
  const int old = g_pp8_stages;
  if (v == 3 || v == 4 || v == 5) g_pp8_stages = v;
  return old;
}

// Runs the convolution on the eight-wave kernel when it is in its class; DRN_ERR_UNSUPPORTED otherwise (drn_conv2d_nhwc_q then
// goes on to the other kernel families).  The class is decided on ONE image's geometry (`HoWo`), so a layer takes the same
// kernel whether its image runs alone or in a batch; every family gives the same bits anyway.
__attribute__((visibility("hidden"))) int drn_pp8_conv_try(const ConvParams& c, int dtype, int cus, hipStream_t st) {
  if (!g_pp8 || dtype != DRN_BF16 || c.out_dt != DRN_BF16 || (c.residual && c.res_dt != DRN_BF16)) return DRN_ERR_UNSUPPORTED;
  if ((c.Cin & 63) || (c.Cout & 7) || c.KH * c.KW > 32 || (c.ldy & 7) || (c.residual && (c.ldres & 7))) return DRN_ERR_UNSUPPORTED;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (!al16(c.X) || !al16(c.Wt) || !al16(c.Y) || (c.residual && !al16(c.residual)) || (c.ldw * 2) % 16 != 0 ||
      (c.scale && !al16(c.scale)) || (c.bias && !al16(c.bias)))
    return DRN_ERR_UNSUPPORTED;
  if ((long)c.Cout * c.ldw * 2 >= 0xFFFFFFF0L || (long)c.Nb * c.H * c.W * c.Cin * 2 >= 0xFFFFFFF0L) return DRN_ERR_UNSUPPORTED;
  const int nslab = c.KH * c.KW * (c.Cin >> 6);
  const long t128 = (((long)c.Ho * c.Wo + 127) / 128) * ((c.Cout + 127) / 128);
  if (g_pp8 == 1) {
    // measured class (tools/conv_bench.py / tools/pp8_probe.py at 800x1216, profiles/r6_*): one image's layer offers at least
    // 5/8 of the CUs a 128x128 tile and the K loop is long enough to amortise the ring's prologue.  (1x1 layers with >= 192 tiles
    // of 256x256 never get here: drn_conv2d_nhwc_q sends them to conv1x1_pp_kernel first - its tile moves half the bytes per MFMA.)
    if (nslab < 4 || t128 * 8 < 5L * cus || c.Cout < 128) return DRN_ERR_UNSUPPORTED;
  }
  Pp8Params p{};
  p.A = c.X; p.a_bytes = (unsigned)((long)c.Nb * c.H * c.W * c.Cin * 2); p.pix_b = (unsigned)(c.Cin * 2);
  p.B = c.Wt; p.ldb_b = (unsigned)(c.ldw * 2);
  p.Mtot = c.Nb * c.Ho * c.Wo; p.N = c.Cout;
  p.HoWo = c.Ho * c.Wo; p.Wo = c.Wo; p.H = c.H; p.W = c.W; p.KH = c.KH; p.KW = c.KW; p.stride = c.stride; p.pad = c.pad; p.dil = c.dil;
  p.spt = c.Cin >> 6;
  p.scale = c.scale; p.bias = c.bias; p.residual = c.residual; p.ldres = c.ldres; p.res_mult = c.res_mult; p.relu = c.relu;
  p.Y = c.Y; p.ldy = c.ldy;
  return launch_pp8_any<PP8_CONV>(p, pp8_wide_ok((long)c.Ho * c.Wo, c.Cout, cus), st);
}

extern "C" {

// relu_(fc(x)) + F.dropout(p) of DiscriminativeAdaptionNeck.forward (box_head.py:88-90) in ONE launch: out [M][ld_out] (bf16) =
// dropout(relu(A [M][lda] . W [N][ldw]^T + bias)), optionally also its transpose outT [N][ld_outT].  See include/drn_wsod.h.
int drn_linear_act_fwd(const void* A, const void* W, const float* bias, const float* mask, unsigned long long seed,
                       const unsigned long long* seed_dev, float drop_p, void* out, long ld_out, void* outT, long ld_outT,
                       int M, int N, int K, long lda, long ldw, int relu, void* stream) {
  if (!A || !W || !out || M < 0 || N < 0 || K <= 0 || lda < K || ldw < K || ld_out < N || (outT && ld_outT < M)) return DRN_ERR_ARG;
  if (M == 0 || N == 0) return DRN_OK;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (!g_pp8 || (K & 63) || (N & 7) || (lda * 2) % 16 != 0 || (ldw * 2) % 16 != 0 || (ld_out & 7) || (outT && (ld_outT & 7)) ||
      !al16(A) || !al16(W) || !al16(out) || (outT && !al16(outT)) || (bias && !al16(bias)) || (mask && !al16(mask)) ||
      (long)M * lda * 2 >= 0xFFFFFFF0L || (long)N * ldw * 2 >= 0xFFFFFFF0L)
    return DRN_ERR_UNSUPPORTED;
  Pp8Params p{};
  p.A = (const char*)A; p.a_bytes = (unsigned)((long)M * lda * 2); p.pix_b = (unsigned)(lda * 2);
  p.B = (const char*)W; p.ldb_b = (unsigned)(ldw * 2);
  p.Mtot = M; p.N = N;
  p.HoWo = M; p.Wo = M; p.H = 1; p.W = M; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1;
  p.spt = K >> 6;
  p.bias = bias; p.relu = relu;
  p.Y = (char*)out; p.ldy = ld_out; p.YT = (char*)outT; p.ldyt = ld_outT;
  p.mask = mask; p.seed = seed; p.seed_dev = seed_dev; p.drop_p = drop_p;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  static int cached = 0;
  if (!cached) {
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cached = prop.multiProcessorCount;
    else cached = 256;
  }
  cus = cached;
  return launch_pp8_any<PP8_FC>(p, pp8_wide_ok(M, N, cus), (hipStream_t)stream);
}

}  // extern "C"
