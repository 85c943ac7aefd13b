// MIL / OICR head kernels for gfx950 (latency-bound, fp32 arithmetic, deterministic reductions:
// fixed-order LDS trees, no float atomics).  Built with -ffp-contract=off.
//
// Replaces (reference file:line):
//   bias_act_fwd/bwd   relu_(fc(x)) + dropout and its autograd   box_head.py:82-91
//   wsddn_fwd_bwd      WSDDNOutputLayers.forward + predict_probs_img + BCE   fast_rcnn.py:493-527,
//                      :317-343, and the autograd of that chain (clamp gradient = 0 outside (1e-6,1-1e-6))
//   oicr_targets       get_pgt (roi_heads_oicr.py:491-567) + label_and_sample_proposals
//                      (roi_heads.py:255-353: pairwise_iou boxes.py:329-361, Matcher matcher.py:61-103)
//   softmax_ce         OICROutputs.softmax_cross_entropy_loss fast_rcnn.py:1087-1096,1128-1144,
//                      predict_probs :1561-1575 and its backward
//   apply_deltas       Box2BoxTransform.apply_deltas box_regression.py:73-110
//   sgd_step           torch.optim.SGD as built by detectron2/solver/build.py:93-137
#include "drn_common.h"
#include <stdlib.h>
#include <float.h>

namespace {

// (the counter-based dropout mask: drn_drop_rule / drn_drop_mult of drn_common.h)

struct ActParams {
  const float* in;     // fwd: split-K partials [splits][M][ld_in]; bwd: upstream grad [M][ld_in]
  int splits; long split_stride;
  const float* bias;   // fwd: [N] or null
  const float* mask;   // explicit dropout multipliers [M][N] or null
  unsigned long long seed; float drop_p;  // used when mask == null and drop_p > 0
  const unsigned long long* seed_dev;     // optional device counter added to `seed` (hipGraph-safe)
  const char* saved;   // bwd: forward output [M][ld_out] (post relu+dropout) or null (no activation)
  char* out; long ld_out;     // [M][ld_out] in out_dtype
  char* outT; long ld_outT;   // [N][ld_outT] in out_dtype or null
  float* colsum;       // bwd: [N] column sums (bias gradient) or null
  const float* colscale;  // bwd: per-column multiplier of grad_out (per-loss upstream grads) or null
  const int* colidx;      // bwd: [N] index into colscale per column (-1 = 0.0), or null (colscale is [N])
  float* colpart;      // bwd: [ceil(M/ACT_ROWS)][N] scratch for the two-stage column sums
  int M, N; long ld_in; int relu; int accumulate_colsum;
  int rows_fwd;  // fwd: rows per block (64, or 4 for skinny outputs without a transposed copy)
  int in_bf16;   // bwd: grad_out holds bf16 (gradients flowing through the conv trunk in the bf16 mode)
  // act_vec_kernel<true, KC > 0> (drn_gemm_nt_act_bwd): grad_out is not read but FORMED here, gA [M][lda] . gB [N][ldb]^T
  // over K = 64 KC (bf16, K-major rows)
  const bf16_t* gA; const bf16_t* gB; long lda, ldb;
};

// 64 columns x ROWS_PER_BLOCK rows per block (64x64 tiles through LDS for the transposed copy).
// bwd: per-block partial column sums go to colpart[blockIdx.y][N] (ceil(M/64) partials); colsum_reduce_kernel adds them in a
// fixed order (deterministic bias gradients, no float atomics).
constexpr int ACT_ROWS = 64;

template <int DT_OUT, int DT_SAVED, bool BWD>
__global__ __launch_bounds__(256) void act_kernel(ActParams p) {
  using EO = ElemOf<DT_OUT>;
  using TO = typename EO::type;
  using ES = ElemOf<DT_SAVED>;
  __shared__ float t[64][65];
  __shared__ float cs[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const int ROWS = BWD ? ACT_ROWS : p.rows_fwd;
  const int RSTEP = ROWS < 64 ? ROWS : 64;
  const int mb0 = blockIdx.y * ROWS;
  const int mb1 = min(mb0 + ROWS, p.M);
  float csum = 0.f;
  float cscale = 1.f;
  if (BWD && p.colscale && n < p.N) {
    const int ci = p.colidx ? p.colidx[n] : n;
    cscale = ci >= 0 ? p.colscale[ci] : 0.f;
  }
  const unsigned long long seed = p.seed + ((!BWD && p.seed_dev) ? p.seed_dev[0] : 0ULL);
  [[maybe_unused]] const DrnDropRule drop = drn_drop_rule(seed, p.drop_p);
  if (!BWD && p.seed_dev && p.drop_p <= 0.f && !p.mask && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    *const_cast<unsigned long long*>(p.seed_dev) += p.seed;  // no dropout here: this launch ADVANCES the mask counter
  for (int mb = mb0; mb < mb1; mb += 64) {
    // (all 16 row steps of a thread unrolled: the skinny backward head - dlogits [2000 x 103] -> dS, dS^T, column sums, 64
    // workgroups - is latency-bound, its loads should all be in flight)
    // backward of a single fp32 input: the thread's 16 rows are fetched up front, unconditionally (clamped indices) - under
    // the row guard below every load is a branch of its own and waits for the previous row's store
    float raw[16];
    const bool pre = BWD && !p.in_bf16 && p.splits == 1;
    if (pre) {
      const int nc = min(n, p.N - 1);
#pragma unroll
      for (int k = 0; k < 16; ++k) raw[k] = p.in[(long)min(mb + ty + 4 * k, p.M - 1) * p.ld_in + nc];
    }
#pragma unroll 16
    for (int k = 0; k < 16; ++k) {
      const int i = ty + 4 * k;
      if (i >= RSTEP) break;
      const int m = mb + i;
      float v = 0.f;
      if (m < p.M && n < p.N) {
        if (!BWD) {
          for (int s = 0; s < p.splits; ++s) v += p.in[(long)s * p.split_stride + (long)m * p.ld_in + n];
          if (p.bias) v += p.bias[n];
          if (p.relu) v = fmaxf(v, 0.f);
          if (p.mask) v *= p.mask[(long)m * p.N + n];
          else if (p.drop_p > 0.f) v *= drn_drop_mult(drop, (unsigned long long)m * p.N + n);
        } else {
          if (p.in_bf16) {
            v = bf16_to_f32(((const bf16_t*)p.in)[(long)m * p.ld_in + n]);
          } else if (pre) {
            v += raw[k];
          } else {
            for (int s = 0; s < p.splits; ++s) v += p.in[(long)s * p.split_stride + (long)m * p.ld_in + n];
          }
          v *= cscale;
          if (p.saved) {
            const float o = ES::ld((const typename ES::type*)p.saved + (long)m * p.ld_out + n);
            float mult = o > 0.f ? 1.f : 0.f;
            if (o > 0.f) {
              if (p.mask) mult = p.mask[(long)m * p.N + n];
              else if (p.drop_p > 0.f) mult = 1.f / (1.f - p.drop_p);
            }
            v *= mult;
          }
          csum += v;
        }
        if (p.out) EO::st((TO*)p.out + (long)m * p.ld_out + n, v);
      }
      t[i][tx] = v;
    }
    if (p.outT) {
      __syncthreads();
#pragma unroll 4
      for (int i = ty; i < 64; i += 4) {
        const int nn = blockIdx.x * 64 + i, m = mb + tx;
        if (nn < p.N && m < p.M) EO::st((TO*)p.outT + (long)nn * p.ld_outT + m, t[tx][i]);
      }
      __syncthreads();
    }
  }
  if (BWD && p.colpart) {
    cs[ty][tx] = csum;
    __syncthreads();
    if (ty == 0 && n < p.N) p.colpart[(long)blockIdx.y * p.N + n] = ((cs[0][tx] + cs[1][tx]) + cs[2][tx]) + cs[3][tx];
  }
}

// Vectorised variant for bf16 outputs (the hot forward / backward calls of fc6 and fc7): four columns per thread -
// 16-B loads of the fp32 input (per split), 8-B bf16 stores, the transposed copy leaves LDS in 16-B stores.  Same
// arithmetic per element as act_kernel (split partials summed in order, bias, ReLU, mask / counter-based dropout;
// backward: ReLU mask of the saved output, column scale, two-stage column sums); step +0.6 %.
// KC > 0 (backward only): the block first FORMS its 64 x 64 tile of grad_out = gA . gB^T over K = 64 KC with the MFMA the
// GEMM kernels use, k ascending (so every element is the bits drn_gemm_nt would have written), parks it in the LDS tile
// and runs the unchanged backward on it: the fp32 [M][N] round trip through HBM and one launch disappear.  For SKINNY K
// (the predictor's dX: K = 128, a 32-MB output for 2 GF): four waves, a 32 x 32 block each, operand fragments straight
// from global memory (16 B per lane and k-step, all of them in flight at once).
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool BWD, int KC = 0>
__global__ __launch_bounds__(256) void act_vec_kernel(ActParams p) {
  constexpr int LDS_AB = KC > 0 ? 2 * 64 * 128 * KC : 0, LDS_T = 64 * 68 * 4;
  __shared__ __attribute__((aligned(16))) char lds_ab[LDS_AB > LDS_T ? LDS_AB : LDS_T];
  float (*t)[68] = (float (*)[68])lds_ab;
  __shared__ float cs[16][64];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int n0 = blockIdx.x * 64 + tx * 4;
  const int mb0 = blockIdx.y * 64;
  const bool nok = n0 < p.N;  // N % 4 == 0: a column group is valid as a whole
  // backward: the saved outputs / mask of this thread's four rows, fetched up front (clamped indices, unconditional loads):
  // they do not depend on the gradient, and inside the row loop each row's loads would wait for the previous row's stores
  uint2 svp[4] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
  f32x4v mkp[4];
  if constexpr (BWD) {
    const int nc = min(n0, p.N - 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long mc = min(mb0 + ty + 16 * k, p.M - 1);
      mkp[k] = f32x4v{1.f, 1.f, 1.f, 1.f};
      if (p.saved) svp[k] = *(const uint2*)((const bf16_t*)p.saved + mc * p.ld_out + nc);
      if (p.mask) mkp[k] = *(const f32x4v*)(p.mask + mc * p.N + nc);
    }
  }
  if constexpr (KC > 0) {
    // operand tiles (64 rows x K of gA and of gB) through LDS: whole rows leave global memory in 16-byte pieces of
    // consecutive lanes (a fragment read straight from global touches 32 rows x 32 B per instruction - 25 us for this
    // launch, address-bound), the 16-byte k-slots of a row XOR-ed with its row so that the fragment reads - 32 rows at
    // one slot - cover all banks.  The fp32 tile `t` reuses the A tile's space once the MFMAs are done.
    constexpr int RB = 128 * KC, SL = 8 * KC;  // row bytes, 16-byte slots per row
    char* la = lds_ab;
    char* lb = lds_ab + 64 * RB;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wm = wv >> 1, wn = wv & 1;
    i32x4_t ga[2 * KC], gb[2 * KC];
#pragma unroll
    for (int j = 0; j < 2 * KC; ++j) {  // rows past M / N are clamped: they only reach tile rows / columns nobody uses
      const int c = threadIdx.x + 256 * j, row = c / SL, slot = c % SL;
      ga[j] = *(const i32x4_t*)(p.gA + (long)min(mb0 + row, p.M - 1) * p.lda + 8 * slot);
      gb[j] = *(const i32x4_t*)(p.gB + (long)min((int)blockIdx.x * 64 + row, p.N - 1) * p.ldb + 8 * slot);
    }
#pragma unroll
    for (int j = 0; j < 2 * KC; ++j) {
      const int c = threadIdx.x + 256 * j, row = c / SL, slot = c % SL;
      *(i32x4_t*)(la + row * RB + ((slot ^ (row & 7)) << 4)) = ga[j];
      *(i32x4_t*)(lb + row * RB + ((slot ^ (row & 7)) << 4)) = gb[j];
    }
    __syncthreads();
    const int ra = wm * 32 + (lane & 31), rb = wn * 32 + (lane & 31);
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4 * KC; ++ks) {
      const int slot = 2 * ks + (lane >> 5);
      const i32x4_t fa = *(const i32x4_t*)(la + ra * RB + ((slot ^ (ra & 7)) << 4));
      const i32x4_t fb = *(const i32x4_t*)(lb + rb * RB + ((slot ^ (rb & 7)) << 4));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, fb), acc, 0, 0, 0);
    }
    __syncthreads();  // everybody is done reading the operand tiles: `t` may overwrite them
    // D layout: lane -> n = lane & 31, register r -> m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) t[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][wn * 32 + (lane & 31)] = acc[r];
    __syncthreads();
  }
  float cscale[4] = {1.f, 1.f, 1.f, 1.f}, csum[4] = {0.f, 0.f, 0.f, 0.f};
  if (BWD && p.colscale && nok) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ci = p.colidx ? p.colidx[n0 + e] : n0 + e;
      cscale[e] = ci >= 0 ? p.colscale[ci] : 0.f;
    }
  }
  const unsigned long long seed = p.seed + ((!BWD && p.seed_dev) ? p.seed_dev[0] : 0ULL);
  [[maybe_unused]] const DrnDropRule drop = drn_drop_rule(seed, p.drop_p);
  if (!BWD && p.seed_dev && p.drop_p <= 0.f && !p.mask && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    *const_cast<unsigned long long*>(p.seed_dev) += p.seed;  // no dropout here: this launch ADVANCES the mask counter
  f32x4v bias4 = {0.f, 0.f, 0.f, 0.f};
  if (!BWD && p.bias && nok) bias4 = *(const f32x4v*)(p.bias + n0);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = ty + 16 * k, m = mb0 + i;
    f32x4v v = {0.f, 0.f, 0.f, 0.f};
    if (m < p.M && nok) {
      if (!BWD) {
        // (fetching the 4 rows x splits partials of a thread up front was tried here too: 17.6 -> 20.4 us - this pass is
        // bandwidth-bound and lives on occupancy, unlike the latency-bound small kernels)
        for (int s = 0; s < p.splits; ++s) v += *(const f32x4v*)(p.in + (long)s * p.split_stride + (long)m * p.ld_in + n0);
        if (p.bias) v += bias4;
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.mask) v *= *(const f32x4v*)(p.mask + (long)m * p.N + n0);
        else if (p.drop_p > 0.f) {
          float dm[4];
          drn_drop_mult4(drop, (unsigned long long)m * p.N + n0, dm);  // (N % 4 == 0 and n0 % 4 == 0: an aligned group)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= dm[e];
        }
      } else {
        if constexpr (KC > 0) v += *(const f32x4v*)&t[i][tx * 4];
        else
          for (int s = 0; s < p.splits; ++s) v += *(const f32x4v*)(p.in + (long)s * p.split_stride + (long)m * p.ld_in + n0);
        const f32x4v mk = mkp[k];
        const uint2 sv = svp[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] *= cscale[e];
          if (p.saved) {
            const float o = bf16_to_f32((bf16_t)((e < 2 ? sv.x : sv.y) >> (16 * (e & 1))));
            float mult = o > 0.f ? 1.f : 0.f;
            if (o > 0.f) {
              if (p.mask) mult = mk[e];
              else if (p.drop_p > 0.f) mult = 1.f / (1.f - p.drop_p);
            }
            v[e] *= mult;
          }
          csum[e] += v[e];
        }
      }
      if (p.out) {
        uint2 o;
        o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        *(uint2*)((bf16_t*)p.out + (long)m * p.ld_out + n0) = o;
      }
    }
    *(f32x4v*)&t[i][tx * 4] = v;
  }
  if (p.outT) {
    __syncthreads();
    const int c = threadIdx.x >> 2, rq = threadIdx.x & 3;  // 4 lanes x 32 B = one 128-B line of a row of outT
    const int nn = blockIdx.x * 64 + c, m = mb0 + rq * 16;
    if (nn < p.N && m < p.M) {
      uint32_t w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        w[k] = (uint32_t)f32_to_bf16(t[rq * 16 + 2 * k][c]) | ((uint32_t)f32_to_bf16(t[rq * 16 + 2 * k + 1][c]) << 16);
      bf16_t* dst = (bf16_t*)p.outT + (long)nn * p.ld_outT + m;
      if (m + 16 <= p.M) {
        ((i32x4_t*)dst)[0] = i32x4_t{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
        ((i32x4_t*)dst)[1] = i32x4_t{(int)w[4], (int)w[5], (int)w[6], (int)w[7]};
      } else {
        for (int k = 0; k < 16 && m + k < p.M; ++k) dst[k] = (bf16_t)(w[k >> 1] >> (16 * (k & 1)));
      }
    }
  }
  if (BWD && p.colpart) {
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[ty][tx * 4 + e] = csum[e];
    __syncthreads();
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x < 64 && n < p.N) {
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) a += cs[q][threadIdx.x];
      p.colpart[(long)blockIdx.y * p.N + n] = a;
    }
  }
}

// the vectorised kernel's preconditions (16-B / 8-B aligned rows everywhere it uses vector accesses)
static inline bool act_vec_ok(const ActParams& p, int out_dtype) {
  auto al = [](const void* q, uintptr_t a) { return (((uintptr_t)q) & (a - 1)) == 0; };
  return out_dtype == DRN_BF16 && !p.in_bf16 && p.rows_fwd == 64 && p.N % 4 == 0 && p.ld_in % 4 == 0 &&
         p.split_stride % 4 == 0 && al(p.in, 16) && (!p.bias || al(p.bias, 16)) && (!p.mask || al(p.mask, 16)) &&
         (!p.out || (p.ld_out % 4 == 0 && al(p.out, 8))) && (!p.saved || (p.ld_out % 4 == 0 && al(p.saved, 8))) &&
         (!p.outT || (p.ld_outT % 8 == 0 && al(p.outT, 16)));
}

__global__ void colsum_reduce_kernel(const float* colpart, int nparts, int N, float* colsum, int accumulate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  // sixteen partials in flight per trip (a load-add-load-add chain is one memory round trip per partial: this launch runs
  // beside the step's HBM-saturating fc6 dW GEMM, where a round trip is ~15 us - 530 us for 32 partials, which pushed the
  // small-tensor SGD behind it past the end of the GEMM and into the next step's start); same order of addition
  float s = 0.f;
  const float prev = accumulate ? colsum[n] : 0.f;
  for (int q0 = 0; q0 < nparts; q0 += 16) {
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = colpart[(long)min(q0 + k, nparts - 1) * N + n];
#pragma unroll
    for (int k = 0; k < 16; ++k) s = q0 + k < nparts ? s + t[k] : s;
  }
  colsum[n] = accumulate ? prev + s : s;
}

// ---------------------------------------------------------------- block reductions (1024 threads)
__device__ __forceinline__ float block_sum_1024(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  const int nw = blockDim.x >> 6;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

struct WsddnParams {
  const float* logits; long ld;   // [M][ld]; cls at col c_cls, det at col c_det
  int c_cls, c_det, K;
  const int* img_off;              // [n_img+1] row offsets
  const float* gt_onehot;          // [n_img][K]
  float* scores;                   // [M][K]
  float* rowsm;                    // [M][K] softmax over classes of the cls logits (kept for the backward)
  float* img_scores;               // [n_img][K] clamped
  float* loss_part;                // [n_img]
  float* dlogits; long ld_d;       // [M][ld_d] (same column offsets) or null
  int n_img; int mean_loss; float loss_scale;
  float* part; int max_blocks;     // [n_img][max_blocks][3][WS_KP] per-block column partials
};

// WSDDN forward + backward as three multi-block launches over row blocks of WS_ROWS rows (grid = blocks x images;
// round 2: 32-row blocks with a parallel combine of the block partials, 55 -> see profiles/r2_09 us for the three stages):
//   A: a = softmax over classes (shuffle reduction inside the LPR lanes that own a row) -> rowsm; per-block online
//      column-softmax partials (block max, block sum of exp relative to it)
//   B: every block re-combines the partials of its image in a fixed order (cmax, csum), writes the scores and its
//      partial column sums
//   C: combine the score sums, image scores / BCE / d loss (first block of the image), analytic backward
// Rows are owned by LPR consecutive lanes (32 for K <= 32, else 64 with two columns per lane); all reductions have
// a fixed order, so the result does not depend on scheduling.
constexpr int WS_ROWS = 32;   // rows per block: 63 blocks for a 2000-proposal image, four row passes per block and stage
constexpr int WS_KP = 128;  // padded column count of the partial buffers

// Where a stage reads its logits: the [M][ld] matrix, or (launch A of drn_mil_oicr_losses) the predictor GEMM's split-K
// partials - summed in split order, plus the bias, exactly what drn_bias_act_fwd writes - with the logit stored on the way,
// so that each one is formed once, by the thread that needs it first.
struct LogitsSrc {
  const float* part; int splits; long split_stride, ld_part;  // [splits][M][ld_part] partial sums of H2 . Wh^T
  const float* bias;                                          // [NH] or null
  float* logits; long ld;                                     // [M][ld]
  unsigned long long* seed_dev; unsigned long long seed_inc;  // the dropout counter this pass advances (or null)
};
// FROM_PARTS: 0 = the logits matrix, 1 = split-K partials with splits <= 8 (straight-line: all loads of all the caller's
// logits can be in flight together), 2 = any number of splits
template <int FROM_PARTS>
__device__ __forceinline__ float logit_at(const LogitsSrc* s, const float* logits, long ld, int r, int c) {
  if (FROM_PARTS == 0) return logits[(long)r * ld + c];
  // eight partials in flight at a time (a runtime-length loop of load + add waits for every load in turn); the caller
  // stores the logit (logit_put) once all of its loads are issued - a store in between would fence the later loads.
  // (loads are unconditional, from clamped indices - a predicated load is a branch, and the scheduler does not gather
  // loads across branches; what an out-of-range slot fetched is dropped by a select)
  const float* src = s->part + (long)r * s->ld_part + c;
  const float b = s->bias ? s->bias[c] : 0.f;
  float v = 0.f;
  const int nq = FROM_PARTS == 1 ? 1 : s->splits;
  for (int q0 = 0; q0 < nq; q0 += 8) {
    float t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = src[(long)min(q0 + q, s->splits - 1) * s->split_stride];
#pragma unroll
    for (int q = 0; q < 8; ++q) v = q0 + q < s->splits ? v + t[q] : v;
  }
  if (s->bias) v += b;
  return v;
}
template <int FROM_PARTS>
__device__ __forceinline__ void logit_put(const LogitsSrc* s, int r, int c, float v) {
  if (FROM_PARTS) s->logits[(long)r * s->ld + c] = v;
}

// (K > 32 - the 80-class configs - runs 64 lanes x 2 columns per row; with 256 threads that was 4 rows per pass and 16-deep
// serial loops over the 63 block partials: 28-30 us per stage against 10-11 us at K = 20.  Those shapes now take 1024
// threads - 16 rows per pass, 4-deep loops; same per-row lane reductions, the block combine adds 16 phase partials
// instead of 4: another fixed summation order for K > 32.)
template <int LPR>
struct WsLanes {
  static constexpr int NT = LPR == 64 ? 1024 : 256;
  static constexpr int RPP = NT / LPR, CPL = LPR == 64 ? 2 : 1;
  static __device__ __forceinline__ float gmax(float v) { for (int o = LPR / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, LPR)); return v; }
  static __device__ __forceinline__ float gsum(float v) { for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPR); return v; }
};

// block-level combine of per-phase column partials (fixed order over phases)
template <int LPR, bool IS_MAX>
__device__ __forceinline__ void ws_colreduce(float (&v)[WsLanes<LPR>::CPL], const int (&col)[WsLanes<LPR>::CPL],
                                             int ph, float (*red)[LPR * WsLanes<LPR>::CPL]) {
  using L = WsLanes<LPR>;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < L::CPL; ++j) red[ph][col[j]] = v[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < L::CPL; ++j) {
    float acc = IS_MAX ? -FLT_MAX : 0.f;
    for (int q = 0; q < L::RPP; ++q) acc = IS_MAX ? fmaxf(acc, red[q][col[j]]) : acc + red[q][col[j]];
    v[j] = acc;
  }
}

// cmax / csum of image `img` from the per-block partials, the same way in every block: phase ph takes the blocks ph,
// ph + RPP, ... and the RPP phase results are merged in phase order through LDS (a serial walk over the 63 blocks of a
// 2000-proposal image would be 126 dependent loads at the head of stages 1 and 2)
template <int LPR>
__device__ __forceinline__ void ws_combine(const WsddnParams& p, int img, int nb, const int (&col)[WsLanes<LPR>::CPL],
                                           const bool (&ok)[WsLanes<LPR>::CPL], int ph,
                                           float (*red)[LPR * WsLanes<LPR>::CPL], float (&cmax)[WsLanes<LPR>::CPL],
                                           float (&csum)[WsLanes<LPR>::CPL]) {
  using L = WsLanes<LPR>;
  const float* part = p.part + (long)img * p.max_blocks * 3 * WS_KP;
  // (eight partial blocks per trip with their loads issued together: same order of max / add as a plain loop over q)
#pragma unroll
  for (int j = 0; j < L::CPL; ++j) {
    cmax[j] = -FLT_MAX;
    {  // (lanes of columns >= K read column K-1: their results are never used)
      for (int q0 = ph; q0 < nb; q0 += 8 * L::RPP) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = part[(min(q0 + k * L::RPP, nb - 1) * 3 + 0) * WS_KP + min(col[j], p.K - 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k) cmax[j] = q0 + k * L::RPP < nb ? fmaxf(cmax[j], t[k]) : cmax[j];
      }
    }
  }
  ws_colreduce<LPR, true>(cmax, col, ph, red);
#pragma unroll
  for (int j = 0; j < L::CPL; ++j) {
    csum[j] = 0.f;
    {
      for (int q0 = ph; q0 < nb; q0 += 8 * L::RPP) {
        float t0[8], t1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int q = min(q0 + k * L::RPP, nb - 1);
          t0[k] = part[(q * 3 + 0) * WS_KP + min(col[j], p.K - 1)];
          t1[k] = part[(q * 3 + 1) * WS_KP + min(col[j], p.K - 1)];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) csum[j] = q0 + k * L::RPP < nb ? csum[j] + t1[k] * expf(t0[k] - cmax[j]) : csum[j];
      }
    }
  }
  ws_colreduce<LPR, false>(csum, col, ph, red);
}

// Every stage is a handful of DEPENDENT memory round trips, not arithmetic: a thread's WS_ROWS / RPP rows are fetched
// together, ahead of the cross-block combine, and kept in registers (stage 0 used to read its det logits twice, stages 1 / 2
// took one round trip per row pass).  Same operations on the same values in the same order as the row-by-row form.
template <int LPR, int STAGE, int FROM_PARTS = 0>
__device__ __forceinline__ void wsddn_stage_body(const WsddnParams& p, const LogitsSrc* src = nullptr) {
  using L = WsLanes<LPR>;
  constexpr int RPP = L::RPP, CPL = L::CPL, NP = WS_ROWS / RPP;
  __shared__ float red[RPP][LPR * CPL];
  const int img = blockIdx.y, blk = blockIdx.x;
  const int r0 = p.img_off[img], r1 = p.img_off[img + 1];
  const int nb = (r1 - r0 + WS_ROWS - 1) / WS_ROWS;
  if (blk >= nb) return;
  const int rb0 = r0 + blk * WS_ROWS, rb1 = min(rb0 + WS_ROWS, r1);
  const int K = p.K;
  const int l = threadIdx.x % LPR, ph = threadIdx.x / LPR;
  bool ok[CPL];
  int col[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) { col[j] = l + j * LPR; ok[j] = col[j] < K; }
  float* part = p.part + ((long)img * p.max_blocks + blk) * 3 * WS_KP;
  if (STAGE == 0) {
    float bmax[CPL], bsum[CPL], xc[NP][CPL], xd[NP][CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) { bmax[j] = -FLT_MAX; bsum[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int r = rb0 + ph + i * RPP, rc = min(r, rb1 - 1);  // clamped: unconditional loads, dropped by the selects
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const bool v = ok[j] && r < rb1;
        const int cc = min(col[j], K - 1);
        const float a = logit_at<FROM_PARTS>(src, p.logits, p.ld, rc, p.c_cls + cc);
        const float d = logit_at<FROM_PARTS>(src, p.logits, p.ld, rc, p.c_det + cc);
        xc[i][j] = v ? a : -FLT_MAX;
        xd[i][j] = v ? d : -FLT_MAX;
      }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int r = rb0 + ph + i * RPP;
      if (r >= rb1) continue;
      float e[CPL], mx = -FLT_MAX;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        if (ok[j]) {
          logit_put<FROM_PARTS>(src, r, p.c_cls + col[j], xc[i][j]);
          logit_put<FROM_PARTS>(src, r, p.c_det + col[j], xd[i][j]);
          bmax[j] = fmaxf(bmax[j], xd[i][j]);
        }
        mx = fmaxf(mx, xc[i][j]);
      }
      mx = L::gmax(mx);
      float se = 0.f;
#pragma unroll
      for (int j = 0; j < CPL; ++j) { e[j] = ok[j] ? expf(xc[i][j] - mx) : 0.f; se += e[j]; }
      se = L::gsum(se);
#pragma unroll
      for (int j = 0; j < CPL; ++j) if (ok[j]) p.rowsm[(long)r * K + col[j]] = e[j] / se;
    }
    ws_colreduce<LPR, true>(bmax, col, ph, red);
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (rb0 + ph + i * RPP < rb1)
#pragma unroll
        for (int j = 0; j < CPL; ++j)
          if (ok[j]) bsum[j] += expf(xd[i][j] - bmax[j]);
    ws_colreduce<LPR, false>(bsum, col, ph, red);
    if (ph == 0)
#pragma unroll
      for (int j = 0; j < CPL; ++j) if (ok[j]) { part[0 * WS_KP + col[j]] = bmax[j]; part[1 * WS_KP + col[j]] = bsum[j]; }
    return;
  }
  // this thread's rows, ahead of the combine: det logits and row softmax (stages 1, 2), scores (stage 2)
  float xd[NP][CPL], rs[NP][CPL], sc[NP][CPL];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int r = rb0 + ph + i * RPP, rc = min(r, rb1 - 1);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const bool v = ok[j] && r < rb1;
      const int cc = min(col[j], K - 1);
      const float a = p.logits[(long)rc * p.ld + p.c_det + cc], b = p.rowsm[(long)rc * K + cc];
      const float c = STAGE == 2 ? p.scores[(long)rc * K + cc] : 0.f;
      xd[i][j] = v ? a : 0.f;
      rs[i][j] = v ? b : 0.f;
      sc[i][j] = v ? c : 0.f;
    }
  }
  float cmax[CPL], csum[CPL];
  ws_combine<LPR>(p, img, nb, col, ok, ph, red, cmax, csum);
  if (STAGE == 1) {
    float S[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) S[j] = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int r = rb0 + ph + i * RPP;
      if (r >= rb1) continue;
#pragma unroll
      for (int j = 0; j < CPL; ++j)
        if (ok[j]) {
          const float b = expf(xd[i][j] - cmax[j]) / csum[j];
          const float s_ = rs[i][j] * b;
          p.scores[(long)r * K + col[j]] = s_;
          S[j] += s_;
        }
    }
    ws_colreduce<LPR, false>(S, col, ph, red);
    if (ph == 0)
#pragma unroll
      for (int j = 0; j < CPL; ++j) if (ok[j]) part[2 * WS_KP + col[j]] = S[j];
    return;
  }
  // STAGE 2: image scores, BCE, d loss / d S_c (clamp passes gradient only inside [1e-6, 1 - 1e-6]), backward
  const float* ipart = p.part + (long)img * p.max_blocks * 3 * WS_KP;
  const float norm = (p.mean_loss ? 1.f / (float)(p.n_img * K) : 1.f) / (float)p.n_img;
  float S[CPL], g[CPL], lsum = 0.f;
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    S[j] = 0.f;
    for (int q0 = ph; q0 < nb; q0 += 8 * RPP) {
      float t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = ipart[(min(q0 + k * RPP, nb - 1) * 3 + 2) * WS_KP + min(col[j], K - 1)];
#pragma unroll
      for (int k = 0; k < 8; ++k) S[j] = q0 + k * RPP < nb ? S[j] + t[k] : S[j];
    }
  }
  ws_colreduce<LPR, false>(S, col, ph, red);
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    g[j] = 0.f;
    if (!ok[j]) continue;
    const float s_ = fminf(fmaxf(S[j], 1e-6f), 1.0f - 1e-6f);
    const float y = p.gt_onehot[img * K + col[j]];
    // F.binary_cross_entropy: -(y*log(s) + (1-y)*log(1-s)), logs clamped at -100
    lsum += -(y * fmaxf(logf(s_), -100.f) + (1.f - y) * fmaxf(logf(1.f - s_), -100.f));
    g[j] = (S[j] >= 1e-6f && S[j] <= 1.0f - 1e-6f) ? (-(y / s_) + (1.f - y) / (1.f - s_)) * norm * p.loss_scale : 0.f;
    if (blk == 0 && ph == 0) p.img_scores[img * K + col[j]] = s_;
  }
  lsum = L::gsum(lsum);
  if (blk == 0 && threadIdx.x == 0) p.loss_part[img] = lsum * norm;
  if (!p.dlogits) return;
  // d cls = g_c*s - a*dot, dot = sum_k g_k s_rk;  d det = g_c*(s - b*S_c)
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int r = rb0 + ph + i * RPP;
    if (r >= rb1) continue;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) dot += g[j] * sc[i][j];
    dot = L::gsum(dot);
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (ok[j]) {
        const float b = expf(xd[i][j] - cmax[j]) / csum[j];
        p.dlogits[(long)r * p.ld_d + p.c_cls + col[j]] = g[j] * sc[i][j] - rs[i][j] * dot;
        p.dlogits[(long)r * p.ld_d + p.c_det + col[j]] = g[j] * (sc[i][j] - b * S[j]);
      }
  }
}

template <int LPR, int STAGE>
__global__ __launch_bounds__(WsLanes<LPR>::NT) void wsddn_stage_kernel(WsddnParams p) { wsddn_stage_body<LPR, STAGE>(p); }

struct TargetParams {
  const float* prev_scores; long ld_s;  // [M][ld_s], class columns 0..K-1 (bg column, if any, ignored)
  const float* prev_boxes; int box_cols;  // [M][box_cols], box_cols = 4 or 4K
  int zero_delta_decode;               // 1: pgt box = apply_deltas(0, prev_boxes[idx]) (box_cols must be 4)
  const float* props;                  // [M][4]
  const int* img_off;                  // [n_img+1]
  const int* gt_classes; const int* gt_count; int gmax;  // [n_img][gmax], [n_img]
  const float* img_scores;             // [n_img][K]
  int K;
  float thr[3]; int lab[4]; int nthr;  // Matcher thresholds/labels (config IOU_THRESHOLDS / IOU_LABELS)
  int* labels; float* weights; int* matched; float* gt_boxes;  // [M], [M], [M], [M][4]
  int* pgt_idx; float* pgt_boxes;      // [n_img][gmax], [n_img][gmax][4]
};

constexpr int MAX_CHAIN_HEADS = 8;
struct TargetMulti { TargetParams h[MAX_CHAIN_HEADS]; };

__device__ __forceinline__ void oicr_targets_body(const TargetParams& p) {
  __shared__ float sv[128][17];
  __shared__ int si[128][17];
  __shared__ float gbox[128][4];
  __shared__ float gw[128];
  __shared__ int gcls[128];
  const int img = blockIdx.x;
  const int r0 = p.img_off[img], r1 = p.img_off[img + 1];
  const int G = p.gt_count[img];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // this thread's first two proposals (all of them for images of <= 2048 proposals): fetched now, used by the labelling
  // pass at the end - they depend on nothing that is computed here
  float pb[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long rc = max(min(r0 + (int)threadIdx.x + 1024 * i, r1 - 1), 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) pb[i][e] = p.props[4 * rc + e];
  }
  // Mining: argmax over the image's rows of the previous scores in each ground-truth class column (first index on
  // ties).  Four classes per trip with their loads in flight together, the 16 wave results of every class parked in LDS,
  // then ONE thread per class finishes its class (the serial form - a class after the other, thread 0 fetching the box of
  // each - was ~3 dependent memory round trips per class).
  for (int g0 = 0; g0 < G; g0 += 4) {
    int cls[4], bi[4];
    float best[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      cls[k] = p.gt_classes[img * p.gmax + min(g0 + k, G - 1)];
      best[k] = -FLT_MAX; bi[k] = 0x7fffffff;
    }
    for (int r = r0 + threadIdx.x; r < r1; r += 2048) {
      float va[4], vb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = p.prev_scores[(long)r * p.ld_s + cls[k]];
        const float b = p.prev_scores[(long)min(r + 1024, r1 - 1) * p.ld_s + cls[k]];
        va[k] = g0 + k < G ? a : -FLT_MAX;
        vb[k] = (g0 + k < G && r + 1024 < r1) ? b : -FLT_MAX;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // ascending r per thread => first index kept on ties
        if (va[k] > best[k]) { best[k] = va[k]; bi[k] = r; }
        if (r + 1024 < r1 && vb[k] > best[k]) { best[k] = vb[k]; bi[k] = r + 1024; }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (g0 + k >= G) break;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best[k], o, 64); const int oi = __shfl_xor(bi[k], o, 64);
        if (ov > best[k] || (ov == best[k] && oi < bi[k])) { best[k] = ov; bi[k] = oi; }
      }
      if (lane == 0) { sv[g0 + k][w] = best[k]; si[g0 + k][w] = bi[k]; }
      if (threadIdx.x == 0) gcls[g0 + k] = cls[k];
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const int g = threadIdx.x;
    const int cls = gcls[g];
    float best = sv[g][0]; int bi = si[g][0];
    for (int q = 1; q < 16; ++q) if (sv[g][q] > best || (sv[g][q] == best && si[g][q] < bi)) { best = sv[g][q]; bi = si[g][q]; }
    if (bi == 0x7fffffff) bi = r0;
    const float* bx = p.prev_boxes + (long)bi * p.box_cols + (p.box_cols == 4 ? 0 : 4 * cls);
    float bb[4] = {bx[0], bx[1], bx[2], bx[3]};
    if (p.zero_delta_decode) {
      // Box2BoxTransform.apply_deltas with all-zero deltas (box_regression.py:73-110), op for op: what a
      // non-regressing refinement head hands to the next stage (equal to the proposal up to 1 ulp)
      const float w = bb[2] - bb[0], h = bb[3] - bb[1];
      const float cx = bb[0] + 0.5f * w, cy = bb[1] + 0.5f * h;
      const float pcx = 0.f * w + cx, pcy = 0.f * h + cy;
      const float pw = expf(0.f) * w, ph = expf(0.f) * h;
      bb[0] = pcx - 0.5f * pw; bb[1] = pcy - 0.5f * ph; bb[2] = pcx + 0.5f * pw; bb[3] = pcy + 0.5f * ph;
    }
    for (int e = 0; e < 4; ++e) { gbox[g][e] = bb[e]; p.pgt_boxes[((long)img * p.gmax + g) * 4 + e] = bb[e]; }
    gw[g] = p.img_scores[img * p.K + cls];
    p.pgt_idx[img * p.gmax + g] = bi - r0;
  }
  __syncthreads();
  auto label_row = [&](int r, float x1, float y1, float x2, float y2) {
    const float a2 = (x2 - x1) * (y2 - y1);
    float best = -1.f; int bg = 0;
    for (int g = 0; g < G; ++g) {
      const float a1 = (gbox[g][2] - gbox[g][0]) * (gbox[g][3] - gbox[g][1]);
      const float iw = fmaxf(fminf(gbox[g][2], x2) - fmaxf(gbox[g][0], x1), 0.f);
      const float ih = fmaxf(fminf(gbox[g][3], y2) - fmaxf(gbox[g][1], y1), 0.f);
      const float inter = iw * ih;
      const float iou = inter > 0.f ? inter / (a1 + a2 - inter) : 0.f;
      if (iou > best) { best = iou; bg = g; }
    }
    int ml = 1;  // Matcher: labels default 1, then per-interval assignment over [-inf, thr.., +inf]
    if (G == 0) { ml = p.lab[0]; best = 0.f; }
    else {
      for (int q = 0; q <= p.nthr; ++q) {
        const float lo = q == 0 ? -INFINITY : p.thr[q - 1];
        const float hi = q == p.nthr ? INFINITY : p.thr[q];
        if (best >= lo && best < hi) ml = p.lab[q];
      }
    }
    int cls = G > 0 ? gcls[bg] : p.K;
    if (ml == 0) cls = p.K;
    if (ml == -1) cls = -1;
    p.labels[r] = cls;
    p.matched[r] = bg;
    p.weights[r] = (G > 0 && cls != -1) ? gw[bg] : 0.f;
    for (int e = 0; e < 4; ++e) p.gt_boxes[4 * (long)r + e] = G > 0 ? gbox[bg][e] : 0.f;
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r0 + threadIdx.x + 1024 * i;
    if (r < r1) label_row(r, pb[i][0], pb[i][1], pb[i][2], pb[i][3]);
  }
  for (int r = r0 + threadIdx.x + 2048; r < r1; r += 1024)
    label_row(r, p.props[4 * (long)r], p.props[4 * (long)r + 1], p.props[4 * (long)r + 2], p.props[4 * (long)r + 3]);
}

__global__ __launch_bounds__(1024) void oicr_targets_kernel(TargetParams p) { oicr_targets_body(p); }
// all refinement heads at once (blockIdx.y = head): head k mines its pseudo ground truth from the probabilities of head
// k-1, which only need the logits of the ONE predictor GEMM - no dependency between the heads' launches
__global__ __launch_bounds__(1024) void oicr_targets_multi_kernel(TargetMulti mp) { oicr_targets_body(mp.h[blockIdx.y]); }

struct CeParams {
  const float* logits; long ld; int col0; int C;  // C = K+1 columns starting at col0
  const int* labels; const float* weights;         // [M] (null => inference: probs only)
  float* probs;                                    // [M][C]
  float* dlogits; long ld_d;                       // [M][ld_d] at col0, or null
  float* loss;                                     // scalar
  int M; float loss_scale;
};

// One wave per row (lane = class column, C <= 128), CE_ROWS rows per 256-thread block.  Kernel 1 writes the
// probabilities and per-block partial (sum w*CE, #valid); kernel 2 re-adds the partials in a fixed order
// (every block gets the same total), writes the loss and the gradient of the logits.
constexpr int CE_ROWS = 16;

struct CeMulti { CeParams h[MAX_CHAIN_HEADS]; float* partial[MAX_CHAIN_HEADS]; };

template <int FROM_PARTS = 0>
__device__ __forceinline__ void ce_rows_body(const CeParams& p, float* partial, const LogitsSrc* src = nullptr) {
  __shared__ float sl[4], sv[4];
  constexpr int NP = CE_ROWS / 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float lsum = 0.f, vsum = 0.f;
  // the wave's four rows are fetched together (one memory round trip instead of four), then the label's logit
  float x0[NP], x1[NP], wt[NP], xl[NP];
  int lab[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int r = blockIdx.x * CE_ROWS + w + 4 * k;
    const bool v = r < p.M;
    const int rc = min(r, p.M - 1);  // clamped: unconditional loads, dropped by the selects
    const float a = logit_at<FROM_PARTS>(src, p.logits, p.ld, rc, p.col0 + min(lane, p.C - 1));
    x0[k] = v && lane < p.C ? a : -FLT_MAX;
    x1[k] = -FLT_MAX;
    if (p.C > 64) {  // uniform
      const float b = logit_at<FROM_PARTS>(src, p.logits, p.ld, rc, p.col0 + min(lane + 64, p.C - 1));
      x1[k] = v && lane + 64 < p.C ? b : -FLT_MAX;
    }
    lab[k] = -1; wt[k] = 0.f;
    if (p.labels) {  // uniform
      const int lb = p.labels[rc];
      const float wb = p.weights[rc];
      lab[k] = v ? lb : -1; wt[k] = v ? wb : 0.f;
    }
  }
  if (p.labels) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int rc = min(blockIdx.x * CE_ROWS + w + 4 * k, p.M - 1);
      const float a = p.logits[(long)rc * p.ld + p.col0 + max(lab[k], 0)];
      xl[k] = lab[k] >= 0 ? a : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int r = blockIdx.x * CE_ROWS + w + 4 * k;
    if (r >= p.M) break;
    if (lane < p.C) logit_put<FROM_PARTS>(src, r, p.col0 + lane, x0[k]);
    if (lane + 64 < p.C) logit_put<FROM_PARTS>(src, r, p.col0 + lane + 64, x1[k]);
    const float mx = wave_max(fmaxf(x0[k], x1[k]));
    const float e0 = lane < p.C ? expf(x0[k] - mx) : 0.f, e1 = lane + 64 < p.C ? expf(x1[k] - mx) : 0.f;
    const float se = wave_sum(e0 + e1);
    if (lane < p.C) p.probs[(long)r * p.C + lane] = e0 / se;
    if (lane + 64 < p.C) p.probs[(long)r * p.C + lane + 64] = e1 / se;
    if (p.labels && lane == 0) {
      const float wk = lab[k] == -1 ? 0.f : wt[k];
      if (lab[k] >= 0) lsum += (logf(se) - (xl[k] - mx)) * wk;
      vsum += wk > 1e-12f ? 1.f : 0.f;
    }
  }
  if (!p.labels) return;
  if (lane == 0) { sl[w] = lsum; sv[w] = vsum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = ((sl[0] + sl[1]) + sl[2]) + sl[3];
    partial[2 * blockIdx.x + 1] = ((sv[0] + sv[1]) + sv[2]) + sv[3];
  }
}
__global__ __launch_bounds__(256) void ce_rows_kernel(CeParams p, float* partial) { ce_rows_body<0>(p, partial); }
__global__ __launch_bounds__(256) void ce_rows_multi_kernel(CeMulti mp) { ce_rows_body<0>(mp.h[blockIdx.y], mp.partial[blockIdx.y]); }

__device__ __forceinline__ void ce_grad_body(const CeParams& p, const float* partial, int nparts) {
  __shared__ float sh[2][256];
  constexpr int NP = CE_ROWS / 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // the rows' labels / weights / probabilities first: they do not depend on the combine below
  int lab[NP];
  float wt[NP], pr0[NP], pr1[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int r = blockIdx.x * CE_ROWS + w + 4 * k;
    const bool v = r < p.M;
    const int rc = min(r, p.M - 1);
    lab[k] = -1; wt[k] = 0.f; pr0[k] = 0.f; pr1[k] = 0.f;
    if (p.dlogits) {  // uniform
      const int lb = p.labels[rc];
      const float wb = p.weights[rc], a = p.probs[(long)rc * p.C + min(lane, p.C - 1)];
      lab[k] = v ? lb : -1; wt[k] = v ? wb : 0.f; pr0[k] = v ? a : 0.f;
      if (p.C > 64) pr1[k] = p.probs[(long)rc * p.C + min(lane + 64, p.C - 1)];
    }
  }
  float l = 0.f, v = 0.f;
  for (int q = threadIdx.x; q < nparts; q += 256) { l += partial[2 * q]; v += partial[2 * q + 1]; }
  sh[0][threadIdx.x] = l; sh[1][threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
    __syncthreads();
  }
  const float L = sh[0][0], V = sh[1][0];
  if (blockIdx.x == 0 && threadIdx.x == 0) p.loss[0] = L / V;
  if (!p.dlogits) return;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int r = blockIdx.x * CE_ROWS + w + 4 * k;
    if (r >= p.M) break;
    const float f = (lab[k] < 0 ? 0.f : wt[k]) / V * p.loss_scale;
    float* d = p.dlogits + (long)r * p.ld_d + p.col0;
    if (lane < p.C) d[lane] = f * (pr0[k] - (lane == lab[k] ? 1.f : 0.f));
    if (lane + 64 < p.C) d[lane + 64] = f * (pr1[k] - (lane + 64 == lab[k] ? 1.f : 0.f));
  }
}
__global__ __launch_bounds__(256) void ce_grad_kernel(CeParams p, const float* partial, int nparts) { ce_grad_body(p, partial, nparts); }
__global__ __launch_bounds__(256) void ce_grad_multi_kernel(CeMulti mp, int nparts) {
  ce_grad_body(mp.h[blockIdx.y], mp.partial[blockIdx.y], nparts);
}

// ---------------------------------------------------------------- the loss tail in six launches (drn_mil_oicr_losses)
// Launch A = WSDDN stage 0 and the softmax of every refinement head reading their logits from the predictor GEMM's split-K
// partials (LogitsSrc): the separate reduce + bias launch and the separate softmax launch are gone, each logit is formed
// and stored exactly once by the thread that needs it first.
template <int LPR>
__global__ __launch_bounds__(WsLanes<LPR>::NT) void mil_stage0_kernel(WsddnParams p, CeMulti cm, LogitsSrc src, int nb_ws,
                                                                       int nb_ce) {
  const int y = blockIdx.y;
  if (y == 0 && blockIdx.x == 0 && threadIdx.x == 0 && src.seed_dev) *src.seed_dev += src.seed_inc;
  if (y < p.n_img) {
    if ((int)blockIdx.x >= nb_ws) return;
    if (src.splits <= 8) wsddn_stage_body<LPR, 0, 1>(p, &src);
    else wsddn_stage_body<LPR, 0, 2>(p, &src);
    return;
  }
  if ((int)blockIdx.x >= nb_ce || threadIdx.x >= 256) return;  // the softmax body is written for four waves
  if (src.splits <= 8) ce_rows_body<1>(cm.h[y - p.n_img], nullptr, &src);
  else ce_rows_body<2>(cm.h[y - p.n_img], nullptr, &src);
}

// OICROutputs.box_reg_loss (fast_rcnn.py:1146-1211): foreground rows only, class-specific columns 4c..4c+3,
// target = Box2BoxTransform.get_deltas(proposal, matched pseudo-GT box) (box_regression.py:38-71), smooth-L1 with
// beta = 0 (= L1), summed and divided by the number of proposals.  Two launches like the CE pair: per-block partial
// sums, then a fixed-order combine + the gradient sign(pred - target)/M in the class columns (zeros elsewhere).
struct BoxRegParams {
  const float* logits; long ld; int col0; int K;   // deltas of this head: columns col0 .. col0 + 4K
  const int* labels; const float* props; const float* gt_boxes;
  float* dlogits; long ld_d; float* loss; int M; float wx, wy, ww, wh, loss_scale;
};

__device__ __forceinline__ void box_target(const float* s, const float* t, float wx, float wy, float ww, float wh,
                                           float (&d)[4]) {
  const float sw = s[2] - s[0], sh = s[3] - s[1];
  const float sx = s[0] + 0.5f * sw, sy = s[1] + 0.5f * sh;
  const float tw = t[2] - t[0], th = t[3] - t[1];
  const float tx = t[0] + 0.5f * tw, ty = t[1] + 0.5f * th;
  d[0] = wx * (tx - sx) / sw; d[1] = wy * (ty - sy) / sh;
  d[2] = ww * logf(tw / sw); d[3] = wh * logf(th / sh);
}

__global__ __launch_bounds__(256) void boxreg_rows_kernel(BoxRegParams p, float* partial) {
  __shared__ float sh[256];
  const int r = blockIdx.x * 256 + threadIdx.x;
  float l = 0.f;
  if (r < p.M) {
    const int lab = p.labels[r];
    if (lab >= 0 && lab < p.K) {
      float d[4];
      box_target(p.props + 4 * (long)r, p.gt_boxes + 4 * (long)r, p.wx, p.wy, p.ww, p.wh, d);
      const float* pr = p.logits + (long)r * p.ld + p.col0 + 4 * lab;
      for (int e = 0; e < 4; ++e) l += fabsf(pr[e] - d[e]);
    }
  }
  sh[threadIdx.x] = l;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void boxreg_grad_kernel(BoxRegParams p, const float* partial, int nparts) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float l = 0.f;
    for (int q = 0; q < nparts; ++q) l += partial[q];
    p.loss[0] = l / (float)p.M;
  }
  if (!p.dlogits) return;
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= p.M) return;
  float* dr = p.dlogits + (long)r * p.ld_d + p.col0;
  for (int c = 0; c < 4 * p.K; ++c) dr[c] = 0.f;
  const int lab = p.labels[r];
  if (lab >= 0 && lab < p.K) {
    float d[4];
    box_target(p.props + 4 * (long)r, p.gt_boxes + 4 * (long)r, p.wx, p.wy, p.ww, p.wh, d);
    const float* pr = p.logits + (long)r * p.ld + p.col0 + 4 * lab;
    for (int e = 0; e < 4; ++e) {
      const float diff = pr[e] - d[e];
      dr[4 * lab + e] = (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) / (float)p.M * p.loss_scale;
    }
  }
}

// mean over n_heads of row softmaxes (fast_rcnn.py:1577-1594)
// bg_first: the heads keep the background in column 0 (PCL); the output is rotated so that it is the last column
// (OICROutputLayers.inference pcl_bg, fast_rcnn.py:1463-1465)
// One thread per row (the sums run over the classes in order, as the oracle's do); 64-thread blocks - 32 blocks at R = 2000
// instead of 8 - and the running means in LDS instead of read-modify-write passes over `probs` in global memory: 44 -> ~10 us of
// every inference pass (profiles/r5_60_tta_dc5_kernel_stats.txt).
constexpr int MSM_THREADS = 64;
// LDS_ACC = false (more than 256 classes - the LVIS-sized heads: C x 256 bytes of LDS would not fit): the running means live
// in the output row itself (read-modify-write per head; the same additions in the same order, so the same bits).
template <bool LDS_ACC>
__global__ __launch_bounds__(MSM_THREADS) void mean_softmax_kernel(const float* logits, long ld, const int* col0s, int n_heads, int C,
                                                                   float* probs, int M, int bg_first) {
  extern __shared__ float msm_acc[];  // [C][MSM_THREADS]: a thread's column - conflict-free
  const int r = blockIdx.x * MSM_THREADS + threadIdx.x;
  if (r >= M) return;
  float* acc = LDS_ACC ? msm_acc + threadIdx.x : probs + (long)r * C;
  const int pitch = LDS_ACC ? MSM_THREADS : 1;
  for (int c = 0; c < C; ++c) acc[c * pitch] = 0.f;
  for (int h = 0; h < n_heads; ++h) {
    const float* row = logits + (long)r * ld + col0s[h];
    float mx = -FLT_MAX;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, row[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(row[c] - mx);
    for (int c = 0; c < C; ++c) {
      const int oc = bg_first ? (c == 0 ? C - 1 : c - 1) : c;
      acc[oc * pitch] += expf(row[c] - mx) / se;
    }
  }
  for (int c = 0; c < C; ++c) probs[(long)r * C + c] = acc[c * pitch] / (float)n_heads;
}

// The same for C <= 64 (every shipped head: 21 / 81 ... up to 64 columns), WAVE per row and lane = class (round 6): the thread-per-row
// form above walks its row three times per head with dependent, uncoalesced loads - 44 us per call at R = 2000, a 4-block launch
// (profiles/r6_30_tta_dc5_kernel_stats.txt: 1.8 % of a TTA image's kernel time).  Here a row is one coalesced load per head, the
// maximum a wave reduction (order-free), and the denominator is summed by every lane in CLASS ORDER from the lanes' values
// (v_readlane, c ascending) - the additions of the form above in the same order on the same values: bit-identical.
__global__ __launch_bounds__(256) void mean_softmax_wave_kernel(const float* logits, long ld, const int* col0s, int n_heads, int C,
                                                                float* probs, int M, int bg_first) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  float acc = 0.f;
  for (int h = 0; h < n_heads; ++h) {
    const float* row = logits + (long)r * ld + col0s[h];
    const float x = lane < C ? row[lane] : -FLT_MAX;
    const float mx = wave_max(x);
    const float e = lane < C ? expf(x - mx) : 0.f;
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), c));
    acc += e / se;
  }
  if (lane < C) {
    const int oc = bg_first ? (lane == 0 ? C - 1 : lane - 1) : lane;
    probs[(long)r * C + oc] = acc / (float)n_heads;
  }
}

static int g_msm_wave = 1;  // drn_tune(DRN_TUNE_MSM_WAVE = 32): 0 = the thread-per-row kernel also for C <= 64 (tests, A/B)
extern "C" __attribute__((visibility("hidden"))) int drn_msm_set_wave(int on) {
  const int old = g_msm_wave;
  g_msm_wave = on != 0;
  return old;
}

// Box2BoxTransform.apply_deltas; deltas == null means all-zero deltas (non-regressing heads)
__global__ void apply_deltas_kernel(const float* deltas, long ld_d, const float* boxes, float* out, int M, int K,
                                    float wx, float wy, float ww, float wh, float clampv) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)M * K) return;
  const int r = i / K, k = i - (long)r * K;
  const float x1 = boxes[4 * (long)r], y1 = boxes[4 * (long)r + 1], x2 = boxes[4 * (long)r + 2], y2 = boxes[4 * (long)r + 3];
  const float w = x2 - x1, h = y2 - y1;
  const float cx = x1 + 0.5f * w, cy = y1 + 0.5f * h;
  float dx = 0.f, dy = 0.f, dw = 0.f, dh = 0.f;
  if (deltas) {
    const float* d = deltas + (long)r * ld_d + 4 * k;
    dx = d[0] / wx; dy = d[1] / wy; dw = fminf(d[2] / ww, clampv); dh = fminf(d[3] / wh, clampv);
  }
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  float* o = out + (long)r * 4 * K + 4 * k;
  o[0] = pcx - 0.5f * pw; o[1] = pcy - 0.5f * ph; o[2] = pcx + 0.5f * pw; o[3] = pcy + 0.5f * ph;
}

// p -= lr * (buf = mom*buf + (g + wd*p)); first step: buf = g + wd*p.  One launch over the flat parameter
// arena (blockIdx.y walks the segments, float4 lanes when the segment offset is 16-B aligned); the bf16
// compute shadow (same flat layout) is refreshed in the same pass, so the weights are read once per step.
// w / momentum / gradient are streamed once per step: non-temporal accesses keep them from evicting the GEMM operands
// (and the freshly written bf16 shadow, which the next fc6 forward reads) from L2 / Infinity Cache (+1.4 % step rate).
template <bool SHADOW, int GDT, bool NT = true>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, float* __restrict__ mom,
                                                  const void* __restrict__ gv, long goff, bf16_t* __restrict__ shadow,
                                                  const SgdSeg* segs, int nseg, float momentum, int first_step,
                                                  float grad_scale) {
  using GT = typename ElemOf<GDT>::type;
  const GT* g = (const GT*)gv - goff;  // arena element j <-> g[j]
  for (int s = blockIdx.y; s < nseg; s += gridDim.y) {
    const SgdSeg sg = segs[s];
    const long tid = blockIdx.x * (long)blockDim.x + threadIdx.x, nthr = (long)gridDim.x * blockDim.x;
    long done = 0;
    if ((sg.off & 3) == 0 && ((sg.off - goff) & 3) == 0) {
      const long nvec = sg.cnt >> 2;
      for (long i = tid; i < nvec; i += nthr) {
        const long j = sg.off + 4 * i;
        const f32x4_t pw = NT ? __builtin_nontemporal_load((const f32x4_t*)(w + j)) : *(const f32x4_t*)(w + j);
        f32x4_t gg;
        if constexpr (GDT == DRN_BF16) {
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          const u32x2_t x = NT ? __builtin_nontemporal_load((const u32x2_t*)(g + j)) : *(const u32x2_t*)(g + j);
          gg = f32x4_t{__builtin_bit_cast(float, x.x << 16), __builtin_bit_cast(float, x.x & 0xffff0000u),
                       __builtin_bit_cast(float, x.y << 16), __builtin_bit_cast(float, x.y & 0xffff0000u)};
        } else {
          gg = NT ? __builtin_nontemporal_load((const f32x4_t*)(g + j)) : *(const f32x4_t*)(g + j);
        }
        f32x4_t mm = {0.f, 0.f, 0.f, 0.f};
        if (!first_step) mm = NT ? __builtin_nontemporal_load((const f32x4_t*)(mom + j)) : *(const f32x4_t*)(mom + j);
        f32x4_t nb, nw;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float d = gg[e] * grad_scale;
          if (sg.wd != 0.f) d = d + sg.wd * pw[e];
          nb[e] = first_step ? d : momentum * mm[e] + d;
          nw[e] = pw[e] - sg.lr * nb[e];
        }
        if (NT) {
          __builtin_nontemporal_store(nb, (f32x4_t*)(mom + j));
          __builtin_nontemporal_store(nw, (f32x4_t*)(w + j));
        } else {
          *(f32x4_t*)(mom + j) = nb;
          *(f32x4_t*)(w + j) = nw;
        }
        if (SHADOW) {
          uint2 o;
          o.x = (uint32_t)f32_to_bf16(nw[0]) | ((uint32_t)f32_to_bf16(nw[1]) << 16);
          o.y = (uint32_t)f32_to_bf16(nw[2]) | ((uint32_t)f32_to_bf16(nw[3]) << 16);
          *(uint2*)(shadow + j) = o;
        }
      }
      done = nvec << 2;
    }
    for (long i = done + tid; i < sg.cnt; i += nthr) {
      const long j = sg.off + i;
      const float pw = w[j];
      float d = ElemOf<GDT>::ld(g + j) * grad_scale;
      if (sg.wd != 0.f) d = d + sg.wd * pw;
      const float b = first_step ? d : momentum * mom[j] + d;
      mom[j] = b;
      const float nw = pw - sg.lr * b;
      w[j] = nw;
      if (SHADOW) shadow[j] = f32_to_bf16(nw);
    }
  }
}

// The same update on a rectangular BLOCK of one 2-D parameter (rows r0..r1, columns c0..c1 of a [.., ld] tensor that starts at
// arena element sg.off): the fc6 weight gradient leaves its GEMM in COLUMN slabs (one exact round of the persistent
// kernel each, run_fc1_tail), and each slab's update starts the moment its GEMM is queued.  Arithmetic, access width
// and cache policy are sgd_kernel's; only the index map differs (a row of the block is a contiguous run of cols / 4
// 16-byte vectors).  c0, cols, ld, sg.off and goff are multiples of 4 (launcher).
template <bool SHADOW, int GDT>
__global__ __launch_bounds__(256) void sgd_block_kernel(float* __restrict__ w, float* __restrict__ mom,
                                                        const void* __restrict__ gv, long goff, bf16_t* __restrict__ shadow,
                                                        const SgdSeg* seg, int r0, int rows, int c0, int cols, long ld,
                                                        float momentum, int first_step, float grad_scale) {
  using GT = typename ElemOf<GDT>::type;
  const GT* g = (const GT*)gv - goff;
  const SgdSeg sg = seg[0];
  const unsigned cv = (unsigned)cols >> 2, nvec = (unsigned)rows * cv;
  const unsigned nthr = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += nthr) {
    const unsigned r = i / cv, c = i - r * cv;
    const long j = sg.off + (long)(r0 + (int)r) * ld + c0 + 4 * (long)c;
    const f32x4_t pw = __builtin_nontemporal_load((const f32x4_t*)(w + j));
    f32x4_t gg;
    if constexpr (GDT == DRN_BF16) {
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      const u32x2_t x = __builtin_nontemporal_load((const u32x2_t*)(g + j));
      gg = f32x4_t{__builtin_bit_cast(float, x.x << 16), __builtin_bit_cast(float, x.x & 0xffff0000u),
                   __builtin_bit_cast(float, x.y << 16), __builtin_bit_cast(float, x.y & 0xffff0000u)};
    } else {
      gg = __builtin_nontemporal_load((const f32x4_t*)(g + j));
    }
    f32x4_t mm = {0.f, 0.f, 0.f, 0.f};
    if (!first_step) mm = __builtin_nontemporal_load((const f32x4_t*)(mom + j));
    f32x4_t nb, nw;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float d = gg[e] * grad_scale;
      if (sg.wd != 0.f) d = d + sg.wd * pw[e];
      nb[e] = first_step ? d : momentum * mm[e] + d;
      nw[e] = pw[e] - sg.lr * nb[e];
    }
    __builtin_nontemporal_store(nb, (f32x4_t*)(mom + j));
    __builtin_nontemporal_store(nw, (f32x4_t*)(w + j));
    if (SHADOW) {
      uint2 o;
      o.x = (uint32_t)f32_to_bf16(nw[0]) | ((uint32_t)f32_to_bf16(nw[1]) << 16);
      o.y = (uint32_t)f32_to_bf16(nw[2]) | ((uint32_t)f32_to_bf16(nw[3]) << 16);
      *(uint2*)(shadow + j) = o;
    }
  }
}

__global__ void sum_small_kernel(const float* in, int n, float scale, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += in[i];
    out[0] = s * scale;
  }
}

}  // namespace

extern "C" {

int drn_bias_act_fwd(const float* partials, int splits, long split_stride, const float* bias, const float* mask,
                     unsigned long long seed, const unsigned long long* seed_dev, float drop_p, void* out, long ld_out,
                     void* outT, long ld_outT, int M, int N, long ld_in, int relu, int out_dtype, void* stream) {
  if (!partials || M < 0 || N < 0 || splits < 1 || (!out && !outT)) return DRN_ERR_ARG;
  if (M == 0 || N == 0) return DRN_OK;
  ActParams p{partials, splits, split_stride, bias, mask, seed, drop_p, seed_dev, nullptr, (char*)out, ld_out, (char*)outT,
              ld_outT, nullptr, nullptr, nullptr, nullptr, M, N, ld_in, relu, 0, 64, 0};
  // skinny outputs (the 103 predictor columns; no transposed copy in that case): 64-row blocks would give only ~64
  // blocks, each thread walking 16 rows x splits partials one row after the other.  4-row blocks: ONE row per thread,
  // all of its split partials in flight at once (16-row blocks took 12.6 us for 8 MB - four dependent row trips; now
  // 6.4 us).  (The same idea for the backward pass - fetch a thread's 16 rows before processing them - needed a full
  // unroll, 253 VGPRs, and measured 18.6 instead of 12.7 us: not kept.)
  if (!outT && (long)((N + 63) / 64) * ((M + 63) / 64) < 256) p.rows_fwd = 4;
  dim3 grid((N + 63) / 64, (M + p.rows_fwd - 1) / p.rows_fwd), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (act_vec_ok(p, out_dtype)) hipLaunchKernelGGL((act_vec_kernel<false>), grid, block, 0, st, p);
  else if (out_dtype == DRN_BF16) hipLaunchKernelGGL((act_kernel<DRN_BF16, DRN_BF16, false>), grid, block, 0, st, p);
  else if (out_dtype == DRN_F32) hipLaunchKernelGGL((act_kernel<DRN_F32, DRN_F32, false>), grid, block, 0, st, p);
  else return DRN_ERR_ARG;
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_bias_act_bwd_splits(const void* grad_out, int grad_dtype, long ld_in, int splits, long split_stride,
                            const float* colscale, const int* colidx, const void* saved_out, const float* mask,
                            float drop_p, void* dpre, long ld_out, void* dpreT, long ld_outT, float* colsum,
                            float* colpart, int accumulate_colsum, int M, int N, int out_dtype, void* stream);

int drn_bias_act_bwd(const void* grad_out, int grad_dtype, long ld_in, const float* colscale, const int* colidx,
                     const void* saved_out, const float* mask, float drop_p,
                     void* dpre, long ld_out, void* dpreT, long ld_outT, float* colsum, float* colpart,
                     int accumulate_colsum, int M, int N, int out_dtype, void* stream) {
  return drn_bias_act_bwd_splits(grad_out, grad_dtype, ld_in, 1, 0, colscale, colidx, saved_out, mask, drop_p, dpre, ld_out,
                                 dpreT, ld_outT, colsum, colpart, accumulate_colsum, M, N, out_dtype, stream);
}

// the same with grad_out given as `splits` fp32 split-K partials (split_stride floats apart) that are summed on load, in
// order - the dX GEMM of fc7 then runs on the 256x256 kernel with a K-split like the forward GEMMs do
int drn_bias_act_bwd_splits(const void* grad_out, int grad_dtype, long ld_in, int splits, long split_stride,
                            const float* colscale, const int* colidx, const void* saved_out, const float* mask,
                            float drop_p, void* dpre, long ld_out, void* dpreT, long ld_outT, float* colsum,
                            float* colpart, int accumulate_colsum, int M, int N, int out_dtype, void* stream) {
  if (!grad_out || M < 0 || N < 0 || (grad_dtype != DRN_F32 && grad_dtype != DRN_BF16)) return DRN_ERR_ARG;
  if (splits < 1 || (splits > 1 && grad_dtype != DRN_F32)) return DRN_ERR_ARG;
  if (colsum && !colpart) return DRN_ERR_ARG;  // colpart: ceil(M/64)*N floats of scratch
  if (M == 0 || N == 0) return DRN_OK;
  ActParams p{(const float*)grad_out, splits, split_stride, nullptr, mask, 0ULL, drop_p, nullptr, (const char*)saved_out, (char*)dpre, ld_out,
              (char*)dpreT, ld_outT, colsum, colscale, colidx, colpart, M, N, ld_in, 1, accumulate_colsum, 64,
              grad_dtype == DRN_BF16};
  const int nparts = (M + ACT_ROWS - 1) / ACT_ROWS;
  dim3 grid((N + 63) / 64, nparts), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (act_vec_ok(p, out_dtype)) hipLaunchKernelGGL((act_vec_kernel<true>), grid, block, 0, st, p);
  else if (out_dtype == DRN_BF16) hipLaunchKernelGGL((act_kernel<DRN_BF16, DRN_BF16, true>), grid, block, 0, st, p);
  else if (out_dtype == DRN_F32) hipLaunchKernelGGL((act_kernel<DRN_F32, DRN_F32, true>), grid, block, 0, st, p);
  else return DRN_ERR_ARG;
  if (colsum)
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 255) / 256), dim3(256), 0, st, colpart, nparts, N, colsum,
                       accumulate_colsum);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// drn_gemm_nt + drn_bias_act_bwd in one launch for a skinny contraction (K = 64, 128, 192 or 256; bf16 operands and
// outputs): dpre = act'(saved_out) .* (A . B^T), its transposed copy and the column sums, bit for bit what the two calls
// produce (the tile is formed with the same MFMA in the same k order; act_vec_kernel<true, K / 64>).
int drn_gemm_nt_act_bwd(const void* A, const void* B, int M, int N, int K, long lda, long ldb, const void* saved_out,
                        const float* mask, float drop_p, void* dpre, long ld_out, void* dpreT, long ld_outT, float* colsum,
                        float* colpart, int accumulate_colsum, void* stream) {
  if (!A || !B || M < 0 || N < 0 || K < 1 || (!dpre && !dpreT)) return DRN_ERR_ARG;
  if (colsum && !colpart) return DRN_ERR_ARG;
  if (M == 0 || N == 0) return DRN_OK;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (K % 64 != 0 || K > 256 || N % 64 != 0 || lda % 8 != 0 || ldb % 8 != 0 || lda < K || ldb < K || !al16(A) || !al16(B))
    return DRN_ERR_UNSUPPORTED;
  ActParams p{nullptr, 1, 0, nullptr, mask, 0ULL, drop_p, nullptr, (const char*)saved_out, (char*)dpre, ld_out,
              (char*)dpreT, ld_outT, colsum, nullptr, nullptr, colpart, M, N, 0, 1, accumulate_colsum, 64, 0,
              (const bf16_t*)A, (const bf16_t*)B, lda, ldb};
  p.in = (const float*)A;  // (only its alignment is looked at below)
  if (!act_vec_ok(p, DRN_BF16)) return DRN_ERR_UNSUPPORTED;
  const int nparts = (M + ACT_ROWS - 1) / ACT_ROWS;
  dim3 grid(N / 64, nparts), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (K / 64) {
    case 1: hipLaunchKernelGGL((act_vec_kernel<true, 1>), grid, block, 0, st, p); break;
    case 2: hipLaunchKernelGGL((act_vec_kernel<true, 2>), grid, block, 0, st, p); break;
    case 3: hipLaunchKernelGGL((act_vec_kernel<true, 3>), grid, block, 0, st, p); break;
    default: hipLaunchKernelGGL((act_vec_kernel<true, 4>), grid, block, 0, st, p); break;
  }
  if (colsum)
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 255) / 256), dim3(256), 0, st, colpart, nparts, N, colsum,
                       accumulate_colsum);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// second stage of the bias-gradient column sums on its own: drn_bias_act_bwd with colsum == NULL and colpart != NULL
// leaves ceil(M/64) x N partials, and this adds them (fixed order) into colsum - so that the optimizer stream can do it
// right in front of the SGD pass instead of the backward's critical path
int drn_colsum_reduce(const float* colpart, int nparts, int N, float* colsum, int accumulate, void* stream) {
  if (!colpart || !colsum || nparts < 1 || N < 1) return DRN_ERR_ARG;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, colpart, nparts, N,
                     colsum, accumulate);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// scratch: n_img * ceil(max_rows/128) * 3 * 128 floats (max_rows = largest proposal count of one image)
int drn_wsddn_fwd_bwd(const float* logits, long ld, int c_cls, int c_det, int K, const int* img_off, int n_img,
                      const float* gt_onehot, float* scores, float* row_softmax, float* img_scores, float* loss_part,
                      float* dlogits, long ld_d, float* scratch, int max_rows, int mean_loss, float loss_scale,
                      void* stream) {
  if (!logits || !img_off || !gt_onehot || !scores || !row_softmax || !img_scores || !loss_part || !scratch || K < 1 ||
      K > 128 || n_img < 1 || max_rows < 1)
    return DRN_ERR_ARG;
  const int nb = (max_rows + WS_ROWS - 1) / WS_ROWS;
  WsddnParams p{logits, ld, c_cls, c_det, K, img_off, gt_onehot, scores, row_softmax, img_scores, loss_part, dlogits, ld_d,
                n_img, mean_loss, loss_scale, scratch, nb};
  dim3 grid(nb, n_img), block(K <= 32 ? 256 : 1024);
  hipStream_t st = (hipStream_t)stream;
  if (K <= 32) {
    hipLaunchKernelGGL((wsddn_stage_kernel<32, 0>), grid, block, 0, st, p);
    hipLaunchKernelGGL((wsddn_stage_kernel<32, 1>), grid, block, 0, st, p);
    hipLaunchKernelGGL((wsddn_stage_kernel<32, 2>), grid, block, 0, st, p);
  } else {
    hipLaunchKernelGGL((wsddn_stage_kernel<64, 0>), grid, block, 0, st, p);
    hipLaunchKernelGGL((wsddn_stage_kernel<64, 1>), grid, block, 0, st, p);
    hipLaunchKernelGGL((wsddn_stage_kernel<64, 2>), grid, block, 0, st, p);
  }
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_oicr_targets(const float* prev_scores, long ld_s, const float* prev_boxes, int box_cols, int zero_delta_decode,
                     const float* props,
                     const int* img_off, int n_img, const int* gt_classes, const int* gt_count, int gmax,
                     const float* img_scores, int K, const float* thresholds, const int* thr_labels, int nthr,
                     int* labels, float* weights, int* matched, float* gt_boxes, int* pgt_idx, float* pgt_boxes,
                     void* stream) {
  if (!prev_scores || !prev_boxes || !props || !img_off || !gt_classes || !gt_count || !img_scores || !labels ||
      !weights || !matched || !gt_boxes || !pgt_idx || !pgt_boxes)
    return DRN_ERR_ARG;
  if (gmax < 1 || gmax > 128 || nthr < 1 || nthr > 3 || (box_cols != 4 && box_cols != 4 * K)) return DRN_ERR_ARG;
  if (zero_delta_decode && box_cols != 4) return DRN_ERR_ARG;
  TargetParams p;
  p.prev_scores = prev_scores; p.ld_s = ld_s; p.prev_boxes = prev_boxes; p.box_cols = box_cols; p.props = props;
  p.zero_delta_decode = zero_delta_decode;
  p.img_off = img_off; p.gt_classes = gt_classes; p.gt_count = gt_count; p.gmax = gmax; p.img_scores = img_scores;
  p.K = K; p.nthr = nthr;
  for (int i = 0; i < 3; ++i) p.thr[i] = i < nthr ? thresholds[i] : 0.f;
  for (int i = 0; i < 4; ++i) p.lab[i] = i <= nthr ? thr_labels[i] : 0;
  p.labels = labels; p.weights = weights; p.matched = matched; p.gt_boxes = gt_boxes; p.pgt_idx = pgt_idx;
  p.pgt_boxes = pgt_boxes;
  hipLaunchKernelGGL(oicr_targets_kernel, dim3(n_img), dim3(1024), 0, (hipStream_t)stream, p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// All n_heads refinement branches in four launches (softmax of every head, pseudo-GT mining + labelling of every head,
// CE partials, CE combine + gradient) instead of three per head.  Per-head outputs are [n_heads] x the single-head
// shape, contiguous.  Non-regressing heads only (pgt boxes = decoded zero deltas of the proposals for heads >= 1).
static void chain_params(const float* logits, long ld, const int* col0s_host, int n_heads, int K, const float* scores0,
                         long ld_s0, const float* props, const int* img_off, int n_img, const int* gt_classes,
                         const int* gt_count, int gmax, const float* img_scores, const float* thresholds,
                         const int* thr_labels, int nthr, float* probs, int* labels, float* weights, int* matched,
                         float* gt_boxes, int* pgt_idx, float* pgt_boxes, float* dlogits, long ld_d, float* losses,
                         float* scratch, int M, float loss_scale, TargetMulti& tm, CeMulti& probs_only, CeMulti& ce) {
  const int C = K + 1, nb = (M + CE_ROWS - 1) / CE_ROWS;
  for (int k = 0; k < n_heads; ++k) {
    float* pk = probs + (long)k * M * C;
    TargetParams& t = tm.h[k];
    t.prev_scores = k == 0 ? scores0 : probs + (long)(k - 1) * M * C;
    t.ld_s = k == 0 ? ld_s0 : C;
    t.prev_boxes = props; t.box_cols = 4; t.zero_delta_decode = k > 0; t.props = props;
    t.img_off = img_off; t.gt_classes = gt_classes; t.gt_count = gt_count; t.gmax = gmax; t.img_scores = img_scores;
    t.K = K; t.nthr = nthr;
    for (int i = 0; i < 3; ++i) t.thr[i] = i < nthr ? thresholds[i] : 0.f;
    for (int i = 0; i < 4; ++i) t.lab[i] = i <= nthr ? thr_labels[i] : 0;
    t.labels = labels + (long)k * M; t.weights = weights + (long)k * M; t.matched = matched + (long)k * M;
    t.gt_boxes = gt_boxes + (long)k * M * 4; t.pgt_idx = pgt_idx + (long)k * n_img * gmax;
    t.pgt_boxes = pgt_boxes + (long)k * n_img * gmax * 4;
    probs_only.h[k] = CeParams{logits, ld, col0s_host[k], C, nullptr, nullptr, pk, nullptr, 0, nullptr, M, loss_scale};
    probs_only.partial[k] = nullptr;
    ce.h[k] = CeParams{logits, ld, col0s_host[k], C, t.labels, t.weights, pk, dlogits, ld_d, losses + k, M, loss_scale};
    ce.partial[k] = scratch + (long)k * 2 * nb;
  }
}

int drn_oicr_refine_chain(const float* logits, long ld, const int* col0s_host, int n_heads, int K, const float* scores0,
                          long ld_s0, const float* props, const int* img_off, int n_img, const int* gt_classes,
                          const int* gt_count, int gmax, const float* img_scores, const float* thresholds,
                          const int* thr_labels, int nthr, float* probs, int* labels, float* weights, int* matched,
                          float* gt_boxes, int* pgt_idx, float* pgt_boxes, float* dlogits, long ld_d, float* losses,
                          float* scratch, int M, float loss_scale, void* stream) {
  if (!logits || !col0s_host || !scores0 || !props || !img_off || !gt_classes || !gt_count || !img_scores || !probs ||
      !labels || !weights || !matched || !gt_boxes || !pgt_idx || !pgt_boxes || !losses || !scratch)
    return DRN_ERR_ARG;
  if (n_heads < 1 || n_heads > MAX_CHAIN_HEADS || K < 1 || K + 1 > 128 || gmax < 1 || gmax > 128 || nthr < 1 || nthr > 3)
    return DRN_ERR_ARG;
  if (M <= 0 || n_img < 1) return M == 0 ? DRN_OK : DRN_ERR_ARG;
  const int nb = (M + CE_ROWS - 1) / CE_ROWS;
  TargetMulti tm;
  CeMulti probs_only, ce;
  chain_params(logits, ld, col0s_host, n_heads, K, scores0, ld_s0, props, img_off, n_img, gt_classes, gt_count, gmax,
               img_scores, thresholds, thr_labels, nthr, probs, labels, weights, matched, gt_boxes, pgt_idx, pgt_boxes, dlogits,
               ld_d, losses, scratch, M, loss_scale, tm, probs_only, ce);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_rows_multi_kernel, dim3(nb, n_heads), dim3(256), 0, st, probs_only);
  hipLaunchKernelGGL(oicr_targets_multi_kernel, dim3(n_img, n_heads), dim3(1024), 0, st, tm);
  hipLaunchKernelGGL(ce_rows_multi_kernel, dim3(nb, n_heads), dim3(256), 0, st, ce);
  hipLaunchKernelGGL(ce_grad_multi_kernel, dim3(nb, n_heads), dim3(256), 0, st, ce, nb);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// The loss tail of a training step with non-regressing refinement heads in SIX launches instead of nine: the predictor's
// split-K reduce + bias (drn_bias_act_fwd, fp32 out), drn_wsddn_fwd_bwd and drn_oicr_refine_chain, bit for bit.
//   A  WSDDN stage 0 + the softmax of every refinement head, both reading the logits from the split-K partials
//   B  WSDDN stage 1        C  WSDDN stage 2 (image scores, BCE, d cls / d det)
//   D  pseudo-GT mining + labels of every head        E, F  CE partials; fixed-order combine + gradient
// (E + F as ONE launch - every block counting V itself, the loss formed by the last-arriving block - was built and measured:
// 21.6 us against 7.8 + 7.4; these kernels are a few dependent memory round trips each, not launch overhead, and the
// last-arriver tail adds one more.  Likewise A as "materialise, barrier, run the stage": 22 us against 21 for the three
// launches it replaced - profiles/r4_37.)
// ws_scratch as drn_wsddn_fwd_bwd; ce_scratch as drn_oicr_refine_chain.  logits [M][ld] is an OUTPUT here: columns
// c_cls..+K, c_det..+K and col0s[k]..+K+1 are written (all NH columns when the heads are exactly these).
int drn_mil_oicr_losses(const float* partials, int splits, long split_stride, long ld_part, const float* bias,
                        unsigned long long seed_inc, unsigned long long* seed_dev, float* logits, long ld, int c_cls,
                        int c_det, int K, const int* img_off, int n_img, const float* gt_onehot, float* scores,
                        float* row_softmax, float* img_scores, float* loss_part, float* ws_scratch, int max_rows,
                        int mean_loss, const int* col0s_host, int n_heads, const float* props, const int* gt_classes,
                        const int* gt_count, int gmax, const float* thresholds, const int* thr_labels, int nthr, float* probs,
                        int* labels, float* weights, int* matched, float* gt_boxes, int* pgt_idx, float* pgt_boxes,
                        float* losses, float* ce_scratch, float* dlogits, long ld_d, int M, float loss_scale,
                        void* stream) {
  if (!partials || !logits || !img_off || !gt_onehot || !scores || !row_softmax || !img_scores || !loss_part ||
      !ws_scratch || !col0s_host || !props || !gt_classes || !gt_count || !probs || !labels || !weights || !matched ||
      !gt_boxes || !pgt_idx || !pgt_boxes || !losses || !ce_scratch)
    return DRN_ERR_ARG;
  if (splits < 1 || K < 1 || K + 1 > 128 || n_img < 1 || max_rows < 1 || n_heads < 1 || n_heads > MAX_CHAIN_HEADS ||
      gmax < 1 || gmax > 128 || nthr < 1 || nthr > 3 || M <= 0)
    return DRN_ERR_ARG;
  const int nb_ws = (max_rows + WS_ROWS - 1) / WS_ROWS, nb_ce = (M + CE_ROWS - 1) / CE_ROWS;
  WsddnParams p{logits, ld, c_cls, c_det, K, img_off, gt_onehot, scores, row_softmax, img_scores, loss_part, dlogits, ld_d,
                n_img, mean_loss, loss_scale, ws_scratch, nb_ws};
  TargetMulti tm;
  CeMulti probs_only, ce;
  chain_params(logits, ld, col0s_host, n_heads, K, scores, K, props, img_off, n_img, gt_classes, gt_count, gmax, img_scores,
               thresholds, thr_labels, nthr, probs, labels, weights, matched, gt_boxes, pgt_idx, pgt_boxes, dlogits, ld_d,
               losses, ce_scratch, M, loss_scale, tm, probs_only, ce);
  LogitsSrc src{partials, splits, split_stride, ld_part, bias, logits, ld, seed_dev, seed_inc};
  hipStream_t st = (hipStream_t)stream;
  dim3 gA(nb_ws > nb_ce ? nb_ws : nb_ce, n_img + n_heads), gW(nb_ws, n_img), bW(K <= 32 ? 256 : 1024);
  if (K <= 32) {
    hipLaunchKernelGGL((mil_stage0_kernel<32>), gA, bW, 0, st, p, probs_only, src, nb_ws, nb_ce);
    hipLaunchKernelGGL((wsddn_stage_kernel<32, 1>), gW, bW, 0, st, p);
    hipLaunchKernelGGL((wsddn_stage_kernel<32, 2>), gW, bW, 0, st, p);
  } else {
    hipLaunchKernelGGL((mil_stage0_kernel<64>), gA, bW, 0, st, p, probs_only, src, nb_ws, nb_ce);
    hipLaunchKernelGGL((wsddn_stage_kernel<64, 1>), gW, bW, 0, st, p);
    hipLaunchKernelGGL((wsddn_stage_kernel<64, 2>), gW, bW, 0, st, p);
  }
  hipLaunchKernelGGL(oicr_targets_multi_kernel, dim3(n_img, n_heads), dim3(1024), 0, st, tm);
  hipLaunchKernelGGL(ce_rows_multi_kernel, dim3(nb_ce, n_heads), dim3(256), 0, st, ce);
  hipLaunchKernelGGL(ce_grad_multi_kernel, dim3(nb_ce, n_heads), dim3(256), 0, st, ce, nb_ce);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// scratch: 2*ceil(M/16) floats (needed when labels != NULL)
int drn_softmax_ce(const float* logits, long ld, int col0, int C, const int* labels, const float* weights,
                   float* probs, float* dlogits, long ld_d, float* loss, float* scratch, int M, float loss_scale,
                   void* stream) {
  if (!logits || !probs || C < 1 || C > 128 || M < 0) return DRN_ERR_ARG;
  if (labels && (!weights || !loss || !scratch)) return DRN_ERR_ARG;
  if (M == 0) return DRN_OK;
  CeParams p{logits, ld, col0, C, labels, weights, probs, dlogits, ld_d, loss, M, loss_scale};
  const int nb = (M + CE_ROWS - 1) / CE_ROWS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(nb), dim3(256), 0, st, p, scratch);
  if (labels) hipLaunchKernelGGL(ce_grad_kernel, dim3(nb), dim3(256), 0, st, p, (const float*)scratch, nb);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// scratch: ceil(M/256) floats
int drn_box_reg_loss(const float* logits, long ld, int col0, int K, const int* labels, const float* props,
                     const float* gt_boxes, const float* weights4_host, float* dlogits, long ld_d, float* loss,
                     float* scratch, int M, float loss_scale, void* stream) {
  if (!logits || !labels || !props || !gt_boxes || !weights4_host || !loss || !scratch || K < 1 || M < 1)
    return DRN_ERR_ARG;
  BoxRegParams p{logits, ld, col0, K, labels, props, gt_boxes, dlogits, ld_d, loss, M, weights4_host[0],
                 weights4_host[1], weights4_host[2], weights4_host[3], loss_scale};
  const int nb = (M + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(boxreg_rows_kernel, dim3(nb), dim3(256), 0, st, p, scratch);
  hipLaunchKernelGGL(boxreg_grad_kernel, dim3(nb), dim3(256), 0, st, p, (const float*)scratch, nb);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_mean_softmax(const float* logits, long ld, const int* col0s_dev, int n_heads, int C, float* probs, int M,
                     int bg_first, void* stream) {
  if (!logits || !col0s_dev || !probs || n_heads < 1) return DRN_ERR_ARG;
  if (M == 0) return DRN_OK;
  if (C < 1) return DRN_ERR_ARG;
  const dim3 grid((M + MSM_THREADS - 1) / MSM_THREADS), block(MSM_THREADS);
  if (C <= 64 && g_msm_wave)
    hipLaunchKernelGGL(mean_softmax_wave_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, col0s_dev, n_heads, C,
                       probs, M, bg_first);
  else if ((size_t)C * MSM_THREADS * sizeof(float) <= 64 * 1024)
    hipLaunchKernelGGL(mean_softmax_kernel<true>, grid, block, (size_t)C * MSM_THREADS * sizeof(float), (hipStream_t)stream, logits,
                       ld, col0s_dev, n_heads, C, probs, M, bg_first);
  else  // (heads with more than 256 classes run instead of being refused)
    hipLaunchKernelGGL(mean_softmax_kernel<false>, grid, block, 0, (hipStream_t)stream, logits, ld, col0s_dev, n_heads, C, probs, M,
                       bg_first);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_apply_deltas(const float* deltas, long ld_d, const float* boxes, float* out, int M, int K, const float* w4,
                     float scale_clamp, void* stream) {
  if (!boxes || !out || !w4 || K < 1) return DRN_ERR_ARG;
  if (M == 0) return DRN_OK;
  const long tot = (long)M * K;
  hipLaunchKernelGGL(apply_deltas_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     deltas, ld_d, boxes, out, M, K, w4[0], w4[1], w4[2], w4[3], scale_clamp);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// workgroups (x) of the optimizer kernel; each runs a grid-stride loop, i.e. lives for the whole launch.  At most two
// 256-thread workgroups (34 VGPRs) fit on a CU beside a resident 256x256 GEMM workgroup - see drn_tune in gemm_conv.hip
static int g_sgd_grid_x = 512;  // measured (tools/overlap_bench.py): 512 -> 6.5 TB/s, 1024 -> 5.8, 256 -> 5.3 on the fc6 slabs
__attribute__((visibility("hidden"))) int drn_sgd_set_grid(int blocks_x) {
  const int old = g_sgd_grid_x;
  if (blocks_x >= 8 && blocks_x <= 65535) g_sgd_grid_x = blocks_x;
  return old;
}

// segs_dev: device array of {int64 off, int64 cnt, float lr, float wd} (24 bytes each).  shadow (optional):
// bf16 array with the arena's flat layout, refreshed in the same pass.
int drn_sgd_step(float* weights, float* momentum_buf, const void* grads, int grad_dtype, long grad_off, void* shadow,
                 int shadow_dtype, const void* segs_dev, int nseg, float momentum, int first_step, float grad_scale,
                 void* stream) {
  if (!weights || !momentum_buf || !grads || !segs_dev || nseg < 1) return DRN_ERR_ARG;
  if (shadow && shadow_dtype != DRN_BF16) return DRN_ERR_ARG;
  if (grad_dtype != DRN_F32 && grad_dtype != DRN_BF16) return DRN_ERR_ARG;
  dim3 grid(g_sgd_grid_x, nseg < 32 ? nseg : 32), block(256);
  hipStream_t st = (hipStream_t)stream;
#define SGD_LAUNCH(SH, GD)                                                                                       \
  hipLaunchKernelGGL((sgd_kernel<SH, GD>), grid, block, 0, st, weights, momentum_buf, grads, grad_off,           \
                     (bf16_t*)shadow, (const SgdSeg*)segs_dev, nseg, momentum, first_step, grad_scale)
  if (shadow) { if (grad_dtype == DRN_BF16) SGD_LAUNCH(true, DRN_BF16); else SGD_LAUNCH(true, DRN_F32); }
  else { if (grad_dtype == DRN_BF16) SGD_LAUNCH(false, DRN_BF16); else SGD_LAUNCH(false, DRN_F32); }
#undef SGD_LAUNCH
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// drn_sgd_step on a rectangular block of ONE 2-D tensor: seg_dev = that tensor's {offset, count, lr, wd} entry (lr / wd are
// read on the device: a captured launch follows the schedule), the block = rows r0 .. r0+rows, columns c0 .. c0+cols of
// its [count / ld][ld] view.
int drn_sgd_step_block(float* weights, float* momentum_buf, const void* grads, int grad_dtype, long grad_off, void* shadow,
                       int shadow_dtype, const void* seg_dev, int r0, int rows, int c0, int cols, long ld, float momentum,
                       int first_step, float grad_scale, void* stream) {
  if (!weights || !momentum_buf || !grads || !seg_dev || r0 < 0 || rows < 0 || c0 < 0 || cols < 0 || ld < c0 + cols)
    return DRN_ERR_ARG;
  if (shadow && shadow_dtype != DRN_BF16) return DRN_ERR_ARG;
  if (grad_dtype != DRN_F32 && grad_dtype != DRN_BF16) return DRN_ERR_ARG;
  if ((c0 & 3) || (cols & 3) || (ld & 3) || (grad_off & 3)) return DRN_ERR_UNSUPPORTED;  // (the tensor's offset: checked by the caller's table)
  if ((long)rows * (cols >> 2) >= (1L << 32)) return DRN_ERR_UNSUPPORTED;
  if (rows == 0 || cols == 0) return DRN_OK;
  dim3 grid(g_sgd_grid_x), block(256);
  hipStream_t st = (hipStream_t)stream;
#define SGD_LAUNCH(SH, GD)                                                                                          \
  hipLaunchKernelGGL((sgd_block_kernel<SH, GD>), grid, block, 0, st, weights, momentum_buf, grads, grad_off,        \
                     (bf16_t*)shadow, (const SgdSeg*)seg_dev, r0, rows, c0, cols, ld, momentum, first_step, grad_scale)
  if (shadow) { if (grad_dtype == DRN_BF16) SGD_LAUNCH(true, DRN_BF16); else SGD_LAUNCH(true, DRN_F32); }
  else { if (grad_dtype == DRN_BF16) SGD_LAUNCH(false, DRN_BF16); else SGD_LAUNCH(false, DRN_F32); }
#undef SGD_LAUNCH
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

__global__ void counter_add_kernel(unsigned long long* c, unsigned long long inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += inc;
}

}  // extern "C"  (kernel above needs C++ linkage)
extern "C" {

// device-side counter (dropout seed) advanced inside the stream / graph: no host involvement per step
int drn_counter_add(unsigned long long* counter, unsigned long long inc, void* stream) {
  if (!counter) return DRN_ERR_ARG;
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, inc);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_sum_small(const float* in, int n, float scale, float* out, void* stream) {
  if (!in || !out || n < 0) return DRN_ERR_ARG;
  hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, n, scale, out);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

}  // extern "C"
