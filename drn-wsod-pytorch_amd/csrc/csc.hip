// CSCROIHeads on the device (projects/WSL/wsl/modeling/roi_heads/roi_heads_csc.py + wsl/layers/csrc/csc/csc_cuda.cu):
//   drn_csc_cpg       image-gradient map of one class: max over the colour channels of |d score_c / d image|, divided by
//                     its maximum                                              roi_heads_csc.py:456-464
//   drn_csc_weights   threshold -> summed-area table -> per-ROI frame / context contrast -> normalisation to [-1, 1] ->
//                     blend with the image-level prediction (one class)      csc_cuda.cu:132-161, :184-350, :398-535
//   drn_csc_loss      the two weighted image-level BCE losses and d loss / d logits through the WSDDN score product,
//                     or (mode 1) d (sum_r score[r, c*]) / d logits, the seed of the image-gradient pass
//                                                                           fast_rcnn.py:887-931, roi_heads_csc.py:441-455
// The reference runs the table, the normalisation and the blend on the HOST (three device<->host copies per class and
// a cudaDeviceSynchronize); here everything stays on the stream.  Integer / index work (threshold, counts, rounded box
// corners) is bit-exact with oracle/csc_ops.c; this file builds with -ffp-contract=off, and `/` / sqrtf() are the
// correctly rounded ones (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt; __fsqrt_rn is the NATIVE sqrt).
#include <float.h>

#include "drn_common.h"

namespace {

constexpr int CSC_T = 1024;

// ---- image-gradient map -----------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void csc_absmax_kernel(const void* dimg, int cpad, int C, long npx, float* map,
                                                         unsigned* gmax) {
  using E = ElemOf<DT>;
  float best = 0.f;
  for (long px = (long)blockIdx.x * blockDim.x + threadIdx.x; px < npx; px += (long)gridDim.x * blockDim.x) {
    const typename E::type* p = (const typename E::type*)dimg + px * cpad;
    float m = fabsf(E::ld(p));
    for (int c = 1; c < C; ++c) m = fmaxf(m, fabsf(E::ld(p + c)));
    map[px] = m;
    best = fmaxf(best, m);
  }
  best = wave_max(best);
  if ((threadIdx.x & 63) == 0) atomicMax(gmax, __float_as_uint(best));  // non-negative floats order like their bits
}

__global__ __launch_bounds__(256) void csc_scale_kernel(float* map, long npx, const unsigned* gmax) {
  const float mx = __uint_as_float(*gmax);
  for (long px = (long)blockIdx.x * blockDim.x + threadIdx.x; px < npx; px += (long)gridDim.x * blockDim.x)
    map[px] = map[px] / mx;
}

// ---- thresholded summed-area table --------------------------------------------------------------------------------
// rows: one workgroup per row, every thread counts a contiguous run, block scan of the run counts, second sweep writes
// the running count.  columns: one thread per column.  Counts are integers < 2^24, so fp32 adds are exact in any order.
__global__ __launch_bounds__(256) void csc_rowscan_kernel(const float* cpg, float* table, int W, float thr) {
  __shared__ int sc[256];
  const int y = blockIdx.x, tid = threadIdx.x;
  const float* src = cpg + (long)y * W;
  float* dst = table + (long)y * W;
  const int per = (W + 255) / 256, x0 = min(W, tid * per), x1 = min(W, x0 + per);
  int cnt = 0;
  for (int x = x0; x < x1; ++x) cnt += src[x] >= thr ? 1 : 0;
  sc[tid] = cnt;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = tid >= o ? sc[tid - o] : 0;
    __syncthreads();
    sc[tid] += v;
    __syncthreads();
  }
  int run = sc[tid] - cnt;
  for (int x = x0; x < x1; ++x) {
    run += src[x] >= thr ? 1 : 0;
    dst[x] = (float)run;
  }
}

__global__ __launch_bounds__(256) void csc_colscan_kernel(float* table, int H, int W) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= W) return;
  float acc = 0.f;
  for (int y = 0; y < H; ++y) {
    acc += table[(long)y * W + x];
    table[(long)y * W + x] = acc;
  }
}

// ---- CSCPool + normalisation + blend (one class, one workgroup) ---------------------------------------------------
__device__ __forceinline__ float csc_box_sum(const float* t, int width, int hs, int ws, int he, int we) {
  const float a1 = t[(long)he * width + we];
  const float a2 = (ws - 1 >= 0) ? t[(long)he * width + (ws - 1)] : 0.f;
  const float a3 = (hs - 1 >= 0) ? t[(long)(hs - 1) * width + we] : 0.f;
  const float a4 = (hs - 1 >= 0 && ws - 1 >= 0) ? t[(long)(hs - 1) * width + (ws - 1)] : 0.f;
  return a1 - a2 - a3 + a4;
}

__device__ __forceinline__ int csc_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// csc_cuda.cu:184-350 (T = float; the expressions written with double literals there are double here)
__device__ float csc_pool_one(const float* table, int height_im, int width_im, const float* roi, int area_sqrt,
                              float context_scale) {
  int wstart = (int)roundf(roi[1]), hstart = (int)roundf(roi[2]), wend = (int)roundf(roi[3]), hend = (int)roundf(roi[4]);
  wstart = csc_clampi(wstart, 0, width_im - 1);
  hstart = csc_clampi(hstart, 0, height_im - 1);
  wend = csc_clampi(wend, 0, width_im - 1);
  hend = csc_clampi(hend, 0, height_im - 1);
  float width_roi = (float)(wend - wstart), height_roi = (float)(hend - hstart);
  float width_roi_inner = (float)(1.0 * width_roi / context_scale);
  float height_roi_inner = (float)(1.0 * height_roi / context_scale);
  float width_roi_outer = (float)(1.0 * width_roi * context_scale);
  float height_roi_outer = (float)(1.0 * height_roi * context_scale);
  const float wcenter = (float)(1.0 * (wend + wstart) / 2.0);
  const float hcenter = (float)(1.0 * (hend + hstart) / 2.0);
  const int wstart_inner = (int)round(wcenter - width_roi_inner / 2.0);
  const int hstart_inner = (int)round(hcenter - height_roi_inner / 2.0);
  const int wend_inner = (int)round(wcenter + width_roi_inner / 2.0);
  const int hend_inner = (int)round(hcenter + height_roi_inner / 2.0);
  const int wstart_outer = (int)round(fmax(wcenter - width_roi_outer / 2.0, 0.0));
  const int hstart_outer = (int)round(fmax(hcenter - height_roi_outer / 2.0, 0.0));
  const int wend_outer = (int)round(fmin(wcenter + width_roi_outer / 2.0, width_im - 1.0));
  const int hend_outer = (int)round(fmin(hcenter + height_roi_outer / 2.0, height_im - 1.0));
  width_roi = (float)(wend - wstart + 1);
  height_roi = (float)(hend - hstart + 1);
  width_roi_inner = (float)(wend_inner - wstart_inner + 1);
  height_roi_inner = (float)(hend_inner - hstart_inner + 1);
  width_roi_outer = (float)(wend_outer - wstart_outer + 1);
  height_roi_outer = (float)(hend_outer - hstart_outer + 1);
  const float sum_roi = csc_box_sum(table, width_im, hstart, wstart, hend, wend);
  const float sum_inner = csc_box_sum(table, width_im, hstart_inner, wstart_inner, hend_inner, wend_inner);
  const float sum_outer = csc_box_sum(table, width_im, hstart_outer, wstart_outer, hend_outer, wend_outer);
  const float area_roi = height_roi * width_roi;
  const float area_inner = height_roi_inner * width_roi_inner;
  const float area_outer = height_roi_outer * width_roi_outer;
  const float area_frame = fmaxf(area_roi - area_inner, 1.f);
  const float area_context = fmaxf(area_outer - area_roi, 1.f);
  const float sum_frame = sum_roi - sum_inner;
  const float sum_context = sum_outer - sum_roi;
  if (area_sqrt)
    return sum_frame / sqrtf(area_frame) - sum_context / sqrtf(area_context);
  return sum_frame / area_frame - sum_context / area_context;
}

// fixed-order block reduction of one value per thread (tree over LDS): same result on every run
template <int OP>  // 0 max, 1 min, 2 sum
__device__ __forceinline__ float csc_block_reduce(float v, float* red) {
  const int tid = threadIdx.x;
  __syncthreads();
  red[tid] = v;
  __syncthreads();
  for (int o = CSC_T / 2; o > 0; o >>= 1) {
    if (tid < o) {
      const float a = red[tid], b = red[tid + o];
      red[tid] = OP == 0 ? fmaxf(a, b) : OP == 1 ? fminf(a, b) : a + b;
    }
    __syncthreads();
  }
  return red[0];
}

__global__ __launch_bounds__(CSC_T) void csc_pool_kernel(const float* table, int H, int W, const float* rois, int M,
                                                         const float* scores, int K, int c, int area_sqrt,
                                                         float context_scale, float* Wout) {
  __shared__ float red[CSC_T];
  const int tid = threadIdx.x;
  float vmax = 0.f, vmin = 0.f, ssum = 0.f;
  for (int r = tid; r < M; r += CSC_T) {
    const float v = csc_pool_one(table, H, W, rois + 5 * (long)r, area_sqrt, context_scale);
    Wout[(long)r * K + c] = v;
    vmax = fmaxf(vmax, v);
    vmin = fminf(vmin, v);
    ssum += scores[(long)r * K + c];
  }
  const float max_value = csc_block_reduce<0>(vmax, red);
  const float min_value = csc_block_reduce<1>(vmin, red);
  const float pred = csc_block_reduce<2>(ssum, red);  // torch.sum(pred_class_logits, dim=0)[c], unclamped
  for (int r = tid; r < M; r += CSC_T) {
    float v = Wout[(long)r * K + c];
    if (max_value > 0 && min_value < 0)
      v = v > 0 ? v / max_value : v / (-min_value);
    else if (max_value > 0 && min_value == 0)
      v = v / max_value;
    else
      v = 1.0f;
    const float a = pred * v, b = (1 - pred) * 1;
    Wout[(long)r * K + c] = a + b;
  }
}

// ---- losses / seeds through the WSDDN product s[r,c] = softmax_c(cls)[r,c] * softmax_r(det)[r,c] -------------------
struct CscLossParams {
  const float* logits; long ld; int c_cls, c_det, K, M;
  const float* scores; const float* rowsm; const float* W; const float* onehot;
  int mode, cstar, mean_loss;
  float* loss; float* dlogits; long ld_d;
};

template <int LPR>
__global__ __launch_bounds__(CSC_T) void csc_loss_kernel(CscLossParams p) {
  constexpr int RPP = CSC_T / LPR, CPL = LPR == 64 ? 2 : 1, NC = LPR * CPL;
  __shared__ float red[RPP][NC];
  const int l = threadIdx.x % LPR, ph = threadIdx.x / LPR, K = p.K, M = p.M;
  int col[CPL];
  bool ok[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) { col[j] = l + j * LPR; ok[j] = col[j] < K; }
  // column reduce over the row phases, fixed order; every thread of a column gets the result
  auto colreduce = [&](float (&v)[CPL], bool is_max) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CPL; ++j) red[ph][col[j]] = v[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      float acc = is_max ? -FLT_MAX : 0.f;
      for (int q = 0; q < RPP; ++q) acc = is_max ? fmaxf(acc, red[q][col[j]]) : acc + red[q][col[j]];
      v[j] = acc;
    }
  };
  auto lanesum = [&](float v) {
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPR);
    return v;
  };
  float cmax[CPL], csum[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) { cmax[j] = -FLT_MAX; csum[j] = 0.f; }
  for (int r = ph; r < M; r += RPP)
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (ok[j]) cmax[j] = fmaxf(cmax[j], p.logits[(long)r * p.ld + p.c_det + col[j]]);
  colreduce(cmax, true);
  for (int r = ph; r < M; r += RPP)
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (ok[j]) csum[j] += expf(p.logits[(long)r * p.ld + p.c_det + col[j]] - cmax[j]);
  colreduce(csum, false);
  float gp[CPL], gn[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) { gp[j] = 0.f; gn[j] = 0.f; }
  if (p.mode == 0) {
    float sp[CPL], sn[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) { sp[j] = 0.f; sn[j] = 0.f; }
    for (int r = ph; r < M; r += RPP)
#pragma unroll
      for (int j = 0; j < CPL; ++j)
        if (ok[j]) {
          const float s = p.scores[(long)r * K + col[j]], w = p.W ? p.W[(long)r * K + col[j]] : 1.f;
          sp[j] += s * fmaxf(w, 0.f);
          sn[j] += s * fmaxf(-w, 0.f);
        }
    colreduce(sp, false);
    colreduce(sn, false);
    const float norm = p.mean_loss ? 1.f / (float)K : 1.f;  // F.binary_cross_entropy reduction; / PL.size(0) = 1 image
    float lp = 0.f, ln = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      if (!ok[j]) continue;
      const float y = p.onehot[col[j]];
      // torch.clamp(x, 1e-20, 1.0 - 1e-20): the upper bound IS 1.0 in fp32; the gradient passes inside the closed range
      const float xp = fminf(fmaxf(sp[j], 1e-20f), 1.0f), xn = fminf(fmaxf(sn[j], 1e-20f), 1.0f);
      // F.binary_cross_entropy: -(y * max(log x, -100) + (1 - y) * max(log1p(-x), -100))
      lp += -(y * fmaxf(logf(xp), -100.f) + (1.f - y) * fmaxf(log1pf(-xp), -100.f));
      ln += -fmaxf(log1pf(-xn), -100.f);  // NL = 0
      // binary_cross_entropy backward: (x - y) / max((1 - x) * x, 1e-12)
      gp[j] = (sp[j] >= 1e-20f && sp[j] <= 1.0f) ? (xp - y) / fmaxf((1.f - xp) * xp, 1e-12f) * norm : 0.f;
      gn[j] = (sn[j] >= 1e-20f && sn[j] <= 1.0f) ? xn / fmaxf((1.f - xn) * xn, 1e-12f) * norm : 0.f;
    }
    lp = lanesum(lp);
    ln = lanesum(ln);
    if (threadIdx.x == 0) { p.loss[0] = lp * norm; p.loss[1] = ln * norm; }
  }
  if (!p.dlogits) return;
  auto gw = [&](int r, int j) -> float {  // d(objective) / d score[r, col[j]]
    if (p.mode == 1) return col[j] == p.cstar ? 1.f : 0.f;
    const float w = p.W ? p.W[(long)r * K + col[j]] : 1.f;
    return gp[j] * fmaxf(w, 0.f) + gn[j] * fmaxf(-w, 0.f);
  };
  float T[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) T[j] = 0.f;
  for (int r = ph; r < M; r += RPP)
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (ok[j]) T[j] += gw(r, j) * p.scores[(long)r * K + col[j]];
  colreduce(T, false);
  // d cls[r,j] = G[r,j] s[r,j] - a[r,j] * sum_k G[r,k] s[r,k];   d det[r,c] = G[r,c] s[r,c] - b[r,c] * T[c]
  for (int r = ph; r < M; r += RPP) {
    float gs[CPL], dot = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { gs[j] = ok[j] ? gw(r, j) * p.scores[(long)r * K + col[j]] : 0.f; dot += gs[j]; }
    dot = lanesum(dot);
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (ok[j]) {
        const float b = expf(p.logits[(long)r * p.ld + p.c_det + col[j]] - cmax[j]) / csum[j];
        p.dlogits[(long)r * p.ld_d + p.c_cls + col[j]] = gs[j] - p.rowsm[(long)r * K + col[j]] * dot;
        p.dlogits[(long)r * p.ld_d + p.c_det + col[j]] = gs[j] - b * T[j];
      }
  }
}

}  // namespace

extern "C" {

// d score_c / d image (NHWC, `cpad` stored channels of which the first C are colours) -> cpg [H*W] fp32.
// scratch: one 32-bit word.  roi_heads_csc.py:456-464 (abs, max over dim 1, divide by the maximum).
int drn_csc_cpg(const void* dimg, int dtype, int cpad, int C, int H, int W, float* cpg, void* scratch, void* stream) {
  if (!dimg || !cpg || !scratch || C < 1 || C > cpad || H < 1 || W < 1) return DRN_ERR_ARG;
  if (dtype != DRN_F32 && dtype != DRN_BF16) return DRN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const long npx = (long)H * W;
  if (hipMemsetAsync(scratch, 0, 4, st) != hipSuccess) return DRN_ERR_LAUNCH;
  const int grid = (int)((npx + 255) / 256 < 2048 ? (npx + 255) / 256 : 2048);
  if (dtype == DRN_F32)
    hipLaunchKernelGGL((csc_absmax_kernel<DRN_F32>), dim3(grid), dim3(256), 0, st, dimg, cpad, C, npx, cpg, (unsigned*)scratch);
  else
    hipLaunchKernelGGL((csc_absmax_kernel<DRN_BF16>), dim3(grid), dim3(256), 0, st, dimg, cpad, C, npx, cpg, (unsigned*)scratch);
  hipLaunchKernelGGL(csc_scale_kernel, dim3(grid), dim3(256), 0, st, cpg, npx, (const unsigned*)scratch);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// one labelled class c: cpg [H*W] -> W[:, c] of W [M][K].  table: [H*W] floats of scratch.  scores [M][K] are the MIL
// scores (their column sum is the image-level prediction the weights are blended with).  csc_cuda.cu:398-535.
int drn_csc_weights(const float* cpg, int H, int W, float fg_threshold, const float* rois, int M, const float* scores,
                    int K, int c, int area_sqrt, float context_scale, float* table, float* Wout, void* stream) {
  if (!cpg || !rois || !scores || !table || !Wout || H < 1 || W < 1 || M < 1 || K < 1 || c < 0 || c >= K) return DRN_ERR_ARG;
  if ((long)H * W >= (1L << 24)) return DRN_ERR_UNSUPPORTED;  // fp32 counts stay exact below 2^24 pixels (as the reference's)
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(csc_rowscan_kernel, dim3(H), dim3(256), 0, st, cpg, table, W, 1.f * fg_threshold);
  hipLaunchKernelGGL(csc_colscan_kernel, dim3((W + 255) / 256), dim3(256), 0, st, table, H, W);
  hipLaunchKernelGGL(csc_pool_kernel, dim3(1), dim3(CSC_T), 0, st, (const float*)table, H, W, rois, M, scores, K, c,
                     area_sqrt, context_scale, Wout);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// mode 0: loss[0] = loss_cls_pos, loss[1] = loss_cls_neg (W may be null = all ones: past WSL.CSC_MAX_ITER) and
//         dlogits[:, cls | det columns] = d (loss_pos + loss_neg) / d logits
// mode 1: dlogits = d (sum_r score[r, cstar]) / d logits (loss untouched)
// One image (the reference head reads image_sizes[0] / gt_classes_img_oh[0]); K <= 128.
int drn_csc_loss(const float* logits, long ld, int c_cls, int c_det, int K, int M, const float* scores,
                 const float* row_softmax, const float* W, const float* onehot, int mode, int cstar, int mean_loss,
                 float* loss, float* dlogits, long ld_d, void* stream) {
  if (!logits || !scores || !row_softmax || K < 1 || K > 128 || M < 1 || (mode != 0 && mode != 1)) return DRN_ERR_ARG;
  if (mode == 0 && (!onehot || !loss)) return DRN_ERR_ARG;
  if (mode == 1 && (!dlogits || cstar < 0 || cstar >= K)) return DRN_ERR_ARG;
  CscLossParams p{logits, ld, c_cls, c_det, K, M, scores, row_softmax, W, onehot, mode, cstar, mean_loss, loss, dlogits, ld_d};
  hipStream_t st = (hipStream_t)stream;
  if (K <= 32)
    hipLaunchKernelGGL((csc_loss_kernel<32>), dim3(1), dim3(CSC_T), 0, st, p);
  else
    hipLaunchKernelGGL((csc_loss_kernel<64>), dim3(1), dim3(CSC_T), 0, st, p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

}  // extern "C"
