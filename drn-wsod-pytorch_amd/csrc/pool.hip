// HBM-bound feature-map kernels for gfx950: input normalisation, 2x2 max-pool, ROIPool / ROIAlign
// (fused with the OICR objectness scaling and written straight into the fc6 GEMM operand layout),
// and a tiled cast+transpose.  NHWC everywhere: a wave reads 64 consecutive channels = one
// 128-B (bf16) / 256-B (f32) line per spatial tap; ROI outputs are re-ordered through LDS so the
// [roi][c*P*P + bin] rows (the reference's NCHW flatten order, box_head.py:85-86) are written in
// full lines too.  Built with -ffp-contract=off: the arithmetic is the oracle's, op for op.
//
// Replaces: GeneralizedRCNNWSL.preprocess_image (projects/WSL/wsl/modeling/meta_arch/rcnn.py:242-249),
// nn.MaxPool2d(2, stride) (resnet_ws.py:214-215,403; vgg.py:99-100), torchvision RoIPool
// (detectron2/modeling/poolers.py:162-165), ROIAlign (detectron2/layers/csrc/ROIAlign/ROIAlign_cuda.cu:65-139)
// and the objectness scaling of roi_heads_oicr.py:342-343.
#include "drn_common.h"
#include <float.h>

namespace {

template <int DT>
__global__ void preprocess_kernel(const float* __restrict__ img, int C, int H, int W, typename ElemOf<DT>::type* out,
                                  int Hp, int Wp, int Cp, float m0, float m1, float m2, float s0, float s1, float s2) {
  const long total = (long)Hp * Wp * Cp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % Cp;
    const long hw = i / Cp;
    const int w = hw % Wp, h = hw / Wp;
    float v = 0.f;
    if (c < C && h < H && w < W) {
      const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
      const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
      v = (img[((long)c * H + h) * W + w] - mean) / sd;
    }
    ElemOf<DT>::st(out + i, v);
  }
}

// Pillow's BILINEAR resample of an 8-bit image (ResizeTransform.apply_image, detectron2/data/transforms/transform.py:101-122, calls
// PIL.Image.resize(..., BILINEAR) for uint8 images; the 16 augmented images of one TTA image take ~0.4 s of it on the host,
// projects/WSL/wsl/modeling/test_time_augmentation_avg.py:68-137).  Pillow resamples in two integer passes - horizontal, the
// result rounded to 8 bits, then vertical - with per-output-position windows [xmin, xmin + n) and coefficients scaled by 2^22
// (Resample.c: precompute_coeffs / normalize_coeffs_8bpc; the host computes them with Pillow's own double arithmetic); here
// one thread forms one output pixel: the horizontal results of its vertical window's rows on the fly, rounded exactly as
// Pillow stores them, then the vertical sum.  Output: fp32 [C][Ho][Wo] (the integers 0 .. 255), optionally mirrored
// (HFlipTransform) - what preprocess_kernel reads.
__global__ void resize_u8_kernel(const unsigned char* __restrict__ src, int H, int W, int C, float* __restrict__ dst, int Ho,
                                 int Wo, const int* __restrict__ xb, const int* __restrict__ xk, int ksx,
                                 const int* __restrict__ yb, const int* __restrict__ yk, int ksy, int flip) {
  constexpr int PB = 22, HALF = 1 << (PB - 1);
  const long total = (long)Ho * Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % Wo), yy = (int)(i / Wo);
    const int xmin = xb ? xb[2 * xx] : xx, xn = xb ? xb[2 * xx + 1] : 1;
    const int ymin = yb ? yb[2 * yy] : yy, yn = yb ? yb[2 * yy + 1] : 1;
    int acc[4] = {HALF, HALF, HALF, HALF};
    for (int y = 0; y < yn; ++y) {
      const unsigned char* row = src + ((long)(ymin + y) * W + xmin) * C;
      int h[4];
      if (xb) {
        int a[4] = {HALF, HALF, HALF, HALF};
        for (int x = 0; x < xn; ++x) {
          const int k = xk[(long)xx * ksx + x];
          for (int c = 0; c < C; ++c) a[c] += (int)row[x * C + c] * k;
        }
        for (int c = 0; c < C; ++c) h[c] = min(max(a[c] >> PB, 0), 255);
      } else {
        for (int c = 0; c < C; ++c) h[c] = row[c];
      }
      if (yb) {
        const int k = yk[(long)yy * ksy + y];
        for (int c = 0; c < C; ++c) acc[c] += h[c] * k;
      } else {
        for (int c = 0; c < C; ++c) acc[c] = h[c];
      }
    }
    const int xo = flip ? Wo - 1 - xx : xx;
    for (int c = 0; c < C; ++c)
      dst[((long)c * Ho + yy) * Wo + xo] = (float)(yb ? min(max(acc[c] >> PB, 0), 255) : acc[c]);
  }
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// one thread = one 16-B channel vector of one output pixel
template <int DT>
__global__ void maxpool2x2_kernel(const char* __restrict__ x, char* __restrict__ y, int Nb, int H, int W, int C,
                                  int Ho, int Wo, int stride) {
  constexpr int ES = EsOf<DT>::value;
  constexpr int V = 16 / ES;
  const int cv = C / V;
  const long total = (long)Nb * Ho * Wo * cv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (i % cv) * V;
    long t = i / cv;
    const int wo = t % Wo; t /= Wo;
    const int ho = t % Ho;
    const int n = t / Ho;
    const char* p00 = x + (((long)(n * H + ho * stride) * W + wo * stride) * C + c) * ES;
    const long dw = (long)C * ES, dh = (long)W * C * ES;
    char* dst = y + (((long)(n * Ho + ho) * Wo + wo) * C + c) * ES;
    if constexpr (DT == DRN_FP8) {
      // the quantised trunk pools post-ReLU tensors only: non-negative e4m3 values order like their bytes
      const u32x4_t a = *(const u32x4_t*)p00, b = *(const u32x4_t*)(p00 + dw);
      const u32x4_t d = *(const u32x4_t*)(p00 + dh), e = *(const u32x4_t*)(p00 + dh + dw);
      auto mx = [](unsigned ua, unsigned ub, unsigned ud, unsigned ue) -> unsigned {
        unsigned o = 0;
#pragma unroll
        for (int k = 0; k < 32; k += 8) {
          const unsigned m0 = max((ua >> k) & 0xffu, (ub >> k) & 0xffu), m1 = max((ud >> k) & 0xffu, (ue >> k) & 0xffu);
          o |= max(m0, m1) << k;
        }
        return o;
      };
      u32x4_t o;
      o.x = mx(a.x, b.x, d.x, e.x);
      o.y = mx(a.y, b.y, d.y, e.y);
      o.z = mx(a.z, b.z, d.z, e.z);
      o.w = mx(a.w, b.w, d.w, e.w);
      *(u32x4_t*)dst = o;
    } else if constexpr (DT == DRN_F32) {
      const f32x4_t a = *(const f32x4_t*)p00, b = *(const f32x4_t*)(p00 + dw);
      const f32x4_t d = *(const f32x4_t*)(p00 + dh), e = *(const f32x4_t*)(p00 + dh + dw);
      f32x4_t o;
      o.x = fmaxf(fmaxf(a.x, b.x), fmaxf(d.x, e.x));
      o.y = fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y));
      o.z = fmaxf(fmaxf(a.z, b.z), fmaxf(d.z, e.z));
      o.w = fmaxf(fmaxf(a.w, b.w), fmaxf(d.w, e.w));
      *(f32x4_t*)dst = o;
    } else {
      const u32x4_t a = *(const u32x4_t*)p00, b = *(const u32x4_t*)(p00 + dw);
      const u32x4_t d = *(const u32x4_t*)(p00 + dh), e = *(const u32x4_t*)(p00 + dh + dw);
      auto mx = [](unsigned ua, unsigned ub, unsigned ud, unsigned ue) -> unsigned {
        // two packed bf16 per dword; bf16 -> f32 is a shift, the max of bf16 values is again bf16
        const float lo = fmaxf(fmaxf(__builtin_bit_cast(float, ua << 16), __builtin_bit_cast(float, ub << 16)),
                               fmaxf(__builtin_bit_cast(float, ud << 16), __builtin_bit_cast(float, ue << 16)));
        const float hi = fmaxf(fmaxf(__builtin_bit_cast(float, ua & 0xffff0000u), __builtin_bit_cast(float, ub & 0xffff0000u)),
                               fmaxf(__builtin_bit_cast(float, ud & 0xffff0000u), __builtin_bit_cast(float, ue & 0xffff0000u)));
        return (__builtin_bit_cast(unsigned, lo) >> 16) | (__builtin_bit_cast(unsigned, hi) & 0xffff0000u);
      };
      u32x4_t o;
      o.x = mx(a.x, b.x, d.x, e.x);
      o.y = mx(a.y, b.y, d.y, e.y);
      o.z = mx(a.z, b.z, d.z, e.z);
      o.w = mx(a.w, b.w, d.w, e.w);
      *(u32x4_t*)dst = o;
    }
  }
}

struct RoiParams {
  const char* feat;  // NHWC
  const float* rois;  // [M][5]
  const float* obj;   // [M] objectness logits or null; output is scaled by (obj + 1)
  char* out;          // [M][ld_out], k = c*P*P + ph*P + pw
  int32_t* argmax;    // [M][C*P*P] (h*W + w, or -1) or null  (ROIPool only)
  int N, H, W, C, P, M;
  float scale;
  long ld_out;
  int sampling_ratio, aligned;
  int lds_px;  // pixels of staging LDS available per block (0 = direct path only)
  char* out_t;   // optional transposed copy [C*P*P][ld_out_t] (column = roi), or null
  long ld_out_t;
  int gpw;       // whole-map kernel: consecutive 8-ROI groups handled by one block (per staged map slice)
  int cpb;       // 64-ROI kernel: consecutive 8-channel chunks handled by one block (bin bounds computed once per block)
  int pf;        // 64-ROI kernel: two map buffers, the next chunk's slice is fetched under this chunk's scan
  int t_c0;      // first channel whose rows of out_t are needed (drn_roi_pool_nhwc_t); kernels may write more
  int c_begin;   // 64-ROI kernel: first channel it handles (the lane-per-bin kernel writes A; this one then only the A^T tail)
  int lane_g;    // lane-per-bin kernel: ROIs per group (one ROI per lane of every wave: <= 64)
  int lane_reps;  // lane-per-bin kernel: groups a block walks with ONE staged slice (large maps: the staging is L2 traffic ~ groups x map)
  int walk;       // walking lane-per-bin kernel: consecutive channel chunks a block walks with ONE window table of its ROIs
  int walk_wp;    // its LDS row pitch in cells (odd)
  const char* cm;  // its chunk-major, order-mapped copy of the map ([N][C/8][H*W] cells of 16 bytes) or null
  unsigned walk_wmagic;  // ceil(2^32 / W): pixel -> row by one v_mul_hi
};

constexpr int RP_CH = 64;    // channels per block = one wave-wide line of NHWC
constexpr int RP_MAXBIN = 64;  // P*P <= 64 (P <= 8)

// MODE 0: RoIPool   MODE 1: ROIAlign (ROIAlign_cuda.cu:65-139 semantics)
template <int DT_IN, int DT_OUT, int MODE>
__global__ __launch_bounds__(256) void roi_kernel(RoiParams p) {
  using EI = ElemOf<DT_IN>;
  using TI = typename EI::type;
  using EO = ElemOf<DT_OUT>;
  using TO = typename EO::type;
  __shared__ float tile[RP_CH][RP_MAXBIN + 1];
  __shared__ int atile[RP_CH][RP_MAXBIN + 1];
  const int m = blockIdx.x;
  const int c0 = blockIdx.y * RP_CH;
  const int cl = threadIdx.x & 63, bg = threadIdx.x >> 6;  // lane = channel, 4 bin groups
  const int c = c0 + cl;
  const float* roi = p.rois + 5 * (long)m;
  const int b = (int)roi[0];
  const int PP = p.P * p.P;
  const float mul = p.obj ? p.obj[m] + 1.f : 1.f;
  const TI* fb = (const TI*)p.feat + (long)b * p.H * p.W * p.C;
  if (MODE == 0) {
    const int x1 = (int)roundf(roi[1] * p.scale), y1 = (int)roundf(roi[2] * p.scale);
    const int x2 = (int)roundf(roi[3] * p.scale), y2 = (int)roundf(roi[4] * p.scale);
    const int rw = max(x2 - x1 + 1, 1), rh = max(y2 - y1 + 1, 1);
    const float bin_h = (float)rh / (float)p.P, bin_w = (float)rw / (float)p.P;
    // LDS path: the union of all bins is the clipped box [y1, y1+rh) x [x1, x1+rw); stage those pixels of this
    // block's 64 channels with 16-B loads (all independent => all in flight), then scan bins out of LDS.
    const int ry0 = min(max(y1, 0), p.H), ry1 = min(max(y1 + rh, 0), p.H);
    const int rx0 = min(max(x1, 0), p.W), rx1 = min(max(x1 + rw, 0), p.W);
    const int rww = rx1 - rx0, npx = (ry1 - ry0) * rww;
    if (p.lds_px > 0 && npx <= p.lds_px && c0 + RP_CH <= p.C) {
      extern __shared__ __attribute__((aligned(16))) char stage[];
      constexpr int ESI = DT_IN == DRN_BF16 ? 2 : 4;
      constexpr int VPL = RP_CH * ESI / 16;  // lanes (16 B each) per pixel
      for (int i = threadIdx.x; i < npx * VPL; i += 256) {
        const int px = i / VPL, v = i - px * VPL;
        const int h = ry0 + px / rww, w = rx0 + px % rww;
        *(i32x4_t*)(stage + ((long)px * RP_CH) * ESI + v * 16) =
            *(const i32x4_t*)((const char*)(fb + ((long)h * p.W + w) * p.C + c0) + v * 16);
      }
      __syncthreads();
      const TI* st = (const TI*)stage;
      for (int bin = bg; bin < PP; bin += 4) {
        const int ph = bin / p.P, pw = bin - ph * p.P;
        int hs = (int)floorf((float)ph * bin_h), he = (int)ceilf((float)(ph + 1) * bin_h);
        int ws = (int)floorf((float)pw * bin_w), we = (int)ceilf((float)(pw + 1) * bin_w);
        hs = min(max(hs + y1, 0), p.H); he = min(max(he + y1, 0), p.H);
        ws = min(max(ws + x1, 0), p.W); we = min(max(we + x1, 0), p.W);
        const bool empty = he <= hs || we <= ws;
        float best = empty ? 0.f : -FLT_MAX;
        int besti = -1;
        for (int h = hs; h < he; ++h)
          for (int w = ws; w < we; ++w) {
            const float v = EI::ld(st + ((h - ry0) * rww + (w - rx0)) * RP_CH + cl);
            if (v > best) { best = v; besti = h * p.W + w; }
          }
        tile[cl][bin] = best * mul;
        atile[cl][bin] = besti;
      }
    } else
    for (int bin = bg; bin < PP; bin += 4) {
      const int ph = bin / p.P, pw = bin - ph * p.P;
      int hs = (int)floorf((float)ph * bin_h), he = (int)ceilf((float)(ph + 1) * bin_h);
      int ws = (int)floorf((float)pw * bin_w), we = (int)ceilf((float)(pw + 1) * bin_w);
      hs = min(max(hs + y1, 0), p.H); he = min(max(he + y1, 0), p.H);
      ws = min(max(ws + x1, 0), p.W); we = min(max(we + x1, 0), p.W);
      const bool empty = he <= hs || we <= ws;
      float best = empty ? 0.f : -FLT_MAX;
      int besti = -1;
      if (c < p.C)
        for (int h = hs; h < he; ++h)
          for (int w = ws; w < we; ++w) {
            const float v = EI::ld(fb + ((long)h * p.W + w) * p.C + c);
            if (v > best) { best = v; besti = h * p.W + w; }
          }
      tile[cl][bin] = best * mul;
      atile[cl][bin] = besti;
    }
  } else {
    const float off = p.aligned ? 0.5f : 0.f;
    const float sw = roi[1] * p.scale - off, sh = roi[2] * p.scale - off;
    const float ew = roi[3] * p.scale - off, eh = roi[4] * p.scale - off;
    float rw = ew - sw, rh = eh - sh;
    if (!p.aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    const float bin_h = rh / (float)p.P, bin_w = rw / (float)p.P;
    const int gh = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(rh / p.P);
    const int gw = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(rw / p.P);
    const float count = (float)max(gh * gw, 1);
    for (int bin = bg; bin < PP; bin += 4) {
      const int ph = bin / p.P, pw = bin - ph * p.P;
      float acc = 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        const float yy = sh + ph * bin_h + (float)(iy + .5f) * bin_h / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float xx = sw + pw * bin_w + (float)(ix + .5f) * bin_w / (float)gw;
          float x = xx, y = yy;
          if (y < -1.0f || y > p.H || x < -1.0f || x > p.W) continue;
          if (y <= 0) y = 0;
          if (x <= 0) x = 0;
          int yl = (int)y, xl = (int)x, yh, xh;
          if (yl >= p.H - 1) { yh = yl = p.H - 1; y = (float)yl; } else yh = yl + 1;
          if (xl >= p.W - 1) { xh = xl = p.W - 1; x = (float)xl; } else xh = xl + 1;
          const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
          if (c < p.C) {
            const float v1 = EI::ld(fb + ((long)yl * p.W + xl) * p.C + c), v2 = EI::ld(fb + ((long)yl * p.W + xh) * p.C + c);
            const float v3 = EI::ld(fb + ((long)yh * p.W + xl) * p.C + c), v4 = EI::ld(fb + ((long)yh * p.W + xh) * p.C + c);
            acc += hy * hx * v1 + hy * lx * v2 + ly * hx * v3 + ly * lx * v4;
          }
        }
      }
      tile[cl][bin] = acc / count * mul;
    }
  }
  __syncthreads();
  // coalesced write-out: k = c*PP + bin is contiguous over this block's 64 channels
  const int nvalid = min(RP_CH, p.C - c0) * PP;
  TO* orow = (TO*)p.out + (long)m * p.ld_out + (long)c0 * PP;
  constexpr int ESO = DT_OUT == DRN_BF16 ? 2 : 4;
  constexpr int VE = 16 / ESO;  // elements per 16-B store
  if ((nvalid % VE) == 0 && ((((long)m * p.ld_out + (long)c0 * PP) * ESO) & 15) == 0 && (((uintptr_t)p.out) & 15) == 0) {
    for (int v = threadIdx.x; v < nvalid / VE; v += 256) {
      float f[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const int i = v * VE + e, lc = i / PP;
        f[e] = tile[lc][i - lc * PP];
      }
      i32x4_t o;
      if constexpr (DT_OUT == DRN_BF16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (int)((uint32_t)f32_to_bf16(f[2 * e]) | ((uint32_t)f32_to_bf16(f[2 * e + 1]) << 16));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __builtin_bit_cast(int, f[e]);
      }
      *(i32x4_t*)((char*)orow + (long)v * 16) = o;
    }
  } else {
    for (int i = threadIdx.x; i < nvalid; i += 256) {
      const int lc = i / PP;
      EO::st(orow + i, tile[lc][i - lc * PP]);
    }
  }
  if (MODE == 0 && p.argmax)
    for (int i = threadIdx.x; i < nvalid; i += 256) {
      const int lc = i / PP;
      p.argmax[(long)m * p.C * PP + (long)c0 * PP + i] = atile[lc][i - lc * PP];
    }
}

// Backward of RoIPool (MODE 0: scatter to the saved arg-max) / ROIAlign (MODE 1: bilinear scatter,
// ROIAlign_cuda.cu:141-250 semantics), fused with the objectness scaling of the forward.  One block = one ROI x 64
// channels like roi_kernel: the [64 x P*P] slice of grad_out is read coalesced into LDS, then lane = channel scatters
// with fp32 atomics into the NHWC gradient map - neighbouring lanes hit neighbouring addresses.  Like the
// reference's CUDA kernels the accumulation order is not fixed (only used when the backbone trains).
struct RoiBwdParams {
  const char* grad_out;  // [M][ld], k = c*P*P + bin
  const float* rois; const float* obj; const int32_t* argmax;
  float* dfeat;          // [N][H][W][C] fp32, zeroed by the launcher
  int N, H, W, C, P, M; float scale; long ld; int sampling_ratio, aligned;
};

template <int DT, int MODE>
__global__ __launch_bounds__(256) void roi_bwd_kernel(RoiBwdParams p) {
  using E = ElemOf<DT>;
  using T = typename E::type;
  __shared__ float tile[RP_CH][RP_MAXBIN + 1];
  __shared__ int atile[RP_CH][RP_MAXBIN + 1];
  const int m = blockIdx.x, c0 = blockIdx.y * RP_CH;
  const int cl = threadIdx.x & 63, bg = threadIdx.x >> 6;
  const int c = c0 + cl;
  const float* roi = p.rois + 5 * (long)m;
  const int b = (int)roi[0];
  const int PP = p.P * p.P;
  const float mul = p.obj ? p.obj[m] + 1.f : 1.f;
  const int nvalid = min(RP_CH, p.C - c0) * PP;
  const T* grow = (const T*)p.grad_out + (long)m * p.ld + (long)c0 * PP;
  for (int i = threadIdx.x; i < nvalid; i += 256) {
    const int lc = i / PP;
    tile[lc][i - lc * PP] = E::ld(grow + i) * mul;
    if (MODE == 0) atile[lc][i - lc * PP] = p.argmax[(long)m * p.C * PP + (long)c0 * PP + i];
  }
  __syncthreads();
  if (c >= p.C) return;
  float* gb = p.dfeat + (long)b * p.H * p.W * p.C + c;
  if (MODE == 0) {
    for (int bin = bg; bin < PP; bin += 4) {
      const int a = atile[cl][bin];
      if (a >= 0) atomicAdd(gb + (long)a * p.C, tile[cl][bin]);
    }
  } else {
    const float off = p.aligned ? 0.5f : 0.f;
    const float sw = roi[1] * p.scale - off, sh = roi[2] * p.scale - off;
    const float ew = roi[3] * p.scale - off, eh = roi[4] * p.scale - off;
    float rw = ew - sw, rh = eh - sh;
    if (!p.aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    const float bin_h = rh / (float)p.P, bin_w = rw / (float)p.P;
    const int gh = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(rh / p.P);
    const int gw = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(rw / p.P);
    const float count = (float)(gh * gw);
    for (int bin = bg; bin < PP; bin += 4) {
      const int ph = bin / p.P, pw = bin - ph * p.P;
      const float g = tile[cl][bin];
      for (int iy = 0; iy < gh; ++iy) {
        const float yy = sh + ph * bin_h + (float)(iy + .5f) * bin_h / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float xx = sw + pw * bin_w + (float)(ix + .5f) * bin_w / (float)gw;
          float x = xx, y = yy;
          if (y < -1.0f || y > p.H || x < -1.0f || x > p.W) continue;
          if (y <= 0) y = 0;
          if (x <= 0) x = 0;
          int yl = (int)y, xl = (int)x, yh, xh;
          if (yl >= p.H - 1) { yh = yl = p.H - 1; y = (float)yl; } else yh = yl + 1;
          if (xl >= p.W - 1) { xh = xl = p.W - 1; x = (float)xl; } else xh = xl + 1;
          const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
          atomicAdd(gb + ((long)yl * p.W + xl) * p.C, g * (hy * hx) / count);
          atomicAdd(gb + ((long)yl * p.W + xh) * p.C, g * (hy * lx) / count);
          atomicAdd(gb + ((long)yh * p.W + xl) * p.C, g * (ly * hx) / count);
          atomicAdd(gb + ((long)yh * p.W + xh) * p.C, g * (ly * lx) / count);
        }
      }
    }
  }
}

// ---- backward of the conv trunk (only when MODEL.BACKBONE.FREEZE_AT < 5) -------------------------------------
// Transposed im2col: out[(ci*KH + kh)*KW + kw][p] = x[n, ho*s + kh*d - pad, wo*s + kw*d - pad, ci] (0 outside),
// p = (n*Ho + ho)*Wo + wo.  It is the K-major B operand of the weight-gradient GEMM  dW[co][ci,kh,kw] = g^T . out^T,
// whose row order makes dW come out in the [Cout, Cin, KH, KW] state_dict layout.  64 pixels x 64 channels per block
// for one tap: channel-contiguous reads, pixel-contiguous writes through an LDS tile.
template <int DT>
__global__ __launch_bounds__(256) void im2col_t_kernel(const char* __restrict__ x, char* __restrict__ out, int Nb, int H,
                                                       int W, int Cin, int ldc, int KH, int KW, int stride, int pad,
                                                       int dil, int Ho, int Wo, long ld_out) {
  using E = ElemOf<DT>;
  using T = typename E::type;
  __shared__ T t[64][66];
  const int tap = blockIdx.z, kh = tap / KW, kw = tap - kh * KW;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int P = Nb * Ho * Wo;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int p = p0 + i, c = c0 + tx;
    T v = (T)0;
    if (p < P && c < Cin) {
      const int wo = p % Wo, ho = (p / Wo) % Ho, n = p / (Wo * Ho);
      const int hi = ho * stride + kh * dil - pad, wi = wo * stride + kw * dil - pad;
      if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = ((const T*)x)[((long)(n * H + hi) * W + wi) * ldc + c];
    }
    t[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, p = p0 + tx;
    if (c < Cin && p < P) ((T*)out)[((long)(c * KH + kh) * KW + kw) * ld_out + p] = t[tx][i];
  }
}

// d(x) of MaxPool2d(2, stride s, padding 0): every input pixel gathers from the (at most 4) windows that contain it
// and takes a window's gradient iff it is that window's first maximum in scan order (torch: `val > maxval`) -
// deterministic, no atomics even for the overlapping stride-1 windows of the dilated configs.
template <int DT>
__global__ void maxpool2x2_bwd_kernel(const char* __restrict__ x, const char* __restrict__ dy, char* __restrict__ dx,
                                      int Nb, int H, int W, int C, int Ho, int Wo, int stride) {
  using E = ElemOf<DT>;
  using T = typename E::type;
  const long total = (long)Nb * H * W * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % C;
    long r = i / C;
    const int w = r % W; r /= W;
    const int h = r % H;
    const int n = r / H;
    float acc = 0.f;
    for (int dh = 0; dh < 2; ++dh) {
      const int hs = h - dh;  // window start row if this pixel is row dh of the window
      if (hs < 0 || hs % stride) continue;
      const int ho = hs / stride;
      if (ho >= Ho) continue;
      for (int dw = 0; dw < 2; ++dw) {
        const int ws = w - dw;
        if (ws < 0 || ws % stride) continue;
        const int wo = ws / stride;
        if (wo >= Wo) continue;
        const T* base = (const T*)x + ((long)(n * H + hs) * W + ws) * C + c;
        float best = E::ld(base);
        int arg = 0;
        const float v1 = E::ld(base + C), v2 = E::ld(base + (long)W * C), v3 = E::ld(base + (long)W * C + C);
        if (v1 > best) { best = v1; arg = 1; }
        if (v2 > best) { best = v2; arg = 2; }
        if (v3 > best) { best = v3; arg = 3; }
        if (arg == dh * 2 + dw) acc += E::ld((const T*)dy + ((long)(n * Ho + ho) * Wo + wo) * C + c);
      }
    }
    E::st((T*)dx + i, acc);
  }
}

template <int DT>
__global__ void add_kernel(const char* __restrict__ a, const char* __restrict__ b, char* __restrict__ out, long n) {
  using E = ElemOf<DT>;
  using T = typename E::type;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    E::st((T*)out + i, E::ld((const T*)a + i) + E::ld((const T*)b + i));
}

// ROIPool specialised for the 7x7 pooler every DRN-WSOD config uses.  One block = one ROI x 256 channels (four
// 64-channel chunks, so the ROI geometry, the 49 bin rectangles and the window's pixel table are computed once);
// per chunk the window pixels are staged in LDS by 16-B loads, the bin maxima come out of LDS, and the
// [c*49 + bin] run (64*49 contiguous outputs) leaves in 16-B stores.  All divisions are by constants.
template <int DT_IN, int DT_OUT>
__global__ __launch_bounds__(256) void roi_pool7_kernel(RoiParams p) {
  using EI = ElemOf<DT_IN>;
  using TI = typename EI::type;
  using EO = ElemOf<DT_OUT>;
  using TO = typename EO::type;
  constexpr int PP = 49, CHUNKS = 4;
  constexpr int ESI = DT_IN == DRN_BF16 ? 2 : 4, ESO = DT_OUT == DRN_BF16 ? 2 : 4;
  constexpr int VPL = RP_CH * ESI / 16, VE = 16 / ESO;
  extern __shared__ __attribute__((aligned(16))) char stage[];  // [lds_px][64] TI
  __shared__ float tile[RP_CH][PP + 1];
  __shared__ int bins[PP][4];
  __shared__ int pxoff[256];
  const int m = blockIdx.x;
  const int cl = threadIdx.x & 63, bg = threadIdx.x >> 6;
  const float* roi = p.rois + 5 * (long)m;
  const int b = (int)roi[0];
  const float mul = p.obj ? p.obj[m] + 1.f : 1.f;
  const TI* fb = (const TI*)p.feat + (long)b * p.H * p.W * p.C;
  const int x1 = (int)roundf(roi[1] * p.scale), y1 = (int)roundf(roi[2] * p.scale);
  const int x2 = (int)roundf(roi[3] * p.scale), y2 = (int)roundf(roi[4] * p.scale);
  const int rw = max(x2 - x1 + 1, 1), rh = max(y2 - y1 + 1, 1);
  const float bin_h = (float)rh / 7.f, bin_w = (float)rw / 7.f;
  const int ry0 = min(max(y1, 0), p.H), ry1 = min(max(y1 + rh, 0), p.H);
  const int rx0 = min(max(x1, 0), p.W), rx1 = min(max(x1 + rw, 0), p.W);
  const int rww = rx1 - rx0, npx = (ry1 - ry0) * rww;  // npx <= lds_px <= 256 guaranteed by the launcher
  if (threadIdx.x < PP) {
    const int ph = threadIdx.x / 7, pw = threadIdx.x - ph * 7;
    int hs = (int)floorf((float)ph * bin_h), he = (int)ceilf((float)(ph + 1) * bin_h);
    int ws = (int)floorf((float)pw * bin_w), we = (int)ceilf((float)(pw + 1) * bin_w);
    bins[threadIdx.x][0] = min(max(hs + y1, 0), p.H); bins[threadIdx.x][1] = min(max(he + y1, 0), p.H);
    bins[threadIdx.x][2] = min(max(ws + x1, 0), p.W); bins[threadIdx.x][3] = min(max(we + x1, 0), p.W);
  }
  if (threadIdx.x < npx) {
    const int hh = threadIdx.x / rww;
    pxoff[threadIdx.x] = (ry0 + hh) * p.W + rx0 + (threadIdx.x - hh * rww);
  }
  __syncthreads();
  for (int ch = 0; ch < CHUNKS; ++ch) {
    const int c0 = (blockIdx.y * CHUNKS + ch) * RP_CH;
    if (c0 >= p.C) break;
    for (int i = threadIdx.x; i < npx * VPL; i += 256) {
      const int px = i / VPL, v = i - px * VPL;
      *(i32x4_t*)(stage + (long)px * (RP_CH * ESI) + v * 16) =
          *(const i32x4_t*)((const char*)(fb + (long)pxoff[px] * p.C + c0) + v * 16);
    }
    __syncthreads();
    const TI* st = (const TI*)stage;
    for (int bin = bg; bin < PP; bin += 4) {
      const int hs = bins[bin][0], he = bins[bin][1], ws = bins[bin][2], we = bins[bin][3];
      float best = (he <= hs || we <= ws) ? 0.f : -FLT_MAX;
      for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) best = fmaxf(best, EI::ld(st + ((h - ry0) * rww + (w - rx0)) * RP_CH + cl));
      tile[cl][bin] = best * mul;
    }
    __syncthreads();
    char* orow = (char*)((TO*)p.out + (long)m * p.ld_out + (long)c0 * PP);
    for (int v = threadIdx.x; v < RP_CH * PP / VE; v += 256) {
      float f[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const int i = v * VE + e, lc = i / PP;
        f[e] = tile[lc][i - lc * PP];
      }
      i32x4_t o;
      if constexpr (DT_OUT == DRN_BF16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (int)((uint32_t)f32_to_bf16(f[2 * e]) | ((uint32_t)f32_to_bf16(f[2 * e + 1]) << 16));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __builtin_bit_cast(int, f[e]);
      }
      *(i32x4_t*)(orow + (long)v * 16) = o;
    }
    __syncthreads();
  }
}

// 7x7 ROIPool, whole-map variant: when a CH-channel slice of one image's feature map fits in LDS (C4/DC5 maps of
// VOC-sized images: 14x14 .. 28x28 pixels) a block stages that slice ONCE and pools ROI_GROUP = 8 consecutive ROIs
// out of it - no per-ROI trip to L2, two barriers per block instead of three per (ROI, chunk).  The [8][CH*49] result
// tile leaves LDS twice: as the 8 row runs of A (16-B stores, k = c*49 + bin) and, when out_t is given, as CH*49
// 16-B column runs of A^T (8 ROIs wide), which replaces the separate 2 x 205 MB transpose pass of the fc6 operand.
// Block ids are XCD-remapped chunk-major, so the 8 blocks that complete one 128-B line of A^T share an XCD's L2.
// two packed bf16 -> two packed int16 with the same ordering (and back: the map is an involution); lets window
// maxima run as v_pk_max_i16 on whole 32-bit words.  -0 orders below +0, NaNs order as large magnitudes.
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int bf16x2_order(int x) {
  const s16x2_t v = __builtin_bit_cast(s16x2_t, x);
  const s16x2_t m = (v >> (short)15) & (short)0x7fff;
  return __builtin_bit_cast(int, (s16x2_t)(v ^ m));
}
__device__ __forceinline__ int pk_max_i16(int a, int b) {
  return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, a), __builtin_bit_cast(s16x2_t, b)));
}

constexpr int ROI_GROUP = 8;
template <int DT, int CH>
__global__ __launch_bounds__(256) void roi_pool7_map_kernel(RoiParams p) {
  using E = ElemOf<DT>;
  using T = typename E::type;
  constexpr int PP = 49, ES = DT == DRN_BF16 ? 2 : 4;
  constexpr int RUN = CH * PP;       // outputs per ROI in this chunk
  constexpr int VPL = CH * ES / 16;  // 16-B vectors per pixel
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = p.H * p.W;
  char* map = smem;                                          // [HW][CH]
  T* tile = (T*)(smem + (((long)HW * CH * ES + 15) & ~15L));  // [ROI_GROUP][RUN]
  __shared__ int hb[ROI_GROUP][7][2], wb[ROI_GROUP][7][2], bidx[ROI_GROUP];
  __shared__ float mulv[ROI_GROUP];
  const int ngroups = (p.M + ROI_GROUP - 1) / ROI_GROUP;
  const int nblk = (ngroups + p.gpw - 1) / p.gpw;  // blocks per channel chunk
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int chunk = logical / nblk, gblk = logical - chunk * nblk;
  const int c0 = chunk * CH;
  const int tid = threadIdx.x;
  const int r = tid >> 5, lane = tid & 31;
  int staged = -1;  // image whose map slice currently sits in LDS
  // several ROI groups per block share one staged map slice: for the large maps of test-time scales the slice is far
  // bigger than a group's output, and re-staging it per group was the whole cost
  for (int gi = 0; gi < p.gpw; ++gi) {
  const int group = gblk * p.gpw + gi;
  if (group >= ngroups) break;
  const int m0 = group * ROI_GROUP;
  const int nr = min(ROI_GROUP, p.M - m0);
  __syncthreads();  // the previous group's tile / bounds are no longer read
  if (tid < ROI_GROUP * 7) {
    const int r = tid / 7, i = tid - r * 7;
    if (r < nr) {
      const float* roi = p.rois + 5 * (long)(m0 + r);
      const int x1 = (int)roundf(roi[1] * p.scale), y1 = (int)roundf(roi[2] * p.scale);
      const int x2 = (int)roundf(roi[3] * p.scale), y2 = (int)roundf(roi[4] * p.scale);
      const int rw = max(x2 - x1 + 1, 1), rh = max(y2 - y1 + 1, 1);
      const float bin_h = (float)rh / 7.f, bin_w = (float)rw / 7.f;
      hb[r][i][0] = min(max((int)floorf((float)i * bin_h) + y1, 0), p.H);
      hb[r][i][1] = min(max((int)ceilf((float)(i + 1) * bin_h) + y1, 0), p.H);
      wb[r][i][0] = min(max((int)floorf((float)i * bin_w) + x1, 0), p.W);
      wb[r][i][1] = min(max((int)ceilf((float)(i + 1) * bin_w) + x1, 0), p.W);
      if (i == 0) {
        bidx[r] = (int)roi[0];
        mulv[r] = p.obj ? p.obj[m0 + r] + 1.f : 1.f;
      }
    }
  }
  __syncthreads();
  const int r = tid >> 5, lane = tid & 31;
  for (int r0 = 0; r0 < nr;) {  // one pass per run of ROIs on the same image (one pass unless a group straddles images)
    const int b = bidx[r0];
    int r1 = r0 + 1;
    while (r1 < nr && bidx[r1] == b) ++r1;
    if (b != staged) {  // uniform over the block
      const char* fb = p.feat + ((long)b * HW * p.C + c0) * ES;
      for (int i = tid; i < HW * VPL; i += 256) {
        const int px = i / VPL, v = i - px * VPL;
        i32x4_t x = *(const i32x4_t*)(fb + (long)px * p.C * ES + v * 16);
        if constexpr (DT == DRN_BF16) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = bf16x2_order(x[e]);
        }
        *(i32x4_t*)(map + (long)i * 16) = x;
      }
      staged = b;
      __syncthreads();
    }
    if (r >= r0 && r < r1) {
      const float mul = mulv[r];
      T* trow = tile + (long)r * RUN;
      if constexpr (DT == DRN_BF16) {
        // lane = (channel octet, bin subset): one 16-B LDS read feeds four packed int16 maxima (8 channels)
        constexpr int NOCT = CH / 8, NSUB = 32 / NOCT;
        const int oct = lane % NOCT, bs = lane / NOCT;
        for (int bin = bs; bin < PP; bin += NSUB) {
          const int ph = bin / 7, pw = bin - ph * 7;
          const int hs = hb[r][ph][0], he = hb[r][ph][1], ws = wb[r][pw][0], we = wb[r][pw][1];
          const int lo = (int)0x80008000u;
          i32x4_t acc = {lo, lo, lo, lo};
          for (int h = hs; h < he; ++h) {
            const char* row = map + ((long)(h * p.W) * CH + oct * 8) * 2;
            for (int w = ws; w < we; ++w) {
              const i32x4_t x = *(const i32x4_t*)(row + (long)w * CH * 2);
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[e] = pk_max_i16(acc[e], x[e]);
            }
          }
          const bool empty = he <= hs || we <= ws;
          bf16_t* dst = trow + (oct * 8) * PP + bin;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t y = (uint32_t)bf16x2_order(acc[e]);
            const float f0 = empty ? 0.f : __builtin_bit_cast(float, y << 16);
            const float f1 = empty ? 0.f : __builtin_bit_cast(float, y & 0xffff0000u);
            dst[(2 * e) * PP] = f32_to_bf16(f0 * mul);
            dst[(2 * e + 1) * PP] = f32_to_bf16(f1 * mul);
          }
        }
      } else {
        for (int u = lane; u < CH * PP; u += 32) {
          const int bin = u / CH, cu = u - bin * CH;
          const int ph = bin / 7, pw = bin - ph * 7;
          const int hs = hb[r][ph][0], he = hb[r][ph][1], ws = wb[r][pw][0], we = wb[r][pw][1];
          float b0 = (he <= hs || we <= ws) ? 0.f : -FLT_MAX;
          for (int h = hs; h < he; ++h)
            for (int w = ws; w < we; ++w) b0 = fmaxf(b0, *(const float*)(map + ((long)(h * p.W + w) * CH + cu) * 4));
          trow[cu * PP + bin] = b0 * mul;
        }
      }
    }
    __syncthreads();
    r0 = r1;
  }
  // A rows: nr contiguous runs of RUN elements
  constexpr int VROW = RUN * ES / 16;
  for (int v = tid; v < nr * VROW; v += 256) {
    const int rr = v / VROW, q = v - rr * VROW;
    *(i32x4_t*)(p.out + ((long)(m0 + rr) * p.ld_out + (long)c0 * PP) * ES + (long)q * 16) =
        *(const i32x4_t*)((const char*)tile + ((long)rr * RUN * ES + (long)q * 16));
  }
  if (p.out_t) {
    T* ot = (T*)p.out_t + (long)c0 * PP * p.ld_out_t + m0;
    if (nr == ROI_GROUP) {  // launcher guarantees 16-B alignment of every 8-ROI run
      for (int idx = tid; idx < RUN; idx += 256) {
        T vals[ROI_GROUP];
#pragma unroll
        for (int rr = 0; rr < ROI_GROUP; ++rr) vals[rr] = tile[(long)rr * RUN + idx];
        i32x4_t* dst = (i32x4_t*)(ot + (long)idx * p.ld_out_t);
#pragma unroll
        for (int q = 0; q < ROI_GROUP * ES / 16; ++q) dst[q] = ((const i32x4_t*)vals)[q];
      }
    } else {
      for (int idx = tid; idx < RUN; idx += 256)
        for (int rr = 0; rr < nr; ++rr) ot[(long)idx * p.ld_out_t + rr] = tile[(long)rr * RUN + idx];
    }
  }
  }  // ROI groups of this block
}

// 7x7 ROIPool, whole-map variant for the training operand pair (A, A^T) in bf16: a block pools 64 consecutive ROIs out
// of an 8-channel slice of the map, so that every row of its A^T tile - 64 ROIs x 2 B - is one FULL 128-byte line
// (the 8-ROI kernel above writes A^T as 16-byte column runs and relies on eight blocks of one XCD meeting in L2 to
// complete a line: 0.39 of the HBM write roofline).  Work items are (ROI, bin) pairs, one 16-byte LDS read per window
// pixel feeds the packed int16 maxima of all 8 channels; 3136 items over 256 threads - no idle lanes, which is what
// made round 1's 64-ROI attempt slower.  The [64][8*49] tile (pitch 792 B: 8-byte aligned rows for the A runs, bank
// spread for the transposed reads) leaves LDS as 64 runs of 784 B of A and 392 full lines of A^T.  Block ids run
// chunk-fastest inside an XCD's contiguous range: the two partial lines at the ends of a 784-byte A run are shared with
// the neighbouring channel chunks, which the same XCD's L2 sees right next in time.
constexpr int ROI_G64 = 64;
constexpr int G64_CH = 8, G64_RUN = G64_CH * 49, G64_PITCH = G64_RUN * 2 + 8, G64_THREADS = 512;
template <int JMAX>  // (ROI, bin) items per thread: ceil(64 * 49 / threads) = 13 / 7 / 4 for 256 / 512 / 1024 threads
__global__ __launch_bounds__(JMAX >= 13 ? 256 : JMAX >= 7 ? 512 : 1024, JMAX >= 13 ? 2 : 4) void roi_pool7_map64_kernel(RoiParams p) {
  constexpr int PP = 49;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = p.H * p.W;
  // the map slice is staged in BANDS of whole rows (p.lds_px pixels at most; one band = the whole map for everything up
  // to ~6700 pixels): larger maps - 75 x 122 for a 1200 x 1951 image - used to fall to the 8-ROI kernel that re-stages
  // the slice per 8 ROIs (1.4 ms per call there).  A (ROI, bin) item keeps its running maximum in registers across the bands.
  const int band_rows = p.lds_px >= HW ? p.H : p.lds_px / p.W;
  const long map_bytes = ((long)(band_rows < p.H ? band_rows * p.W : HW) * 16 + 15) & ~15L;
  // [band pixels][8 channels, order-mapped bf16]; p.pf: two such buffers, chunk cc scans buffer cc & 1
  char* tile = smem + (p.pf ? 2 : 1) * map_bytes;            // [64][G64_PITCH]
  __shared__ unsigned char hb[ROI_G64][7][2], wb[ROI_G64][7][2];
  __shared__ int bidx[ROI_G64];
  __shared__ float mulv[ROI_G64];
  const int nchunks = (p.C - p.c_begin) / G64_CH, nblk = nchunks / p.cpb;   // blocks per ROI group
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int group = logical / nblk, cb = logical - group * nblk;
  const int m0 = group * ROI_G64;
  const int nr = min(ROI_G64, p.M - m0);
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int i = tid; i < ROI_G64 * 7; i += nthr) {
    const int r = i / 7, k = i - r * 7;
    if (r < nr) {
      const float* roi = p.rois + 5 * (long)(m0 + r);
      const int x1 = (int)roundf(roi[1] * p.scale), y1 = (int)roundf(roi[2] * p.scale);
      const int x2 = (int)roundf(roi[3] * p.scale), y2 = (int)roundf(roi[4] * p.scale);
      const int rw = max(x2 - x1 + 1, 1), rh = max(y2 - y1 + 1, 1);
      const float bin_h = (float)rh / 7.f, bin_w = (float)rw / 7.f;
      hb[r][k][0] = (unsigned char)min(max((int)floorf((float)k * bin_h) + y1, 0), p.H);
      hb[r][k][1] = (unsigned char)min(max((int)ceilf((float)(k + 1) * bin_h) + y1, 0), p.H);
      wb[r][k][0] = (unsigned char)min(max((int)floorf((float)k * bin_w) + x1, 0), p.W);
      wb[r][k][1] = (unsigned char)min(max((int)ceilf((float)(k + 1) * bin_w) + x1, 0), p.W);
      if (k == 0) {
        bidx[r] = (int)roi[0];
        mulv[r] = p.obj ? p.obj[m0 + r] + 1.f : 1.f;
      }
    }
  }
  __syncthreads();
  // runs of ROIs on the same image, as a bit mask of run ends (bit r: ROI r is the last of its run).  The walk that
  // used to find each run's end - `while (bidx[r1] == b) ++r1`: 64 dependent LDS reads in every thread, per chunk -
  // was 40-50 us of the 141-us launch (knock-outs, profiles/r2_24_*)
  __shared__ unsigned long long runmask;
  if (tid < 64) {
    const bool last = tid + 1 >= nr || bidx[tid + 1] != bidx[tid];
    const unsigned long long m = __ballot(last && tid < nr);
    if (tid == 0) runmask = m;
  }
  __syncthreads();
  const unsigned long long runs = runmask;
  // Per-thread item table, computed ONCE per block: item j of this thread is (ROI, bin) number tid + j * nthr of the
  // group, whatever the chunk - its window, its tile / A offsets, its scale and its "empty bin" flag do not depend on
  // the channels.  (They used to be re-derived - two divisions, four byte loads, the 64-bit output address - in the scan,
  // again in the epilogue and again in the A store loop of every chunk: the launch is VALU-issue bound, ~1400
  // instructions per thread and chunk at 4 cycles each.)
  int win[JMAX];    // hs | he << 8 | ws << 16 | we << 24 (map coordinates)
  int meta[JMAX];   // r | bin << 8 | empty << 16 | valid << 17
#pragma unroll
  for (int j = 0; j < JMAX; ++j) {
    const int it = tid + j * nthr;
    const bool valid = it < nr * PP;
    const int r = valid ? it / PP : 0, bin = valid ? it - r * PP : 0;
    const int ph = bin / 7, pw = bin - ph * 7;
    const int hs = hb[r][ph][0], he = hb[r][ph][1], ws = wb[r][pw][0], we = wb[r][pw][1];
    win[j] = hs | he << 8 | ws << 16 | we << 24;
    meta[j] = r | bin << 8 | ((he <= hs || we <= ws) ? 1 << 16 : 0) | (valid ? 1 << 17 : 0);
  }
  // p.pf (whole map in one band, <= 2 pixels per thread): the slice of the NEXT chunk (first run's image) is fetched
  // into registers at the top of a chunk and moved into the other map buffer behind the scan: no chunk but the first
  // waits for a global load, and the wait sits in front of this chunk's stores in program order (vmcnt counts loads
  // and stores in order), so the stores drain under the next chunk's scan.
  i32x4_t pfr[2];
  const int b_first = bidx[0];
  auto fetch = [&](int c0_) {
    const char* src = p.feat + ((long)b_first * HW * p.C + c0_) * 2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int px = tid + q * nthr;
      if (px < HW) pfr[q] = *(const i32x4_t*)(src + (long)px * p.C * 2);
    }
  };
  auto stash = [&](char* dst) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int px = tid + q * nthr;
      if (px < HW) {
        i32x4_t x = pfr[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = bf16x2_order(x[e]);
        *(i32x4_t*)(dst + (long)px * 16) = x;
      }
    }
  };
  if (p.pf) {
    fetch(p.c_begin + cb * p.cpb * G64_CH);
    stash(smem);
    __syncthreads();
  }
  typedef int i32x2_t __attribute__((ext_vector_type(2)));
  for (int cc = 0; cc < p.cpb; ++cc) {  // the block's channel chunks: same ROIs, same bin bounds
    const int c0 = p.c_begin + (cb * p.cpb + cc) * G64_CH;
    // the packed table stays packed: without this the compiler hoists every unpacked field (and every product with a
    // pitch) out of the chunk loop - ~60 more live registers, i.e. spills at the 128 that two blocks per CU allow
#pragma unroll
    for (int j = 0; j < JMAX; ++j) asm volatile("" : "+v"(win[j]), "+v"(meta[j]));
    char* map = p.pf ? smem + (cc & 1) * map_bytes : smem;
    if (p.pf && cc + 1 < p.cpb) fetch(c0 + G64_CH);
    for (int r0 = 0; r0 < nr;) {  // one pass per run of ROIs on the same image
      const int b = bidx[r0];
      const int r1 = r0 + __builtin_ctzll(runs >> r0) + 1;
      const char* fb = p.feat + ((long)b * HW * p.C + c0) * 2;
      const int lo = (int)0x80008000u;
      i32x4_t acc[JMAX];
#pragma unroll
      for (int j = 0; j < JMAX; ++j) acc[j] = i32x4_t{lo, lo, lo, lo};
      for (int y0 = 0; y0 < p.H; y0 += band_rows) {
        const int y1 = min(p.H, y0 + band_rows), npx = (y1 - y0) * p.W;
        const char* fbb = fb + (long)y0 * p.W * p.C * 2;
        if (!(p.pf && r0 == 0)) {  // (prefetch mode: the first run's slice is in LDS already, behind a barrier)
          // four pixels per thread and trip, all four loads issued before the first conversion: at real map sizes (50x76:
          // 3800 pixels of 16 bytes, 2 KB apart) the one-pixel loop waited out a full memory latency per pixel -
          // 47 of the launch's 256 us there (knock-outs, profiles/r3_23_roi_large_maps.txt)
          // (JMAX <= 4 = the 1024-thread variant these maps take; in the 512-thread variant of the 14x14 .. 38x38 maps the
          // extra live registers would spill at its 128-register cap, and its slice comes from the prefetch path anyway)
          if (JMAX > 4) {
            for (int px = tid; px < npx; px += nthr) {
              i32x4_t x = *(const i32x4_t*)(fbb + (long)px * p.C * 2);
#pragma unroll
              for (int e = 0; e < 4; ++e) x[e] = bf16x2_order(x[e]);
              *(i32x4_t*)(map + (long)px * 16) = x;
            }
          } else
          for (int px0 = tid; px0 < npx; px0 += 4 * nthr) {
            i32x4_t x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int px = px0 + q * nthr;
              if (px < npx) x[q] = *(const i32x4_t*)(fbb + (long)px * p.C * 2);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int px = px0 + q * nthr;
              if (px < npx) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[q][e] = bf16x2_order(x[q][e]);
                *(i32x4_t*)(map + (long)px * 16) = x[q];
              }
            }
          }
          __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
          const int r = meta[j] & 0xff;
          if ((meta[j] >> 17 & 1) && r >= r0 && r < r1) {
            const int hs = max(win[j] & 0xff, y0), he = min(win[j] >> 8 & 0xff, y1);
            const int ws = win[j] >> 16 & 0xff, we = win[j] >> 24 & 0xff;
            for (int h = hs; h < he; ++h) {
              const char* row = map + (long)((h - y0) * p.W) * 16;
              int w = ws;
              if (JMAX <= 4)
              for (; w + 1 < we; w += 2) {  // two pixels per trip, both reads in flight before the first max
                const i32x4_t x0 = *(const i32x4_t*)(row + w * 16), x1 = *(const i32x4_t*)(row + w * 16 + 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][e] = pk_max_i16(pk_max_i16(acc[j][e], x0[e]), x1[e]);
              }
              for (; w < we; ++w) {
                const i32x4_t x = *(const i32x4_t*)(row + w * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][e] = pk_max_i16(acc[j][e], x[e]);
              }
            }
          }
        }
        if (!(p.pf && r1 == nr)) __syncthreads();  // the band may be replaced (prefetch mode, last run: this buffer rests for two chunks)
      }
#pragma unroll
      for (int j = 0; j < JMAX; ++j) {
        const int r = meta[j] & 0xff, bin = meta[j] >> 8 & 0xff;
        if ((meta[j] >> 17 & 1) && r >= r0 && r < r1) {
          const bool empty = meta[j] >> 16 & 1;
          const float mul = mulv[r];
          bf16_t* dst = (bf16_t*)(tile + r * G64_PITCH) + bin;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t y = (uint32_t)bf16x2_order(acc[j][e]);
            const float f0 = empty ? 0.f : __builtin_bit_cast(float, y << 16);
            const float f1 = empty ? 0.f : __builtin_bit_cast(float, y & 0xffff0000u);
            dst[(2 * e) * PP] = f32_to_bf16(f0 * mul);
            dst[(2 * e + 1) * PP] = f32_to_bf16(f1 * mul);
          }
        }
      }
      r0 = r1;
    }
    if (p.pf && cc + 1 < p.cpb) stash(smem + ((cc + 1) & 1) * map_bytes);  // (that buffer's readers: chunk cc - 1, two barriers ago)
    __syncthreads();  // the tile is complete (and the next chunk's slice visible)
    // A: nr runs of 784 bytes (49 x 16 B), rows of the tile are 8-byte aligned; piece j of this thread = its item j
    char* oa = p.out + ((long)m0 * p.ld_out + (long)c0 * PP) * 2;
    const int ld2 = (int)(p.ld_out * 2);  // (64 rows x ld_out x 2 B fits 31 bits: ld_out < 16 M elements)
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      if (meta[j] >> 17 & 1) {
        const int r = meta[j] & 0xff, bin = meta[j] >> 8 & 0xff;
        const char* src = tile + r * G64_PITCH + bin * 16;
        const i32x2_t a = *(const i32x2_t*)src, b2 = *(const i32x2_t*)(src + 8);
        *(i32x4_t*)(oa + (r * ld2 + bin * 16)) = i32x4_t{a[0], a[1], b2[0], b2[1]};
      }
    }
    if (p.out_t && c0 + G64_CH > p.t_c0) {  // (round 3: the fc6 dW reads A itself; only the peeled tail columns keep an A^T)
      char* ot = p.out_t + ((long)c0 * PP * p.ld_out_t + m0) * 2;
      if (nr == ROI_G64) {
        // (k row, 8-ROI octet): 8 lanes write one full 128-byte line; piece i of this thread is row (tid >> 3) + i * nthr / 8
        const int q = tid & 7;
        const char* src = tile + (8 * q) * G64_PITCH + (tid >> 3) * 2;
        char* dst = ot + (long)(tid >> 3) * p.ld_out_t * 2 + q * 16;
        const long dstep = (long)(nthr >> 3) * p.ld_out_t * 2;
#pragma unroll
        for (int i = 0; i < JMAX; ++i) {
          if (tid + i * nthr < G64_RUN * 8) {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              w[k] = (uint32_t)(*(const bf16_t*)(src + (2 * k) * G64_PITCH)) |
                     ((uint32_t)(*(const bf16_t*)(src + (2 * k + 1) * G64_PITCH)) << 16);
            *(i32x4_t*)dst = i32x4_t{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
          }
          src += (nthr >> 3) * 2;
          dst += dstep;
        }
      } else {
        for (int idx = tid; idx < G64_RUN; idx += nthr)
          for (int rr = 0; rr < nr; ++rr)
            ((bf16_t*)(ot + (long)idx * p.ld_out_t * 2))[rr] = *(const bf16_t*)(tile + (long)rr * G64_PITCH + idx * 2);
      }
    }
    if (cc + 1 < p.cpb) {
      // the tile is free for the next chunk: every wave's LDS reads have returned (their data went into the stores).  A raw
      // barrier - __syncthreads() would also wait (vmcnt) for the A / A^T stores just issued, which are meant to drain
      // under the next chunk's scan
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }  // channel chunks of this block
}

// 7x7 ROIPool, LANE-PER-BIN variant (round 4): the training operand A in bf16.
// What bounded the 64-ROI kernel above (88 us for the 201 MB of A at the bench shape = 0.29 of the HBM roofline, its
// traffic 1.03x algorithmic): ~240 VALU instructions per (ROI, bin, 8 channels) item - a per-pixel window loop whose
// trip counts differ lane by lane and that waits out the LDS latency at every pixel (ISA: ds_read_b128, s_waitcnt
// lgkmcnt(0), four v_pk_max_i16, six loop-control instructions), an epilogue that scatters every item's 8 channels into a
// [ROI][channel][bin] LDS tile as eight 2-byte writes, and a second pass that reads the tile back for the stores.
// Here a WAVE owns one ROI and lane l < 49 owns bin l:
//   * channel c's 49 bins sit in 49 consecutive lanes, and A[r][c * 49 + bin] is exactly that order: every channel
//     leaves as ONE 98-byte run per store instruction (global_store_short / _short_d16_hi on the packed pair) - no LDS
//     tile, no transposition, no second pass, no barrier per chunk;
//   * the bin windows are computed once per ROI and serve all NCK 8-channel chunks of the block's slice: per window
//     pixel one address and NCK independent 16-byte LDS reads (all in flight together) feed NCK x 4 v_pk_max_i16;
//   * window loops run to the wave's LARGEST window with clamped coordinates (a pixel read twice does not change a
//     maximum): uniform trip counts, no divergence, nothing waits per pixel.
// LDS: [NCK][H*W][16 B] (8 channels of a pixel, order-mapped bf16), staged once per run of same-image ROIs of the block.
// Bit-identical to the kernels above (same maxima, same fp32 scaling, same RNE conversion).  15 of 64 lanes idle.
// VD (round 5): dwords of one LDS cell - 4 = 8 channels of a pixel in 16 bytes (every map whose 8-channel slice fits the LDS),
// 2 = 4 channels in 8 bytes: maps of up to ~19 700 cells, i.e. the shipped dilated-C5 recipe's stride-8 feature map of a
// real-size image (99 x 151 at 800 x 1216: an 8-channel slice is 240 KB).  Those maps used to fall to the 64-ROI kernel in
// row bands: 1.95 ms per pooling launch at R = 2000, 46 % of a DC5 inference pass (profiles/r5_25_infer800_r50dc5_kernel_stats.txt).
template <int NCK, int NWV = 8, int VD = 4>
__global__ __launch_bounds__(NWV * 64) void roi_pool7_lane_kernel(RoiParams p) {
  typedef int cellv __attribute__((ext_vector_type(VD)));
  constexpr int CB = VD * 4, CH = VD * 2;  // bytes / channels of a cell
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = p.H * p.W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NWV;
  // window pixels per trip of the scan: with one or two chunks per block (large maps: windows of 4-15 pixels a side) four
  // clamped pixels of a row go out together - four independent LDS reads in flight instead of one per trip
  constexpr int UNR = NCK <= 2 ? 4 : 1;
  const int nslice = p.C / (CH * NCK);
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int group = logical / nslice, sl = logical - group * nslice;
  const int c0 = sl * CH * NCK;
  const int ph = lane / 7, pw = lane - ph * 7;
  const bool is_bin = lane < 49;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned cstride = (unsigned)HW * (unsigned)CB;
  int cur_img = -1;
  for (int rep = 0; rep < p.lane_reps; ++rep) {  // (the staged slice carries over from group to group while the image stays)
  const int m0 = (group * p.lane_reps + rep) * p.lane_g;
  if (m0 >= p.M) break;
  const int nr = min(p.lane_g, p.M - m0);
  // every wave keeps the group's ROIs in its lanes (lane l: ROI m0 + l; <= 64 per group): box corners on the map, image
  // index, scale - one round of loads per group instead of a dependent scalar load chain per ROI; a ROI's values reach all
  // lanes through v_readlane (the ROI index is wave-uniform)
  int vx1 = 0, vy1 = 0, vx2 = 0, vy2 = 0, vimg = -1;
  float vmul = 1.f;
  if (lane < nr) {
    const float* roi = p.rois + 5 * (long)(m0 + lane);
    vimg = (int)roi[0];
    vx1 = (int)roundf(roi[1] * p.scale);
    vy1 = (int)roundf(roi[2] * p.scale);
    vx2 = (int)roundf(roi[3] * p.scale);
    vy2 = (int)roundf(roi[4] * p.scale);
    vmul = p.obj ? p.obj[m0 + lane] + 1.f : 1.f;
  }
  // runs of ROIs on the same image as a bit mask of run ends (one run in all but ragged batches)
  const int nxt = __shfl_down(vimg, 1, 64);
  const unsigned long long runs = __ballot(lane < nr && (lane + 1 >= nr || nxt != vimg));
  for (int r0 = 0; r0 < nr;) {
    const int b = __builtin_amdgcn_readlane(vimg, r0);
    const int r1 = r0 + __builtin_ctzll(runs >> r0) + 1;
    if (b != cur_img) {
      if (cur_img >= 0) __syncthreads();  // every wave is done with the previous image's slice
      // (built and measured: staging from a chunk-major copy of the map, [N][C / 8][H * W][8] - every slice one contiguous run
      // instead of 16 bytes of each 2-KB pixel - moved the launch by 2-5 % at 43x58 .. 63x92: the scan bounds it, not the
      // staging's sector over-fetch; the extra entry points were removed again)
      const char* fb = p.feat + ((long)b * HW * p.C + c0) * 2;
      for (int idx = tid; idx < HW * NCK; idx += NW * 64) {
        const int px = idx / NCK, c = idx - px * NCK;
        cellv x = *(const cellv*)(fb + (long)px * p.C * 2 + c * CB);
#pragma unroll
        for (int e = 0; e < VD; ++e) x[e] = bf16x2_order(x[e]);
        *(cellv*)(smem + ((long)c * HW + px) * CB) = x;
      }
      __syncthreads();
      cur_img = b;
    }
    for (int r = r0 + wave; r < r1; r += NW) {
      const int x1 = __builtin_amdgcn_readlane(vx1, r), y1 = __builtin_amdgcn_readlane(vy1, r);
      const int x2 = __builtin_amdgcn_readlane(vx2, r), y2 = __builtin_amdgcn_readlane(vy2, r);
      const float mul = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vmul), r));
      const int rw = max(x2 - x1 + 1, 1), rh = max(y2 - y1 + 1, 1);
      const float bin_h = (float)rh / 7.f, bin_w = (float)rw / 7.f;
      const int hs = min(max((int)floorf((float)ph * bin_h) + y1, 0), p.H);
      const int he = min(max((int)ceilf((float)(ph + 1) * bin_h) + y1, 0), p.H);
      const int ws = min(max((int)floorf((float)pw * bin_w) + x1, 0), p.W);
      const int we = min(max((int)ceilf((float)(pw + 1) * bin_w) + x1, 0), p.W);
      const bool empty = he <= hs || we <= ws;
      const int nh = (is_bin && !empty) ? he - hs : 0, nw = (is_bin && !empty) ? we - ws : 0;
      int max_nh = 0, max_nw = 0;  // the wave's largest window (uniform)
      while (__ballot(max_nh < nh) != 0) ++max_nh;
      while (__ballot(max_nw < nw) != 0) ++max_nw;
      const int lo = (int)0x80008000u;
      cellv acc[NCK];
#pragma unroll
      for (int c = 0; c < NCK; ++c)
#pragma unroll
        for (int e = 0; e < VD; ++e) acc[c][e] = lo;
      // (built and measured: the same loops with every read PREDICATED on the lane's own window instead of clamped - a third
      // of the LDS bytes - are slower at every map size, 61 -> 67 us at 14x14 and 350 -> 404 us at 63x92: the exec-mask
      // bookkeeping and the re-initialised operands cost more issue slots than the reads cost LDS cycles)
      // UNRH window rows per trip (4-channel cells - the largest maps, windows of 5-20 pixels a side: two rows x four clamped
      // pixels = eight independent LDS reads in flight instead of four; the scan of those maps is bound by the reads' latency,
      // not by LDS bandwidth: 1432 us against ~270 us of LDS cycles at 99x151, profiles/r5_26_*)
      constexpr int UNRH = VD == 2 ? 2 : 1;
      const int wlast = min(max(we - 1, 0), p.W - 1), wfirst = min(max(ws, 0), p.W - 1);
      for (int hi = 0; hi < max_nh; hi += UNRH) {
        unsigned arow[UNRH];
#pragma unroll
        for (int v = 0; v < UNRH; ++v) {
          const int hr = min(max(min(hs + hi + v, he - 1), 0), p.H - 1);
          arow[v] = lds0 + (unsigned)(hr * p.W) * (unsigned)CB;
        }
        for (int wi = 0; wi < max_nw; wi += UNR) {
          cellv x[UNRH][UNR][NCK];
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int wc = min(wfirst + wi + u, wlast);  // clamped: a pixel read twice does not change a maximum
#pragma unroll
            for (int v = 0; v < UNRH; ++v) {
              const unsigned a = arow[v] + (unsigned)wc * (unsigned)CB;
#pragma unroll
              for (int c = 0; c < NCK; ++c)
                x[v][u][c] = *(__attribute__((address_space(3))) const cellv*)(uintptr_t)(a + (unsigned)c * cstride);
            }
          }
#pragma unroll
          for (int v = 0; v < UNRH; ++v)
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
              for (int c = 0; c < NCK; ++c)
#pragma unroll
                for (int e = 0; e < VD; ++e) acc[c][e] = pk_max_i16(acc[c][e], x[v][u][c][e]);
        }
      }
      if (is_bin) {
        bf16_t* dst = (bf16_t*)p.out + (long)(m0 + r) * p.ld_out + (long)c0 * 49 + lane;

#pragma unroll
        for (int c = 0; c < NCK; ++c)
#pragma unroll
          for (int e = 0; e < VD; ++e) {
            // (an empty bin is +0 in both halves; one packed conversion - v_cvt_pk_bf16_f32, RNE like f32_to_bf16 - and the
            // two halves of its result leave through global_store_short / global_store_short_d16_hi)
            const uint32_t y = empty ? 0u : (uint32_t)bf16x2_order(acc[c][e]);
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t f = f32x2_t{__builtin_bit_cast(float, y << 16), __builtin_bit_cast(float, y & 0xffff0000u)} * mul;
            const uint32_t o = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
            dst[(c * CH + 2 * e) * 49] = (bf16_t)(o & 0xffffu);
            dst[(c * CH + 2 * e + 1) * 49] = (bf16_t)(o >> 16);
          }
      }
    }
    r0 = r1;
  }
  }  // groups of this block
}

// 7x7 ROIAlign, LANE-PER-BIN (round 6): bf16 in / bf16 out, the forward of detectron2/layers/roi_align.py:22-59 ->
// ROIAlign_forward (detectron2/layers/csrc/ROIAlign/ROIAlign_cuda.cu:65-139; pre_calc + accumulate of ROIAlign_cpu.cpp), fused with
// the objectness scaling like the RoIPool kernels.  The generic kernel (roi_kernel MODE 1: block = ROI x 64 channels, lane =
// channel) computes every sample's four bilinear weights - ~40 VALU instructions of float clamping - once per LANE, i.e. 64
// times per 64 channels, and fetches its four taps from global memory per sample: 455-500 us for 2000 ROIs on the 14x14x1024
// map, 0.05 of the HBM roof (profiles/r6_11_roi_align.txt).  Here, as in roi_pool7_lane_kernel: a block stages NCK 8-channel
// slices of the whole map in LDS once per group of ROIs, a WAVE owns one ROI and lane l < 49 owns bin l; the sampling grid
// (gh x gw, adaptive or fixed) is uniform over the ROI, a sample's weights and its four cell addresses are computed ONCE per
// bin and serve all NCK x 8 channels of the block's slices (4 x NCK 16-byte LDS reads, 7 fp32 operations per channel), and a
// channel's 49 bins leave as one 98-byte run per store instruction.
// Same operations on the same values in the same order as roi_kernel<.., 1> (w = hy * hx ..; ((w1 v1 + w2 v2) + w3 v3) + w4 v4;
// samples outside [-1, H] x [-1, W] skipped; acc / count * (objectness + 1); one RNE conversion): bit-identical outputs.
template <int NCK>
__global__ __launch_bounds__(512) void roi_align7_lane_kernel(RoiParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = p.H * p.W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = 8;
  const int nslice = p.C / (8 * NCK);
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int group = logical / nslice, sl = logical - group * nslice;
  const int c0 = sl * 8 * NCK;
  const int ph = lane / 7, pw = lane - ph * 7;
  const bool is_bin = lane < 49;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned cstride = (unsigned)HW * 16u;
  const int m0 = group * p.lane_g;
  if (m0 >= p.M) return;
  const int nr = min(p.lane_g, p.M - m0);
  // the group's ROIs in the lanes of every wave (lane l: ROI m0 + l)
  float fx1 = 0.f, fy1 = 0.f, fx2 = 0.f, fy2 = 0.f, vmul = 1.f;
  int vimg = -1;
  if (lane < nr) {
    const float* roi = p.rois + 5 * (long)(m0 + lane);
    vimg = (int)roi[0];
    fx1 = roi[1]; fy1 = roi[2]; fx2 = roi[3]; fy2 = roi[4];
    vmul = p.obj ? p.obj[m0 + lane] + 1.f : 1.f;
  }
  const int nxt = __shfl_down(vimg, 1, 64);
  const unsigned long long runs = __ballot(lane < nr && (lane + 1 >= nr || nxt != vimg));
  auto bcast = [&](float v, int r) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r)); };
  int cur_img = -1;
  for (int r0 = 0; r0 < nr;) {
    const int b = __builtin_amdgcn_readlane(vimg, r0);
    const int r1 = r0 + __builtin_ctzll(runs >> r0) + 1;
    if (b != cur_img) {
      if (cur_img >= 0) __syncthreads();
      const char* fb = p.feat + ((long)b * HW * p.C + c0) * 2;
      for (int idx = tid; idx < HW * NCK; idx += NW * 64) {
        const int px = idx / NCK, c = idx - px * NCK;
        *(i32x4_t*)(smem + ((long)c * HW + px) * 16) = *(const i32x4_t*)(fb + (long)px * p.C * 2 + c * 16);
      }
      __syncthreads();
      cur_img = b;
    }
    for (int r = r0 + wave; r < r1; r += NW) {
      const float off = p.aligned ? 0.5f : 0.f;
      const float sw = bcast(fx1, r) * p.scale - off, sh = bcast(fy1, r) * p.scale - off;
      const float ew = bcast(fx2, r) * p.scale - off, eh = bcast(fy2, r) * p.scale - off;
      const float mul = bcast(vmul, r);
      float rw = ew - sw, rh = eh - sh;
      if (!p.aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
      const float bin_h = rh / 7.f, bin_w = rw / 7.f;
      const int gh = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(rh / 7);
      const int gw = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(rw / 7);
      const float count = (float)max(gh * gw, 1);
      float acc[NCK][8];
#pragma unroll
      for (int c = 0; c < NCK; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        const float yy = sh + ph * bin_h + (float)(iy + .5f) * bin_h / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float xx = sw + pw * bin_w + (float)(ix + .5f) * bin_w / (float)gw;
          float x = xx, y = yy;
          if (!is_bin || y < -1.0f || y > p.H || x < -1.0f || x > p.W) continue;
          if (y <= 0) y = 0;
          if (x <= 0) x = 0;
          int yl = (int)y, xl = (int)x, yh, xh;
          if (yl >= p.H - 1) { yh = yl = p.H - 1; y = (float)yl; } else yh = yl + 1;
          if (xl >= p.W - 1) { xh = xl = p.W - 1; x = (float)xl; } else xh = xl + 1;
          const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
          const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
          const unsigned a1 = lds0 + (unsigned)(yl * p.W + xl) * 16u, a2 = lds0 + (unsigned)(yl * p.W + xh) * 16u;
          const unsigned a3 = lds0 + (unsigned)(yh * p.W + xl) * 16u, a4 = lds0 + (unsigned)(yh * p.W + xh) * 16u;
          typedef __attribute__((address_space(3))) const i32x4_t* lds_v4;
#pragma unroll
          for (int c = 0; c < NCK; ++c) {
            const i32x4_t q1 = *(lds_v4)(uintptr_t)(a1 + (unsigned)c * cstride), q2 = *(lds_v4)(uintptr_t)(a2 + (unsigned)c * cstride);
            const i32x4_t q3 = *(lds_v4)(uintptr_t)(a3 + (unsigned)c * cstride), q4 = *(lds_v4)(uintptr_t)(a4 + (unsigned)c * cstride);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              auto f = [&](const i32x4_t& q) {
                const uint32_t wd = (uint32_t)q[e >> 1];
                return __builtin_bit_cast(float, (e & 1) ? (wd & 0xffff0000u) : (wd << 16));
              };
              acc[c][e] += w1 * f(q1) + w2 * f(q2) + w3 * f(q3) + w4 * f(q4);
            }
          }
        }
      }
      if (is_bin) {
        bf16_t* dst = (bf16_t*)p.out + (long)(m0 + r) * p.ld_out + (long)c0 * 49 + lane;
#pragma unroll
        for (int c = 0; c < NCK; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) dst[(c * 8 + e) * 49] = f32_to_bf16(acc[c][e] / count * mul);
      }
    }
    r0 = r1;
  }
}

// Chunk-major copy of a bf16 NHWC map for the walking kernel below: [N][H*W][C] -> [N][C/8][H*W] cells of 16 bytes (8 channels of a
// pixel), values already order-mapped (bf16x2_order).  A staged slice is then ONE contiguous run instead of 16 bytes of every
// pixel's 2-KB line (64 lines per wave instruction: ~28 us of a 156-us pooling launch at 50x76 - profiles/r5_32_roi_walk_knockouts.txt).
// 32 pixels x 32 chunks per block through LDS: 512-byte runs on both sides.
__global__ __launch_bounds__(256) void roi_chunk_major_kernel(const char* __restrict__ feat, char* __restrict__ cm, int HW, int C) {
  __shared__ i32x4_t tile[32][33];
  const int nchunks = C >> 3;
  const int px0 = blockIdx.x * 32, ch0 = blockIdx.y * 32, img = blockIdx.z;
  const char* src = feat + (long)img * HW * C * 2;
  char* dst = cm + (long)img * nchunks * HW * 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + 256 * i, pl = idx >> 5, cl = idx & 31;
    i32x4_t x = {0, 0, 0, 0};
    if (px0 + pl < HW && ch0 + cl < nchunks) x = *(const i32x4_t*)(src + ((long)(px0 + pl) * C + (ch0 + cl) * 8) * 2);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = bf16x2_order(x[e]);
    tile[pl][cl] = x;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + 256 * i, cl = idx >> 5, pl = idx & 31;
    if (px0 + pl < HW && ch0 + cl < nchunks) *(i32x4_t*)(dst + ((long)(ch0 + cl) * HW + px0 + pl) * 16) = tile[pl][cl];
  }
}

// 7x7 ROIPool, lane-per-bin, WALKING variant (round 5) for maps whose 8-channel slice leaves at most two blocks per CU
// (43x58 and larger: test-time scales, real-size training images).  Knock-outs of the kernel above at 50x76 / R = 2000
// (profiles/r5_32_roi_walk_knockouts.txt): 71 us of its 215 are the staging (one 16-byte piece of every pixel's 2-KB line per
// load - and one load, one wait, one LDS write per trip), ~90 us the scan, ~27 us the stores; the scan is instruction-issue
// bound (conflict-free or broadcast LDS addresses: -10 %; six more VALU instructions per read: +18 %), and with ROIs sorted by
// size the launch takes 1.6x as long - whatever unit waits for its largest ROIs sets the time.  Here
//   * a block keeps its group of ROIs and walks `p.walk` CONSECUTIVE channel chunks, re-staging the slice between them: the
//     walked chunks are the 16-byte pieces of ONE 128-byte line per pixel (walk = 8), so the first chunk's staging brings the
//     lines into the XCD's L2 and the other seven hit there; eight loads are in flight per thread;
//   * the bin bounds are computed once per block into an LDS table: per ROI 7 row entries (first row's LDS offset, rows - 1)
//     and 7 column entries (first column's offset, columns - 1) - a lane fetches the two entries of its bin, 72 bytes per ROI
//     instead of ~100 VALU instructions per (ROI, chunk);
//   * the waves of a block take ROIs from a shared counter (one LDS atomic per ROI and chunk, fetched one ROI ahead) instead
//     of a fixed share: the barrier at the end of a chunk waits for one ROI, not for the wave with the largest eight;
//   * the slice's rows have an ODD pitch in cells: W is even for every map here, so rows alone moved a lane by even cell
//     counts (4-byte banks: 16-byte cells map to 16 bank groups).
// Same maxima over the same pixels, same scaling / conversion as the kernels above: bit-identical outputs.
// NSG: sub-groups of 64 ROIs per block (image indices of a sub-group sit in the lanes of every wave).
constexpr int WALK_TAB = 18;  // table dwords per ROI: [0..7] rows by ph, [8..15] columns by pw, [16] largest window, [17] scale
template <int NWV, int NSG, int VD, int SB, int OCC = 1>  // SB: slice cells per thread (>= ceil(H * W / threads)); OCC: blocks per CU the registers must allow
__global__ __launch_bounds__(NWV * 64, OCC) void roi_pool7_walk_kernel(RoiParams p) {
  typedef int cellv __attribute__((ext_vector_type(VD)));
  constexpr int CB = VD * 4, CH = VD * 2, NT = NWV * 64, G = 64 * NSG;
  // (two window rows per trip - eight reads in flight - measured slower for 8-channel cells: 187 vs 174 us at 50x76, the rows are
  // rounded up to pairs)
  constexpr int UNR = 4, UNRH = VD == 2 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = p.H * p.W;
  const int tid = threadIdx.x, lane = tid & 63;
  const int Wp = p.walk_wp;  // LDS row pitch in cells
  const int ncg = p.C / (CH * p.walk);  // chunk groups
  const int ngroups = gridDim.x / ncg;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  // chunk-group major: the blocks of one XCD share few slices
  const int cg = logical / ngroups, group = logical - cg * ngroups;
  const int ph = lane / 7, pw = lane - ph * 7;  // (lanes 49..63: ph = 7 / 8 / 9 -> the table's entry 7, an empty row)
  const bool is_bin = lane < 49;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned WCB = (unsigned)Wp * CB;
  unsigned* tab = (unsigned*)(smem + (size_t)p.H * Wp * CB);
  unsigned* ctr = tab + G * WALK_TAB;
  const int m0 = group * G;
  // ---- once per block: the window table ---------------------------------------------------------------------------------
  for (int t = tid; t < G * 8; t += NT) {
    const int rl = t >> 3, e = t & 7;
    const int m = m0 + rl;
    unsigned rowe = 0x80000000u, cole = 0x80000000u;
    int nh = 0, nw = 0;
    float mul = 1.f;
    if (m < p.M && e < 7) {
      const float* roi = p.rois + 5 * (long)m;
      const int x1 = (int)roundf(roi[1] * p.scale), y1 = (int)roundf(roi[2] * p.scale);
      const int x2 = (int)roundf(roi[3] * p.scale), y2 = (int)roundf(roi[4] * p.scale);
      mul = p.obj ? p.obj[m] + 1.f : 1.f;
      const int rw = max(x2 - x1 + 1, 1), rh = max(y2 - y1 + 1, 1);
      const float bin_h = (float)rh / 7.f, bin_w = (float)rw / 7.f;
      const int hs = min(max((int)floorf((float)e * bin_h) + y1, 0), p.H);
      const int he = min(max((int)ceilf((float)(e + 1) * bin_h) + y1, 0), p.H);
      const int ws = min(max((int)floorf((float)e * bin_w) + x1, 0), p.W);
      const int we = min(max((int)ceilf((float)(e + 1) * bin_w) + x1, 0), p.W);
      nh = max(he - hs, 0);
      nw = max(we - ws, 0);
      // (an empty row / column of bins reads some valid pixel and drops it)
      rowe = (unsigned)min(hs, p.H - 1) * WCB | (unsigned)max(nh - 1, 0) << 20 | (nh == 0 ? 0x80000000u : 0u);
      cole = (unsigned)min(ws, p.W - 1) * CB | (unsigned)max(nw - 1, 0) << 20 | (nw == 0 ? 0x80000000u : 0u);
    }
    // the ROI's largest window: over its 8 entries = 8 consecutive lanes
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
      nh = max(nh, __shfl_xor(nh, d, 64));
      nw = max(nw, __shfl_xor(nw, d, 64));
    }
    tab[rl * WALK_TAB + e] = rowe;
    tab[rl * WALK_TAB + 8 + e] = cole;
    if (e == 0) {
      tab[rl * WALK_TAB + 16] = (unsigned)nh | (unsigned)nw << 16;
      tab[rl * WALK_TAB + 17] = __builtin_bit_cast(unsigned, mul);
    }
  }
  // image runs of every sub-group, as a bit mask of run ends (one run in all but ragged batches)
  int vimg[NSG], nrs[NSG];
  unsigned long long runs[NSG];
#pragma unroll
  for (int sg = 0; sg < NSG; ++sg) {
    const int ms = m0 + sg * 64;
    const int nr = max(min(64, p.M - ms), 0);
    nrs[sg] = nr;
    vimg[sg] = lane < nr ? (int)p.rois[5 * (long)(ms + lane)] : -1;
    const int nxt = __shfl_down(vimg[sg], 1, 64);
    runs[sg] = __ballot(lane < nr && (lane + 1 >= nr || nxt != vimg[sg]));
  }
  // ---- the walk -----------------------------------------------------------------------------------------------------------
  // one image in the whole group (every batch but ragged ones): the NEXT chunk's slice is fetched into registers under this
  // chunk's scan - two blocks of a CU start together and run the same phases, so without it both stage (the memory pipe busy,
  // VALU idle) and both scan (the reverse) at the same time
  bool single = true;
#pragma unroll
  for (int sg = 0; sg < NSG; ++sg)
    single = single && __ballot(lane < nrs[sg] && vimg[sg] != __builtin_amdgcn_readlane(vimg[0], 0)) == 0;
  cellv pf[SB];
  // source of a slice: the chunk-major, order-mapped copy when the caller gave a workspace (one contiguous run), else the NHWC
  // map itself (16 bytes of every pixel's line)
  const bool from_cm = p.cm != nullptr;
  const long src_pitch = from_cm ? CB : (long)p.C * 2;
  auto load_slice = [&](int b, int chunk) {
    const char* fb = from_cm ? p.cm + ((long)b * (p.C / CH) + chunk) * HW * CB : p.feat + ((long)b * HW * p.C + (long)chunk * CH) * 2;
    // (no branch around a load or a write, not even a uniform one: the wait-count pass then puts s_waitcnt vmcnt(0) in front of
    // every load; a thread past the end re-reads / re-writes the last cell)
#pragma unroll
    for (int k = 0; k < SB; ++k) pf[k] = *(const cellv*)(fb + (long)min(tid + k * NT, HW - 1) * src_pitch);
  };
  auto write_slice = [&]() {
#pragma unroll
    for (int k = 0; k < SB; ++k) {
      cellv x = pf[k];
#pragma unroll
      for (int e = 0; e < VD; ++e) x[e] = from_cm ? x[e] : bf16x2_order(x[e]);
      const unsigned px = (unsigned)min(tid + k * NT, HW - 1), py = __umulhi(px, p.walk_wmagic);  // px / W
      *(cellv*)(smem + (size_t)(py * Wp + (px - py * p.W)) * CB) = x;
    }
  };
  if (single && nrs[0] > 0) load_slice(__builtin_amdgcn_readlane(vimg[0], 0), cg * p.walk);
  bool staged = false;
  for (int cc = 0; cc < p.walk; ++cc) {
    const int chunk = cg * p.walk + cc;
    int cur_img = -1;
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) {
      const int nr = nrs[sg];
      for (int r0 = 0; r0 < nr;) {
        const int b = __builtin_amdgcn_readlane(vimg[sg], r0);
        const int r1 = r0 + __builtin_ctzll(runs[sg] >> r0) + 1;
        // every wave is done with the previous run (slice and counter): its LDS reads have returned (their data went into the
        // stores).  Raw barriers - __syncthreads() would also drain the A stores just issued (vmcnt)
        if (staged) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        if (tid == 0) *ctr = (unsigned)r0;
        bool fresh = false;
        if (b != cur_img) {
          if (!single) load_slice(b, chunk);
          write_slice();
          cur_img = b;
          fresh = true;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        staged = true;
        if (single && fresh && cc + 1 < p.walk) load_slice(b, chunk + 1);
        // ROIs r0 .. r1-1 of this sub-group, taken from the counter one ahead of the one being scanned
        unsigned nxt = 0;
        if (lane == 0) nxt = atomicAdd(ctr, 1u);
        for (;;) {
          const int r = __builtin_amdgcn_readfirstlane(nxt);
          if (r >= r1) break;
          if (lane == 0) nxt = atomicAdd(ctr, 1u);
          const unsigned* te = tab + (sg * 64 + r) * WALK_TAB;
          const unsigned rowe = te[ph > 7 ? 7 : ph], cole = te[8 + pw];
          const unsigned uni = __builtin_amdgcn_readfirstlane(te[16]);
          const float mul = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(te[17]));
          const int max_nh = uni & 0xffff, max_nw = uni >> 16;
          const unsigned nhm1 = rowe >> 20 & 0x7ff, nwm1 = cole >> 20 & 0x7ff;
          const unsigned org = lds0 + (rowe & 0xfffff) + (cole & 0xfffff);
          const bool empty = (int)(rowe | cole) < 0;
          const int lo = (int)0x80008000u;
          cellv acc;
#pragma unroll
          for (int e = 0; e < VD; ++e) acc[e] = lo;
          for (int hi = 0; hi < max_nh; hi += UNRH) {
            unsigned arow[UNRH];
#pragma unroll
            for (int v = 0; v < UNRH; ++v) arow[v] = org + min((unsigned)(hi + v), nhm1) * WCB;
            for (int wi = 0; wi < max_nw; wi += UNR) {
              cellv x[UNRH][UNR];
#pragma unroll
              for (int u = 0; u < UNR; ++u) {
                const unsigned co = min((unsigned)(wi + u), nwm1) * CB;
#pragma unroll
                for (int v = 0; v < UNRH; ++v)
                  x[v][u] = *(__attribute__((address_space(3))) const cellv*)(uintptr_t)(arow[v] + co);
              }
#pragma unroll
              for (int v = 0; v < UNRH; ++v)
#pragma unroll
                for (int u = 0; u < UNR; ++u)
#pragma unroll
                  for (int e = 0; e < VD; ++e) acc[e] = pk_max_i16(acc[e], x[v][u][e]);
            }
          }
          // (built and measured: the item's 8 x 49 values - one 784-byte run of A - through a per-wave LDS scratch as ONE 16-byte store
          // per lane instead of eight 2-byte stores: 176.0 vs 173.8 us at 50x76, 137 vs 125 at 43x58 (the scratch costs the third
          // block per CU) - eight ds_write_b16 + a ds_read_b128 take the issue slots the eight stores took)
          if (is_bin) {
            bf16_t* dst = (bf16_t*)p.out + (long)(m0 + sg * 64 + r) * p.ld_out + (long)chunk * CH * 49 + lane;
#pragma unroll
            for (int e = 0; e < VD; ++e) {
              // (an empty bin is +0 in both halves; one packed conversion - v_cvt_pk_bf16_f32, RNE like f32_to_bf16)
              const uint32_t y = empty ? 0u : (uint32_t)bf16x2_order(acc[e]);
              typedef float f32x2_t __attribute__((ext_vector_type(2)));
              typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
              const f32x2_t f = f32x2_t{__builtin_bit_cast(float, y << 16), __builtin_bit_cast(float, y & 0xffff0000u)} * mul;
              const uint32_t o = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
              dst[(2 * e) * 49] = (bf16_t)(o & 0xffffu);
              dst[(2 * e + 1) * 49] = (bf16_t)(o >> 16);
            }
          }
        }
        r0 = r1;
      }
    }
  }
}

// 7x7 ROIPool from a SPARSE TABLE of block maxima (round 6) - the maps whose slice leaves room for 4 channels per cell only (the
// shipped dilated-C5 recipe's stride-8 map of a real-size image: 99 x 151 x 2048, ~3750 cells per ROI).  The window kernels above
// read every cell of every ROI: 15.4 G bf16 elements through LDS and a packed maximum per two of them, 920 us for 2000 ROIs
// (roi_pool7_lane_kernel<1, 16, 2>, profiles/r6_11_roi_align.txt) - a compute bound that no schedule removes.  A maximum is
// idempotent, so the windows can share work instead: with T[k][l][y][x] = max over rows y .. y + 2^k - 1 and columns x .. x + 2^l - 1,
// a bin [hs, he) x [ws, we) whose sides are within [2^k, 2^(k+1)] x [2^l, 2^(l+1)] is the maximum of FOUR table cells
// (rows hs and he - 2^k, columns ws and we - 2^l: the blocks overlap, which a maximum does not mind) instead of ~77.  A ROI's 49
// bins differ by at most one row / column, so ONE level pair (k, l) = floor(log2) of its smallest non-empty bin serves all of
// them (2^k <= nh <= 2^k + 1 <= 2^(k+1)); ROIs clipped by the map edge or beyond level 4 (bins of 32+ cells) take
// ceil(nh / 2^k) x ceil(nw / 2^l) cells in a loop.  Work per block = (4-channel slice, image, l):
//   * the ROIs of its class are listed (by k) in LDS from the one-byte class codes roi_st_prep_kernel left; no ROI: exit;
//   * the slice - one contiguous run of the chunk-major, order-mapped copy of the map - lands in LDS and l doubling steps
//     along the rows make T[0][l] in place (a thread keeps its cells in registers: per step one LDS read of the partner
//     cell, one barrier, one write, one barrier);
//   * for k = 0 .. 4: doubling steps down the columns up to level k, then the waves pool the listed ROIs of level k: wave = ROI,
//     lane = bin as in roi_pool7_lane_kernel (every channel leaves as one 98-byte run), the bin's four cell coordinates come
//     packed in ONE dword per lane from the record the prep kernel wrote (fetched eight ROIs at a time).
// 30 doubling steps per slice serve ALL ROIs (~2.5 LDS passes over the slice each) against ~77 reads per (ROI, bin) before.
// Same maxima over the same cells, same scaling and conversion: bit-identical to the other RoIPool kernels.
constexpr int ST_LEVELS = 5;  // levels 0 .. 4: blocks of 1 .. 16 rows / columns
constexpr int ST_BATCH = 8;   // ROI records a wave fetches per round

// One wave per ROI: the record [64 dwords] - lanes 0 .. 48: y0 | y1 << 8 | x0 << 16 | x1 << 24 (first / last block row, first / last
// block column of the bin at the ROI's level; an empty bin: y0 = 1 > y1 = 0), lane 60: most blocks per bin (rows | columns << 8),
// lane 62: the objectness scale, lane 63: k | l << 4 - and the class byte (image * 5 + l) * 5 + k (255: image index out of range).
__global__ __launch_bounds__(256) void roi_st_prep_kernel(RoiParams p, unsigned* __restrict__ rec, unsigned char* __restrict__ cls) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.M) return;
  const float* roi = p.rois + 5 * (long)m;
  const int b = (int)roi[0];
  const int x1 = (int)roundf(roi[1] * p.scale), y1 = (int)roundf(roi[2] * p.scale);
  const int x2 = (int)roundf(roi[3] * p.scale), y2 = (int)roundf(roi[4] * p.scale);
  const float mul = p.obj ? p.obj[m] + 1.f : 1.f;
  const int rw = max(x2 - x1 + 1, 1), rh = max(y2 - y1 + 1, 1);
  const float bin_h = (float)rh / 7.f, bin_w = (float)rw / 7.f;
  const int ph = lane / 7, pw = lane - ph * 7;
  const bool is_bin = lane < 49;
  const int hs = min(max((int)floorf((float)ph * bin_h) + y1, 0), p.H);
  const int he = min(max((int)ceilf((float)(ph + 1) * bin_h) + y1, 0), p.H);
  const int ws = min(max((int)floorf((float)pw * bin_w) + x1, 0), p.W);
  const int we = min(max((int)ceilf((float)(pw + 1) * bin_w) + x1, 0), p.W);
  const int nh = he - hs, nw = we - ws;
  const bool empty = !is_bin || nh <= 0 || nw <= 0;
  int hmin = is_bin && nh > 0 ? nh : 0x7fff, wmin = is_bin && nw > 0 ? nw : 0x7fff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    hmin = min(hmin, __shfl_xor(hmin, o, 64));
    wmin = min(wmin, __shfl_xor(wmin, o, 64));
  }
  if (hmin == 0x7fff) hmin = 1;
  if (wmin == 0x7fff) wmin = 1;
  const int k = min(31 - __builtin_clz(hmin), ST_LEVELS - 1), l = min(31 - __builtin_clz(wmin), ST_LEVELS - 1);
  int nr = empty ? 0 : (nh + (1 << k) - 1) >> k, nc = empty ? 0 : (nw + (1 << l) - 1) >> l;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    nr = max(nr, __shfl_xor(nr, o, 64));
    nc = max(nc, __shfl_xor(nc, o, 64));
  }
  unsigned r = 1u;  // empty: y0 = 1, y1 = 0, x0 = x1 = 0
  if (!empty) r = (unsigned)hs | (unsigned)(he - (1 << k)) << 8 | (unsigned)ws << 16 | (unsigned)(we - (1 << l)) << 24;
  if (lane == 60) r = (unsigned)nr | (unsigned)nc << 8;
  if (lane == 61) r = (unsigned)b;
  if (lane == 62) r = __builtin_bit_cast(unsigned, mul);
  if (lane == 63) r = (unsigned)k | (unsigned)l << 4;
  rec[(long)m * 64 + lane] = r;
  if (lane == 0) cls[m] = (b >= 0 && b < p.N) ? (unsigned char)((b * ST_LEVELS + l) * ST_LEVELS + k) : (unsigned char)255;
}

// chunk-major copy with cells of VD dwords (roi_chunk_major_kernel is the VD = 4 form; kept separate: its 16-byte tile is the
// walking kernel's measured path)
template <int VD>
__global__ __launch_bounds__(256) void roi_chunk_major_vd_kernel(const char* __restrict__ feat, char* __restrict__ cm, int HW, int C) {
  typedef int cellv __attribute__((ext_vector_type(VD)));
  constexpr int CB = VD * 4, CH = VD * 2;
  __shared__ cellv tile[32][33];
  const int nchunks = C / CH;
  const int px0 = blockIdx.x * 32, ch0 = blockIdx.y * 32, img = blockIdx.z;
  const char* src = feat + (long)img * HW * C * 2;
  char* dst = cm + (long)img * nchunks * HW * CB;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + 256 * i, pl = idx >> 5, cl = idx & 31;
    cellv x;
#pragma unroll
    for (int e = 0; e < VD; ++e) x[e] = 0;
    if (px0 + pl < HW && ch0 + cl < nchunks) x = *(const cellv*)(src + ((long)(px0 + pl) * C + (ch0 + cl) * CH) * 2);
#pragma unroll
    for (int e = 0; e < VD; ++e) x[e] = bf16x2_order(x[e]);
    tile[pl][cl] = x;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + 256 * i, cl = idx >> 5, pl = idx & 31;
    if (px0 + pl < HW && ch0 + cl < nchunks) *(cellv*)(dst + ((long)(ch0 + cl) * HW + px0 + pl) * CB) = tile[pl][cl];
  }
}

__device__ unsigned long long g_st_prof[8];  // PROF builds (tools/roi_st_probe.py): shader-clock cycles of block phases as thread 0 sees them
template <int VD, int SB, bool PROF = false>  // SB: slice cells per thread (>= ceil(H * W / 1024))
__global__ __launch_bounds__(1024) void roi_pool7_st_kernel(RoiParams p, const unsigned* __restrict__ rec, const unsigned char* __restrict__ cls) {
  typedef int cellv __attribute__((ext_vector_type(VD)));
  typedef __attribute__((address_space(3))) const cellv* lds_cell_t;
  constexpr int CB = VD * 4, CH = VD * 2, NT = 1024, NW = 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = p.H * p.W, W = p.W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned short* list = (unsigned short*)(smem + (size_t)HW * CB);
  int* cnt = (int*)(list + ((p.M + 7) & ~7));  // [0..4] ROIs per level, [8..12] fill cursors
  constexpr int SCR = (CH * 98 + 15) & ~15;    // a wave's output run of one ROI: CH channels x 49 bins
  char* scr = (char*)(cnt + 16) + wave * SCR;
  const int last = (HW - 1) * CB;              // byte offset of the slice's last cell
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int nbl = p.N * ST_LEVELS;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);  // the blocks of one slice (its images and levels) share an XCD's L2
  const int sl = logical / nbl, bl = logical - sl * nbl;
  const int b = bl / ST_LEVELS, l = bl - b * ST_LEVELS;
  const int c0 = sl * CH;
  unsigned long long tp[5] = {0, 0, 0, 0, 0}, t0 = 0, t1;
  if constexpr (PROF) t0 = __builtin_amdgcn_s_memtime();
#define ST_CLK(K) do { if constexpr (PROF) { t1 = __builtin_amdgcn_s_memtime(); tp[K] += t1 - t0; t0 = t1; } } while (0)
  // ---- the ROIs of this (image, l), by k ------------------------------------------------------------------------------------
  if (tid < 16) cnt[tid] = 0;
  __syncthreads();
  for (int m = tid; m < p.M; m += NT) {
    const unsigned d = (unsigned)cls[m] - (unsigned)(bl * ST_LEVELS);
    if (d < (unsigned)ST_LEVELS) atomicAdd(&cnt[d], 1);
  }
  __syncthreads();
  int start[ST_LEVELS + 1];
  start[0] = 0;
#pragma unroll
  for (int k = 0; k < ST_LEVELS; ++k) start[k + 1] = start[k] + __builtin_amdgcn_readfirstlane(cnt[k]);
  if (start[ST_LEVELS] == 0) return;
  ST_CLK(0);
  // ---- the slice: one contiguous run of the chunk-major copy -> registers -> LDS --------------------------------------------
  cellv own[SB];
  const char* src = p.cm + ((long)b * (p.C / CH) + sl) * HW * CB;
#pragma unroll
  for (int j = 0; j < SB; ++j) own[j] = *(const cellv*)(src + (long)min(tid + j * NT, HW - 1) * CB);
  for (int m = tid; m < p.M; m += NT) {
    const unsigned d = (unsigned)cls[m] - (unsigned)(bl * ST_LEVELS);
    if (d < (unsigned)ST_LEVELS) {
      int st0 = start[0];
#pragma unroll
      for (int k = 1; k < ST_LEVELS; ++k) st0 = d == (unsigned)k ? start[k] : st0;
      list[st0 + atomicAdd(&cnt[8 + d], 1)] = (unsigned short)m;
    }
  }
#pragma unroll
  for (int j = 0; j < SB; ++j) *(cellv*)(smem + (size_t)min(tid + j * NT, HW - 1) * CB) = own[j];
  __syncthreads();
  ST_CLK(1);
  // ---- doubling steps: cell i takes the maximum with cell i + stride (stride = s cells along a row, s * W cells down a column).
  // No edge cases: a partner beyond the row's end is the next row's cell, one beyond the map the last cell - real values in cells
  // that cover no valid block (x + 2s > W or y + 2s > H: never looked up, never the partner of a valid cell at a later level)
  int off[SB];
#pragma unroll
  for (int j = 0; j < SB; ++j) off[j] = min(tid + j * NT, HW - 1) * CB;
  auto step = [&](int stride_bytes) {
    cellv o[SB];
#pragma unroll
    for (int j = 0; j < SB; ++j) o[j] = *(lds_cell_t)(uintptr_t)(lds0 + (unsigned)min(off[j] + stride_bytes, last));
#pragma unroll
    for (int j = 0; j < SB; ++j)
#pragma unroll
      for (int e = 0; e < VD; ++e) own[j][e] = pk_max_i16(own[j][e], o[j][e]);
    __syncthreads();  // every partner has been read (and: every wave is done pooling the previous level out of the table)
#pragma unroll
    for (int j = 0; j < SB; ++j) *(cellv*)(smem + off[j]) = own[j];
    __syncthreads();
  };
  for (int s = 1; s < (1 << l); s <<= 1) step(s * CB);
  ST_CLK(2);
  // ---- level by level down the columns; the ROIs of each level ------------------------------------------------------------------
  const int T = 1 << l;
  int curk = 0;
#pragma unroll 1
  for (int k = 0; k < ST_LEVELS; ++k) {
    int seg0 = start[0], seg1 = start[1];
#pragma unroll
    for (int q = 1; q < ST_LEVELS; ++q) {
      seg0 = k == q ? start[q] : seg0;
      seg1 = k == q ? start[q + 1] : seg1;
    }
    if (seg0 == seg1) continue;
    for (; curk < k; ++curk) step((W << curk) * CB);
    ST_CLK(3);
    const int S = 1 << k;
    for (int base = seg0 + wave; base < seg1; base += NW * ST_BATCH) {
      int my = 0;
      if (lane < ST_BATCH) my = list[min(base + lane * NW, seg1 - 1)];
      unsigned rq[ST_BATCH];
#pragma unroll
      for (int q = 0; q < ST_BATCH; ++q) rq[q] = rec[(long)__builtin_amdgcn_readlane(my, q) * 64 + lane];
#pragma unroll
      for (int q = 0; q < ST_BATCH; ++q) {
        if (base + q * NW >= seg1) break;
        const int m = __builtin_amdgcn_readlane(my, q);
        const unsigned r = rq[q];
        const unsigned meta = (unsigned)__builtin_amdgcn_readlane((int)r, 60);
        const float mul = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)r, 62));
        const int max_nr = meta & 0xff, max_nc = meta >> 8 & 0xff;
        const int y0 = r & 0xff, y1 = r >> 8 & 0xff, x0 = r >> 16 & 0xff, x1 = r >> 24;
        const int keep = y1 < y0 ? 0 : -1;  // an empty bin is +0
        cellv acc;
        if (max_nr <= 2 && max_nc <= 2) {
          const unsigned r0 = __umul24((unsigned)y0, (unsigned)W), r1 = __umul24((unsigned)y1, (unsigned)W);
          const cellv a = *(lds_cell_t)(uintptr_t)(lds0 + (r0 + (unsigned)x0) * CB);
          const cellv bq = *(lds_cell_t)(uintptr_t)(lds0 + (r0 + (unsigned)x1) * CB);
          const cellv c = *(lds_cell_t)(uintptr_t)(lds0 + (r1 + (unsigned)x0) * CB);
          const cellv d = *(lds_cell_t)(uintptr_t)(lds0 + (r1 + (unsigned)x1) * CB);
#pragma unroll
          for (int e = 0; e < VD; ++e) acc[e] = pk_max_i16(pk_max_i16(a[e], bq[e]), pk_max_i16(c[e], d[e]));
        } else {
#pragma unroll
          for (int e = 0; e < VD; ++e) acc[e] = (int)0x80008000u;
          for (int i = 0; i < max_nr; ++i) {
            const unsigned row = __umul24((unsigned)min(y0 + i * S, y1), (unsigned)W);
            for (int j = 0; j < max_nc; ++j) {
              const cellv x = *(lds_cell_t)(uintptr_t)(lds0 + (row + (unsigned)min(x0 + j * T, x1)) * CB);
#pragma unroll
              for (int e = 0; e < VD; ++e) acc[e] = pk_max_i16(acc[e], x[e]);
            }
          }
        }
        // the ROI's CH x 49 values are ONE run of A: through the wave's LDS scratch (2-byte writes at [channel][bin]) they leave as
        // one 8- / 16-byte store per lane instead of CH 2-byte stores (a wave's LDS operations execute in order: the scratch is
        // reused from ROI to ROI without a wait)
        if (lane < 49) {
          unsigned short* sp = (unsigned short*)scr + lane;
#pragma unroll
          for (int e = 0; e < VD; ++e) {
            const uint32_t y = (uint32_t)(bf16x2_order(acc[e]) & keep);
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t f = f32x2_t{__builtin_bit_cast(float, y << 16), __builtin_bit_cast(float, y & 0xffff0000u)} * mul;
            const uint32_t o = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
            sp[(2 * e) * 49] = (unsigned short)(o & 0xffffu);
            sp[(2 * e + 1) * 49] = (unsigned short)(o >> 16);
          }
          const cellv v = *(const volatile cellv*)(scr + lane * CB);
          *(cellv*)(p.out + ((long)m * p.ld_out + (long)c0 * 49) * 2 + lane * CB) = v;
        }
      }
    }
    if constexpr (PROF) {
      __syncthreads();  // (a profile build waits for the level's slowest wave here; the product waits in the next step)
      ST_CLK(4);
    }
  }
  if constexpr (PROF) {
    if (tid == 0) {
      for (int q = 0; q < 5; ++q) atomicAdd(&g_st_prof[q], tp[q]);
      atomicAdd(&g_st_prof[5], 1ull);
      atomicAdd(&g_st_prof[6], (unsigned long long)start[ST_LEVELS]);
    }
  }
#undef ST_CLK
}

static int g_roi_st_prof = 0;  // drn_tune(31, 10 / 11): profile builds on / off; (31, 12): print and clear the counters
static int g_roi_st = 1;  // drn_tune(DRN_TUNE_ROI_ST = 31): 0 = off, 1 = where it is faster (default: large maps with enough ROIs), 2 = every map whose slice fits
static size_t roi_st_align(size_t x) { return (x + 255) & ~(size_t)255; }
// cells of VD dwords for this map under the sparse-table kernel (0: not its shape)
static int roi_st_vd(int N, int H, int W, int C, int M) {
  if (!g_roi_st || H < 2 || H > 255 || W < 2 || W > 255 || N < 1 || N > 10 || M < 64 || M > 16384) return 0;
  const size_t hw = (size_t)H * W;
  if (hw > 30 * 1024) return 0;
  const size_t list = (size_t)((M + 7) & ~7) * 2 + 64;  // ROI list + counters; then a 208- / 400- / 784-byte scratch per wave
  const bool fits8 = C % 8 == 0 && hw * 16 + list + 16 * 784 <= 160 * 1024, fits4 = C % 4 == 0 && hw * 8 + list + 16 * 400 <= 160 * 1024;
  // 2 channels per cell: the stride-8 maps of the largest test-time scales (1200 x 1600: 150 x 200 cells), twice the blocks
  const bool fits2 = C % 2 == 0 && hw * 4 + list + 16 * 208 <= 160 * 1024;
  if (!fits8 && !fits4) return fits2 && (g_roi_st == 2 || M >= 400) ? 1 : 0;
  if (g_roi_st == 2) return fits8 ? 4 : fits4 ? 2 : 0;
  // default: where the table's fixed cost (staging + <= 8 doubling steps per block, ~HW) is below what the window kernels spend
  // reading every ROI's cells (~M x ROI area): profiles/r6_17_roi_st.txt, r6_18 (R = 250 / 1000 / 4000)
  if (fits8) return (M >= 600 && hw >= 3000) || (M >= 1500 && hw >= 1800) ? 4 : 0;
  return fits4 && M >= 400 ? 2 : 0;
}
static size_t roi_st_ws_bytes(int N, int H, int W, int C, int M) {
  return roi_st_align((size_t)N * H * W * C * 2) + roi_st_align((size_t)M * 256) + roi_st_align((size_t)M);
}
static bool launch_roi_st(const RoiParams& p0, hipStream_t st, void* ws, size_t ws_bytes) {
  RoiParams p = p0;
  const int vd = roi_st_vd(p.N, p.H, p.W, p.C, p.M);
  if (!vd || !ws || (((uintptr_t)ws) & 15) || ws_bytes < roi_st_ws_bytes(p.N, p.H, p.W, p.C, p.M)) return false;
  char* cm = (char*)ws;
  unsigned* rec = (unsigned*)(cm + roi_st_align((size_t)p.N * p.H * p.W * p.C * 2));
  unsigned char* cls = (unsigned char*)rec + roi_st_align((size_t)p.M * 256);
  const int HW = p.H * p.W, ch = vd * 2, nchunks = p.C / ch;
  const int sb = (HW + 1023) / 1024;
  const void* fn = nullptr;
#define ST_PICK(VD_, SB_) fn = (const void*)roi_pool7_st_kernel<VD_, SB_>
  if (vd == 1) { if (sb <= 20) ST_PICK(1, 20); else ST_PICK(1, 30); }
  else if (vd == 2) { if (sb <= 10) ST_PICK(2, 10); else if (sb <= 15) ST_PICK(2, 15); else ST_PICK(2, 20); }
  else { if (sb <= 5) ST_PICK(4, 5); else ST_PICK(4, 10); }
#undef ST_PICK
  if (g_roi_st_prof && vd == 2 && sb > 10 && sb <= 15) fn = (const void*)roi_pool7_st_kernel<2, 15, true>;
  if (g_roi_st_prof && vd == 4 && sb > 5) fn = (const void*)roi_pool7_st_kernel<4, 10, true>;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
  hipLaunchKernelGGL(roi_st_prep_kernel, dim3((p.M + 3) / 4), dim3(256), 0, st, p, rec, cls);
  const dim3 cgrid((HW + 31) / 32, (nchunks + 31) / 32, p.N);
  if (vd == 1) hipLaunchKernelGGL(roi_chunk_major_vd_kernel<1>, cgrid, dim3(256), 0, st, p.feat, cm, HW, p.C);
  else if (vd == 2) hipLaunchKernelGGL(roi_chunk_major_vd_kernel<2>, cgrid, dim3(256), 0, st, p.feat, cm, HW, p.C);
  else hipLaunchKernelGGL(roi_chunk_major_vd_kernel<4>, cgrid, dim3(256), 0, st, p.feat, cm, HW, p.C);
  p.cm = cm;
  p.walk_wmagic = (unsigned)((0x100000000ull + (unsigned)p.W - 1) / (unsigned)p.W);
  p.out_t = nullptr;
  const size_t smem = (size_t)HW * vd * 4 + (size_t)((p.M + 7) & ~7) * 2 + 64 + 16 * (vd == 1 ? 208 : vd == 2 ? 400 : 784);
  const dim3 grid((unsigned)nchunks * p.N * ST_LEVELS), block(1024);
  const unsigned* rec_c = rec;
  const unsigned char* cls_c = cls;
  void* args[] = {(void*)&p, (void*)&rec_c, (void*)&cls_c};
  return hipLaunchKernel(fn, grid, block, args, smem, st) == hipSuccess;
}

static int g_roi_lane = 1;  // drn_tune(DRN_TUNE_ROI_LANE = 19): 0 = the 64-ROI kernel writes A as before
// A (all channels) through the lane-per-bin kernel; false when the map slice of even ONE chunk does not fit
// chunks per block: as many as fit 38 KB (four 8-wave blocks per CU), else 76 KB (two), else one chunk in <= 154 KB; 0: none fits
static int roi_lane_chunks(int H, int W, int C) {
  if (C % 8) return 0;
  const size_t per_chunk = (size_t)H * W * 16;
  size_t budget = 38 * 1024;
  for (int pass = 0; pass < 3; ++pass, budget = pass == 1 ? 76 * 1024 : 154 * 1024)
    for (int k = 8; k >= 1; k >>= 1)
      if ((C / 8) % k == 0 && per_chunk * k <= budget) return k;
  return 0;
}

static int g_roi_walk_nsg = 2;  // sub-groups of 64 ROIs per block of the walking kernel on one-block-per-CU maps (tests: DRN_TUNE_ROI_LANE = 3 -> 1)
static int g_roi_lane_reps = 0;  // drn_tune(DRN_TUNE_ROI_LANE_REPS = 22): groups per block on one-block-per-CU maps (0 = default: 4, fewer while < 2 rounds of blocks)
static int cu_count_pool_fwd();
static bool roi_walk_applies(int H, int W, int C);
static bool launch_roi_lane(const RoiParams& p0, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0) {
  RoiParams p = p0;
  if (!g_roi_lane || p.C % 8) return false;
  size_t per_chunk = (size_t)p.H * p.W * 16;
  int nck = roi_lane_chunks(p.H, p.W, p.C);
  int vd = 4;
  if (!nck && p.C % 4 == 0 && (size_t)p.H * p.W * 8 <= 154 * 1024) {  // 4-channel cells: one 8-byte-per-pixel chunk per block
    nck = 1;
    vd = 2;
    per_chunk = (size_t)p.H * p.W * 8;
  }
  if (launch_roi_st(p, st, ws, ws_bytes)) return true;  // large maps: four table cells per bin instead of the window's ~77
  if (!nck) return false;
  const size_t smem = per_chunk * nck;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)roi_pool7_lane_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)roi_pool7_lane_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)roi_pool7_lane_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)roi_pool7_lane_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)roi_pool7_lane_kernel<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)roi_pool7_lane_kernel<1, 16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)roi_pool7_lane_kernel<1, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
      return false;
    attr = true;
  }
  if (nck == 1 && vd == 4 && g_roi_lane != 2 && roi_walk_applies(p.H, p.W, p.C)) {
    // one chunk per block: the walking kernel
    const size_t lds_max = 160 * 1024;
    // slice + window table + counter
    auto need = [&](int nsg_, int wp_) { return (size_t)p.H * wp_ * 16 + (size_t)64 * nsg_ * WALK_TAB * 4 + 16; };
    const bool big1 = need(1, p.W | 1) > 80 * 1024;  // one block per CU: 16 waves
    // the largest maps: first the odd pitch goes, then the second sub-group of ROIs
    // 128 ROIs per block (64 with DRN_TUNE_ROI_LANE = 3, for tests) where the table fits: half the stagings and barriers per item
    int wp = p.W | 1, nsg = g_roi_walk_nsg == 2 && (big1 || need(2, wp) <= 80 * 1024) ? 2 : 1;
    if (need(nsg, wp) > lds_max) wp = p.W;
    if (need(nsg, wp) > lds_max) nsg = 1;
    const size_t wsmem = need(nsg, wp);
    const int ngr = (p.M + 64 * nsg - 1) / (64 * nsg);
    const int nchunks = p.C / 8;
    // chunks per block: 8 (the pieces of one 128-byte line per pixel) unless fewer fill the rounds of blocks better - a VALU-bound
    // block per CU (two of the 8-wave blocks), so a grid of 1.5 rounds takes the time of 2 (75x122 / R = 1500: 12 groups x 16
    // chunk groups = 192 blocks: 244 us; 2 chunks per block = 768 blocks = 3 rounds: 204 us, profiles/r5_40_roi_walk_big.txt)
    const long slots = (long)cu_count_pool_fwd() * (big1 ? 1 : 2);
    int walk = 1;
    double best = -1.0;
    for (int w = 8, lg = 0; w >= 1; w >>= 1, ++lg) {
      if (nchunks % w != 0) continue;
      if (g_roi_lane_reps > 0 && w > g_roi_lane_reps) continue;
      const long grid = (long)ngr * (nchunks / w);
      const double score = (double)grid / (double)((grid + slots - 1) / slots * slots) * (1.0 - 0.015 * lg);
      if (g_roi_lane_reps > 0) { walk = w; break; }  // (knob: the largest admissible walk <= its value)
      if (score > best) best = score, walk = w;
    }
    p.walk = walk;
    p.walk_wp = wp;
    p.walk_wmagic = (unsigned)((0x100000000ull + (unsigned)p.W - 1) / (unsigned)p.W);
    p.out_t = nullptr;
    p.cm = nullptr;
    if (ws && ws_bytes >= (size_t)p.N * p.H * p.W * p.C * 2 && (((uintptr_t)ws) & 15) == 0) {
      const dim3 cgrid((p.H * p.W + 31) / 32, (nchunks + 31) / 32, p.N);
      hipLaunchKernelGGL(roi_chunk_major_kernel, cgrid, dim3(256), 0, st, p.feat, (char*)ws, p.H * p.W, p.C);
      p.cm = (const char*)ws;
    }
    // block shape: one block per CU -> 16 waves; two blocks per CU -> 16-wave blocks (eight waves per SIMD, <= 64 VGPRs) for slices
    // of up to 3072 cells, else 8-wave blocks (43x58: 111.9 vs 122.5 us; 50x76: 170.6 vs 161.9 us - profiles/r5_47_roi_walk_nsg2.txt)
    const bool w16 = !big1 && p.H * p.W <= 3072;
    const int cells = p.H * p.W, nt = big1 || w16 ? 1024 : 512, sb = (cells + nt - 1) / nt;
    const dim3 wgrid((unsigned)ngr * (nchunks / walk)), wblock(nt);
    const void* fn = nullptr;
#define WALK_PICK(NWV_, NSG_, SB_, OCC_) fn = (const void*)roi_pool7_walk_kernel<NWV_, NSG_, 4, SB_, OCC_>
    if (w16) {
      if (nsg == 2) WALK_PICK(16, 2, 3, 2); else WALK_PICK(16, 1, 3, 2);
    } else if (!big1) {
      if (nsg == 2) { if (sb <= 8) WALK_PICK(8, 2, 8, 1); else WALK_PICK(8, 2, 10, 1); }
      else { if (sb <= 8) WALK_PICK(8, 1, 8, 1); else WALK_PICK(8, 1, 10, 1); }
    } else {
      if (nsg == 2) { if (sb <= 6) WALK_PICK(16, 2, 6, 1); else WALK_PICK(16, 2, 10, 1); }
      else { if (sb <= 6) WALK_PICK(16, 1, 6, 1); else WALK_PICK(16, 1, 10, 1); }
    }
#undef WALK_PICK
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    void* args[] = {(void*)&p};
    if (hipLaunchKernel(fn, wgrid, wblock, args, wsmem, st) != hipSuccess) return false;
    return true;
  }
  // ROIs per block: 32 (four per wave) - the staging of the slice is then ~1/8 of the block's output bytes at 14x14; large
  // maps (one chunk of 60+ KB per block) take 64 so that the slice is staged half as often
  p.lane_g = smem > 38 * 1024 ? 64 : 32;
  int ngroups = (p.M + p.lane_g - 1) / p.lane_g;
  // one block per CU (slices beyond 76 KB): every group of 64 ROIs re-stages the slice from L2 - 32 groups x 64 slices x
  // 120 KB = 250 MB at 50x76.  A block walks `lane_reps` groups with one staged slice as long as >= 2 rounds of blocks remain
  p.lane_reps = 1;
  if (smem > 76 * 1024) {
    const long blocks1 = (long)ngroups * (p.C / (2 * vd * nck));
    // (4-channel cells - the DC5 stride-8 map: a slice is staged 8 bytes per 4-KB pixel, i.e. a whole 128-byte line per cell from
    // the Infinity Cache: 7.9 GB per launch with 4 groups per staged slice; with all of a slice's groups on one block - still
    // two rounds of blocks - the launch went from 1242 to 948 us, profiles/r5_28_*)
    int reps = g_roi_lane_reps > 0 ? g_roi_lane_reps : (vd == 2 ? 32 : 4);
    while (reps > 1 && blocks1 / reps < 2L * cu_count_pool_fwd()) reps >>= 1;
    p.lane_reps = reps;
    ngroups = (ngroups + reps - 1) / reps;
  }
  const bool big = smem > 76 * 1024;  // one block per CU: 16 waves
  // (one block per CU - maps beyond ~4700 pixels: with a block per group of 64 ROIs this kernel measured 372 vs 325 us for the
  // 64-ROI kernel at 63x92 and went there only for maps that kernel stages in two row bands; with four groups per staged
  // slice it is 293 vs 330 us at 63x92 and 260 vs 368 us at 75x122 and takes every map whose chunk fits)
  const dim3 grid((unsigned)ngroups * (p.C / (2 * vd * nck))), block(big ? 1024 : 512);
  p.out_t = nullptr;  // (A only; the caller launches the 64-ROI kernel for the A^T tail chunks)
  if (vd == 2) {
    if (big) hipLaunchKernelGGL((roi_pool7_lane_kernel<1, 16, 2>), grid, block, smem, st, p);
    else hipLaunchKernelGGL((roi_pool7_lane_kernel<1, 8, 2>), grid, block, smem, st, p);
  } else if (nck == 8) hipLaunchKernelGGL(roi_pool7_lane_kernel<8>, grid, block, smem, st, p);
  else if (nck == 4) hipLaunchKernelGGL(roi_pool7_lane_kernel<4>, grid, block, smem, st, p);
  else if (nck == 2) hipLaunchKernelGGL(roi_pool7_lane_kernel<2>, grid, block, smem, st, p);
  else if (big) hipLaunchKernelGGL((roi_pool7_lane_kernel<1, 16>), grid, block, smem, st, p);
  else hipLaunchKernelGGL(roi_pool7_lane_kernel<1>, grid, block, smem, st, p);
  return true;
}

// ROIAlign through roi_align7_lane_kernel; false when no 8-channel slice of the map fits the LDS
static bool launch_roi_align_lane(const RoiParams& p0, hipStream_t st) {
  RoiParams p = p0;
  if (p.C % 8) return false;
  int nck = roi_lane_chunks(p.H, p.W, p.C);
  if (!nck) return false;
  // (64 fp32 accumulators per lane at 8 chunks: 4 chunks per block keep the wave under 128 registers - two blocks per CU)
  if (nck > 4) nck = 4;
  const size_t smem = (size_t)p.H * p.W * 16 * nck;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)roi_align7_lane_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)roi_align7_lane_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)roi_align7_lane_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
      return false;
    attr = true;
  }
  p.lane_g = smem > 38 * 1024 ? 64 : 32;
  const int ngroups = (p.M + p.lane_g - 1) / p.lane_g;
  const dim3 grid((unsigned)ngroups * (p.C / (8 * nck))), block(512);
  p.out_t = nullptr;
  if (nck == 4) hipLaunchKernelGGL(roi_align7_lane_kernel<4>, grid, block, smem, st, p);
  else if (nck == 2) hipLaunchKernelGGL(roi_align7_lane_kernel<2>, grid, block, smem, st, p);
  else hipLaunchKernelGGL(roi_align7_lane_kernel<1>, grid, block, smem, st, p);
  return true;
}

// maps the walking kernel takes: one 8-channel chunk per block (roi_lane_chunks == 1 beyond the 38-KB class) that fits with its table
static bool roi_walk_applies(int H, int W, int C) {
  return C % 8 == 0 && W >= 2 && roi_lane_chunks(H, W, C) == 1 &&
         (size_t)H * W * 16 + 64 * WALK_TAB * 4 + 16 <= 160 * 1024 && (H * W + 1023) / 1024 <= 10;
}

static int cu_count_pool();
static int cu_count_pool_fwd() { return cu_count_pool(); }
static int cu_count_pool() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
            ? pr.multiProcessorCount : 256;
  }
  return n;
}

// Chunks per block: stand-alone the launch gets faster with 4-8 (141 -> 117-125 us at 14x14 / R = 2000: bin bounds and item
// table once per block, next slice prefetched), but INSIDE the training step it runs beside the optimizer pass and the
// trunk's conv chain, and 512 long-lived blocks - a static partition of the work - lose to 4096 short ones that the
// dispatcher balances over whichever CUs are free: same-box A/B of the whole step 646 img/s (1), 643 (2), 634 (8) against
// 646 with the previous kernel (profiles/r2_26_roi_ab.txt).  Default 1; the knob stays for stand-alone pooling (inference).
static int g_roi_cpb = 1;  // drn_tune(DRN_TUNE_ROI_CPB): most 8-channel chunks per block of the 64-ROI kernel (power of two)
static int g_roi_pf = 1;   // drn_tune(DRN_TUNE_ROI_PREFETCH): 0/1 - second map buffer, next chunk's slice fetched under the scan
static int g_roi_map64 = 512;  // drn_tune(DRN_TUNE_ROI_MAP64): 0 = off, else threads per block (256 / 512 / 1024)

static int g_roi_lds_kb = 154;  // drn_tune(DRN_TUNE_ROI_LDS_KB = 15)
static bool launch_roi_map64(const RoiParams& p0, hipStream_t st) {
  RoiParams p = p0;
  if (!g_roi_map64 || p.C % G64_CH || p.H > 255 || p.W > 255) return false;
  // LDS a block may take for its map slice + result tile (+ ~1.5 KB static): 154 KB = one block per CU with the whole
  // slice of maps up to ~80x80; DRN_TUNE_ROI_LDS_KB = 76 stages larger maps in bands so that TWO blocks share a CU
  const size_t tile_b = (size_t)ROI_G64 * G64_PITCH, budget = (size_t)g_roi_lds_kb * 1024 - tile_b;
  size_t map_b = ((size_t)p.H * p.W * 16 + 15) & ~(size_t)15;
  p.lds_px = p.H * p.W;
  if (map_b > budget) {  // bands of whole rows
    const int rows = (int)(budget / ((size_t)p.W * 16));
    if (rows < 8) return false;
    p.lds_px = rows * p.W;
    map_b = ((size_t)p.lds_px * 16 + 15) & ~(size_t)15;
  }
  size_t smem = map_b + tile_b;
  {
    // prefetch mode: whole map in one band, two buffers within the same blocks-per-CU class, <= 2 pixels per thread
    const int e = g_roi_pf;
    const size_t cls = smem <= 76 * 1024 ? 76 * 1024 : 156 * 1024;
    const int thr = smem > 76 * 1024 && g_roi_map64 == 512 ? 1024 : g_roi_map64;
    p.pf = e && p.lds_px == p.H * p.W && smem + map_b <= cls && p.H * p.W <= 2 * thr;
    if (p.pf) smem += map_b;
  }
  // <= 76 KB: two blocks per CU (the 14x14 .. 38x38 maps).  Up to 156 KB - the 40x60 .. 63x100 maps of real-size training
  // images - ONE block per CU still stages its 8-channel map slice once per 64 ROIs; the 8-ROI whole-map kernel that
  // these maps used to fall to re-stages it per 8 ROIs (2.9 GB through L2 per call at 63x92: 1.5 ms, half of the eager
  // step at 1000x1464, `profiles/r2_15_*`)
  if (smem > 156 * 1024) return false;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)roi_pool7_map64_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)roi_pool7_map64_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)roi_pool7_map64_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
      return false;
    attr = true;
  }
  const int ngroups = (p.M + ROI_G64 - 1) / ROI_G64;
  // one block per CU (maps beyond ~38x38): 1024 threads - the window scans are latency-bound and eight waves per CU hide
  // little of it (63x92 map, 2000 proposals: 467 -> 394 us); two blocks per CU: the tuned 512
  const int threads = smem > 76 * 1024 && g_roi_map64 == 512 ? 1024 : g_roi_map64;
  // channel chunks per block (tune knob, default 1): the bin bounds of a 64-ROI group and the per-thread item table are
  // paid once per `cpb` chunks instead of once per chunk; largest power of two <= the knob that still leaves two blocks
  // for every CU
  int cpb = g_roi_cpb;
  const int nchunks = (p.C - p.c_begin) / G64_CH;
  while (cpb > 1 && (nchunks % cpb || (long)(nchunks / cpb) * ngroups < 2L * cu_count_pool())) cpb >>= 1;
  p.cpb = cpb;
  const dim3 grid((nchunks / cpb) * ngroups), block(threads);
  if (threads >= 1024) hipLaunchKernelGGL(roi_pool7_map64_kernel<4>, grid, block, smem, st, p);
  else if (threads >= 512) hipLaunchKernelGGL(roi_pool7_map64_kernel<7>, grid, block, smem, st, p);
  else hipLaunchKernelGGL(roi_pool7_map64_kernel<13>, grid, block, smem, st, p);
  return true;
}

// bf16 -> bf16 transpose with 16-B global accesses on both sides (the A -> A^T copy of the fc6 operand is
// 2 x 205 MB per step): 64x64 tile, rows read as 8-element vectors, written transposed into LDS, re-read as vectors.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                             int rows, int cols, long ld_in, long ld_out) {
  __shared__ bf16_t t[64][72];  // [c][r], row pitch 144 B keeps the 16-B reads aligned and spreads banks
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int v = threadIdx.x & 7, rr = threadIdx.x >> 3;  // 8 vectors per 64-element row, 32 rows per pass
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r = r0 + rr + 32 * pass, c = c0 + v * 8;
    i32x4_t x = {0, 0, 0, 0};
    if (r < rows && c + 8 <= cols) x = *(const i32x4_t*)(in + (long)r * ld_in + c);
    else if (r < rows)
      for (int e = 0; e < 8; ++e)
        if (c + e < cols) ((bf16_t*)&x)[e] = in[(long)r * ld_in + c + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[v * 8 + e][rr + 32 * pass] = ((const bf16_t*)&x)[e];
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int c = c0 + rr + 32 * pass, r = r0 + v * 8;
    if (c >= cols) continue;
    const i32x4_t x = *(const i32x4_t*)(&t[rr + 32 * pass][v * 8]);
    if (r + 8 <= rows) *(i32x4_t*)(out + (long)c * ld_out + r) = x;
    else
      for (int e = 0; e < 8; ++e)
        if (r + e < rows) out[(long)c * ld_out + r + e] = ((const bf16_t*)&x)[e];
  }
}

// out[c][r] = (T_OUT) in[r][c]; 64x64 tiles through LDS, both sides coalesced.
template <int DT_IN, int DT_OUT>
__global__ __launch_bounds__(256) void transpose_kernel(const char* __restrict__ in, char* __restrict__ out, int rows,
                                                        int cols, long ld_in, long ld_out) {
  using EI = ElemOf<DT_IN>;
  using EO = ElemOf<DT_OUT>;
  __shared__ float t[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    t[i][tx] = (r < rows && c < cols) ? EI::ld((const typename EI::type*)in + (long)r * ld_in + c) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) EO::st((typename EO::type*)out + (long)c * ld_out + r, t[tx][i]);
  }
}

template <int DT_IN, int DT_OUT>
__global__ void cast2d_kernel(const char* __restrict__ in, char* __restrict__ out, int rows, int cols, long ld_in,
                              long ld_out) {
  using EI = ElemOf<DT_IN>;
  using EO = ElemOf<DT_OUT>;
  const long total = (long)rows * cols;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols, c = i - r * cols;
    EO::st((typename EO::type*)out + r * ld_out + c, EI::ld((const typename EI::type*)in + r * ld_in + c));
  }
}

inline int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int drn_transpose2d(const void* in, void* out, int rows, int cols, long ld_in, long ld_out, int in_dtype,
                               int out_dtype, void* stream);

template <int DT, int CH>
static bool launch_roi_map(const RoiParams& p, hipStream_t st, size_t lds_budget) {
  const int es = DT == DRN_BF16 ? 2 : 4;
  if (p.C % CH) return false;
  const size_t smem = (((size_t)p.H * p.W * CH * es + 15) & ~(size_t)15) + (size_t)ROI_GROUP * CH * 49 * es;
  if (smem > lds_budget) return false;
  const int ngroups = (p.M + ROI_GROUP - 1) / ROI_GROUP;
  // groups per block: keep the bytes staged per block (H*W pixels) below the bytes it writes (8 ROIs x 49 bins x 2
  // copies per group) - 1 for the 14x14 training map, up to 10 for a 75x100 map
  RoiParams q = p;
  q.gpw = (p.H * p.W + 783) / 784;
  if (q.gpw > 16) q.gpw = 16;
  if (q.gpw < 1) q.gpw = 1;
  const int nblk = (ngroups + q.gpw - 1) / q.gpw;
  auto k = roi_pool7_map_kernel<DT, CH>;
  static bool attr = false;
  if (!attr && smem > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return false;
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3((p.C / CH) * nblk), dim3(256), smem, st, q);
  return true;
}

extern "C" {

__attribute__((visibility("hidden"))) int drn_roi_set_chunks(int cpb) {
  const int old = g_roi_cpb;
  if (cpb >= 1 && cpb <= 64 && (cpb & (cpb - 1)) == 0) g_roi_cpb = cpb;
  return old;
}

static int g_roi_map64_a = 0;  // drn_tune(DRN_TUNE_ROI_MAP64_A = 14): 1 = the 64-ROI kernel also for A alone (no A^T)
__attribute__((visibility("hidden"))) int drn_roi_set_map64_a(int on) {
  const int old = g_roi_map64_a;
  g_roi_map64_a = on != 0;
  return old;
}

__attribute__((visibility("hidden"))) int drn_roi_set_lds_kb(int kb) {
  const int old = g_roi_lds_kb;
  if (kb >= 60 && kb <= 154) g_roi_lds_kb = kb;
  return old;
}

__attribute__((visibility("hidden"))) int drn_roi_set_lane_reps(int reps) {
  const int old = g_roi_lane_reps;
  if (reps >= 0 && reps <= 64) g_roi_lane_reps = reps;
  return old;
}
__attribute__((visibility("hidden"))) int drn_roi_set_lane(int on) {
  const int old = g_roi_lane;
  g_roi_walk_nsg = on == 3 ? 1 : 2;
  g_roi_lane = on < 0 ? 0 : on == 3 ? 1 : on > 2 ? 2 : on;
  return old;
}

__attribute__((visibility("hidden"))) int drn_roi_set_st(int on) {
  const int old = g_roi_st;
  if (on >= 0 && on <= 2) g_roi_st = on;
  if (on == 10 || on == 11) g_roi_st_prof = on == 10;
  if (on == 12) {
    unsigned long long h[8];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_st_prof), sizeof(h)) != hipSuccess) return -1;
    const double n = h[5] ? (double)h[5] : 1.0;
    fprintf(stderr, "roi_st profile: %llu blocks, %.1f ROIs each | shader-clock cycles per block: scan %.0f  slice %.0f  row steps %.0f  column steps %.0f  pooling %.0f\n",
            h[5], (double)h[6] / n, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n);
    for (auto& x : h) x = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_st_prof), h, sizeof(h)) != hipSuccess) return -1;
  }
  return old;
}

__attribute__((visibility("hidden"))) int drn_roi_set_prefetch(int on) {
  const int old = g_roi_pf;
  g_roi_pf = on != 0;
  return old;
}

__attribute__((visibility("hidden"))) int drn_roi_set_map64(int on) {
  const int old = g_roi_map64;
  g_roi_map64 = on == 1 ? 512 : (on == 0 || on == 256 || on == 512 || on == 1024) ? on : old;
  return old;
}

int drn_resize_bilinear_u8(const void* src_hwc, int H, int W, int C, float* dst_chw, int Ho, int Wo, const int* xbounds,
                           const int* xcoef, int ksx, const int* ybounds, const int* ycoef, int ksy, int flip, void* stream) {
  if (!src_hwc || !dst_chw || H <= 0 || W <= 0 || C < 1 || C > 4 || Ho <= 0 || Wo <= 0) return DRN_ERR_ARG;
  if ((xbounds && (!xcoef || ksx < 1)) || (ybounds && (!ycoef || ksy < 1))) return DRN_ERR_ARG;
  if ((!xbounds && Wo != W) || (!ybounds && Ho != H)) return DRN_ERR_ARG;  // no pass in a direction: the size stays
  const long total = (long)Ho * Wo;
  hipLaunchKernelGGL(resize_u8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)src_hwc, H, W, C, dst_chw, Ho, Wo, xbounds, xcoef, ksx, ybounds, ycoef, ksy, flip);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_preprocess_nhwc(const float* img_chw, int C, int H, int W, void* out_nhwc, int Hp, int Wp, int Cp,
                        const float* mean3, const float* std3, int dtype, void* stream) {
  if (!img_chw || !out_nhwc || C > 3 || C < 1 || Cp < C || H > Hp || W > Wp) return DRN_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)Hp * Wp * Cp;
  if (dtype == DRN_BF16)
    hipLaunchKernelGGL(preprocess_kernel<DRN_BF16>, dim3(grid_for(total, 256)), dim3(256), 0, st, img_chw, C, H, W,
                       (bf16_t*)out_nhwc, Hp, Wp, Cp, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  else if (dtype == DRN_F32)
    hipLaunchKernelGGL(preprocess_kernel<DRN_F32>, dim3(grid_for(total, 256)), dim3(256), 0, st, img_chw, C, H, W,
                       (float*)out_nhwc, Hp, Wp, Cp, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  else
    return DRN_ERR_ARG;
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_maxpool2x2_nhwc(const void* x, void* y, int Nb, int H, int W, int C, int stride, int dtype, void* stream) {
  if (!x || !y || (stride != 1 && stride != 2) || H < 2 || W < 2) return DRN_ERR_ARG;
  const int es = drn_esize(dtype);
  if ((C * es) % 16) return DRN_ERR_ARG;
  const int Ho = (H - 2) / stride + 1, Wo = (W - 2) / stride + 1;
  const long total = (long)Nb * Ho * Wo * (C * es / 16);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DRN_BF16)
    hipLaunchKernelGGL(maxpool2x2_kernel<DRN_BF16>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const char*)x,
                       (char*)y, Nb, H, W, C, Ho, Wo, stride);
  else if (dtype == DRN_F32)
    hipLaunchKernelGGL(maxpool2x2_kernel<DRN_F32>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const char*)x,
                       (char*)y, Nb, H, W, C, Ho, Wo, stride);
  else if (dtype == DRN_FP8)
    hipLaunchKernelGGL(maxpool2x2_kernel<DRN_FP8>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const char*)x,
                       (char*)y, Nb, H, W, C, Ho, Wo, stride);
  else
    return DRN_ERR_ARG;
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_im2col_t(const void* x, void* out, int Nb, int H, int W, int Cin, int ldc, int KH, int KW, int stride, int pad,
                 int dil, long ld_out, int dtype, void* stream) {
  if (!x || !out || Nb < 1 || Cin < 1 || ldc < Cin || KH < 1 || KW < 1 || stride < 1 || dil < 1) return DRN_ERR_ARG;
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  if (Ho < 1 || Wo < 1 || ld_out < (long)Nb * Ho * Wo) return DRN_ERR_ARG;
  dim3 grid((Nb * Ho * Wo + 63) / 64, (Cin + 63) / 64, KH * KW), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DRN_BF16)
    hipLaunchKernelGGL(im2col_t_kernel<DRN_BF16>, grid, block, 0, st, (const char*)x, (char*)out, Nb, H, W, Cin, ldc, KH,
                       KW, stride, pad, dil, Ho, Wo, ld_out);
  else if (dtype == DRN_F32)
    hipLaunchKernelGGL(im2col_t_kernel<DRN_F32>, grid, block, 0, st, (const char*)x, (char*)out, Nb, H, W, Cin, ldc, KH,
                       KW, stride, pad, dil, Ho, Wo, ld_out);
  else
    return DRN_ERR_ARG;
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_maxpool2x2_bwd_nhwc(const void* x, const void* dy, void* dx, int Nb, int H, int W, int C, int stride, int dtype,
                            void* stream) {
  if (!x || !dy || !dx || (stride != 1 && stride != 2) || H < 2 || W < 2) return DRN_ERR_ARG;
  const int Ho = (H - 2) / stride + 1, Wo = (W - 2) / stride + 1;
  const long total = (long)Nb * H * W * C;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DRN_BF16)
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel<DRN_BF16>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const char*)x,
                       (const char*)dy, (char*)dx, Nb, H, W, C, Ho, Wo, stride);
  else if (dtype == DRN_F32)
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel<DRN_F32>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const char*)x,
                       (const char*)dy, (char*)dx, Nb, H, W, C, Ho, Wo, stride);
  else
    return DRN_ERR_ARG;
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// props[M][4] <- rois[M][1:5] and (optional) words_dst[n_words] <- words_src: the two hand-overs from the staged next
// batch to the buffers the heads read (proposal boxes for pseudo-GT mining / IoU labelling; the image-level label block),
// in ONE launch in front of the pooling kernel.  See include/drn_wsod.h.
__global__ void stage_rois_kernel(const float* __restrict__ boxes, const float* __restrict__ logits, float batch_index,
                                  float* __restrict__ rois, float* __restrict__ obj, float* __restrict__ props, int M) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float x0 = boxes[4 * (long)i], y0 = boxes[4 * (long)i + 1], x1 = boxes[4 * (long)i + 2], y1 = boxes[4 * (long)i + 3];
  float* r = rois + 5 * (long)i;
  r[0] = batch_index; r[1] = x0; r[2] = y0; r[3] = x1; r[4] = y1;
  if (props) { float* q = props + 4 * (long)i; q[0] = x0; q[1] = y0; q[2] = x1; q[3] = y1; }
  if (obj) obj[i] = logits[i];
}

__global__ void stage_heads_kernel(const float* __restrict__ rois, float* __restrict__ props, int M,
                                   const int* __restrict__ words_src, int* __restrict__ words_dst, int n_words) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M * 4) props[i] = rois[(i >> 2) * 5 + 1 + (i & 3)];
  if (i < n_words) words_dst[i] = words_src[i];
}

int drn_stage_heads_inputs(const float* rois, float* props, int M, const int* words_src, int* words_dst, int n_words,
                           void* stream) {
  if (M < 0 || n_words < 0 || (M > 0 && (!rois || !props)) || (n_words > 0 && (!words_src || !words_dst))) return DRN_ERR_ARG;
  const long n = (long)M * 4 > n_words ? (long)M * 4 : n_words;
  if (n == 0) return DRN_OK;
  hipLaunchKernelGGL(stage_heads_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, rois, props, M,
                     words_src, words_dst, n_words);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// boxes [M][4] + objectness logits [M] of ONE image -> the pooler's rois [M][5] = (batch index, x0, y0, x1, y1)
// (convert_boxes_to_pooler_format, detectron2/modeling/poolers.py:69-96), a contiguous copy of the logits and of the boxes:
// one launch for what was torch.full + two torch.cat + two copies in front of every forward
int drn_stage_rois(const float* boxes, const float* logits, float batch_index, float* rois, float* obj, float* props, int M,
                   void* stream) {
  if (M < 0 || (M > 0 && (!boxes || !rois))) return DRN_ERR_ARG;
  if ((obj != nullptr) != (logits != nullptr)) return DRN_ERR_ARG;
  if (M == 0) return DRN_OK;
  hipLaunchKernelGGL(stage_rois_kernel, dim3(grid_for(M, 256)), dim3(256), 0, (hipStream_t)stream, boxes, logits, batch_index,
                     rois, obj, props, M);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_add(const void* a, const void* b, void* out, long n, int dtype, void* stream) {
  if (!a || !b || !out || n < 0) return DRN_ERR_ARG;
  if (n == 0) return DRN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DRN_BF16)
    hipLaunchKernelGGL(add_kernel<DRN_BF16>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const char*)a, (const char*)b, (char*)out, n);
  else if (dtype == DRN_F32)
    hipLaunchKernelGGL(add_kernel<DRN_F32>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const char*)a, (const char*)b, (char*)out, n);
  else
    return DRN_ERR_ARG;
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_roi_pool_nhwc_t(const void* feat, const float* rois, const float* objectness, void* out, void* out_t,
                        int32_t* argmax, int N, int H, int W, int C, int P, int M, float spatial_scale, long ld_out,
                        long ld_out_t, int mode, int sampling_ratio, int aligned, int in_dtype, int out_dtype,
                        int t_first_channel, void* stream);
int drn_roi_pool_nhwc_ws(const void* feat, const float* rois, const float* objectness, void* out, void* out_t,
                         int32_t* argmax, int N, int H, int W, int C, int P, int M, float spatial_scale, long ld_out,
                         long ld_out_t, int mode, int sampling_ratio, int aligned, int in_dtype, int out_dtype,
                         int t_first_channel, void* workspace, long workspace_bytes, void* stream);

// mode 0 = RoIPool, 1 = ROIAlign. in_dtype = feature dtype, out_dtype = pooled dtype.
int drn_roi_pool_nhwc(const void* feat, const float* rois, const float* objectness, void* out, void* out_t,
                      int32_t* argmax, int N, int H, int W, int C, int P, int M, float spatial_scale, long ld_out,
                      long ld_out_t, int mode, int sampling_ratio, int aligned, int in_dtype, int out_dtype,
                      void* stream) {
  return drn_roi_pool_nhwc_t(feat, rois, objectness, out, out_t, argmax, N, H, W, C, P, M, spatial_scale, ld_out, ld_out_t,
                             mode, sampling_ratio, aligned, in_dtype, out_dtype, 0, stream);
}

// The same with a hint: rows of out_t below channel t_first_channel need not be written (the 64-ROI training kernel then
// skips its A^T store loop for those channel chunks; every other path writes all of out_t).
int drn_roi_pool_nhwc_t(const void* feat, const float* rois, const float* objectness, void* out, void* out_t,
                        int32_t* argmax, int N, int H, int W, int C, int P, int M, float spatial_scale, long ld_out,
                        long ld_out_t, int mode, int sampling_ratio, int aligned, int in_dtype, int out_dtype,
                        int t_first_channel, void* stream) {
  return drn_roi_pool_nhwc_ws(feat, rois, objectness, out, out_t, argmax, N, H, W, C, P, M, spatial_scale, ld_out, ld_out_t, mode,
                              sampling_ratio, aligned, in_dtype, out_dtype, t_first_channel, nullptr, 0, stream);
}

// Bytes of workspace with which drn_roi_pool_nhwc_ws pools this shape faster (0: the shape takes a kernel that needs none).
long drn_roi_pool_workspace_bytes(int N, int H, int W, int C, int P, int M, int mode, int has_argmax, int in_dtype, int out_dtype) {
  if (N < 1 || H < 1 || W < 1 || C < 1) return 0;
  if (mode != 0 || P != 7 || has_argmax || in_dtype != DRN_BF16 || out_dtype != DRN_BF16 || M < ROI_G64 || C % G64_CH != 0) return 0;
  if (g_roi_lane != 0 && roi_st_vd(N, H, W, C, M)) return (long)roi_st_ws_bytes(N, H, W, C, M);
  return g_roi_lane != 0 && g_roi_lane != 2 && roi_walk_applies(H, W, C) ? (long)N * H * W * C * 2 : 0;
}

// The same with a caller-owned workspace (drn_roi_pool_workspace_bytes; null / too small: as without): maps whose 8-channel slice
// leaves one chunk per block are first copied chunk-major into it, so that the walking kernel stages contiguous runs.
int drn_roi_pool_nhwc_ws(const void* feat, const float* rois, const float* objectness, void* out, void* out_t,
                         int32_t* argmax, int N, int H, int W, int C, int P, int M, float spatial_scale, long ld_out,
                         long ld_out_t, int mode, int sampling_ratio, int aligned, int in_dtype, int out_dtype,
                         int t_first_channel, void* workspace, long workspace_bytes, void* stream) {
  if (!feat || !rois || !out || P < 1 || P * P > RP_MAXBIN || M < 0 || (mode != 0 && mode != 1)) return DRN_ERR_ARG;
  if (workspace_bytes < 0) return DRN_ERR_ARG;
  if (t_first_channel < 0) return DRN_ERR_ARG;
  if (ld_out < (long)C * P * P || (out_t && ld_out_t < M)) return DRN_ERR_ARG;
  if (M == 0) return DRN_OK;
  RoiParams p{(const char*)feat, rois, objectness, (char*)out, argmax, N, H, W, C, P, M, spatial_scale, ld_out,
              sampling_ratio, aligned, 0, (char*)out_t, ld_out_t};
  p.t_c0 = out_t ? t_first_channel : 0;
  dim3 grid(M, (C + RP_CH - 1) / RP_CH), block(256);
  hipStream_t st = (hipStream_t)stream;
  {
    // whole-map path (see roi_pool7_map_kernel): 7x7 ROIPool, same in/out dtype, no argmax, 16-B aligned runs
    const int es = drn_esize(out_dtype);
    const bool al = ((ld_out * es) % 16) == 0 && (((uintptr_t)out) & 15) == 0 && (((uintptr_t)feat) & 15) == 0 &&
                    (!out_t || (((ld_out_t * es) % 16) == 0 && (((uintptr_t)out_t) & 15) == 0));
    if (mode == 0 && P == 7 && !argmax && in_dtype == out_dtype && al) {
      bool done = false;
      // channel slice per block: 32 wide when two blocks fit a CU (the 14x14 .. 28x28 training maps), else the widest
      // slice whose map fits at all - the 43x58 .. 75x100 maps of test-time scales need 16 or 8 channels and most of
      // a CU's LDS (one block per CU), which still beats the per-ROI window kernels by 3-4x there
      const size_t two = 80 * 1024, one = 156 * 1024;
      if (in_dtype == DRN_BF16 && M >= ROI_G64) {  // the training operand (pair), and A alone at inference (46 vs 83 us at
        // 14x14, 194 vs 433 us at 50x76 against the 8-ROI whole-map kernels: tools/roi_a_alone_bench.py)
        // round 4: A from the lane-per-bin kernel; the 64-ROI kernel - full 128-byte A^T lines - then only for the channel
        // chunks whose A^T rows the fc6 dW still reads (the tail its peel takes; it writes their A runs again, same values)
        // (built and measured: the tail's A^T rows as 2-byte stores from the lane kernel itself - 49 partial lines per
        // instruction - cost 40 us for 4.7 MB at the bench shape; the 64-ROI kernel's full lines cost ~8 us as a launch)
        const int cb = out_t ? p.t_c0 / G64_CH * G64_CH : C;
        const bool few_t = !out_t || (long)(C - cb) * 8 <= C;
        if (few_t && C % G64_CH == 0 && launch_roi_lane(p, st, workspace, (size_t)workspace_bytes)) {
          done = true;
          if (out_t && cb < C) {
            RoiParams q = p;
            q.c_begin = cb;
            done = launch_roi_map64(q, st);
            if (!done) done = launch_roi_map64(p, st);  // (cannot happen for shapes the lane kernel took)
          }
        } else if (out_t || g_roi_map64_a) {
          done = launch_roi_map64(p, st);
        }
      }
      if (done) {
      } else if (in_dtype == DRN_BF16)
        // (maps too large for two 8-ROI blocks per CU - inference at real image sizes, no A^T - also take the 64-ROI
        // kernel: the 8-ROI one would re-stage its map slice per 8 proposals)
        done = launch_roi_map<DRN_BF16, 32>(p, st, two) || launch_roi_map<DRN_BF16, 64>(p, st, two) ||
               launch_roi_map<DRN_BF16, 16>(p, st, two) || launch_roi_map<DRN_BF16, 8>(p, st, two) ||
               (M >= ROI_G64 && launch_roi_map64(p, st)) ||
               launch_roi_map<DRN_BF16, 32>(p, st, one) || launch_roi_map<DRN_BF16, 16>(p, st, one) ||
               launch_roi_map<DRN_BF16, 8>(p, st, one);
      else if (in_dtype == DRN_F32)
        done = launch_roi_map<DRN_F32, 32>(p, st, two) || launch_roi_map<DRN_F32, 16>(p, st, two) ||
               launch_roi_map<DRN_F32, 8>(p, st, two) || launch_roi_map<DRN_F32, 4>(p, st, two) ||
               launch_roi_map<DRN_F32, 16>(p, st, one) || launch_roi_map<DRN_F32, 8>(p, st, one) ||
               launch_roi_map<DRN_F32, 4>(p, st, one);
      if (done) {
        DRN_CHECK_LAUNCH();
        return DRN_OK;
      }
    }
  }
  // ROIAlign, bf16 -> bf16, P = 7, channels in chunks of 8, a slice of the map in LDS: the lane-per-bin form
  if (mode == 1 && P == 7 && !out_t && in_dtype == DRN_BF16 && out_dtype == DRN_BF16 && g_roi_lane && (((uintptr_t)feat) & 15) == 0 &&
      M >= 32 && launch_roi_align_lane(p, st)) {
    DRN_CHECK_LAUNCH();
    return DRN_OK;
  }
  if (out_t) {  // general shapes: pool into `out`, then the transpose pass
    p.out_t = nullptr;
    int rc = drn_roi_pool_nhwc(feat, rois, objectness, out, nullptr, argmax, N, H, W, C, P, M, spatial_scale, ld_out, 0,
                               mode, sampling_ratio, aligned, in_dtype, out_dtype, stream);
    if (rc != DRN_OK) return rc;
    return drn_transpose2d(out, out_t, M, C * P * P, ld_out, ld_out_t, out_dtype, out_dtype, stream);
  }
  // ROIPool on a full 64-channel chunk stages the box window in LDS: up to 256 pixels (25 KB bf16 / 64 KB f32... capped)
  size_t smem = 0;
  if (mode == 0 && C % RP_CH == 0) {
    const int es = drn_esize(in_dtype);
    int px = H * W < 256 ? H * W : 256;
    if ((size_t)px * RP_CH * es > 32 * 1024) px = 32 * 1024 / (RP_CH * es);
    p.lds_px = px;
    smem = (size_t)px * RP_CH * es;
  }
  // fast path: 7x7 ROIPool, no argmax wanted, whole map fits the staging tile, channels in full 64-wide chunks,
  // 16-B aligned output rows
  const int eso = drn_esize(out_dtype);
  if (mode == 0 && P == 7 && !argmax && C % RP_CH == 0 && H * W <= p.lds_px && p.lds_px > 0 && H * W <= 256 &&
      ((ld_out * eso) % 16) == 0 && (((uintptr_t)out) & 15) == 0 && (in_dtype == out_dtype)) {
    dim3 g7(M, (C / RP_CH + 3) / 4);
    if (in_dtype == DRN_BF16) hipLaunchKernelGGL((roi_pool7_kernel<DRN_BF16, DRN_BF16>), g7, block, smem, st, p);
    else hipLaunchKernelGGL((roi_pool7_kernel<DRN_F32, DRN_F32>), g7, block, smem, st, p);
    DRN_CHECK_LAUNCH();
    return DRN_OK;
  }
#define RP_LAUNCH(DI, DO, MD) hipLaunchKernelGGL((roi_kernel<DI, DO, MD>), grid, block, smem, st, p)
  if (in_dtype == DRN_BF16 && out_dtype == DRN_BF16) { if (mode == 0) RP_LAUNCH(DRN_BF16, DRN_BF16, 0); else RP_LAUNCH(DRN_BF16, DRN_BF16, 1); }
  else if (in_dtype == DRN_F32 && out_dtype == DRN_F32) { if (mode == 0) RP_LAUNCH(DRN_F32, DRN_F32, 0); else RP_LAUNCH(DRN_F32, DRN_F32, 1); }
  else if (in_dtype == DRN_F32 && out_dtype == DRN_BF16) { if (mode == 0) RP_LAUNCH(DRN_F32, DRN_BF16, 0); else RP_LAUNCH(DRN_F32, DRN_BF16, 1); }
  else if (in_dtype == DRN_BF16 && out_dtype == DRN_F32) { if (mode == 0) RP_LAUNCH(DRN_BF16, DRN_F32, 0); else RP_LAUNCH(DRN_BF16, DRN_F32, 1); }
  else return DRN_ERR_ARG;
#undef RP_LAUNCH
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// d(feat) of drn_roi_pool_nhwc: grad_out [M][ld_g] (k = c*P*P + bin, fp32 or bf16) -> dfeat [N][H][W][C] fp32 (zeroed
// here).  mode 0 needs the arg-max the forward returned; `objectness` as in the forward (fused scaling).
int drn_roi_pool_backward_nhwc(const void* grad_out, const float* rois, const float* objectness, const int32_t* argmax,
                               float* dfeat, int N, int H, int W, int C, int P, int M, float spatial_scale, long ld_g,
                               int mode, int sampling_ratio, int aligned, int grad_dtype, void* stream) {
  if (!grad_out || !rois || !dfeat || P < 1 || P * P > RP_MAXBIN || M < 0 || (mode != 0 && mode != 1)) return DRN_ERR_ARG;
  if ((mode == 0 && !argmax) || ld_g < (long)C * P * P || N < 1 || H < 1 || W < 1 || C < 1) return DRN_ERR_ARG;
  if (grad_dtype != DRN_F32 && grad_dtype != DRN_BF16) return DRN_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dfeat, 0, sizeof(float) * (size_t)N * H * W * C, st) != hipSuccess) return DRN_ERR_LAUNCH;
  if (M == 0) return DRN_OK;
  RoiBwdParams p{(const char*)grad_out, rois, objectness, argmax, dfeat, N, H, W, C, P, M, spatial_scale, ld_g,
                 sampling_ratio, aligned};
  dim3 grid(M, (C + RP_CH - 1) / RP_CH), block(256);
#define RB_LAUNCH(DT, MD) hipLaunchKernelGGL((roi_bwd_kernel<DT, MD>), grid, block, 0, st, p)
  if (grad_dtype == DRN_BF16) { if (mode == 0) RB_LAUNCH(DRN_BF16, 0); else RB_LAUNCH(DRN_BF16, 1); }
  else { if (mode == 0) RB_LAUNCH(DRN_F32, 0); else RB_LAUNCH(DRN_F32, 1); }
#undef RB_LAUNCH
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_transpose2d(const void* in, void* out, int rows, int cols, long ld_in, long ld_out, int in_dtype,
                    int out_dtype, void* stream) {
  if (!in || !out || rows < 0 || cols < 0) return DRN_ERR_ARG;
  if (rows == 0 || cols == 0) return DRN_OK;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64), block(256);
  hipStream_t st = (hipStream_t)stream;
#define TR_LAUNCH(DI, DO) hipLaunchKernelGGL((transpose_kernel<DI, DO>), grid, block, 0, st, (const char*)in, (char*)out, rows, cols, ld_in, ld_out)
  if (in_dtype == DRN_BF16 && out_dtype == DRN_BF16 && (ld_in % 8) == 0 && (ld_out % 8) == 0 &&
      ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0)
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, block, 0, st, (const bf16_t*)in, (bf16_t*)out, rows, cols, ld_in, ld_out);
  else if (in_dtype == DRN_BF16 && out_dtype == DRN_BF16) TR_LAUNCH(DRN_BF16, DRN_BF16);
  else if (in_dtype == DRN_F32 && out_dtype == DRN_F32) TR_LAUNCH(DRN_F32, DRN_F32);
  else if (in_dtype == DRN_F32 && out_dtype == DRN_BF16) TR_LAUNCH(DRN_F32, DRN_BF16);
  else if (in_dtype == DRN_BF16 && out_dtype == DRN_F32) TR_LAUNCH(DRN_BF16, DRN_F32);
  else return DRN_ERR_ARG;
#undef TR_LAUNCH
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

// out[r][c] = cast(in[r][c]) with independent leading dimensions (refreshes padded compute shadows).
int drn_cast2d(const void* in, void* out, int rows, int cols, long ld_in, long ld_out, int in_dtype, int out_dtype,
               void* stream) {
  if (!in || !out || rows < 0 || cols < 0) return DRN_ERR_ARG;
  if (rows == 0 || cols == 0) return DRN_OK;
  const long total = (long)rows * cols;
  dim3 grid(grid_for(total, 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define CA_LAUNCH(DI, DO) hipLaunchKernelGGL((cast2d_kernel<DI, DO>), grid, block, 0, st, (const char*)in, (char*)out, rows, cols, ld_in, ld_out)
  if (in_dtype == DRN_BF16 && out_dtype == DRN_BF16) CA_LAUNCH(DRN_BF16, DRN_BF16);
  else if (in_dtype == DRN_F32 && out_dtype == DRN_F32) CA_LAUNCH(DRN_F32, DRN_F32);
  else if (in_dtype == DRN_F32 && out_dtype == DRN_BF16) CA_LAUNCH(DRN_F32, DRN_BF16);
  else if (in_dtype == DRN_BF16 && out_dtype == DRN_F32) CA_LAUNCH(DRN_BF16, DRN_F32);
  else return DRN_ERR_ARG;
#undef CA_LAUNCH
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

}  // extern "C"
