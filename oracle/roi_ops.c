/*
 * ORACLE (test infrastructure, not product code).
 *
 * Plain-C CPU restatement of the region ops on the DRN-WSOD hot path:
 *   - RoIPool fwd/bwd      : torchvision.ops.RoIPool semantics (source NOT in /root/reference;
 *                            un-pinned external dependency, call site
 *                            detectron2/modeling/poolers.py:162-165; semantics restated in
 *                            SURVEY.md Appendix C.1).  PARITY UNPINNED for this op.
 *   - ROIAlign fwd/bwd     : follows detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:20-114
 *                            (bilinear pre-calc), :116-218 (forward), :220-395 (backward).
 *                            Pinned against oracle/_ref (the reference file compiled as-is) and
 *                            the goldens of tests/layers/test_roi_align.py:26-39.
 *   - nms / batched_nms    : torchvision.ops.nms / boxes.batched_nms semantics (SURVEY Appendix
 *                            C.2/C.3; call sites detectron2/layers/nms.py:10-29).  PARITY UNPINNED.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * All arithmetic is fp32, layouts are the reference's (NCHW features, [M,5] rois).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------ RoIPool (Appendix C.1) */
void oracle_roi_pool_forward(const float* input, const float* rois, int num_rois, int channels,
                             int height, int width, int pooled_h, int pooled_w,
                             float spatial_scale, float* output, int32_t* argmax) {
#pragma omp parallel for schedule(static)
  for (int n = 0; n < num_rois; ++n) {
    const float* roi = rois + 5 * n;
    const int b = (int)roi[0];
    /* C round(): half away from zero */
    const int x1 = (int)roundf(roi[1] * spatial_scale);
    const int y1 = (int)roundf(roi[2] * spatial_scale);
    const int x2 = (int)roundf(roi[3] * spatial_scale);
    const int y2 = (int)roundf(roi[4] * spatial_scale);
    const int rw = imax(x2 - x1 + 1, 1);
    const int rh = imax(y2 - y1 + 1, 1);
    const float bin_h = (float)rh / (float)pooled_h;
    const float bin_w = (float)rw / (float)pooled_w;
    for (int c = 0; c < channels; ++c) {
      const float* plane = input + ((size_t)b * channels + c) * height * width;
      for (int ph = 0; ph < pooled_h; ++ph) {
        int hs = (int)floorf((float)ph * bin_h);
        int he = (int)ceilf((float)(ph + 1) * bin_h);
        hs = imin(imax(hs + y1, 0), height);
        he = imin(imax(he + y1, 0), height);
        for (int pw = 0; pw < pooled_w; ++pw) {
          int ws = (int)floorf((float)pw * bin_w);
          int we = (int)ceilf((float)(pw + 1) * bin_w);
          ws = imin(imax(ws + x1, 0), width);
          we = imin(imax(we + x1, 0), width);
          const int empty = (he <= hs) || (we <= ws);
          float best = empty ? 0.f : -FLT_MAX;
          int besti = -1;
          for (int h = hs; h < he; ++h)
            for (int w = ws; w < we; ++w) {
              const float v = plane[h * width + w];
              if (v > best) { best = v; besti = h * width + w; }
            }
          const size_t o = (((size_t)n * channels + c) * pooled_h + ph) * pooled_w + pw;
          output[o] = best;
          if (argmax) argmax[o] = besti;
        }
      }
    }
  }
}

/* grad_input must be zero-initialised by the caller; deterministic sequential scatter-add. */
void oracle_roi_pool_backward(const float* grad_out, const float* rois, const int32_t* argmax,
                              int num_rois, int channels, int height, int width, int pooled_h,
                              int pooled_w, float* grad_input) {
  for (int n = 0; n < num_rois; ++n) {
    const int b = (int)rois[5 * n];
    for (int c = 0; c < channels; ++c) {
      float* gplane = grad_input + ((size_t)b * channels + c) * height * width;
      for (int p = 0; p < pooled_h * pooled_w; ++p) {
        const size_t o = ((size_t)n * channels + c) * pooled_h * pooled_w + p;
        const int a = argmax[o];
        if (a >= 0) gplane[a] += grad_out[o];
      }
    }
  }
}

/* ------------------------------------------------------------------ ROIAlign */
typedef struct { int p1, p2, p3, p4; float w1, w2, w3, w4; } precalc_t;

/* ROIAlign_cpu.cpp:20-114 */
static void precalc_bilinear(int height, int width, int pooled_h, int pooled_w, int grid_h,
                             int grid_w, float start_h, float start_w, float bin_h, float bin_w,
                             precalc_t* pc) {
  int idx = 0;
  for (int ph = 0; ph < pooled_h; ph++)
    for (int pw = 0; pw < pooled_w; pw++)
      for (int iy = 0; iy < grid_h; iy++) {
        const float yy = start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)grid_h;
        for (int ix = 0; ix < grid_w; ix++) {
          const float xx = start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)grid_w;
          float x = xx, y = yy;
          precalc_t q;
          if (y < -1.0 || y > height || x < -1.0 || x > width) {
            memset(&q, 0, sizeof q);
            pc[idx++] = q;
            continue;
          }
          if (y <= 0) y = 0;
          if (x <= 0) x = 0;
          int y_low = (int)y, x_low = (int)x, y_high, x_high;
          if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; }
          else y_high = y_low + 1;
          if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; }
          else x_high = x_low + 1;
          const float ly = y - y_low, lx = x - x_low;
          const float hy = 1.f - ly, hx = 1.f - lx;
          q.p1 = y_low * width + x_low;  q.p2 = y_low * width + x_high;
          q.p3 = y_high * width + x_low; q.p4 = y_high * width + x_high;
          q.w1 = hy * hx; q.w2 = hy * lx; q.w3 = ly * hx; q.w4 = ly * lx;
          pc[idx++] = q;
        }
      }
}

static void roi_geometry(const float* roi, float scale, int aligned, int pooled_h, int pooled_w,
                         int sampling_ratio, float* start_h, float* start_w, float* bin_h,
                         float* bin_w, int* grid_h, int* grid_w) {
  const float off = aligned ? 0.5f : 0.0f;
  const float sw = roi[1] * scale - off, sh = roi[2] * scale - off;
  const float ew = roi[3] * scale - off, eh = roi[4] * scale - off;
  float rw = ew - sw, rh = eh - sh;
  if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  *start_h = sh; *start_w = sw;
  *bin_h = rh / (float)pooled_h;
  *bin_w = rw / (float)pooled_w;
  *grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / pooled_h);
  *grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / pooled_w);
}

/* ROIAlign_cpu.cpp:116-218 */
void oracle_roi_align_forward(const float* input, const float* rois, int num_rois, int channels,
                              int height, int width, int pooled_h, int pooled_w,
                              float spatial_scale, int sampling_ratio, int aligned,
                              float* output) {
#pragma omp parallel for schedule(dynamic, 8)
  for (int n = 0; n < num_rois; ++n) {
    const float* roi = rois + 5 * n;
    const int b = (int)roi[0];
    float sh, sw, bh, bw; int gh, gw;
    roi_geometry(roi, spatial_scale, aligned, pooled_h, pooled_w, sampling_ratio, &sh, &sw, &bh,
                 &bw, &gh, &gw);
    const float count = (float)imax(gh * gw, 1);
    const int npc = imax(gh * gw, 0) * pooled_h * pooled_w;
    precalc_t* pc = (precalc_t*)malloc(sizeof(precalc_t) * (size_t)imax(npc, 1));
    precalc_bilinear(height, width, pooled_h, pooled_w, gh, gw, sh, sw, bh, bw, pc);
    for (int c = 0; c < channels; ++c) {
      const float* plane = input + ((size_t)b * channels + c) * height * width;
      int k = 0;
      for (int p = 0; p < pooled_h * pooled_w; ++p) {
        float acc = 0.f;
        for (int s = 0; s < gh * gw; ++s) {
          const precalc_t q = pc[k++];
          acc += q.w1 * plane[q.p1] + q.w2 * plane[q.p2] + q.w3 * plane[q.p3] + q.w4 * plane[q.p4];
        }
        output[((size_t)n * channels + c) * pooled_h * pooled_w + p] = acc / count;
      }
    }
    free(pc);
  }
}

/* ROIAlign_cpu.cpp:220-395; grad_input zero-initialised by the caller. */
void oracle_roi_align_backward(const float* grad_out, const float* rois, int num_rois,
                               int channels, int height, int width, int pooled_h, int pooled_w,
                               float spatial_scale, int sampling_ratio, int aligned,
                               float* grad_input) {
  for (int n = 0; n < num_rois; ++n) {
    const float* roi = rois + 5 * n;
    const int b = (int)roi[0];
    float sh, sw, bh, bw; int gh, gw;
    roi_geometry(roi, spatial_scale, aligned, pooled_h, pooled_w, sampling_ratio, &sh, &sw, &bh,
                 &bw, &gh, &gw);
    const float count = (float)(gh * gw);
    const int npc = imax(gh * gw, 0) * pooled_h * pooled_w;
    precalc_t* pc = (precalc_t*)malloc(sizeof(precalc_t) * (size_t)imax(npc, 1));
    precalc_bilinear(height, width, pooled_h, pooled_w, gh, gw, sh, sw, bh, bw, pc);
    for (int c = 0; c < channels; ++c) {
      float* gplane = grad_input + ((size_t)b * channels + c) * height * width;
      int k = 0;
      for (int p = 0; p < pooled_h * pooled_w; ++p) {
        const float g = grad_out[((size_t)n * channels + c) * pooled_h * pooled_w + p];
        for (int s = 0; s < gh * gw; ++s) {
          const precalc_t q = pc[k++];
          gplane[q.p1] += g * q.w1 / count;
          gplane[q.p2] += g * q.w2 / count;
          gplane[q.p3] += g * q.w3 / count;
          gplane[q.p4] += g * q.w4 / count;
        }
      }
    }
    free(pc);
  }
}

/* ------------------------------------------------------------------ NMS (Appendix C.2) */
typedef struct { float s; int64_t i; } sitem_t;
static int cmp_desc_stable(const void* a, const void* b) {
  const sitem_t* x = (const sitem_t*)a; const sitem_t* y = (const sitem_t*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return (x->i > y->i) - (x->i < y->i); /* stable: lower original index first */
}

/* returns number kept; keep[] holds original indices in descending-score order */
int64_t oracle_nms(const float* boxes, const float* scores, int64_t n, float thr, int64_t* keep) {
  if (n <= 0) return 0;
  sitem_t* order = (sitem_t*)malloc(sizeof(sitem_t) * (size_t)n);
  uint8_t* dead = (uint8_t*)calloc((size_t)n, 1);
  float* area = (float*)malloc(sizeof(float) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    order[i].s = scores[i]; order[i].i = i;
    area[i] = (boxes[4 * i + 2] - boxes[4 * i]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
  }
  qsort(order, (size_t)n, sizeof(sitem_t), cmp_desc_stable);
  int64_t nk = 0;
  for (int64_t a = 0; a < n; ++a) {
    const int64_t i = order[a].i;
    if (dead[i]) continue;
    keep[nk++] = i;
    const float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
    for (int64_t bb = a + 1; bb < n; ++bb) {
      const int64_t j = order[bb].i;
      if (dead[j]) continue;
      const float xx1 = fmaxf(ix1, boxes[4 * j]), yy1 = fmaxf(iy1, boxes[4 * j + 1]);
      const float xx2 = fminf(ix2, boxes[4 * j + 2]), yy2 = fminf(iy2, boxes[4 * j + 3]);
      const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
      const float inter = w * h;
      const float ovr = inter / (area[i] + area[j] - inter);
      if (ovr > thr) dead[j] = 1;
    }
  }
  free(order); free(dead); free(area);
  return nk;
}

/* Appendix C.3 (torchvision 0.6 batched_nms): offset boxes by class * (max coord + 1). */
int64_t oracle_batched_nms(const float* boxes, const float* scores, const int64_t* idxs,
                           int64_t n, float thr, int64_t* keep) {
  if (n <= 0) return 0;
  float mx = boxes[0];
  for (int64_t i = 1; i < 4 * n; ++i) mx = fmaxf(mx, boxes[i]);
  float* ob = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    const float off = (float)idxs[i] * (mx + 1.f);
    for (int k = 0; k < 4; ++k) ob[4 * i + k] = boxes[4 * i + k] + off;
  }
  const int64_t nk = oracle_nms(ob, scores, n, thr, keep);
  free(ob);
  return nk;
}
