/* ORACLE (test infrastructure only) - CPU restatement of the reference's CSC op.
 *
 * Restates projects/WSL/wsl/layers/csrc/csc/csc_cuda.cu:
 *   binary_and_integral_cpu  :132-161   thresholded CPG map -> summed-area table (fp32 counts)
 *   CSCPool                  :184-350   per-ROI  frame / context contrast on that table
 *   csc_forward_cuda         :352-552   per labelled class: table, CSCPool, max/min normalisation to [-1, 1],
 *                                       blend with the image-level prediction
 * The reference has NO CPU implementation of this op (csc.h dispatches to CUDA only), so it cannot be run in the build
 * container: PARITY UNPINNED for this file.  It is pinned indirectly by (a) a brute-force check that every box sum
 * read from the table equals the directly counted sum (tests/test_oracle_golden.py) and (b) the model-level golden, in
 * which the unmodified reference CSCROIHeads runs with this function standing in for `_C.csc_forward`.
 *
 * Arithmetic notes (csc_cuda.cu is instantiated with T = float):
 *   - `1.0 * x / context_scale`, `x / 2.0`, `max(.., 0.0)`, `min(.., width_im - 1.0)` are DOUBLE expressions there
 *     (double literals), their results are stored to float variables or passed to round(double);
 *   - round() is half-away-from-zero; sqrt / division on float are correctly rounded in CUDA's default mode.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static const float kMIN_SCORE = (float)(-1.0 * 1e20);

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* csc_cuda.cu:132-161 */
void oracle_csc_integral(const float* src, float* sum, int height, int width, float threshold) {
  float s = 0;
  for (int x = 0; x < width; x++) {
    s += (src[x] >= threshold) ? 1.f : 0.f;
    sum[x] = s;
  }
  for (int y = 1; y < height; y++) {
    const float* sr = src + (long)y * width;
    float* su = sum + (long)y * width;
    s = 0;
    for (int x = 0; x < width; x++) {
      s += (sr[x] >= threshold) ? 1.f : 0.f;
      su[x] = su[x - width] + s;
    }
  }
}

static float box_sum(const float* t, int width, int hs, int ws, int he, int we) {
  const float a1 = t[(long)he * width + we];
  const float a2 = (ws - 1 >= 0) ? t[(long)he * width + (ws - 1)] : 0.f;
  const float a3 = (hs - 1 >= 0) ? t[(long)(hs - 1) * width + we] : 0.f;
  const float a4 = (hs - 1 >= 0 && ws - 1 >= 0) ? t[(long)(hs - 1) * width + (ws - 1)] : 0.f;
  return a1 - a2 - a3 + a4;
}

/* csc_cuda.cu:184-350: one ROI.  out_boxes (optional, 12 ints): the three integer boxes, for the brute-force pin */
float oracle_csc_pool_one(const float* table, int height_im, int width_im, const float* roi, int area_sqrt,
                          float context_scale, int* out_boxes) {
  int wstart = (int)roundf(roi[1]), hstart = (int)roundf(roi[2]), wend = (int)roundf(roi[3]), hend = (int)roundf(roi[4]);
  wstart = clampi(wstart, 0, width_im - 1);
  hstart = clampi(hstart, 0, height_im - 1);
  wend = clampi(wend, 0, width_im - 1);
  hend = clampi(hend, 0, height_im - 1);
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
  if (out_boxes) {
    const int b[12] = {hstart, wstart, hend, wend, hstart_inner, wstart_inner, hend_inner, wend_inner,
                       hstart_outer, wstart_outer, hend_outer, wend_outer};
    for (int i = 0; i < 12; ++i) out_boxes[i] = b[i];
  }
  const float sum_roi = box_sum(table, width_im, hstart, wstart, hend, wend);
  const float sum_inner = box_sum(table, width_im, hstart_inner, wstart_inner, hend_inner, wend_inner);
  const float sum_outer = box_sum(table, width_im, hstart_outer, wstart_outer, hend_outer, wend_outer);
  const float area_roi = height_roi * width_roi;
  const float area_inner = height_roi_inner * width_roi_inner;
  const float area_outer = height_roi_outer * width_roi_outer;
  const float area_frame = fmaxf(area_roi - area_inner, 1.f);
  const float area_context = fmaxf(area_outer - area_roi, 1.f);
  const float sum_frame = sum_roi - sum_inner;
  const float sum_context = sum_outer - sum_roi;
  if (area_sqrt) return sum_frame / sqrtf(area_frame) - sum_context / sqrtf(area_context);
  return sum_frame / area_frame - sum_context / area_context;
}

/* csc_cuda.cu:352-552.  cpgs [B][K][H][W], labels / preds [B][K], rois [R][5] -> W [R][K] (ones where unlabelled).
 * As in the reference every labelled (b, c) pair scores ALL R rois against image b's map (it is run with B = 1). */
void oracle_csc_forward(const float* cpgs, const float* labels, const float* preds, const float* rois, int B, int K,
                        int H, int Wd, int R, float fg_threshold, int area_sqrt, float context_scale, float* W) {
  for (long i = 0; i < (long)R * K; ++i) W[i] = 1.f;
  float* table = (float*)malloc(sizeof(float) * (size_t)H * (size_t)Wd);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < K; ++c) {
      const int li = b * K + c;
      const float label_value = labels[li], pred_value = preds[li];
      if (label_value < 0.5f) continue;
      const float max_val = 1.f;
      oracle_csc_integral(cpgs + (long)li * H * Wd, table, H, Wd, max_val * fg_threshold);
      for (int r = 0; r < R; ++r)
        W[(long)r * K + c] = oracle_csc_pool_one(table, H, Wd, rois + 5 * (long)r, area_sqrt, context_scale, 0);
      float max_value = 0, min_value = 0;
      for (int r = 0; r < R; ++r) {
        const float v = W[(long)r * K + c];
        if (v > max_value) max_value = v;
        if (v < min_value && v != kMIN_SCORE) min_value = v;
      }
      if (max_value > 0 && min_value < 0) {
        for (int r = 0; r < R; ++r) {
          float v = W[(long)r * K + c];
          v = (v == kMIN_SCORE) ? -1.f : (v > 0 ? v / max_value : v / (-min_value));
          W[(long)r * K + c] = v;
        }
      } else if (max_value > 0 && min_value == 0) {
        for (int r = 0; r < R; ++r) {
          float v = W[(long)r * K + c];
          v = (v == kMIN_SCORE) ? -1.f : v / max_value;
          W[(long)r * K + c] = v;
        }
      } else {
        for (int r = 0; r < R; ++r) W[(long)r * K + c] = 1.0f;
      }
      for (int r = 0; r < R; ++r) {
        const float a = pred_value * W[(long)r * K + c];
        const float bb = (1 - pred_value) * 1;
        W[(long)r * K + c] = a + bb;
      }
    }
  free(table);
}
