/*
 * drn_wsod.h — C ABI of the MI355X-native DRN-WSOD / OICR hot path (libdrn_wsod_hip.so).
 *
 * Boundary contract:
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless a name ends in _host;
 *   - every entry point enqueues work on the given hipStream_t (passed as void*), never
 *     synchronises, never allocates, retains no pointers after returning; graph-capturable;
 *   - returns 0 (DRN_OK) or a negative code: -1 invalid argument, -2 launch failure,
 *     -3 unsupported; never throws across the ABI;
 *   - dtype codes: 0 = fp32 (parity mode, exact fp32 MFMA), 1 = bf16 (fast mode, fp32 accumulate);
 *   - feature maps are NHWC, matrices row-major with an explicit leading dimension in ELEMENTS.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference
 * repository root).  INTEGRATION.md shows the Python-side binding a maintainer would add.
 */
#ifndef DRN_WSOD_H_
#define DRN_WSOD_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRN_OK 0
#define DRN_ERR_ARG (-1)
#define DRN_ERR_LAUNCH (-2)
#define DRN_ERR_UNSUPPORTED (-3)
#define DRN_F32 0
#define DRN_BF16 1

/* ---- backbone -------------------------------------------------------------------------------- */

/* GeneralizedRCNNWSL.preprocess_image, projects/WSL/wsl/modeling/meta_arch/rcnn.py:242-249 +
 * ImageList.from_tensors, detectron2/structures/image_list.py:57-119.
 * img_chw [C][H][W] f32 -> out_nhwc [Hp][Wp][Cp] (one image slot), (x-mean)/std, zero padded. */
int drn_preprocess_nhwc(const float* img_chw, int C, int H, int W, void* out_nhwc, int Hp, int Wp, int Cp,
                        const float* mean3_host, const float* std3_host, int dtype, void* stream);

/* ResizeTransform.apply_image on an 8-bit image (detectron2/data/transforms/transform.py:101-122: PIL.Image.resize(size,
 * BILINEAR)) [+ HFlipTransform], as DatasetMapperTTAAVG applies them 16 times per image
 * (projects/WSL/wsl/modeling/test_time_augmentation_avg.py:68-137), on the device.  Pillow (un-vendored, un-pinned dependency
 * of the reference; its published algorithm - Resample.c - is restated, and pinned bit for bit against the installed Pillow by
 * the tests): a horizontal then a vertical integer pass; output position i reads source positions [bounds[2i], bounds[2i] +
 * bounds[2i+1]) with coefficients coef[i * ks + j] scaled by 2^22, each pass rounds ((1 << 21) + sum) >> 22 and clips to 8 bits.
 * The caller computes bounds / coef with Pillow's double arithmetic (host side: ops.pil_bilinear_coeffs) and passes them as
 * DEVICE pointers; a NULL bounds pointer skips that pass (its size must not change).
 * src_hwc uint8 [H][W][C] (C <= 4), dst_chw fp32 [C][Ho][Wo] holding the integers 0 .. 255; flip != 0 mirrors the columns. */
int drn_resize_bilinear_u8(const void* src_hwc, int H, int W, int C, float* dst_chw, int Ho, int Wo, const int* xbounds,
                           const int* xcoef, int ksx, const int* ybounds, const int* ycoef, int ksy, int flip, void* stream);

/* Conv2d.forward = F.conv2d -> FrozenBatchNorm2d -> relu_ [+ residual add before the relu],
 * detectron2/layers/wrappers.py:94-99, detectron2/layers/batch_norm.py:45-65,
 * projects/WSL/wsl/modeling/backbone/resnet_ws.py:217-237, vgg.py:104-122.
 * x NHWC [Nb][H][W][Cin]; w [Cout][ldw] with k = (kh*KW + kw)*Cin + ci, rows zero-padded to a
 * multiple of 128 bytes; y [Nb*Ho*Wo][ldy]; y = act(conv*scale[c] + bias[c] (+ residual)). */
int drn_conv2d_nhwc(const void* x, const void* w, void* y, const float* scale, const float* bias,
                    const void* residual, int Nb, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                    int pad, int dil, long ldw, long ldy, long ldres, int relu, int dtype, void* stream);

/* The same convolution for the QUANTISED trunk (BASELINE configs[4], "fp8 MFMA conv path"; the reference is fp32 only -
 * SURVEY F5 - so the definition of record is the fp32 conv above with operands and activations rounded to fp8):
 * `dtype` (DRN_F32 / DRN_BF16 / DRN_FP8 = OCP e4m3fn, one byte) is the element type of x and w - DRN_FP8 runs
 * v_mfma_f32_32x32x16_fp8_fp8 with fp32 accumulation - while y is stored as `out_dtype` and the residual read as
 * `res_dtype` and multiplied by `res_mult`.  Per-tensor / per-output-channel quantisation scales live in the caller's
 * affine: with x stored as x*s_x, w[c] as w[c]*s_w[c], y as y*s_y and the residual as r*s_r the caller passes
 * scale[c] = bn_scale[c]*s_y/(s_x*s_w[c]), bias[c] = bn_bias[c]*s_y, res_mult = s_y/s_r (s_y = 1 for a bf16/fp32 y).
 * fp8 stores round to nearest even and saturate at +-448.  Cin*esize must be a multiple of 16 bytes. */
#define DRN_FP8 2
int drn_conv2d_nhwc_q(const void* x, const void* w, void* y, const float* scale, const float* bias,
                      const void* residual, int Nb, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                      int pad, int dil, long ldw, long ldy, long ldres, int relu, int dtype, int out_dtype,
                      int res_dtype, float res_mult, void* stream);

/* The tail of a 64-channel bottleneck block on a large map as ONE launch (BottleneckBlock.forward,
 * projects/WSL/wsl/modeling/backbone/resnet_ws.py:217-237: conv2 -> relu -> conv3 -> + shortcut -> relu [-> max pool]):
 * y = pool(act3((conv1x1(act2(conv3x3(x) * scale2 + bias2)) * scale3 + bias3) + residual * res_mult)), bf16;
 * x [Nb][H][W][64], w2 [64][ldw2] (3x3, pad = dil = stride = 1, packed as for drn_conv2d_nhwc), w3 [256][ldw3] (1x1 over
 * 64 channels), residual [Nb*H*W][256] or NULL, y [Nb*H*W][256].  The 3x3's output - rounded to bf16 exactly as
 * drn_conv2d_nhwc would store it - stays in LDS: bit for bit the result of the separate calls.
 * w3 == NULL: no 1x1 stage (the deep stem's last 3x3: y and the residual have 64 channels; needs pool).
 * pool != 0: nn.MaxPool2d(2, 2) (resnet_ws.py:214-215) applied in the epilogue - y is the pooled map
 * [Nb][(H - 2) / 2 + 1][(W - 2) / 2 + 1][channels], the full-resolution output is never written; needs the last ReLU.
 * Maps of >= 32768 pixels per image (the LDS-resident-patch kernel's class); DRN_ERR_UNSUPPORTED otherwise (callers then
 * run the separate launches). */
int drn_conv3x3_pw_nhwc(const void* x, const void* w2, const float* scale2, const float* bias2, int relu2, const void* w3,
                        const float* scale3, const float* bias3, const void* residual, void* y, int Nb, int H, int W,
                        long ldw2, long ldw3, float res_mult, int relu3, int pool, void* stream);

/* nn.MaxPool2d(kernel_size=2, stride=s, padding=0), resnet_ws.py:214-215,403; vgg.py:99-100.  DRN_FP8: non-negative
 * (post-ReLU) values only - they order like their bytes. */
int drn_maxpool2x2_nhwc(const void* x, void* y, int Nb, int H, int W, int C, int stride, int dtype, void* stream);

/* ---- backward of the conv trunk (MODEL.BACKBONE.FREEZE_AT < 5; torch.autograd of F.conv2d / max_pool2d) ---- *
 * conv dgrad runs drn_conv2d_nhwc on the flipped/transposed packed weights (every trained conv has stride 1,
 * resnet_ws.py:148-150), the FrozenBN-affine/ReLU backward runs drn_bias_act_bwd, and the weight gradient is
 * drn_gemm_nt(g^T, im2col_t): */

/* out[(ci*KH + kh)*KW + kw][p] = x[n, ho*s + kh*d - pad, wo*s + kw*d - pad, ci]  (x NHWC with channel stride ldc,
 * p = (n*Ho + ho)*Wo + wo, zeros outside the image; columns beyond p are left untouched). */
int drn_im2col_t(const void* x, void* out, int Nb, int H, int W, int Cin, int ldc, int KH, int KW, int stride, int pad,
                 int dil, long ld_out, int dtype, void* stream);

/* d(x) of nn.MaxPool2d(2, stride, 0): gradient goes to each window's first maximum (torch semantics). */
int drn_maxpool2x2_bwd_nhwc(const void* x, const void* dy, void* dx, int Nb, int H, int W, int C, int stride, int dtype,
                            void* stream);

/* out = a + b (gradient fan-in of a residual block). */
int drn_add(const void* a, const void* b, void* out, long n, int dtype, void* stream);

/* Hand-over from the staged NEXT batch to the buffers the heads read, one launch in front of the pooling kernel:
 * props[M][4] <- rois[M][1:5] (the proposal boxes `label_and_sample_proposals` / `get_pgt` work on,
 * roi_heads_oicr.py:320-421: the reference keeps them in the Instances it carries along) and, when n_words > 0,
 * words_dst[0:n_words] <- words_src (the image-level label block of `get_image_level_gt`, roi_heads_oicr.py:287-318).
 * Replaces two tensor copies of the host framework on the launch stream. */
int drn_stage_heads_inputs(const float* rois, float* props, int M, const int* words_src, int* words_dst, int n_words,
                           void* stream);

/* convert_boxes_to_pooler_format for ONE image (detectron2/modeling/poolers.py:69-96) in one launch: boxes [M][4] (+ objectness
 * logits [M] or NULL) -> rois [M][5] = (batch_index, x0, y0, x1, y1), obj [M] (NULL iff logits is NULL) and, when props != NULL, a
 * contiguous copy of the boxes for the pseudo-GT mining / box decoding (roi_heads_oicr.py:320-421 reads `proposals` there). */
int drn_stage_rois(const float* boxes, const float* logits, float batch_index, float* rois, float* obj, float* props, int M,
                   void* stream);

/* ---- region pooling -------------------------------------------------------------------------- */

/* ROIPooler.forward single-level path, detectron2/modeling/poolers.py:191-226:
 * mode 0 = torchvision RoIPool (poolers.py:162-165),
 * mode 1 = ROIAlign (detectron2/layers/csrc/ROIAlign/ROIAlign.h:54-128 roi_align_forward),
 * fused with `box_features * (objectness_logits + 1)` (roi_heads_oicr.py:342-343) when objectness != NULL.
 * feat NHWC; rois [M][5] = (batch_idx, x0, y0, x1, y1); out [M][ld_out] with column
 * c*P*P + ph*P + pw (the NCHW flatten order of box_head.py:85-86); argmax [M][C*P*P] or NULL.
 * out_t (optional, same dtype as out): the transposed copy [C*P*P][ld_out_t] (column = roi) that the fc6 dW GEMM
 * consumes as its K-major operand; written by the same launch when the feature map fits in LDS. */
int drn_roi_pool_nhwc(const void* feat, const float* rois, const float* objectness, void* out, void* out_t,
                      int32_t* argmax, int N, int H, int W, int C, int P, int M, float spatial_scale, long ld_out,
                      long ld_out_t, int mode, int sampling_ratio, int aligned, int in_dtype, int out_dtype,
                      void* stream);

/* drn_roi_pool_nhwc with a hint for out_t: only the rows of channels >= t_first_channel (row index c*P*P + bin) are
 * needed - the fc6 weight gradient reads `out` itself through drn_gemm_tn and keeps a transposed copy only for the few
 * trailing columns its tail-balancing launch takes (drn_gemm_nt_main_cols).  The 64-ROI training kernel skips the A^T
 * store loop of the other channel chunks; every other kernel writes all of out_t (a superset). */
int drn_roi_pool_nhwc_t(const void* feat, const float* rois, const float* objectness, void* out, void* out_t,
                        int32_t* argmax, int N, int H, int W, int C, int P, int M, float spatial_scale, long ld_out,
                        long ld_out_t, int mode, int sampling_ratio, int aligned, int in_dtype, int out_dtype,
                        int t_first_channel, void* stream);

/* drn_roi_pool_nhwc_t with a caller-owned scratch buffer (round 5).  Feature maps whose 8-channel slice leaves one chunk per
 * workgroup (43x58 cells and more: test-time scales, real-size training images; poolers.py:191-226 at those sizes) are first
 * copied chunk-major - [N][C/8][H*W] cells of 16 bytes - into `workspace`, so that the pooling kernel stages contiguous runs
 * instead of 16 bytes of every pixel's line (50x76 / R = 2000: 156 -> ~130 us).  Round 6: large maps with enough ROIs (the rule
 * is DRN_TUNE_ROI_ST's) are pooled from a SPARSE TABLE of block maxima built in LDS out of that copy - four table cells per bin
 * instead of every cell of its window (the stride-8 dilated-C5 map of an 800x1216 image, R = 2000: 929 -> 314 us; 1200x1600:
 * 4433 -> 677 us); the workspace then also holds a 256-byte record and a class byte per ROI.  drn_roi_pool_workspace_bytes
 * returns the size that helps for a shape (0: no workspace is used); workspace NULL or smaller: exactly drn_roi_pool_nhwc_t.
 * The workspace is read and written on `stream` only and holds nothing between calls; results are bit-identical with and
 * without it. */
long drn_roi_pool_workspace_bytes(int N, int H, int W, int C, int P, int M, int mode, int has_argmax, int in_dtype,
                                  int out_dtype);
int drn_roi_pool_nhwc_ws(const void* feat, const float* rois, const float* objectness, void* out, void* out_t,
                         int32_t* argmax, int N, int H, int W, int C, int P, int M, float spatial_scale, long ld_out,
                         long ld_out_t, int mode, int sampling_ratio, int aligned, int in_dtype, int out_dtype,
                         int t_first_channel, void* workspace, long workspace_bytes, void* stream);


/* Backward of the above w.r.t. the feature map: torchvision RoIPool's backward (scatter to the arg-max the forward
 * returned) and roi_align_backward (detectron2/layers/csrc/ROIAlign/ROIAlign.h:93-128, ROIAlign_cuda.cu:141-250),
 * with the forward's objectness scaling applied to grad_out.  grad_out [M][ld_g] (column c*P*P + bin, fp32 or bf16);
 * dfeat [N][H][W][C] fp32 is zeroed and accumulated with fp32 atomics like the reference's CUDA kernels. */
int drn_roi_pool_backward_nhwc(const void* grad_out, const float* rois, const float* objectness, const int32_t* argmax,
                               float* dfeat, int N, int H, int W, int C, int P, int M, float spatial_scale, long ld_g,
                               int mode, int sampling_ratio, int aligned, int grad_dtype, void* stream);

/* out[c][r] = cast(in[r][c]) — builds the K-major operands of the dW GEMMs. */
int drn_transpose2d(const void* in, void* out, int rows, int cols, long ld_in, long ld_out, int in_dtype,
                    int out_dtype, void* stream);

/* out[r][c] = cast(in[r][c]) with independent leading dimensions (padded compute shadows of weights). */
int drn_cast2d(const void* in, void* out, int rows, int cols, long ld_in, long ld_out, int in_dtype, int out_dtype,
               void* stream);

/* ---- dense contractions ---------------------------------------------------------------------- */

/* Which output columns drn_gemm_nt's persistent 256x256 launch keeps for an [M, N] output: [0, n0); the columns from n0 on
 * are peeled into a small-tile launch first (tail balancing, DRN_TUNE_GEMM_TAIL_SPLIT).  n0 == N when nothing is peeled.
 * Lets a caller that issues several row slabs of one product (the fc6 weight gradient: F.linear's dW, box_head.py:82-91)
 * compute ALL slabs' peeled columns in one launch and hand the slabs' main columns - exact rounds - to drn_gemm_nt. */
long drn_gemm_nt_main_cols(int M, int N, int splits);

/* nn.Linear / F.linear and its autograd (fc6/fc7 of box_head.py:82-91, predictors of
 * fast_rcnn.py:453-461,1316-1327).  C[s][M][N] (fp32) = A[M][K] * B[N][K]^T over K-split s.
 * K*esize must be a multiple of 128 bytes (callers zero-pad K), lda/ldb multiples of 16 bytes.
 * c_dtype: DRN_F32, or DRN_BF16 (splits == 1, no accumulate) for weight-gradient buckets that are exchanged
 * between GPUs in bf16. */
int drn_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, int dtype,
                int c_dtype, int splits, long c_split_stride, int accumulate, void* stream);

/* drn_gemm_nt_pair: two independent drn_gemm_nt problems (bf16 operands, fp32 outputs; splits / accumulate per problem
 * as there) in ONE persistent launch of the 256x256 kernel - the workgroups of every XCD are divided between the two
 * problems in proportion to their work.  Serves pairs of launches that each leave CUs idle: the fc7 weight gradient and
 * the fc7 input gradient of the explicit backward (torch.autograd of box_head.py:82-91's fc2: dW = dY^T X, dX = dY W,
 * both from dY).  Bit-identical to the two separate calls. */
int drn_gemm_nt_pair(const void* A0, const void* B0, void* C0, int M0, int N0, int K0, long lda0, long ldb0, long ldc0,
                     int splits0, long stride0, int accumulate0, const void* A1, const void* B1, void* C1, int M1, int N1,
                     int K1, long lda1, long ldb1, long ldc1, int splits1, long stride1, int accumulate1, void* stream);

/* drn_gemm_tn: C[M,N] = A[M,K] . Bt[K,N] with the second operand given K-MAJOR (Bt row-major [kb_rows][ldb]; rows
 * kb_rows..K-1 - the K padding - are read as zeros and need not exist).  bf16 operands, fp32 accumulate, C fp32
 * (splits / accumulate as drn_gemm_nt) or bf16.  The fc6 weight gradient dW = dP1^T . A (the autograd of
 * box_head.py:82-91's fc1) reads the pooled matrix A [R][C*49] through it directly, so the pooling launch no longer
 * writes a transposed copy A^T.  256x256 ping-pong kernel with transposing LDS reads (ds_read_b64_tr_b16); results are
 * bit-identical to drn_gemm_nt on a materialised transpose.  K % 64 == 0, (ldb * 2) % 16 == 0, ldb >= N. */
int drn_gemm_tn(const void* A, const void* Bt, void* C, int M, int N, int K, int kb_rows, long lda, long ldb, long ldc,
                int c_dtype, int splits, long c_split_stride, int accumulate, void* stream);

/* The same pair of reference operations (dW = dY^T X of box_head.py:82-91's fc1 through torch.autograd, then
 * torch.optim.SGD.step on it: detectron2/solver/build.py:93-137, projects/WSL/tools/train_net.py:104-113) in the TN operand
 * form of drn_gemm_tn and with the gradient rounded to bf16 exactly as the unfused pair does (drn_gemm_tn into a bf16 bucket,
 * then drn_sgd_step / drn_sgd_step_block reading it): G = A[M][K] . Bt[K][N] is written to grad_bucket (bf16, [M][ldc]) and
 *   d = G*grad_scale + wd*W;  buf = first_step ? d : momentum*buf + d;  W -= lr*buf;  shadow = bf16(W)
 * is applied by the SAME launch: every workgroup of the persistent 256x256 kernel updates the tile it finished last while it
 * multiplies the next one (one 8-row chunk per K slab, loads / stores interleaved with the LDS-DMA pipeline), so the
 * optimizer's HBM traffic is a steady stream under the MFMA work and the gradient is read back from L2.  Bit-identical to
 * the unfused pair.  Shape class: K >= 128 with K % 64 == 0 (with fewer than 32 K slabs - fewer than 1985 proposals - the
 * chunks of a tile's update that find no slab follow the mainloop, exposed), M, N multiples of 256, a bf16 shadow, 16-byte
 * aligned pointers and pitches, more tiles than compute units; DRN_ERR_UNSUPPORTED for everything outside it (callers then run
 * the unfused pair), DRN_ERR_ARG only for null / negative arguments. */
int drn_gemm_tn_sgd(const void* A, const void* Bt, void* grad_bucket, int M, int N, int K, int kb_rows, long lda, long ldb,
                    long ldc, float* weights, float* momentum_buf, void* shadow, long ld_w, const void* seg_dev,
                    float momentum, int first_step, float grad_scale, void* stream);

/* tuning / test hook: pin the GEMM tile to 64, 128 or 256 (0 = heuristic); returns the previous setting. */
int drn_gemm_set_tile(int tile);

/* tuning knobs for A/B measurements and tests (defaults = the measured best); returns the previous value, -1 for an
 * unknown knob.  DRN_TUNE_GEMM_PERSISTENT (1): 0/1 - 256x256 GEMM launches with more (tile, K-split) work items than CUs
 * run as ONE resident workgroup per CU that loops over its share (default 1; same arithmetic, bit-identical results).
 * DRN_TUNE_SGD_GRID (2): workgroups (x) of the optimizer kernel (default 512).
 * DRN_TUNE_GEMM_GROUP_ROWS (3): tile rows per group of the 256x256 GEMM's XCD patch mapping (0 = heuristic). */
#define DRN_TUNE_GEMM_PERSISTENT 1
#define DRN_TUNE_SGD_GRID 2
#define DRN_TUNE_GEMM_GROUP_ROWS 3
#define DRN_TUNE_CONV_KSPLIT 5 /* 0/1: 32x32 wave-K-split conv kernel for latency-bound small-map layers (default 1) */
#define DRN_TUNE_ROI_MAP64 4 /* 64-ROI x 8-channel whole-map ROIPool for the bf16 (A, A^T) pair: 0 = off, else threads per workgroup (256 / 512 / 1024, default 512) */
#define DRN_TUNE_GEMM_TAIL_SPLIT 6 /* 0/1: a persistent 256x256 launch whose last round would be < 3/8 full runs an exact number of rounds; the peeled tile columns go to the small-tile kernel first (default 1; bit-identical) */
#define DRN_TUNE_CONV_KS_TILES 7 /* largest number of 64x64 tiles of ONE image's layer that still runs on the wave-K-split conv kernel (0 = default: CUs / 4) */
#define DRN_TUNE_CONV_K2_TILES 8 /* largest number of 64x64 tiles of ONE image's layer that runs two K-groups per tile (conv_nhwc_k2_kernel); -1 = default (2 x CUs), 0 = off */
#define DRN_TUNE_ROI_CPB 10 /* 64-ROI ROIPool: most 8-channel chunks one workgroup walks (power of two, default 1; halved until two workgroups per CU remain): bin bounds / item table once per workgroup - faster stand-alone (4-8), slower inside the training step */
#define DRN_TUNE_ROI_PREFETCH 11 /* 0/1 (default 1): 64-ROI ROIPool keeps two map-slice buffers and fetches the next chunk's slice under the scan */
#define DRN_TUNE_GEMM_PINGPONG 12 /* 0/1 (default 1): bf16 256x256 GEMMs run the ping-pong mainloop - the two waves of a SIMD half a phase apart, four [reads + DMA | 8 MFMAs] phases per K slab, half-tile LDS-DMA spread over the slab; bit-identical to the lock-step pipeline it replaces (0) */
#define DRN_TUNE_ROI_LDS_KB 15 /* 60..154 (default 154): LDS a 64-ROI pooling block may take for map slice + tile; 76 stages larger maps in row bands so that two blocks share a CU */
#define DRN_TUNE_ROI_MAP64_A 14 /* 0/1 (default 0): the 64-ROI ROIPool kernel also when only A is asked for (no A^T) */
#define DRN_TUNE_FP8_K64 13 /* 0/1 (default 1): fp8 convolutions multiply with v_mfma_scale_f32_32x32x64_f8f6f4 (unit scales; the fp8 MFMA rate) instead of the K = 16 non-scaled form (bf16 rate); same exact products, another fp32 summation order */
#define DRN_TUNE_GEMM_NWG 18 /* resident workgroups of persistent 256x256 launches (multiple of 8; 0 = default: one per CU) - for launches on a CU-masked stream */
#define DRN_TUNE_SGDP_EPILOGUE 20 /* 0/1 (default 1): the tile epilogue of drn_gemm_tn_sgd moves the bf16 gradient tile LDS -> global four 16-byte pieces per trip instead of one (A/B knob; bit-identical) */
#define DRN_TUNE_ROI_LANE_REPS 22 /* lane-per-bin ROIPool on maps that leave one block per CU: groups of 64 ROIs a block walks with one staged map slice (0 = default: 4, halved while fewer than two rounds of blocks would remain; 1 = a block per group) */
#define DRN_TUNE_MSM_WAVE 32 /* 0/1 (default 1): drn_mean_softmax for heads of <= 64 columns as a wave per row / lane per class (one coalesced load per head, the denominator summed in class order: bit-identical); 0 = the thread-per-row kernel */
#define DRN_TUNE_ROI_ST 31 /* RoIPool from a sparse table of block maxima (four table cells per bin instead of the window's cells; needs the workspace of drn_roi_pool_workspace_bytes): 0 = off, 1 (default) = where it beats the window kernels (maps of >= 1800 cells with >= 1500 ROIs, >= 3000 cells with >= 600 ROIs, and every map whose LDS slice holds only 4 channels per cell - the dilated-C5 stride-8 map of a real-size image - with >= 400 ROIs), 2 = every map whose slice fits (>= 64 ROIs) */
#define DRN_TUNE_ROI_LANE 19 /* 0/1 (default 1): the bf16 training operand A from the lane-per-bin ROIPool kernel (a wave per ROI, lane = bin: every channel leaves as one 98-byte run per store instruction); 0 = the 64-ROI kernel writes A */
#define DRN_TUNE_CONV_RING 23 /* register-ring conv kernels (conv_ring.hip; bf16, Cin % 64 == 0, layers beyond the latency-bound small maps): 0 = off (the tiles of gemm_conv.hip), 1 = default (64x64 tile, class by measurement), 64 / 128 = pin the 64x64 / 128x128 tile (any other value is ignored: drn_tune returns the unchanged setting); every tile gives the same bits as the 64x64 / 128x128 tiled kernel */
#define DRN_TUNE_CONV_PP 24 /* 1x1 / stride-1 bf16 convs of large maps on the 256x256 ping-pong GEMM mainloop with the conv epilogue (conv1x1_pp_kernel): 0 = off, 1 = default (Cout >= 256 and >= 192 tiles of 256x256 per image, or >= 100 tiles with K >= 1024), n > 1 = at least n tiles, any Cout; same bits as the tiled kernels */
#define DRN_TUNE_CONV_PATCH 9 /* 0 = never use the LDS-resident-patch kernel for 3x3 / 64 -> 64 channel convs; 1 = default (maps of >= 32768 pixels); > 1 = that many pixels per image at least */
#define DRN_TUNE_PP8 25 /* eight-wave ping-pong kernel (pp8.hip: 128x128 or 256x128 tile, two waves per SIMD half a phase apart; bf16 convs with Cin % 64 == 0 and drn_linear_act_fwd): 0 = off, 1 = default class (measured per layer shape, see drn_pp8_conv_try), 2 = every layer in the kernel's class; same bits as the tiled kernels */
#define DRN_TUNE_PP8_STAGES 26 /* 3 / 4 / 5 (default 5): 32-KB LDS stages of the 128x128 form's ring = 1 / 2 / 3 K slabs in flight (A/B knob; bit-identical) */
#define DRN_TUNE_PP8_VARIANT 27 /* schedule variant of that kernel (A/B knob; bit-identical): 0 = a slab's four DMA pieces in the fragment-read phase, 1 (default) = two there and two between the MFMAs, 2 = all between the MFMAs, + 4 = no s_setprio around the MFMAs, + 8 = profile build (shader-clock split of the mainloop) */
#define DRN_TUNE_PP8_PROFILE 28 /* any value: print (stderr) and clear the per-phase shader-clock sums the profile builds (DRN_TUNE_PP8_VARIANT + 8) accumulated for workgroup 0; returns 0 */
#define DRN_TUNE_PP8_WIDE_VARIANT 30 /* schedule variant of the 256x128 form (A/B knob; bit-identical): bit 0 = all six DMA pieces of a slab in the second fragment-read phase (else three there, three between the MFMAs), 4 = no s_setprio, 8 = profile build; default 4 */
#define DRN_TUNE_PP8_WIDE 29 /* the 256x128 form of that kernel (wave tile 64x64, three 48-KB stages): 0 = never, 1 = default (layers that give it >= 5/8 of the CUs' worth of tiles per image), 2 = always */
int drn_tune(int knob, int value);

/* relu_(fc(x)) + F.dropout(p) of DiscriminativeAdaptionNeck.forward (projects/WSL/wsl/modeling/roi_heads/box_head.py:82-91)
 * as ONE launch for bf16 operands: out [M][ld_out] (bf16) = dropout(relu(A [M][lda] . W [N][ldw]^T + bias)), and optionally
 * its transpose outT [N][ld_outT] (rows m >= M of outT are left untouched).  Same dropout rule as drn_bias_act_fwd (explicit
 * mask [M][N], else counter-based from seed (+ *seed_dev) when drop_p > 0; this entry never advances the counter); the
 * product is accumulated over K in one fp32 chain per element (= drn_gemm_nt with splits == 1 followed by
 * drn_bias_act_fwd, bit for bit).  K % 64 == 0, N % 8 == 0, 16-byte aligned rows; DRN_ERR_UNSUPPORTED outside that class
 * (the caller then runs the two-launch form). */
int drn_linear_act_fwd(const void* A, const void* W, const float* bias, const float* mask, unsigned long long seed,
                       const unsigned long long* seed_dev, float drop_p, void* out, long ld_out, void* outT, long ld_outT,
                       int M, int N, int K, long lda, long ldw, int relu, void* stream);

/* relu_(fc(x)) + F.dropout(p), box_head.py:88-90: sums split-K partials, adds bias, ReLU, dropout
 * (explicit multiplier mask [M][N] if given, else counter-based mask from seed (+ *seed_dev) when drop_p > 0; a launch
 * WITHOUT dropout (drop_p == 0, no mask) that is given seed_dev advances that counter: *seed_dev += seed - the heads'
 * logits pass does this behind the two dropout layers, so a replayed hipGraph draws fresh masks without a launch of its own);
 * writes out [M][ld_out] and/or its transpose outT [N][ld_outT]. */
int drn_bias_act_fwd(const float* partials, int splits, long split_stride, const float* bias, const float* mask,
                     unsigned long long seed, const unsigned long long* seed_dev, float drop_p, void* out, long ld_out,
                     void* outT, long ld_outT, int M, int N, long ld_in, int relu, int out_dtype, void* stream);

/* *counter += inc on the stream (the dropout seed lives on the device so a replayed hipGraph draws fresh masks). */
int drn_counter_add(unsigned long long* counter, unsigned long long inc, void* stream);

/* autograd of the above: dpre = grad_out * dropout_mult * (saved_out > 0); colsum[n] = sum_m dpre (bias
 * gradient, fixed summation order); saved_out == NULL means "no activation"; colscale [N] (optional)
 * multiplies grad_out per column (per-loss upstream gradients stay on the device): column n uses
 * colscale[colidx ? colidx[n] : n] (index -1 = 0); colpart = scratch of
 * ceil(M/64)*N floats for the two-stage (deterministic) column sums, required with colsum. */
int drn_bias_act_bwd(const void* grad_out, int grad_dtype, long ld_in, const float* colscale, const int* colidx,
                     const void* saved_out, const float* mask, float drop_p,
                     void* dpre, long ld_out, void* dpreT, long ld_outT, float* colsum, float* colpart,
                     int accumulate_colsum, int M, int N, int out_dtype, void* stream);

/* drn_bias_act_bwd with grad_out given as `splits` fp32 split-K partials, `split_stride` floats apart, summed on load in
 * order (the dX GEMM in front of it can then use a K-split like the forward GEMMs). */
int drn_bias_act_bwd_splits(const void* grad_out, int grad_dtype, long ld_in, int splits, long split_stride,
                            const float* colscale, const int* colidx, const void* saved_out, const float* mask,
                            float drop_p, void* dpre, long ld_out, void* dpreT, long ld_outT, float* colsum,
                            float* colpart, int accumulate_colsum, int M, int N, int out_dtype, void* stream);

/* drn_gemm_nt (fp32 out, one split) + drn_bias_act_bwd in ONE launch for a skinny contraction - the predictor's dX
 * feeding fc7's activation backward (box_head.py:82-91 autograd behind fast_rcnn.py:493-527): dpre = act'(saved_out) .*
 * (A [M][lda] . B [N][ldb]^T), its transposed copy and the bias-gradient column sums, bit for bit what the two calls
 * produce; the fp32 [M][N] product never goes to memory.  bf16 operands and outputs, K in {64, 128, 192, 256},
 * N % 64 == 0, 16-byte aligned rows; DRN_ERR_UNSUPPORTED otherwise (callers then run the two calls). */
int drn_gemm_nt_act_bwd(const void* A, const void* B, int M, int N, int K, long lda, long ldb, const void* saved_out,
                        const float* mask, float drop_p, void* dpre, long ld_out, void* dpreT, long ld_outT, float* colsum,
                        float* colpart, int accumulate_colsum, void* stream);

/* Second stage of drn_bias_act_bwd's column sums (bias gradients) on its own: with colsum == NULL and colpart != NULL
 * drn_bias_act_bwd only leaves the ceil(M/64) x N per-block partials; this adds them in a fixed order into colsum
 * (accumulate != 0: += ).  Lets the optimizer stream finish the bias gradients right in front of the SGD pass. */
int drn_colsum_reduce(const float* colpart, int nparts, int N, float* colsum, int accumulate, void* stream);

/* ---- MIL / OICR head ------------------------------------------------------------------------- */

/* WSDDNOutputLayers.forward (fast_rcnn.py:493-527) + predict_probs_img (:689-700) +
 * WSDDNOutputs.binary_cross_entropy_loss (:317-329) + their autograd.  logits [M][ld] fp32 with the
 * cls / det heads at columns c_cls / c_det; img_off [n_img+1] row offsets.  loss = sum(loss_part).
 * scratch: n_img * ceil(max_rows/32) * 384 floats, max_rows = largest per-image proposal count. */
int drn_wsddn_fwd_bwd(const float* logits, long ld, int c_cls, int c_det, int K, const int* img_off, int n_img,
                      const float* gt_onehot, float* scores, float* row_softmax, float* img_scores, float* loss_part,
                      float* dlogits, long ld_d, float* scratch, int max_rows, int mean_loss, float loss_scale,
                      void* stream);

/* OICRROIHeads.get_pgt (roi_heads_oicr.py:491-567) + ROIHeads.label_and_sample_proposals
 * (roi_heads.py:255-353; pairwise_iou structures/boxes.py:329-361; Matcher modeling/matcher.py:61-103).
 * prev_boxes [M][box_cols] with box_cols = 4 (Boxes path) or 4K (class-specific decoded boxes);
 * zero_delta_decode = 1 applies apply_deltas(0, .) to the selected box (what a non-regressing head passes on). */
int drn_oicr_targets(const float* prev_scores, long ld_s, const float* prev_boxes, int box_cols, int zero_delta_decode,
                     const float* props,
                     const int* img_off, int n_img, const int* gt_classes, const int* gt_count, int gmax,
                     const float* img_scores, int K, const float* thresholds_host, const int* thr_labels_host,
                     int nthr, int* labels, float* weights, int* matched, float* gt_boxes, int* pgt_idx,
                     float* pgt_boxes, void* stream);

/* The whole refinement cascade of OICRROIHeads._forward_box (roi_heads_oicr.py:372-395: for each branch k,
 * get_pgt on the previous branch's scores -> label_and_sample_proposals -> OICROutputs.losses) for n_heads
 * NON-regressing branches in four launches: = drn_softmax_ce(probs only) / drn_oicr_targets / drn_softmax_ce per head,
 * bit for bit.  scores0 [M][ld_s0] = the WSDDN scores feeding branch 0.  Per-head outputs are [n_heads] x the
 * single-head arrays, contiguous; scratch: n_heads * 2*ceil(M/16) floats; col0s_host: host array [n_heads]. */
int drn_oicr_refine_chain(const float* logits, long ld, const int* col0s_host, int n_heads, int K, const float* scores0,
                          long ld_s0, const float* props, const int* img_off, int n_img, const int* gt_classes,
                          const int* gt_count, int gmax, const float* img_scores, const float* thresholds,
                          const int* thr_labels, int nthr, float* probs, int* labels, float* weights, int* matched,
                          float* gt_boxes, int* pgt_idx, float* pgt_boxes, float* dlogits, long ld_d, float* losses,
                          float* scratch, int M, float loss_scale, void* stream);

/* The loss tail of one training step of OICRROIHeads._forward_box with non-regressing refinement branches
 * (roi_heads_oicr.py:351-395: predictor bias -> WSDDNOutputs.losses -> the refinement cascade) in SIX launches:
 * = drn_bias_act_fwd (fp32 logits from the predictor GEMM's split-K partials [splits][M][ld_part] + bias) +
 * drn_wsddn_fwd_bwd + drn_oicr_refine_chain (nine), bit for bit: WSDDN stage 0 and the heads' softmax read their logits
 * from the partials.  logits [M][ld] is an OUTPUT: columns c_cls..+K, c_det..+K and col0s_host[k]..+K+1 are written.
 * seed_dev (optional): the device-side dropout counter this pass advances by seed_inc (what the fp32 drn_bias_act_fwd
 * call without dropout does).  ws_scratch as drn_wsddn_fwd_bwd, ce_scratch as drn_oicr_refine_chain.  The WSDDN scores
 * feed branch 0. */
int drn_mil_oicr_losses(const float* partials, int splits, long split_stride, long ld_part, const float* bias,
                        unsigned long long seed_inc, unsigned long long* seed_dev, float* logits, long ld, int c_cls,
                        int c_det, int K, const int* img_off, int n_img, const float* gt_onehot, float* scores,
                        float* row_softmax, float* img_scores, float* loss_part, float* ws_scratch, int max_rows,
                        int mean_loss, const int* col0s_host, int n_heads, const float* props, const int* gt_classes,
                        const int* gt_count, int gmax, const float* thresholds, const int* thr_labels, int nthr, float* probs,
                        int* labels, float* weights, int* matched, float* gt_boxes, int* pgt_idx, float* pgt_boxes,
                        float* losses, float* ce_scratch, float* dlogits, long ld_d, int M, float loss_scale,
                        void* stream);

/* OICROutputs.softmax_cross_entropy_loss (fast_rcnn.py:1087-1096,1128-1144), predict_probs (:1561-1575)
 * and the backward: loss = sum_r w_r CE_r / #{w_r > 1e-12}.  labels == NULL => probabilities only.
 * scratch: 2*ceil(M/16) floats (two-stage deterministic reduction), required with labels. */
int drn_softmax_ce(const float* logits, long ld, int col0, int C, const int* labels, const float* weights,
                   float* probs, float* dlogits, long ld_d, float* loss, float* scratch, int M, float loss_scale,
                   void* stream);

/* OICROutputs.box_reg_loss, fast_rcnn.py:1146-1211 (WSL.REFINE_REG heads) with Box2BoxTransform.get_deltas,
 * detectron2/modeling/box_regression.py:38-71, smooth-L1 beta = 0: loss = sum_fg |pred - target| / M, and its
 * gradient written to dlogits columns col0 .. col0+4K.  scratch: ceil(M/256) floats. */
int drn_box_reg_loss(const float* logits, long ld, int col0, int K, const int* labels, const float* props,
                     const float* gt_boxes, const float* weights4_host, float* dlogits, long ld_d, float* loss,
                     float* scratch, int M, float loss_scale, void* stream);

/* OICROutputLayers.predict_probs_K, fast_rcnn.py:1577-1594.  bg_first = 1: the heads keep the background in their
 * column 0 (PCL) and the output is rotated so that it is the last column (`pcl_bg`, fast_rcnn.py:1463-1465). */
int drn_mean_softmax(const float* logits, long ld, const int* col0s_dev, int n_heads, int C, float* probs, int M,
                     int bg_first, void* stream);

/* Box2BoxTransform.apply_deltas, detectron2/modeling/box_regression.py:73-110 (deltas NULL = zeros). */
int drn_apply_deltas(const float* deltas, long ld_d, const float* boxes, float* out, int M, int K,
                     const float* weights4_host, float scale_clamp, void* stream);

int drn_sum_small(const float* in, int n, float scale, float* out, void* stream);

/* ---- optimizer ------------------------------------------------------------------------------- */

/* torch.optim.SGD(momentum) with the per-parameter groups of detectron2/solver/build.py:93-137, applied to a
 * flat parameter arena.  segs_dev: array of {int64 offset, int64 count, float lr, float weight_decay}.
 * shadow (optional, bf16, same flat layout) is refreshed in the same pass.  grads: fp32, or bf16 (gradient buckets
 * that were all-reduced in bf16); arena element j reads grads[j - grad_off], so a bucket buffer that only covers
 * one tensor can be passed with that tensor's arena offset. */
int drn_sgd_step(float* weights, float* momentum_buf, const void* grads, int grad_dtype, long grad_off, void* shadow,
                 int shadow_dtype, const void* segs_dev, int nseg, float momentum, int first_step, float grad_scale,
                 void* stream);

/* The same update (torch.optim.SGD.step, detectron2/solver/build.py:93-137; projects/WSL/tools/train_net.py:104-113 calls
 * it once per iteration) on a rectangular BLOCK of one 2-D parameter: rows r0 .. r0+rows, columns c0 .. c0+cols of the
 * [count / ld][ld] view of the tensor that seg_dev ({offset, count, lr, wd}, ONE entry) describes.  The fc6 weight
 * gradient is produced in column slabs of exactly one round of the persistent GEMM each, and a slab's update is
 * issued the moment its GEMM is queued - the optimizer pass of step t then trails the weight-gradient GEMM by one round
 * instead of by half of it.  Per-element arithmetic is drn_sgd_step's (bit-identical results); c0, cols, ld, grad_off
 * multiples of 4. */
int drn_sgd_step_block(float* weights, float* momentum_buf, const void* grads, int grad_dtype, long grad_off, void* shadow,
                       int shadow_dtype, const void* seg_dev, int r0, int rows, int c0, int cols, long ld, float momentum,
                       int first_step, float grad_scale, void* stream);

/* ---- inference tail -------------------------------------------------------------------------- */

/* fast_rcnn_inference_single_image, fast_rcnn.py:88-141 + batched_nms, detectron2/layers/nms.py:10-29. */
long drn_detect_workspace_bytes(int cap);
int drn_detect_topk(const float* boxes, const float* scores, int R, int K, int nreg, float img_h, float img_w,
                    float score_thresh, float nms_thresh, int topk, void* workspace, long workspace_bytes, int cap,
                    int* keep_ids, int* n_keep, void* stream);
int drn_detect_gather(const void* workspace, long workspace_bytes, int cap, const int* keep_ids, const int* n_keep,
                      int topk, float* out_boxes, float* out_scores, int* out_classes, int* out_rows, void* stream);

/* ---- test-time augmentation ------------------------------------------- */

/* GeneralizedRCNNWithTTAAVG._get_augmented_boxes (projects/WSL/wsl/modeling/test_time_augmentation_avg.py:269-294):
 * fold one augmentation's predictions into the running averages.  boxes [n_boxes][4] (the [R, 4K] prediction viewed
 * as boxes, in the augmented image's coordinates) are mapped back through HFlipTransform.inverse (flip_w = width of
 * the augmented image, < 0 = not flipped) and ResizeTransform.inverse (sx = w / new_w, sy = h / new_h, float32 like
 * the reference's numpy code) and added to acc_boxes; scores [n_scores] are added to acc_scores.  first = 1 starts the
 * sums, n_final = number of augmentations on the last call (the means are written then), 0 otherwise. */
int drn_tta_accumulate(const float* boxes, const float* scores, float* acc_boxes, float* acc_scores, long n_boxes,
                       long n_scores, float sx, float sy, float flip_w, int first, int n_final, void* stream);

/* ---- PCL refinement (PCLROIHeads) ------------------------------------------------------------------
 * The reference computes these on the HOST: the targets in numpy + scikit-learn after a device->host copy of the scores
 * (projects/WSL/wsl/modeling/roi_heads/third_party/pcl.py:26-200, called from fast_rcnn.py:1725-1745), the loss in C++ on
 * the CPU (projects/WSL/wsl/layers/csrc/pcl_loss/pcl_loss.h:52-131 always dispatches to pcl_loss_cpu.cpp:8-117).  One
 * image per call (the reference asserts a batch of one, pcl.py:94,149); R <= 4096, K <= 128.
 *
 * drn_pcl_adjacency: `_build_graph` (pcl.py:78-87): bit r2 of word adj[r1][r2/32] = IoU(box r1, box r2) > iou_thr
 * (pairwise_iou, detectron2/structures/boxes.py:329-361).  adj: [R, ceil(R/32)] uint32.  Shared by every branch. */
int drn_pcl_adjacency(const float* boxes, int R, float iou_thr, uint32_t* adj, void* stream);

/* drn_pcl_refine: every refinement branch b < n_branch in one call.  logits [R, ld] fp32 holds branch b's K+1 columns
 * at cols[b] (host array; column 0 of a branch = background, class c at 1 + c - the PCL convention).  Branch 0 clusters
 * on wsddn_scores [R, ld_ws] (class c at column c), branch b > 0 on the softmax of branch b-1 (roi_heads_pcl.py:321-334).
 * onehot [K]: image-level labels.  Writes: probs [n_branch, R, K+1] (row softmax, fast_rcnn.py:1561-1575);
 * labels / cls_w / assign [n_branch, R] (pcl.py:166-181; assign = -1 for background rows); the proposal clusters
 * pc_labels / pc_probs / pc_count / img_w / pc_rows (row of the centre box) / pc_scores [n_branch, pmax] with their
 * number in n_pc [n_branch] (pmax >= 5 * labelled classes, <= 640); losses [n_branch] = pcl_loss forward summed over
 * classes / R (pcl_loss.py:51); dlogits [R, ld] at the same columns = d loss / d logits (pcl_loss_cpu.cpp:60-115 through
 * the softmax).  k-means and equal-degree ties follow the fixed definitions of oracle/pcl_oracle.py (the reference's
 * scikit-learn draw / numpy sort order are not functions of the inputs). */
int drn_pcl_refine(const float* logits, int ld, const int* cols, int n_branch, int K, const float* wsddn_scores,
                   int ld_ws, const float* boxes, const uint32_t* adj, const float* onehot, int R, float* probs,
                   int* labels, float* cls_w, int* assign, int* pc_labels, float* pc_probs, int* pc_count,
                   float* img_w, int* pc_rows, float* pc_scores, int* n_pc, int pmax, float* losses, float* dlogits,
                   void* stream);

/* ---- CSC head (CSCROIHeads) -------------------------------------------------------------------------
 * projects/WSL/wsl/modeling/roi_heads/roi_heads_csc.py:423-510 and wsl/layers/csrc/csc/csc_cuda.cu.  One image per
 * call (the reference head reads image_sizes[0] / gt_classes_img_oh[0]); K <= 128.
 *
 * drn_csc_cpg: the per-class map of `_forward_cpg` (roi_heads_csc.py:456-464) from d (sum_r score[r, c]) / d image
 * (NHWC, `cpad` stored channels, the first C are colours; dtype DRN_F32 / DRN_BF16): |.|, max over the C channels,
 * divided by the maximum of the map -> cpg [H*W] fp32.  scratch: 4 bytes. */
int drn_csc_cpg(const void* dimg, int dtype, int cpad, int C, int H, int W, float* cpg, void* scratch, void* stream);

/* drn_csc_weights: `csc_forward_cuda` for ONE labelled class c (csc_cuda.cu:398-535): cpg >= fg_threshold ->
 * summed-area table (binary_and_integral_cpu :132-161; exact integer counts, H*W < 2^24) -> CSCPool (:184-350) per ROI
 * (rois [M][5] = batch index + box in image coordinates) -> max / min normalisation to [-1, 1] -> blend with the
 * image-level prediction sum_r scores[r][c] -> W[:, c] of W [M][K].  table: H*W floats of scratch.  The reference does
 * the table, the normalisation and the blend on the host; this stays on the stream. */
int drn_csc_weights(const float* cpg, int H, int W, float fg_threshold, const float* rois, int M, const float* scores,
                    int K, int c, int area_sqrt, float context_scale, float* table, float* Wout, void* stream);

/* drn_csc_loss: CSCOutputs.csc_loss (fast_rcnn.py:887-931) on the WSDDN scores (scores / row_softmax [M][K] as written
 * by drn_wsddn_fwd_bwd, logits as given to it).
 * mode 0: loss[0] = loss_cls_pos = BCE(clamp(sum_r s * max(W, 0)), onehot), loss[1] = loss_cls_neg =
 *         BCE(clamp(sum_r s * max(-W, 0)), 0) (W NULL = ones: past WSL.CSC_MAX_ITER) and, when dlogits != NULL, the
 *         cls / det columns of dlogits [M][ld_d] = d (loss[0] + loss[1]) / d logits.
 * mode 1: dlogits = d (sum_r scores[r][cstar]) / d logits - the seed of the image-gradient pass
 *         (roi_heads_csc.py:441-455, grad_outputs[:, c] = 1). */
int drn_csc_loss(const float* logits, long ld, int c_cls, int c_det, int K, int M, const float* scores,
                 const float* row_softmax, const float* W, const float* onehot, int mode, int cstar, int mean_loss,
                 float* loss, float* dlogits, long ld_d, void* stream);

/* ---- launch plans (round 3): the frozen trunk's whole layer sequence in ONE call ----
 * Replaces the per-layer Python walk of `ResNet.forward` / `BasicStem.forward` / `BottleneckBlock.forward` /
 * `BasicBlock.forward` (projects/WSL/wsl/modeling/backbone/resnet_ws.py:479-502, :405-416, :217-237, :93-112) and
 * `VGG16.forward` / `PlainBlock.forward` (vgg.py:213-231, :104-122) when nothing has to be kept for a backward pass.
 * A plan is an array of ops over numbered activation SLOTS (buffers the caller owns; an op never writes its own input
 * or residual slot).  Geometry per op follows from the input size; every op runs through drn_conv2d_nhwc_q /
 * drn_maxpool2x2_nhwc, so the kernels, their selection and the results are those of the per-layer calls. */
#define DRN_TRUNK_CONV 0
#define DRN_TRUNK_MAXPOOL 1
#define DRN_TRUNK_MAX_SLOTS 16
#define DRN_TRUNK_KIND_MASK 0xff
#define DRN_TRUNK_FUSE_POOL 0x200 /* flag on a conv op: its output is read by the NEXT op alone, a 2x2 / stride-2 max pool - may run inside the conv's launch (drn_conv3x3_pw_nhwc with pool = 1; with DRN_TRUNK_FUSE_NEXT on the op before it: three ops, one launch) */
#define DRN_TRUNK_FUSE_NEXT 0x100 /* flag on a 3x3 / 64 -> 64 conv op: its output is read by the NEXT op alone, a 1x1 conv to 256 channels - the executor may run the pair as one drn_conv3x3_pw_nhwc launch (the dst slot is then not written) */
typedef struct DrnTrunkOp {
  int kind;           /* DRN_TRUNK_CONV | DRN_TRUNK_MAXPOOL, optionally | DRN_TRUNK_FUSE_NEXT */
  int src, dst, res;  /* slot indices; res = -1: no residual (conv only) */
  const void* w;      /* conv: packed weights [cout][ldw] as for drn_conv2d_nhwc_q */
  const float* scale; /* per-cout affine (folded FrozenBN / quantisation scales) or NULL */
  const float* bias;
  int cin, cout, ksize, stride, pad, dil, relu; /* cin = stored channels of the input slot; pool: stride only */
  long ldw;
  int dtype, out_dtype, res_dtype; /* element types of x / w, of y, of the residual (pool: dtype) */
  float res_mult;
} DrnTrunkOp;

/* Bytes every slot must hold for an [Nb, H, W, C0] input of element type in_dtype sitting in slot in_slot
 * (slot_bytes[n_slots]; 0 for slots the plan never writes), and the (h, w, c) each slot holds when the plan ends
 * (slot_hwc[3 * n_slots], may be NULL).  Host-only: launches nothing. */
int drn_trunk_shapes(const DrnTrunkOp* ops_host, int n_ops, int n_slots, int in_slot, int Nb, int H, int W, int C0,
                     int in_dtype, long* slot_bytes_host, int* slot_hwc_host);

/* Enqueue the whole plan.  slots_host[n_slots]: device pointers (16-byte aligned), each at least as large as
 * drn_trunk_shapes reported. */
int drn_trunk_forward(const DrnTrunkOp* ops_host, int n_ops, int n_slots, int in_slot, void* const* slots_host, int Nb,
                      int H, int W, int C0, int in_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DRN_WSOD_H_ */
