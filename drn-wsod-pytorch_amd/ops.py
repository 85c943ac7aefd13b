"""Tensor-level wrappers over the C ABI (include/drn_wsod.h).  torch is used only to own device
memory and streams; every computation below is a hand-written HIP kernel.  No CPU path exists:
passing CPU tensors raises."""
import ctypes
import math

import torch

from . import _cabi as C

SCALE_CLAMP = math.log(1000.0 / 16)  # detectron2/modeling/box_regression.py:9
GEMM_TIMING = None  # set to a list by bench.py to time GEMM launches with HIP events
HBM_TIMING = None   # likewise for the two HBM-bound kernels of the step (pooling launch, optimizer launches) - in-step figures


FP8 = torch.float8_e4m3fn  # OCP e4m3fn: gfx950's native fp8
FP8_MAX = 448.0


def esize(dtype):
    return 1 if dtype == FP8 else 2 if dtype == torch.bfloat16 else 4


def kpad(k, dtype):
    """K rounded up to a whole number of 128-byte slabs (GEMM contract)."""
    q = 128 // esize(dtype)
    return (k + q - 1) // q * q


def _2d(t):
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D tensor expected"
    return t.stride(0)


def gemm_set_tile(tile):
    """pin the GEMM tile (64/128/256; 0 = heuristic); returns the previous setting"""
    return C.lib().drn_gemm_set_tile(int(tile))


TUNE_GEMM_PERSISTENT, TUNE_SGD_GRID, TUNE_GEMM_GROUP_ROWS, TUNE_ROI_MAP64, TUNE_CONV_KSPLIT, TUNE_GEMM_TAIL_SPLIT, TUNE_CONV_KS_TILES, TUNE_CONV_K2_TILES, TUNE_CONV_PATCH = 1, 2, 3, 4, 5, 6, 7, 8, 9
TUNE_CONV_RING = 23
TUNE_ROI_ST = 31
TUNE_MSM_WAVE = 32
TUNE_ROI_LANE = 19
TUNE_CONV_PP = 24
TUNE_PP8, TUNE_PP8_STAGES, TUNE_PP8_VARIANT, TUNE_PP8_PROFILE, TUNE_PP8_WIDE, TUNE_PP8_WIDE_VARIANT = 25, 26, 27, 28, 29, 30
TUNE_ROI_CPB, TUNE_ROI_PREFETCH, TUNE_GEMM_PINGPONG, TUNE_FP8_K64, TUNE_ROI_MAP64_A, TUNE_ROI_LDS_KB = 10, 11, 12, 13, 14, 15


def tune(knob, value):
    """drn_tune: set a tuning knob (A/B measurements, tests); returns the previous value"""
    return C.lib().drn_tune(int(knob), int(value))


def gemm_nt(A, B, M, N, K, out=None, splits=1, accumulate=False):
    """C[s,M,N] (fp32) = A[M,:K] @ B[N,:K]^T.  A, B: 2-D row-major device tensors of the compute dtype
    whose leading dimension may exceed K (zero padded up to kpad)."""
    lda, ldb = _2d(A), _2d(B)
    assert A.dtype == B.dtype
    if out is None:
        out = torch.empty((splits, M, N), dtype=torch.float32, device=A.device)
    assert out.dtype == torch.float32 or (out.dtype == torch.bfloat16 and splits == 1 and not accumulate)
    ldc = out.stride(-2)
    sstride = out.stride(0) if out.dim() == 3 else 0
    assert out.dim() == 3 or splits == 1
    if GEMM_TIMING is not None:  # bench.py: HIP events on the launching stream around this launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    C.call("drn_gemm_nt", C.ptr(A), C.ptr(B), C.ptr(out), M, N, K, lda, ldb, ldc, C.dt(A.dtype), C.dt(out.dtype), splits,
           sstride, int(accumulate), C.stream())
    if GEMM_TIMING is not None:
        e1.record()
        GEMM_TIMING.append((e0, e1, 2.0 * M * N * K, (M, N, K)))
    return out


def gemm_nt_pair(g0, g1):
    """two independent gemm_nt problems in one persistent launch (drn_gemm_nt_pair); g = dict(A, B, M, N, K, out,
    splits=1, accumulate=False) with fp32 `out` [splits, M, N]"""
    args = []
    for g in (g0, g1):
        A, B, out = g["A"], g["B"], g["out"]
        assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and out.dtype == torch.float32 and out.dim() == 3
        s_ = int(g.get("splits", 1))
        args += [C.ptr(A), C.ptr(B), C.ptr(out), int(g["M"]), int(g["N"]), int(g["K"]), _2d(A), _2d(B), out.stride(-2), s_,
                 out.stride(0), int(bool(g.get("accumulate", False)))]
    if GEMM_TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    C.call("drn_gemm_nt_pair", *args, C.stream())
    if GEMM_TIMING is not None:
        e1.record()
        GEMM_TIMING.append((e0, e1, 2.0 * (g0["M"] * g0["N"] * g0["K"] + g1["M"] * g1["N"] * g1["K"]), ("pair",)))


def gemm_tn(A, Bt, M, N, K, kb_rows, out=None, splits=1, accumulate=False):
    """C[s,M,N] = A[M,:K] @ Bt[:K,:N] with the second operand K-major (Bt [kb_rows, ldb] row-major; rows kb_rows..K-1
    count as zeros and need not exist): drn_gemm_tn, bf16 operands.  The fc6 weight gradient reads the pooled matrix
    through it, so no transposed copy of it is written."""
    assert A.dtype == torch.bfloat16 and Bt.dtype == torch.bfloat16 and Bt.shape[0] >= kb_rows
    lda, ldb = _2d(A), _2d(Bt)
    if out is None:
        out = torch.empty((splits, M, N), dtype=torch.float32, device=A.device)
    assert out.dtype == torch.float32 or (out.dtype == torch.bfloat16 and splits == 1 and not accumulate)
    ldc = out.stride(-2)
    sstride = out.stride(0) if out.dim() == 3 else 0
    assert out.dim() == 3 or splits == 1
    if GEMM_TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    C.call("drn_gemm_tn", C.ptr(A), C.ptr(Bt), C.ptr(out), M, N, K, int(kb_rows), lda, ldb, ldc, C.dt(out.dtype), splits,
           sstride, int(accumulate), C.stream())
    if GEMM_TIMING is not None:
        e1.record()
        GEMM_TIMING.append((e0, e1, 2.0 * M * N * K, (M, N, K)))
    return out


def gemm_tn_sgd(A, Bt, M, N, K, kb_rows, bucket, weights, mom, shadow, seg_dev, momentum, first_step, grad_scale=1.0):
    """drn_gemm_tn_sgd: bucket[M, N] (bf16) = A[M,:K] @ Bt[:K,:N] and, in the same launch, the SGD step of weights[M, N] /
    mom / shadow (2-D views with a common row pitch) with that gradient.  Returns False when the shape is outside the
    kernel's class (nothing was launched: run gemm_tn + sgd_step_block instead)."""
    assert A.dtype == torch.bfloat16 and Bt.dtype == torch.bfloat16 and bucket.dtype == torch.bfloat16
    assert weights.dtype == torch.float32 and mom.dtype == torch.float32 and shadow.dtype == torch.bfloat16
    assert _2d(weights) == _2d(mom) == _2d(shadow)
    if GEMM_TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = C.lib().drn_gemm_tn_sgd(C.ptr(A), C.ptr(Bt), C.ptr(bucket), M, N, K, int(kb_rows), _2d(A), _2d(Bt), _2d(bucket),
                                 C.ptr(weights), C.ptr(mom), C.ptr(shadow), _2d(weights), C.ptr(seg_dev), float(momentum),
                                 int(bool(first_step)), float(grad_scale), C.stream())
    if rc == -3:
        return False
    if rc != 0:
        raise C.DrnError("drn_gemm_tn_sgd failed (%d)" % rc)
    if GEMM_TIMING is not None:
        e1.record()
        GEMM_TIMING.append((e0, e1, 2.0 * M * N * K, ("tn_sgd", M, N, K)))
    return True


def stage_heads_inputs(rois, props, words_src=None, words_dst=None):
    """props[M,4] <- rois[M,1:5]; words_dst <- words_src (int32 blocks of equal length), one launch"""
    assert rois.dtype == torch.float32 and props.dtype == torch.float32 and rois.is_contiguous() and props.is_contiguous()
    assert rois.shape[1] == 5 and props.shape == (rois.shape[0], 4)
    n = 0
    if words_src is not None:
        assert words_src.dtype == torch.int32 and words_dst.dtype == torch.int32 and words_src.numel() == words_dst.numel()
        n = words_src.numel()
    C.call("drn_stage_heads_inputs", C.ptr(rois), C.ptr(props), rois.shape[0], C.ptr(words_src), C.ptr(words_dst), n,
           C.stream())


def gemm_nt_main_cols(M, N, splits=1):
    """columns [0, n0) that drn_gemm_nt keeps for its persistent launch (n0 == N: no tail balancing for this shape)"""
    fn = C.lib().drn_gemm_nt_main_cols
    return int(fn(int(M), int(N), int(splits)))


def conv2d_nhwc(x, w_packed, cout, kh, kw, stride=1, pad=0, dil=1, scale=None, bias=None, residual=None, relu=False):
    """x [N,H,W,Cin] NHWC contiguous; w_packed [Cout, ldw]; returns y [N,Ho,Wo,Cout]."""
    assert x.is_contiguous() and x.dim() == 4
    n, h, w, cin = x.shape
    ho = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=x.dtype, device=x.device)
    if residual is not None:
        assert residual.shape == y.shape and residual.is_contiguous()
    C.call("drn_conv2d_nhwc", C.ptr(x), C.ptr(w_packed), C.ptr(y), C.ptr(scale), C.ptr(bias), C.ptr(residual), n, h, w,
           cin, cout, kh, kw, stride, pad, dil, _2d(w_packed), cout, cout, int(relu), C.dt(x.dtype), C.stream())
    return y


def conv2d_nhwc_q(x, w_packed, cout, kh, kw, stride, pad, dil, scale, bias, out_dtype, residual=None, res_mult=1.0,
                  relu=False):
    """drn_conv2d_nhwc_q: x / w_packed in one element type (fp32, bf16 or fp8 e4m3fn), y stored as out_dtype, an
    optional residual of any of the three types multiplied by res_mult before the add (see include/drn_wsod.h for how
    the quantisation scales enter `scale`, `bias` and `res_mult`)."""
    assert x.is_contiguous() and x.dim() == 4 and x.dtype == w_packed.dtype
    n, h, w, cin = x.shape
    ho = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=out_dtype, device=x.device)
    if residual is not None:
        assert residual.shape == y.shape and residual.is_contiguous()
    C.call("drn_conv2d_nhwc_q", C.ptr(x), C.ptr(w_packed), C.ptr(y), C.ptr(scale), C.ptr(bias), C.ptr(residual), n, h, w,
           cin, cout, kh, kw, stride, pad, dil, _2d(w_packed), cout, cout, int(relu), C.dt(x.dtype), C.dt(out_dtype),
           C.dt(residual.dtype) if residual is not None else 0, float(res_mult), C.stream())
    return y


def conv3x3_pw_nhwc(x, w2_packed, scale2, bias2, relu2, w3_packed=None, scale3=None, bias3=None, residual=None, res_mult=1.0,
                    relu3=True, pool=False, out=None):
    """drn_conv3x3_pw_nhwc: 3x3 (64 -> 64, pad 1) -> act [-> 1x1 (64 -> 256) + residual -> act] [-> 2x2 / stride-2 max pool] as
    one launch; x [N,H,W,64] bf16.  Returns y, or None when the shape is outside the kernel's class (run the separate ops)."""
    assert x.is_contiguous() and x.dim() == 4 and x.shape[3] == 64 and x.dtype == torch.bfloat16
    n, h, w, _ = x.shape
    cy = 256 if w3_packed is not None else 64
    ho, wo = ((h - 2) // 2 + 1, (w - 2) // 2 + 1) if pool else (h, w)
    y = out if out is not None else torch.empty((n, ho, wo, cy), dtype=x.dtype, device=x.device)
    if residual is not None:
        assert residual.shape == (n, h, w, cy) and residual.is_contiguous() and residual.dtype == x.dtype
    rc = C.lib().drn_conv3x3_pw_nhwc(C.ptr(x), C.ptr(w2_packed), C.ptr(scale2), C.ptr(bias2), int(relu2), C.ptr(w3_packed),
                                     C.ptr(scale3), C.ptr(bias3), C.ptr(residual), C.ptr(y), n, h, w, _2d(w2_packed),
                                     _2d(w3_packed) if w3_packed is not None else 0, float(res_mult), int(relu3), int(pool),
                                     C.stream())
    if rc == -3:
        return None
    if rc != 0:
        raise C.DrnError("drn_conv3x3_pw_nhwc failed (%d)" % rc)
    return y


def maxpool2x2_nhwc(x, stride):
    n, h, w, c = x.shape
    ho, wo = (h - 2) // stride + 1, (w - 2) // stride + 1
    y = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
    C.call("drn_maxpool2x2_nhwc", C.ptr(x), C.ptr(y), n, h, w, c, stride, C.dt(x.dtype), C.stream())
    return y


def preprocess_nhwc(images, mean, std, dtype, cpad):
    """images: list of [3,H,W] f32 device tensors -> ([N,Hmax,Wmax,cpad] NHWC, sizes)."""
    hmax = max(int(i.shape[1]) for i in images)
    wmax = max(int(i.shape[2]) for i in images)
    out = torch.empty((len(images), hmax, wmax, cpad), dtype=dtype, device=images[0].device)
    m, s = C.host_floats(mean), C.host_floats(std)
    for k, im in enumerate(images):
        im = im.contiguous().float()
        C.call("drn_preprocess_nhwc", C.ptr(im), im.shape[0], im.shape[1], im.shape[2], C.ptr(out[k]), hmax, wmax, cpad,
               ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p), C.dt(dtype), C.stream())
    return out, [(int(i.shape[1]), int(i.shape[2])) for i in images]


def pil_bilinear_coeffs(in_size, out_size):
    """Pillow's resampling windows and 22-bit coefficients of a BILINEAR resize of `in_size` positions to `out_size`
    (Resample.c: precompute_coeffs + normalize_coeffs_8bpc), with Pillow's double arithmetic in Pillow's operation order:
    -> (bounds int32 [out, 2] = (first source position, count), coef int32 [out, ksize], ksize)"""
    import numpy as np

    scale = float(in_size) / out_size
    fscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * fscale  # (the triangle filter's support is 1)
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / fscale
    center = 0.0 + (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)  # (int) truncates, and the operands are > -1
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):  # (column by column: the same summation order as Pillow's loop over x)
        a = np.abs(((x + xmin).astype(np.float64) - center + 0.5) * ss)
        w = np.where((a < 1.0) & (x < xmax), 1.0 - a, 0.0)
        kk[:, x] = w
        ww = ww + w
    nz = ww != 0.0
    kk[nz] = kk[nz] / ww[nz, None]
    coef = np.where(kk < 0, (-0.5 + kk * (1 << 22)).astype(np.int64), (0.5 + kk * (1 << 22)).astype(np.int64)).astype(np.int32)
    return np.stack([xmin, xmax], 1).astype(np.int32), coef, ksize


_RESIZE_COEFFS = {}


def resize_bilinear_u8(img_hwc, new_h, new_w, flip=False, out=None):
    """img_hwc: uint8 [H, W, C] device tensor -> fp32 [C, new_h, new_w] holding exactly the bytes
    PIL.Image.fromarray(img).resize((new_w, new_h), BILINEAR) produces (mirrored left-right when `flip`): ResizeTransform
    [+ HFlipTransform] of the TTA mapper on the device (drn_resize_bilinear_u8)."""
    assert img_hwc.dtype == torch.uint8 and img_hwc.dim() == 3 and img_hwc.is_contiguous() and img_hwc.is_cuda
    h, w, c = img_hwc.shape
    dev = img_hwc.device

    def tables(n_in, n_out):
        if n_in == n_out:
            return None, None, 0
        key = (n_in, n_out, dev)
        t = _RESIZE_COEFFS.get(key)
        if t is None:
            if len(_RESIZE_COEFFS) > 256:
                _RESIZE_COEFFS.clear()
            b, k, ks = pil_bilinear_coeffs(n_in, n_out)
            t = _RESIZE_COEFFS[key] = (torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), ks)
        return t

    xb, xk, ksx = tables(w, new_w)
    yb, yk, ksy = tables(h, new_h)
    if out is None:
        out = torch.empty((c, new_h, new_w), dtype=torch.float32, device=dev)
    C.call("drn_resize_bilinear_u8", C.ptr(img_hwc), h, w, c, C.ptr(out), new_h, new_w, C.ptr(xb), C.ptr(xk), ksx, C.ptr(yb),
           C.ptr(yk), ksy, int(bool(flip)), C.stream())
    return out


ROI_WORKSPACE = True  # tools / tests: False = pool without the chunk-major scratch copy (same results)


def roi_pool_nhwc(feat, rois, objectness, P, scale, mode=0, sampling_ratio=0, aligned=False, out=None, out_dtype=None,
                  want_argmax=False, out_t=None, t_first_channel=0):
    """feat [N,H,W,C]; rois [M,5] f32; -> out [M, ld] (first C*P*P columns valid, k = c*P*P + bin); out_t (optional,
    [C*P*P, ld_t]) receives the transposed copy in the same call - with t_first_channel > 0 only its rows from channel
    t_first_channel on are guaranteed (drn_roi_pool_nhwc_t)."""
    n, h, w, c = feat.shape
    m = rois.shape[0]
    out_dtype = out_dtype or feat.dtype
    if out is None:
        out = torch.zeros((m, kpad(c * P * P, out_dtype)), dtype=out_dtype, device=feat.device)
    arg = torch.empty((m, c * P * P), dtype=torch.int32, device=feat.device) if want_argmax else None
    if out_t is not None:
        assert out_t.dtype == out.dtype
    if HBM_TIMING is not None:  # bench.py: HIP events on the launching stream around this launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    # scratch for the chunk-major copy of large maps (drn_roi_pool_nhwc_ws): from torch's caching allocator on the current stream,
    # nothing is kept between calls; 0 bytes for every shape whose kernel does not use one (the bench shape among them)
    ws_bytes = C.lib().drn_roi_pool_workspace_bytes(n, h, w, c, P, m, mode, int(want_argmax), C.dt(feat.dtype), C.dt(out.dtype))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=feat.device) if ws_bytes > 0 and ROI_WORKSPACE else None
    C.call("drn_roi_pool_nhwc_ws", C.ptr(feat), C.ptr(rois), C.ptr(objectness), C.ptr(out), C.ptr(out_t), C.ptr(arg), n, h,
           w, c, P, m, float(scale), _2d(out), _2d(out_t) if out_t is not None else 0, mode, sampling_ratio,
           int(aligned), C.dt(feat.dtype), C.dt(out.dtype), int(t_first_channel), C.ptr(ws), ws_bytes if ws is not None else 0,
           C.stream())
    if HBM_TIMING is not None:
        e1.record()
        es = esize(out.dtype)
        t_rows = 0 if out_t is None else (c - min(c, t_first_channel // 8 * 8)) * P * P  # rows of out_t really written
        nbytes = m * c * P * P * es + m * t_rows * es + feat.numel() * esize(feat.dtype) + rois.numel() * 4
        HBM_TIMING.append((e0, e1, nbytes, ("roi_pool", m, c * P * P, out_t is not None)))
    return (out, arg) if want_argmax else out


def stage_rois(boxes, logits, batch_index=0.0):
    """boxes [M, 4] f32 (+ logits [M] f32 or None) of ONE image -> (rois [M, 5], obj [M] or None, props [M, 4]) in one launch
    (drn_stage_rois: convert_boxes_to_pooler_format + the contiguous copies the heads read)"""
    assert boxes.is_cuda and boxes.dtype == torch.float32 and boxes.dim() == 2 and boxes.is_contiguous() and boxes.shape[1] == 4
    M = boxes.shape[0]
    rois = torch.empty((M, 5), dtype=torch.float32, device=boxes.device)
    props = torch.empty((M, 4), dtype=torch.float32, device=boxes.device)
    obj = None
    if logits is not None:
        assert (logits.is_cuda and logits.device == boxes.device and logits.dtype == torch.float32 and logits.dim() == 1 and
                logits.is_contiguous() and logits.shape[0] == M), "objectness logits: a contiguous f32 [M] tensor on the boxes' device"
        obj = torch.empty((M,), dtype=torch.float32, device=boxes.device)
    C.call("drn_stage_rois", C.ptr(boxes), C.ptr(logits), float(batch_index), C.ptr(rois), C.ptr(obj), C.ptr(props), M, C.stream())
    return rois, obj, props


def im2col_t(x, cin, kh, kw, stride, pad, dil, out=None):
    """x [N,H,W,Cpad] NHWC -> [cin*kh*kw, kpad(N*Ho*Wo)] (row (ci*kh + i)*kw + j), zero padded columns."""
    n, h, w, cp = x.shape
    ho = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    if out is None:
        out = torch.zeros((cin * kh * kw, kpad(n * ho * wo, x.dtype)), dtype=x.dtype, device=x.device)
    C.call("drn_im2col_t", C.ptr(x), C.ptr(out), n, h, w, cin, cp, kh, kw, stride, pad, dil, _2d(out), C.dt(x.dtype),
           C.stream())
    return out


def maxpool2x2_bwd_nhwc(x, dy, stride):
    n, h, w, c = x.shape
    assert dy.dtype == x.dtype and dy.is_contiguous() and x.is_contiguous()
    dx = torch.empty_like(x)
    C.call("drn_maxpool2x2_bwd_nhwc", C.ptr(x), C.ptr(dy), C.ptr(dx), n, h, w, c, stride, C.dt(x.dtype), C.stream())
    return dx


def add(a, b, out=None):
    assert a.dtype == b.dtype and a.numel() == b.numel() and a.is_contiguous() and b.is_contiguous()
    if out is None:
        out = torch.empty_like(a)
    C.call("drn_add", C.ptr(a), C.ptr(b), C.ptr(out), a.numel(), C.dt(a.dtype), C.stream())
    return out


def roi_pool_backward_nhwc(grad_out, rois, objectness, feat_shape, P, scale, mode=0, sampling_ratio=0, aligned=False,
                           argmax=None):
    """grad_out [M, >= C*P*P] -> d(feat) [N,H,W,C] fp32 (see drn_roi_pool_backward_nhwc)."""
    n, h, w, c = feat_shape
    m = rois.shape[0]
    dfeat = torch.empty((n, h, w, c), dtype=torch.float32, device=grad_out.device)
    C.call("drn_roi_pool_backward_nhwc", C.ptr(grad_out), C.ptr(rois), C.ptr(objectness), C.ptr(argmax), C.ptr(dfeat),
           n, h, w, c, P, m, float(scale), _2d(grad_out), mode, sampling_ratio, int(aligned), C.dt(grad_out.dtype),
           C.stream())
    return dfeat


def transpose2d(inp, rows, cols, out=None, out_dtype=None):
    out_dtype = out_dtype or inp.dtype
    if out is None:
        out = torch.zeros((cols, kpad(rows, out_dtype)), dtype=out_dtype, device=inp.device)
    C.call("drn_transpose2d", C.ptr(inp), C.ptr(out), rows, cols, _2d(inp), _2d(out), C.dt(inp.dtype), C.dt(out.dtype),
           C.stream())
    return out


def cast2d(inp, rows, cols, out):
    C.call("drn_cast2d", C.ptr(inp), C.ptr(out), rows, cols, _2d(inp), _2d(out), C.dt(inp.dtype), C.dt(out.dtype),
           C.stream())
    return out


def counter_add(counter, inc=1):
    C.call("drn_counter_add", C.ptr(counter), int(inc), C.stream())


def bias_act_fwd(partials, M, N, bias=None, relu=True, mask=None, seed=0, drop_p=0.0, out=None, outT=None, seed_dev=None):
    splits = partials.shape[0] if partials.dim() == 3 else 1
    sstride = partials.stride(0) if partials.dim() == 3 else 0
    ref = out if out is not None else outT
    C.call("drn_bias_act_fwd", C.ptr(partials), splits, sstride, C.ptr(bias), C.ptr(mask), int(seed), C.ptr(seed_dev),
           float(drop_p),
           C.ptr(out), _2d(out) if out is not None else 0, C.ptr(outT), _2d(outT) if outT is not None else 0, M, N,
           partials.stride(-2), int(relu), C.dt(ref.dtype), C.stream())


def linear_act_fwd(A, W, M, N, K, bias=None, relu=True, mask=None, seed=0, drop_p=0.0, out=None, outT=None, seed_dev=None):
    """drn_linear_act_fwd: out [M, N] (bf16) = dropout(relu(A[M,:K] @ W[N,:K]^T + bias)) and optionally its transpose, ONE launch
    (no split-K partials, no second pass).  Returns False when the shape is outside the kernel's class (the caller then runs
    gemm_nt + bias_act_fwd)."""
    assert A.dtype == W.dtype == torch.bfloat16 and out is not None and out.dtype == torch.bfloat16
    rc = C.lib().drn_linear_act_fwd(C.ptr(A), C.ptr(W), C.ptr(bias), C.ptr(mask), int(seed), C.ptr(seed_dev), float(drop_p),
                                    C.ptr(out), _2d(out), C.ptr(outT), _2d(outT) if outT is not None else 0, M, N, K, _2d(A),
                                    _2d(W), int(relu), C.stream())
    if rc == -3:
        return False
    if rc != 0:
        raise C.DrnError("drn_linear_act_fwd failed (%d)" % rc)
    return True


def bias_act_bwd(grad_out, M, N, saved=None, mask=None, drop_p=0.0, colscale=None, dpre=None, dpreT=None, colsum=None,
                 accumulate_colsum=False, colpart=None, colidx=None):
    ref = dpre if dpre is not None else dpreT
    if saved is not None and dpre is not None:
        assert _2d(saved) == _2d(dpre)
    ld_out = _2d(dpre) if dpre is not None else (_2d(saved) if saved is not None else 0)
    if colsum is not None and colpart is None:
        colpart = torch.empty(((M + 63) // 64, N), dtype=torch.float32, device=grad_out.device)
    if grad_out.dim() == 3:  # [splits, M, ld]: fp32 split-K partials of the dX GEMM, summed on load
        C.call("drn_bias_act_bwd_splits", C.ptr(grad_out), C.dt(grad_out.dtype), grad_out.stride(-2), grad_out.shape[0],
               grad_out.stride(0), C.ptr(colscale), C.ptr(colidx), C.ptr(saved), C.ptr(mask), float(drop_p), C.ptr(dpre),
               ld_out, C.ptr(dpreT), _2d(dpreT) if dpreT is not None else 0, C.ptr(colsum), C.ptr(colpart),
               int(accumulate_colsum), M, N, C.dt(ref.dtype), C.stream())
        return
    C.call("drn_bias_act_bwd", C.ptr(grad_out), C.dt(grad_out.dtype), _2d(grad_out), C.ptr(colscale), C.ptr(colidx),
           C.ptr(saved), C.ptr(mask), float(drop_p),
           C.ptr(dpre), ld_out, C.ptr(dpreT), _2d(dpreT) if dpreT is not None else 0, C.ptr(colsum), C.ptr(colpart),
           int(accumulate_colsum), M, N, C.dt(ref.dtype), C.stream())


def gemm_nt_act_bwd(A, B, M, N, K, saved=None, mask=None, drop_p=0.0, dpre=None, dpreT=None, colsum=None,
                    accumulate_colsum=False, colpart=None):
    """drn_gemm_nt_act_bwd: bias_act_bwd of the product A . B^T without the product going to memory (skinny K, bf16).
    Returns False when the shape is outside the kernel's class (the caller then runs gemm_nt + bias_act_bwd)."""
    ref = dpre if dpre is not None else dpreT
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and ref.dtype == torch.bfloat16
    if saved is not None and dpre is not None:
        assert _2d(saved) == _2d(dpre)
    ld_out = _2d(dpre) if dpre is not None else (_2d(saved) if saved is not None else 0)
    if colsum is not None and colpart is None:
        colpart = torch.empty(((M + 63) // 64, N), dtype=torch.float32, device=A.device)
    rc = C.lib().drn_gemm_nt_act_bwd(C.ptr(A), C.ptr(B), M, N, K, _2d(A), _2d(B), C.ptr(saved), C.ptr(mask), float(drop_p),
                                     C.ptr(dpre), ld_out, C.ptr(dpreT), _2d(dpreT) if dpreT is not None else 0,
                                     C.ptr(colsum), C.ptr(colpart), int(accumulate_colsum), C.stream())
    if rc == -3:
        return False
    if rc != 0:
        raise C.DrnError("drn_gemm_nt_act_bwd failed (%d)" % rc)
    return True


def colsum_reduce(colpart, nparts, N, colsum, accumulate=False):
    """finish the two-stage column sums that bias_act_bwd(colsum=None, colpart=...) left as per-block partials"""
    C.call("drn_colsum_reduce", C.ptr(colpart), int(nparts), int(N), C.ptr(colsum), int(accumulate), C.stream())


def wsddn_fwd_bwd(logits, c_cls, c_det, K, img_off, n_img, gt_onehot, dlogits=None, mean_loss=True, loss_scale=1.0,
                  max_rows=None, return_rowsm=False):
    M = logits.shape[0]
    max_rows = max_rows or M
    dev = logits.device
    scores = torch.empty((M, K), dtype=torch.float32, device=dev)
    img_scores = torch.empty((n_img, K), dtype=torch.float32, device=dev)
    loss_part = torch.empty((n_img,), dtype=torch.float32, device=dev)
    rowsm = torch.empty((M, K), dtype=torch.float32, device=dev)
    scratch = torch.empty((n_img * ((max_rows + 31) // 32) * 384,), dtype=torch.float32, device=dev)
    C.call("drn_wsddn_fwd_bwd", C.ptr(logits), _2d(logits), c_cls, c_det, K, C.ptr(img_off), n_img, C.ptr(gt_onehot),
           C.ptr(scores), C.ptr(rowsm), C.ptr(img_scores), C.ptr(loss_part), C.ptr(dlogits),
           _2d(dlogits) if dlogits is not None else 0, C.ptr(scratch), int(max_rows), int(mean_loss), float(loss_scale),
           C.stream())
    if return_rowsm:
        return scores, img_scores, loss_part, rowsm
    return scores, img_scores, loss_part


def csc_cpg(dimg_nhwc, n_colours, out=None):
    """drn_csc_cpg: d score / d image [1, H, W, Cpad] -> the normalised map [H, W] fp32 (roi_heads_csc.py:456-464)"""
    n, H, W, cp = dimg_nhwc.shape
    assert n == 1 and dimg_nhwc.is_contiguous()
    if out is None:
        out = torch.empty((H, W), dtype=torch.float32, device=dimg_nhwc.device)
    scratch = torch.empty((1,), dtype=torch.int32, device=dimg_nhwc.device)
    C.call("drn_csc_cpg", C.ptr(dimg_nhwc), C.dt(dimg_nhwc.dtype), cp, n_colours, H, W, C.ptr(out), C.ptr(scratch), C.stream())
    return out


def csc_weights(cpg, fg_threshold, rois5, scores, c, area_sqrt, context_scale, W_out, table=None):
    """drn_csc_weights: column c of the CSC weights W_out [M, K] from one class map (csc_cuda.cu:398-535)"""
    H, Wd = cpg.shape
    M, K = scores.shape
    assert cpg.is_contiguous() and rois5.is_contiguous() and scores.is_contiguous() and W_out.is_contiguous()
    if table is None:
        table = torch.empty((H, Wd), dtype=torch.float32, device=cpg.device)
    C.call("drn_csc_weights", C.ptr(cpg), H, Wd, float(fg_threshold), C.ptr(rois5), M, C.ptr(scores), K, int(c),
           int(area_sqrt), float(context_scale), C.ptr(table), C.ptr(W_out), C.stream())
    return W_out


def csc_loss(logits, c_cls, c_det, K, scores, rowsm, W, onehot, mean_loss, dlogits=None, seed_class=None):
    """drn_csc_loss.  seed_class None: the two CSC losses [2] (+ their dlogits); else d (sum_r s[r, seed_class]) / d logits"""
    M = scores.shape[0]
    mode = 0 if seed_class is None else 1
    loss = torch.empty((2,), dtype=torch.float32, device=scores.device) if mode == 0 else None
    C.call("drn_csc_loss", C.ptr(logits), _2d(logits), c_cls, c_det, K, M, C.ptr(scores), C.ptr(rowsm), C.ptr(W),
           C.ptr(onehot), mode, int(seed_class or 0), int(mean_loss), C.ptr(loss), C.ptr(dlogits),
           _2d(dlogits) if dlogits is not None else 0, C.stream())
    return loss


def oicr_targets(prev_scores, prev_boxes, props, img_off, n_img, gt_classes, gt_count, img_scores, K,
                 thresholds=(0.5,), labels=(0, 1), zero_delta_decode=False):
    M = props.shape[0]
    dev = props.device
    gmax = gt_classes.shape[1]
    out = dict(labels=torch.empty((M,), dtype=torch.int32, device=dev),
               weights=torch.empty((M,), dtype=torch.float32, device=dev),
               matched=torch.empty((M,), dtype=torch.int32, device=dev),
               gt_boxes=torch.empty((M, 4), dtype=torch.float32, device=dev),
               pgt_idx=torch.empty((n_img, gmax), dtype=torch.int32, device=dev),
               pgt_boxes=torch.empty((n_img, gmax, 4), dtype=torch.float32, device=dev))
    th, lb = C.host_floats(thresholds), C.host_ints(labels)
    C.call("drn_oicr_targets", C.ptr(prev_scores), _2d(prev_scores), C.ptr(prev_boxes), prev_boxes.shape[1],
           int(zero_delta_decode), C.ptr(props),
           C.ptr(img_off), n_img, C.ptr(gt_classes), C.ptr(gt_count), gmax, C.ptr(img_scores), K,
           ctypes.cast(th, ctypes.c_void_p), ctypes.cast(lb, ctypes.c_void_p), len(thresholds), C.ptr(out["labels"]),
           C.ptr(out["weights"]), C.ptr(out["matched"]), C.ptr(out["gt_boxes"]), C.ptr(out["pgt_idx"]),
           C.ptr(out["pgt_boxes"]), C.stream())
    return out


def oicr_refine_chain(logits, col0s, K, scores0, props, img_off, n_img, gt_classes, gt_count, img_scores,
                      thresholds=(0.5,), labels=(0, 1), dlogits=None, loss_scale=1.0):
    """All (non-regressing) refinement branches at once; returns per-head (targets dict, probs, loss) views."""
    M, dev, nh = props.shape[0], props.device, len(col0s)
    gmax = gt_classes.shape[1]
    nb = (M + 15) // 16
    e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
    probs = e((nh, M, K + 1), torch.float32)
    out = dict(labels=e((nh, M), torch.int32), weights=e((nh, M), torch.float32), matched=e((nh, M), torch.int32),
               gt_boxes=e((nh, M, 4), torch.float32), pgt_idx=e((nh, n_img, gmax), torch.int32),
               pgt_boxes=e((nh, n_img, gmax, 4), torch.float32))
    losses = e((nh,), torch.float32)
    scratch = e((nh * 2 * nb,), torch.float32)
    th, lb, c0 = C.host_floats(thresholds), C.host_ints(labels), C.host_ints(col0s)
    vp = lambda a: ctypes.cast(a, ctypes.c_void_p)
    C.call("drn_oicr_refine_chain", C.ptr(logits), _2d(logits), vp(c0), nh, K, C.ptr(scores0), _2d(scores0),
           C.ptr(props), C.ptr(img_off), n_img, C.ptr(gt_classes), C.ptr(gt_count), gmax, C.ptr(img_scores),
           vp(th), vp(lb), len(thresholds),
           C.ptr(probs), C.ptr(out["labels"]), C.ptr(out["weights"]), C.ptr(out["matched"]), C.ptr(out["gt_boxes"]),
           C.ptr(out["pgt_idx"]), C.ptr(out["pgt_boxes"]),
           C.ptr(dlogits), _2d(dlogits) if dlogits is not None else 0, C.ptr(losses), C.ptr(scratch), M,
           float(loss_scale), C.stream())
    return [({k: v[i] for k, v in out.items()}, probs[i], losses[i: i + 1]) for i in range(nh)]


def mil_oicr_losses(partials, bias, logits, c_cls, c_det, K, img_off, n_img, gt_onehot, col0s, props, gt_classes,
                    gt_count, thresholds=(0.5,), labels=(0, 1), dlogits=None, mean_loss=True, loss_scale=1.0,
                    max_rows=None, seed_inc=0, seed_dev=None):
    """drn_mil_oicr_losses: predictor split-K reduce + bias -> `logits` (written), WSDDN forward/backward and the whole
    non-regressing refinement cascade in six launches.  Returns (scores, img_scores, loss_part, chain) with `chain` as
    oicr_refine_chain returns it."""
    M, dev, nh = props.shape[0], props.device, len(col0s)
    max_rows = max_rows or M
    gmax = gt_classes.shape[1]
    nb = (M + 15) // 16
    e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
    scores, rowsm = e((M, K), torch.float32), e((M, K), torch.float32)
    img_scores, loss_part = e((n_img, K), torch.float32), e((n_img,), torch.float32)
    ws_scratch = e((n_img * ((max_rows + 31) // 32) * 384,), torch.float32)
    probs = e((nh, M, K + 1), torch.float32)
    out = dict(labels=e((nh, M), torch.int32), weights=e((nh, M), torch.float32), matched=e((nh, M), torch.int32),
               gt_boxes=e((nh, M, 4), torch.float32), pgt_idx=e((nh, n_img, gmax), torch.int32),
               pgt_boxes=e((nh, n_img, gmax, 4), torch.float32))
    losses = e((nh,), torch.float32)
    ce_scratch = e((nh * 2 * nb,), torch.float32)
    th, lb, c0 = C.host_floats(thresholds), C.host_ints(labels), C.host_ints(col0s)
    vp = lambda a: ctypes.cast(a, ctypes.c_void_p)
    splits = partials.shape[0] if partials.dim() == 3 else 1
    sstride = partials.stride(0) if partials.dim() == 3 else 0
    C.call("drn_mil_oicr_losses", C.ptr(partials), splits, sstride, partials.stride(-2), C.ptr(bias), int(seed_inc),
           C.ptr(seed_dev), C.ptr(logits), _2d(logits), c_cls, c_det, K, C.ptr(img_off), n_img, C.ptr(gt_onehot),
           C.ptr(scores), C.ptr(rowsm), C.ptr(img_scores), C.ptr(loss_part), C.ptr(ws_scratch), int(max_rows),
           int(mean_loss), vp(c0), nh, C.ptr(props), C.ptr(gt_classes), C.ptr(gt_count), gmax, vp(th), vp(lb),
           len(thresholds), C.ptr(probs), C.ptr(out["labels"]), C.ptr(out["weights"]), C.ptr(out["matched"]),
           C.ptr(out["gt_boxes"]), C.ptr(out["pgt_idx"]), C.ptr(out["pgt_boxes"]), C.ptr(losses), C.ptr(ce_scratch),
           C.ptr(dlogits), _2d(dlogits) if dlogits is not None else 0, M, float(loss_scale), C.stream())
    chain = [({k: v[i] for k, v in out.items()}, probs[i], losses[i: i + 1]) for i in range(nh)]
    return scores, img_scores, loss_part, chain


def softmax_ce(logits, col0, ncol, labels=None, weights=None, dlogits=None, loss_scale=1.0):
    M = logits.shape[0]
    probs = torch.empty((M, ncol), dtype=torch.float32, device=logits.device)
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device) if labels is not None else None
    scratch = torch.empty((2 * ((M + 15) // 16),), dtype=torch.float32, device=logits.device) if labels is not None else None
    C.call("drn_softmax_ce", C.ptr(logits), _2d(logits), col0, ncol, C.ptr(labels), C.ptr(weights), C.ptr(probs),
           C.ptr(dlogits), _2d(dlogits) if dlogits is not None else 0, C.ptr(loss), C.ptr(scratch), M, float(loss_scale),
           C.stream())
    return probs, loss


def box_reg_loss(logits, col0, K, labels, props, gt_boxes, weights=(10.0, 10.0, 5.0, 5.0), dlogits=None, loss_scale=1.0):
    M = logits.shape[0]
    loss = torch.zeros((1,), dtype=torch.float32, device=logits.device)
    scratch = torch.empty(((M + 255) // 256,), dtype=torch.float32, device=logits.device)
    w = C.host_floats(weights)
    C.call("drn_box_reg_loss", C.ptr(logits), _2d(logits), col0, K, C.ptr(labels), C.ptr(props), C.ptr(gt_boxes),
           ctypes.cast(w, ctypes.c_void_p), C.ptr(dlogits), _2d(dlogits) if dlogits is not None else 0, C.ptr(loss),
           C.ptr(scratch), M, float(loss_scale), C.stream())
    return loss


_COL0_DEV = {}
COL0_CACHE = True  # tools: False = the column table is uploaded per call (a host sync per inference pass)


def mean_softmax(logits, col0s, ncol, bg_first=False):
    M = logits.shape[0]
    probs = torch.empty((M, ncol), dtype=torch.float32, device=logits.device)
    # the column table lives on the device once per (columns, device): built per call it was a pageable H2D copy, which waits for
    # the stream's queued work - every inference pass ended in a host sync (1.3 ms of a TTA pass, profiles/r5_53_tta_host_profile.txt)
    key = (tuple(int(c) for c in col0s), str(logits.device))
    cd = _COL0_DEV.get(key) if COL0_CACHE else None
    if cd is None:
        cd = _COL0_DEV[key] = torch.tensor(list(key[0]), dtype=torch.int32, device=logits.device)
    C.call("drn_mean_softmax", C.ptr(logits), _2d(logits), C.ptr(cd), len(col0s), ncol, C.ptr(probs), M, int(bg_first),
           C.stream())
    return probs


def apply_deltas(deltas, boxes, K, weights=(10.0, 10.0, 5.0, 5.0), col0=0):
    """deltas: [M, ld] fp32 (class-specific deltas start at column col0) or None (zeros) -> [M, 4K]."""
    M = boxes.shape[0]
    out = torch.empty((M, 4 * K), dtype=torch.float32, device=boxes.device)
    w = C.host_floats(weights)
    dp = None if deltas is None else deltas.data_ptr() + 4 * col0
    C.call("drn_apply_deltas", dp, _2d(deltas) if deltas is not None else 0, C.ptr(boxes), C.ptr(out), M, K,
           ctypes.cast(w, ctypes.c_void_p), float(SCALE_CLAMP), C.stream())
    return out


def sum_small(x, scale=1.0):
    out = torch.empty((1,), dtype=torch.float32, device=x.device)
    C.call("drn_sum_small", C.ptr(x), x.numel(), float(scale), C.ptr(out), C.stream())
    return out


def sgd_step(weights, momentum_buf, grads, segs_dev, nseg, momentum, first_step, grad_scale=1.0, shadow=None,
             grad_off=0):
    """grads: the fp32 gradient arena, or (grad_off > 0) a bucket buffer - fp32 or bf16 - whose element 0 is arena
    element grad_off."""
    if HBM_TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    C.call("drn_sgd_step", C.ptr(weights), C.ptr(momentum_buf), C.ptr(grads), C.dt(grads.dtype), int(grad_off),
           C.ptr(shadow), C.dt(shadow.dtype) if shadow is not None else 0, C.ptr(segs_dev), nseg, float(momentum),
           int(first_step), float(grad_scale), C.stream())
    if HBM_TIMING is not None:
        e1.record()
        # bytes are the caller's to know (the segment table lives on the device): keyed by (segments, bucket offset)
        HBM_TIMING.append((e0, e1, None, ("sgd", int(nseg), int(grad_off), str(grads.dtype), shadow is not None)))


def sgd_step_block(weights, momentum_buf, grads, seg_dev, r0, rows, c0, cols, ld, momentum, first_step, grad_scale=1.0,
                   shadow=None, grad_off=0):
    """drn_sgd_step on rows r0 .. r0+rows, columns c0 .. c0+cols of the [., ld] tensor described by seg_dev (ONE
    {offset, count, lr, wd} entry on the device); weights / momentum_buf / shadow are the flat arenas, grads as in
    sgd_step."""
    if HBM_TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    C.call("drn_sgd_step_block", C.ptr(weights), C.ptr(momentum_buf), C.ptr(grads), C.dt(grads.dtype), int(grad_off),
           C.ptr(shadow), C.dt(shadow.dtype) if shadow is not None else 0, C.ptr(seg_dev), int(r0), int(rows), int(c0),
           int(cols), int(ld), float(momentum), int(first_step), float(grad_scale), C.stream())
    if HBM_TIMING is not None:
        e1.record()
        HBM_TIMING.append((e0, e1, int(rows) * int(cols), ("sgd_block", int(r0), int(rows), int(c0), int(cols),
                                                            str(grads.dtype), shadow is not None)))


def detect_topk(boxes, scores, image_shape, score_thresh, nms_thresh, topk):
    """Single image: boxes [R, 4*nreg] f32, scores [R, K+1] f32 -> (boxes [n,4], scores [n], classes [n] i64,
    rows [n] i64) exactly as fast_rcnn_inference_single_image."""
    R, K = scores.shape[0], scores.shape[1] - 1
    nreg = boxes.shape[1] // 4
    dev = boxes.device
    cap = max(R * K, 1)
    nbytes = C.lib().drn_detect_workspace_bytes(cap)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    keep = torch.empty((topk,), dtype=torch.int32, device=dev)
    nk = torch.zeros((1,), dtype=torch.int32, device=dev)
    boxes = boxes.contiguous().float()
    scores = scores.contiguous().float()
    C.call("drn_detect_topk", C.ptr(boxes), C.ptr(scores), R, K, nreg, float(image_shape[0]), float(image_shape[1]),
           float(score_thresh), float(nms_thresh), topk, C.ptr(ws), nbytes, cap, C.ptr(keep), C.ptr(nk), C.stream())
    ob = torch.empty((topk, 4), dtype=torch.float32, device=dev)
    os_ = torch.empty((topk,), dtype=torch.float32, device=dev)
    oc = torch.empty((topk,), dtype=torch.int32, device=dev)
    orow = torch.empty((topk,), dtype=torch.int32, device=dev)
    C.call("drn_detect_gather", C.ptr(ws), nbytes, cap, C.ptr(keep), C.ptr(nk), topk, C.ptr(ob), C.ptr(os_), C.ptr(oc),
           C.ptr(orow), C.stream())
    n = int(nk.item())
    return ob[:n], os_[:n], oc[:n].long(), orow[:n].long()


def tta_accumulate(boxes, scores, acc_boxes, acc_scores, sx, sy, flip_w, first, n_final):
    """fold one augmentation's [R, 4K] boxes / [R, K+1] scores into the running TTA averages (see the header)"""
    assert boxes.is_contiguous() and scores.is_contiguous() and boxes.dtype == torch.float32 and scores.dtype == torch.float32
    assert acc_boxes.shape == boxes.shape and acc_scores.shape == scores.shape
    C.call("drn_tta_accumulate", C.ptr(boxes), C.ptr(scores), C.ptr(acc_boxes), C.ptr(acc_scores), boxes.numel() // 4,
           scores.numel(), float(sx), float(sy), float(flip_w), int(first), int(n_final), C.stream())


def pcl_adjacency(boxes, iou_thr=0.4):
    """[R, ceil(R/32)] uint32 bit matrix of IoU(boxes, boxes) > iou_thr (third_party/pcl.py:78-87)"""
    R = boxes.shape[0]
    assert boxes.dtype == torch.float32 and boxes.is_contiguous()
    adj = torch.empty((R, (R + 31) // 32), dtype=torch.int32, device=boxes.device)
    C.call("drn_pcl_adjacency", C.ptr(boxes), R, float(iou_thr), C.ptr(adj), C.stream())
    return adj


def pcl_refine(logits, col0s, K, wsddn_scores, boxes, adj, onehot, dlogits):
    """PCL targets + loss + d loss / d logits of every refinement branch of ONE image (see include/drn_wsod.h).
    Returns a list of per-branch dicts (targets, clusters, probs, loss view)."""
    R, dev, nb = boxes.shape[0], boxes.device, len(col0s)
    pmax = min(640, 5 * K)
    e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
    probs = e((nb, R, K + 1), torch.float32)
    row = dict(labels=e((nb, R), torch.int32), cls_loss_weights=e((nb, R), torch.float32),
               gt_assignment=e((nb, R), torch.int32))
    pc = dict(pc_labels=e((nb, pmax), torch.int32), pc_probs=e((nb, pmax), torch.float32),
              pc_count=e((nb, pmax), torch.int32), img_cls_loss_weights=e((nb, pmax), torch.float32),
              pc_rows=e((nb, pmax), torch.int32), pc_scores=e((nb, pmax), torch.float32))
    n_pc = e((nb,), torch.int32)
    losses = e((nb,), torch.float32)
    c0 = C.host_ints(col0s)
    C.call("drn_pcl_refine", C.ptr(logits), _2d(logits), ctypes.cast(c0, ctypes.c_void_p), nb, K, C.ptr(wsddn_scores),
           _2d(wsddn_scores), C.ptr(boxes), C.ptr(adj), C.ptr(onehot), R, C.ptr(probs),
           C.ptr(row["labels"]), C.ptr(row["cls_loss_weights"]), C.ptr(row["gt_assignment"]), C.ptr(pc["pc_labels"]),
           C.ptr(pc["pc_probs"]), C.ptr(pc["pc_count"]), C.ptr(pc["img_cls_loss_weights"]), C.ptr(pc["pc_rows"]),
           C.ptr(pc["pc_scores"]), C.ptr(n_pc), pmax, C.ptr(losses), C.ptr(dlogits), C.stream())
    out = []
    for b in range(nb):
        d = {k: v[b] for k, v in row.items()}
        d.update({k: v[b] for k, v in pc.items()})
        d.update(n_pc=n_pc[b: b + 1], probs=probs[b], loss=losses[b: b + 1])
        out.append(d)
    return out
