"""ORACLE — CPU fp32 restatement of the DRN-WSOD / OICR hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module;
the product package (drn-wsod-pytorch_amd/) never does and fails loudly without its HIP library.

Every function cites the reference file:line it restates (paths relative to /root/reference).
Parameters are held in a flat dict keyed by the reference's state_dict names
(`backbone.res4.2.conv2.norm.weight`, `roi_heads.box_head.fc1.weight`, ...), so a reference
state_dict drops in unchanged (tests/golden/gen_golden.py does exactly that to pin this file).

Pinning status:
  * everything restating code that lives in /root/reference is pinned by tests/golden/*.npz,
    generated from the imported reference by tests/golden/gen_golden.py;
  * ROIAlign is additionally pinned by oracle/_ref (the reference C++ compiled as-is);
  * RoIPool / nms / batched_nms restate torchvision (absent, un-pinned dependency):
    PARITY UNPINNED for those three (semantics in SURVEY.md Appendix C).
"""
import ctypes
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        import importlib.util

        spec = importlib.util.spec_from_file_location("_oracle_build", os.path.join(_HERE, "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        fresh = os.path.exists(mod.ORACLE_SO) and all(
            os.path.getmtime(mod.ORACLE_SO) >= os.path.getmtime(os.path.join(_HERE, f)) for f in mod.ORACLE_SOURCES)
        so = mod.ORACLE_SO if fresh else mod.build_oracle()
        _LIB = ctypes.CDLL(so)
        _LIB.oracle_nms.restype = ctypes.c_int64
        _LIB.oracle_batched_nms.restype = ctypes.c_int64
        _LIB.oracle_csc_pool_one.restype = ctypes.c_float
    return _LIB


def _fp(t):
    return ctypes.c_void_p(t.data_ptr())


# --------------------------------------------------------------------------------------------
# Region ops (C restatements in roi_ops.c)
# --------------------------------------------------------------------------------------------
def roi_pool_forward(feat: torch.Tensor, rois: torch.Tensor, P: int, scale: float):
    """torchvision RoIPool (call site detectron2/modeling/poolers.py:162-165).
    feat NCHW f32, rois [M,5] -> (out [M,C,P,P] f32, argmax [M,C,P,P] i32)."""
    feat = feat.contiguous().float()
    rois = rois.contiguous().float()
    M = rois.shape[0]
    N, C, H, W = feat.shape
    out = torch.empty(M, C, P, P, dtype=torch.float32)
    arg = torch.empty(M, C, P, P, dtype=torch.int32)
    _lib().oracle_roi_pool_forward(_fp(feat), _fp(rois), M, C, H, W, P, P, ctypes.c_float(scale), _fp(out), _fp(arg))
    return out, arg


def roi_pool_backward(grad_out, rois, argmax, shape):
    N, C, H, W = shape
    g = torch.zeros(N, C, H, W, dtype=torch.float32)
    M, _, P, _ = grad_out.shape
    go = grad_out.contiguous().float()
    _lib().oracle_roi_pool_backward(_fp(go), _fp(rois.contiguous().float()), _fp(argmax.contiguous()), M, C, H, W, P,
                                    P, _fp(g))
    return g


class _RoIPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, P, scale):
        out, arg = roi_pool_forward(feat, rois, P, scale)
        ctx.save_for_backward(rois, arg)
        ctx.shape = tuple(feat.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        rois, arg = ctx.saved_tensors
        return roi_pool_backward(g, rois, arg, ctx.shape), None, None, None


def roi_pool(feat, rois, P, scale):
    return _RoIPoolFn.apply(feat, rois, P, scale)


def roi_align_forward(feat, rois, P, scale, sampling_ratio=0, aligned=False):
    """detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:116-218."""
    feat = feat.contiguous().float()
    rois = rois.contiguous().float()
    M = rois.shape[0]
    N, C, H, W = feat.shape
    out = torch.empty(M, C, P, P, dtype=torch.float32)
    _lib().oracle_roi_align_forward(_fp(feat), _fp(rois), M, C, H, W, P, P, ctypes.c_float(scale), int(sampling_ratio),
                                    int(bool(aligned)), _fp(out))
    return out


def roi_align_backward(grad_out, rois, shape, P, scale, sampling_ratio=0, aligned=False):
    """detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:286-395."""
    N, C, H, W = shape
    g = torch.zeros(N, C, H, W, dtype=torch.float32)
    go = grad_out.contiguous().float()
    rois = rois.contiguous().float()
    _lib().oracle_roi_align_backward(_fp(go), _fp(rois), rois.shape[0], C, H, W, P, P, ctypes.c_float(scale),
                                     int(sampling_ratio), int(bool(aligned)), _fp(g))
    return g


class _RoIAlignFn(torch.autograd.Function):
    """detectron2/layers/roi_align.py:22-59 (_ROIAlign)."""

    @staticmethod
    def forward(ctx, feat, rois, P, scale, sampling_ratio, aligned):
        ctx.save_for_backward(rois)
        ctx.args = (tuple(feat.shape), P, scale, sampling_ratio, aligned)
        return roi_align_forward(feat, rois, P, scale, sampling_ratio, aligned)

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        shape, P, scale, sr, al = ctx.args
        return roi_align_backward(g, rois, shape, P, scale, sr, al), None, None, None, None, None


def roi_align(feat, rois, P, scale, sampling_ratio=0, aligned=False):
    return _RoIAlignFn.apply(feat, rois, P, scale, sampling_ratio, aligned)


def nms(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    """torchvision.ops.nms."""
    n = boxes.shape[0]
    keep = torch.empty(max(n, 1), dtype=torch.int64)
    b = boxes.contiguous().float()
    s = scores.contiguous().float()
    k = _lib().oracle_nms(_fp(b), _fp(s), ctypes.c_int64(n), ctypes.c_float(thr), _fp(keep))
    return keep[:k].clone()


def batched_nms(boxes, scores, idxs, thr):
    """detectron2/layers/nms.py:10-29 over torchvision 0.6 batched_nms."""
    assert boxes.shape[-1] == 4
    n = boxes.shape[0]
    if n < 40000:
        if n == 0:
            return torch.empty(0, dtype=torch.int64)
        keep = torch.empty(n, dtype=torch.int64)
        b = boxes.contiguous().float()
        s = scores.contiguous().float()
        i = idxs.contiguous().to(torch.int64)
        k = _lib().oracle_batched_nms(_fp(b), _fp(s), _fp(i), ctypes.c_int64(n), ctypes.c_float(thr), _fp(keep))
        return keep[:k].clone()
    # nms.py:19-29 — per-class loop, result re-sorted by score
    mask = torch.zeros(n, dtype=torch.bool)
    for c in torch.unique(idxs).tolist():
        m = (idxs == c).nonzero().view(-1)
        k = nms(boxes[m], scores[m], thr)
        mask[m[k]] = True
    keep = mask.nonzero().view(-1)
    return keep[torch.argsort(scores[keep], descending=True, stable=True)]


# --------------------------------------------------------------------------------------------
# Structures / box math
# --------------------------------------------------------------------------------------------
def pairwise_iou(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """detectron2/structures/boxes.py:329-361. b1 [N,4], b2 [M,4] -> [N,M]."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh = wh.clamp(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype))


def matcher(iou: torch.Tensor, thresholds=(0.5,), labels=(0, 1)):
    """detectron2/modeling/matcher.py:61-103 (allow_low_quality_matches=False, roi_heads.py:207-211)."""
    if iou.numel() == 0:
        return (torch.zeros(iou.shape[1], dtype=torch.int64), torch.full((iou.shape[1],), labels[0], dtype=torch.int8))
    th = [-float("inf")] + list(thresholds) + [float("inf")]
    vals, matches = iou.max(dim=0)
    out = torch.ones_like(matches, dtype=torch.int8)
    for l, lo, hi in zip(labels, th[:-1], th[1:]):
        out[(vals >= lo) & (vals < hi)] = l
    return matches, out


_SCALE_CLAMP = math.log(1000.0 / 16)


def get_deltas(src, tgt, weights=(10.0, 10.0, 5.0, 5.0)):
    """detectron2/modeling/box_regression.py:38-71."""
    sw = src[:, 2] - src[:, 0]
    sh = src[:, 3] - src[:, 1]
    sx = src[:, 0] + 0.5 * sw
    sy = src[:, 1] + 0.5 * sh
    tw = tgt[:, 2] - tgt[:, 0]
    th = tgt[:, 3] - tgt[:, 1]
    tx = tgt[:, 0] + 0.5 * tw
    ty = tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)),
                       dim=1)


def apply_deltas(deltas, boxes, weights=(10.0, 10.0, 5.0, 5.0)):
    """detectron2/modeling/box_regression.py:73-110."""
    boxes = boxes.to(deltas.dtype)
    w = boxes[:, 2] - boxes[:, 0]
    h = boxes[:, 3] - boxes[:, 1]
    cx = boxes[:, 0] + 0.5 * w
    cy = boxes[:, 1] + 0.5 * h
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = torch.clamp(deltas[:, 2::4] / ww, max=_SCALE_CLAMP)
    dh = torch.clamp(deltas[:, 3::4] / wh, max=_SCALE_CLAMP)
    pcx = dx * w[:, None] + cx[:, None]
    pcy = dy * h[:, None] + cy[:, None]
    pw = torch.exp(dw) * w[:, None]
    ph = torch.exp(dh) * h[:, None]
    out = torch.zeros_like(deltas)
    out[:, 0::4] = pcx - 0.5 * pw
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw
    out[:, 3::4] = pcy + 0.5 * ph
    return out


# --------------------------------------------------------------------------------------------
# Config
# --------------------------------------------------------------------------------------------
@dataclass
class OracleCfg:
    arch: str = "wsr50"  # wsr18 | wsr50 | wsr101 | vgg16
    out_feature: str = "res5"  # res4 (C4) | res5 (DC5 / C5) | plain5
    res5_dilation: int = 2  # MODEL.RESNETS.RES5_DILATION / MODEL.VGG.CONV5_DILATION
    stem_out: int = 64
    res2_out: int = 256
    num_classes: int = 20
    refine_num: int = 3
    refine_reg: Tuple[bool, ...] = (False, False, False)
    dan_dim: Tuple[int, int] = (2048, 4096)
    pooler_type: str = "ROIPool"  # ROIPool | ROIAlign | ROIAlignV2
    pooler_res: int = 7
    sampling_ratio: int = 0
    mean_loss: bool = True
    iou_thresholds: Tuple[float, ...] = (0.5,)
    iou_labels: Tuple[int, ...] = (0, 1)
    bbox_weights: Tuple[float, ...] = (10.0, 10.0, 5.0, 5.0)
    pixel_mean: Tuple[float, ...] = (102.9801, 115.9465, 122.7717)
    pixel_std: Tuple[float, ...] = (1.0, 1.0, 1.0)
    score_thresh: float = 1e-5
    nms_thresh: float = 0.3
    topk: int = 100
    dropout: float = 0.5
    # solver (projects/WSL/configs/PascalVOC-Detection/oicr_WSR_50_DC5_1x.yaml:37-46)
    base_lr: float = 0.01
    momentum: float = 0.9
    weight_decay: float = 0.0005
    bias_lr_factor: float = 2.0
    weight_decay_bias: float = 0.0
    width_per_group: int = 64  # MODEL.RESNETS.WIDTH_PER_GROUP (bottleneck width of res2)
    heads: str = "oicr"  # MODEL.ROI_HEADS.NAME: "oicr" (OICRROIHeads) | "pcl" (PCLROIHeads, oracle/pcl_oracle.py) |
                         # "wsddn" (WSDDNROIHeads: refine_num = 0) | "csc" (CSCROIHeads: refine_num = 0)
    # bf16 "fast mode" of the product (no reference counterpart: SURVEY F5 - the reference is fp32 only).  True = the
    # same fp32 algorithm with every value the product STORES in bf16 rounded to bf16 (round-to-nearest-even) at the same
    # point: the normalised image, every conv output after its fused epilogue, conv / fc weights as GEMM operands, the
    # pooled fc6 operand, fc6 / fc7 activations, the pre-activation gradients fed to the dW / dX GEMMs and the fc6
    # weight-gradient bucket.  Accumulation, biases, FrozenBN affines, logits, losses, SGD state stay fp32.  This is what
    # the full-size bf16 parity tests compare the HIP path with (tests/test_bench_mode_gpu.py).
    emulate_bf16: bool = False
    # CSCROIHeads (heads == "csc"; roi_heads_csc.py:104-118 constants, WSL.CSC_MAX_ITER).  csc_iter is the head's
    # `self.iter` STATE (roi_heads_csc.py:106,263): roi_heads_train advances it once per training forward.
    csc_tau: float = 0.7
    csc_fg_threshold: float = 0.1
    csc_context_scale: float = 1.8
    csc_area_sqrt: bool = True
    csc_max_iter: int = 35000
    csc_iter: int = 0
    # fp8 conv trunk of the product (BASELINE configs[4]; again no reference counterpart).  {conv name relative to the
    # backbone ("stem.conv1", "res2.0.shortcut", ...): per-tensor scale s_y of that conv's stored output}; the conv whose
    # scale entry is 1.0 and that is named by fp8_last writes bf16.  Emulated like emulate_bf16: the fp32 algorithm with
    # each stored activation rounded to OCP e4m3fn at scale s_y (round-to-nearest-even, saturating at 448) and each conv
    # weight rounded per output channel at scale 448 / max|w[c]|; the image and stem.conv1's weights are bf16.
    fp8_scales: Optional[dict] = None
    fp8_last: str = ""

    @property
    def blocks(self):
        return {"wsr18": [2, 2, 2, 2], "wsr50": [3, 4, 6, 3], "wsr101": [3, 4, 23, 3]}.get(self.arch)

    @property
    def n_stages(self):
        return {"res2": 1, "res3": 2, "res4": 3, "res5": 4}[self.out_feature]


class _RoundFwd(torch.autograd.Function):
    """value rounded to bf16 (RNE, as v_cvt_pk_bf16_f32 / torch do); gradient passes through (the product's backward
    GEMMs read the rounded stored value, which is what autograd sees downstream of this node)"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """identity forward; the GRADIENT is rounded to bf16 (pre-activation gradients are stored in bf16 for the dW / dX
    GEMMs; the fc6 weight gradient is rounded once into its bf16 bucket)"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _q(x, cfg):
    return _RoundFwd.apply(x) if getattr(cfg, "emulate_bf16", False) else x


FP8_MAX = 448.0


def _fp8_round(t, s):
    """value of t after being stored as fp8 e4m3fn at scale s"""
    return (t * s).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).to(torch.float32) / s


def _qt(t, cfg, name):
    """stored value of the output of conv `name` (backbone-relative module name)"""
    sc = getattr(cfg, "fp8_scales", None)
    if not sc:
        return _q(t, cfg)
    if name == cfg.fp8_last:
        return t.to(torch.bfloat16).to(torch.float32)
    return _fp8_round(t, sc[name])


def _qb(x, cfg):
    return _RoundBwd.apply(x) if getattr(cfg, "emulate_bf16", False) else x


def _bn(x, p, prefix, eps=1e-5):
    """detectron2/layers/batch_norm.py:45-65 (FrozenBatchNorm2d, inference branch)."""
    return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"], p[prefix + ".weight"],
                        p[prefix + ".bias"], training=False, eps=eps)


def _conv_bn(x, p, prefix, stride=1, padding=0, dilation=1, cfg=None):
    """detectron2/layers/wrappers.py:63-99 (Conv2d: conv -> norm -> activation)."""
    w = p[prefix + ".weight"]
    if getattr(cfg, "fp8_scales", None):
        if prefix.endswith("stem.conv1"):
            w = w.to(torch.bfloat16).to(torch.float32)  # bf16 image x bf16 weights
        else:
            s_w = FP8_MAX / w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-12)
            w = (w * s_w).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).to(torch.float32) / s_w
    else:
        w = _q(w, cfg)
    x = F.conv2d(x, w, p.get(prefix + ".bias"), stride=stride, padding=padding, dilation=dilation)
    if prefix + ".norm.weight" in p:
        x = _bn(x, p, prefix + ".norm")
    return x


def resnet_ws_stage_plan(cfg: OracleCfg):
    """projects/WSL/wsl/modeling/backbone/resnet_ws.py:649-703: per stage (name, nblocks, dilation,
    pool_stride_of_last_block or None)."""
    plan = []
    for idx in range(cfg.n_stages):
        stage_idx = idx + 2
        dilation = cfg.res5_dilation if stage_idx in (4, 5) else 1
        first_stride = 2 if idx == 0 or (stage_idx == 3 and cfg.res5_dilation == 1) else 1
        has_pool = stage_idx in (2, 3)
        plan.append(("res%d" % stage_idx, cfg.blocks[idx], dilation, first_stride if has_pool else None))
    return plan


def resnet_ws_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, cfg: OracleCfg, prefix="backbone."):
    """BasicStem resnet_ws.py:405-416; BottleneckBlock.forward :217-237; BasicBlock.forward :87-112;
    ResNet.forward :479-502."""
    s = prefix + "stem."
    # bf16 / fp8 emulation: every conv's fused epilogue (affine [+ residual] [+ ReLU]) stores a rounded value
    q = lambda t, name: _qt(t, cfg, name[len(prefix):])
    x = q(F.relu(_conv_bn(x, p, s + "conv1", stride=2, padding=1, cfg=cfg)), s + "conv1")
    x = q(F.relu(_conv_bn(x, p, s + "conv2", padding=1, cfg=cfg)), s + "conv2")
    x = q(F.relu(_conv_bn(x, p, s + "conv3", padding=1, cfg=cfg)), s + "conv3")
    x = F.max_pool2d(x, 2, 2)
    feats = {"stem": x}
    for name, nblk, dil, pool in resnet_ws_stage_plan(cfg):
        for b in range(nblk):
            bp = "%s%s.%d." % (prefix, name, b)
            if cfg.arch == "wsr18":
                out = q(F.relu(_conv_bn(x, p, bp + "conv1", padding=dil, dilation=dil, cfg=cfg)), bp + "conv1")
                out = _conv_bn(out, p, bp + "conv2", padding=dil, dilation=dil, cfg=cfg)
                last = bp + "conv2"
            else:
                out = q(F.relu(_conv_bn(x, p, bp + "conv1", cfg=cfg)), bp + "conv1")
                out = q(F.relu(_conv_bn(out, p, bp + "conv2", padding=dil, dilation=dil, cfg=cfg)), bp + "conv2")
                out = _conv_bn(out, p, bp + "conv3", cfg=cfg)
                last = bp + "conv3"
            sc = q(_conv_bn(x, p, bp + "shortcut", cfg=cfg), bp + "shortcut") if (bp + "shortcut.weight") in p else x
            x = q(F.relu(out + sc), last)
            if pool is not None and b == nblk - 1:
                x = F.max_pool2d(x, 2, pool)
        feats[name] = x
    return feats


def vgg16_forward(p, x, cfg: OracleCfg, prefix="backbone."):
    """projects/WSL/wsl/modeling/backbone/vgg.py:104-122 (PlainBlock.forward), :125-231 (VGG16)."""
    d = cfg.res5_dilation
    plan = [("plain1", 2, 1, 2), ("plain2", 2, 1, 2), ("plain3", 3, 1, 2), ("plain4", 3, 1, 1 if d == 2 else 2),
            ("plain5", 3, d, None)]
    feats = {}
    for name, nconv, dil, pool in plan:
        for i in range(nconv):
            x = _q(F.relu(_conv_bn(x, p, "%s%s.0.conv%d" % (prefix, name, i + 1), padding=dil, dilation=dil, cfg=cfg)), cfg)
        if pool is not None:
            x = F.max_pool2d(x, 2, pool)
        feats[name] = x
    return feats


def backbone_forward(p, x, cfg):
    if cfg.arch == "vgg16":
        return vgg16_forward(p, x, cfg)["plain5"]
    return resnet_ws_forward(p, x, cfg)[cfg.out_feature]


def backbone_stride(cfg: OracleCfg) -> int:
    if cfg.arch == "vgg16":
        return 8 if cfg.res5_dilation == 2 else 16
    s = 4
    for _, _, _, pool in resnet_ws_stage_plan(cfg):
        s *= pool or 1
    return s


def preprocess_image(images: Sequence[torch.Tensor], cfg: OracleCfg):
    """projects/WSL/wsl/modeling/meta_arch/rcnn.py:242-249 + detectron2/structures/image_list.py:57-119
    (size_divisibility 0: zero-pad bottom/right to the batch max)."""
    mean = torch.tensor(cfg.pixel_mean).view(-1, 1, 1)
    std = torch.tensor(cfg.pixel_std).view(-1, 1, 1)
    ims = [(im.float() - mean) / std for im in images]
    H = max(i.shape[1] for i in ims)
    W = max(i.shape[2] for i in ims)
    out = torch.zeros(len(ims), ims[0].shape[0], H, W)
    for k, im in enumerate(ims):
        out[k, :, : im.shape[1], : im.shape[2]] = im
    if getattr(cfg, "fp8_scales", None):
        out = out.to(torch.bfloat16).to(torch.float32)  # the fp8 trunk extends the bf16 mode: bf16 image
    return _q(out, cfg), [(i.shape[1], i.shape[2]) for i in ims]


# --------------------------------------------------------------------------------------------
# ROI heads
# --------------------------------------------------------------------------------------------
def boxes_to_rois(box_lists: Sequence[torch.Tensor]) -> torch.Tensor:
    """detectron2/modeling/poolers.py:62-96 convert_boxes_to_pooler_format."""
    return torch.cat([torch.cat((torch.full((len(b), 1), float(i)), b.float()), dim=1) for i, b in enumerate(box_lists)],
                     dim=0)


def pool_features(feat, box_lists, cfg: OracleCfg):
    """detectron2/modeling/poolers.py:191-226 single-level fast path."""
    rois = boxes_to_rois(box_lists)
    scale = 1.0 / backbone_stride(cfg)
    if cfg.pooler_type == "ROIPool":
        return roi_pool(feat, rois, cfg.pooler_res, scale)
    if cfg.pooler_type == "ROIAlign":
        return roi_align(feat, rois, cfg.pooler_res, scale, cfg.sampling_ratio, False)
    if cfg.pooler_type == "ROIAlignV2":
        return roi_align(feat, rois, cfg.pooler_res, scale, cfg.sampling_ratio, True)
    raise ValueError("Unknown pooler type: {}".format(cfg.pooler_type))


def dan_forward(p, x, cfg, training, dropout_masks=None, prefix="roi_heads.box_head."):
    """projects/WSL/wsl/modeling/roi_heads/box_head.py:82-91. dropout_masks: optional list of two
    {0, 1/(1-p)} multiplier tensors (injected so GPU and oracle share the mask; F8)."""
    x = torch.flatten(x, start_dim=1)
    for k in (1, 2):
        w = _q(p[prefix + "fc%d.weight" % k], cfg)
        if k == 1:
            w = _qb(w, cfg)  # the fc6 weight gradient is rounded once into its bf16 bucket (FusedSGD.enable_pipelined)
        # bf16 mode: the pre-activation gradient is stored in bf16 for the dW / dX GEMMs, the bias gradient is its
        # fp32 column sum (bias_act_bwd) - hence the rounding node sits between the GEMM and the bias add
        if getattr(cfg, "emulate_bf16", False):
            x = F.relu(_qb(F.linear(x, w), cfg) + p[prefix + "fc%d.bias" % k])
        else:
            x = F.relu(F.linear(x, w, p[prefix + "fc%d.bias" % k]))
        if training:
            if dropout_masks is not None:
                x = x * dropout_masks[k - 1]
            elif cfg.dropout > 0:
                x = F.dropout(x, p=cfg.dropout, training=True)
        x = _q(x, cfg)  # H1 / H2 are stored in bf16 after bias + ReLU + dropout
    return x


def _head_linear(x, w, b, cfg):
    """predictor Linear; bf16 mode: bf16 weight operand, logits stay fp32, their gradient is stored in bf16 for the
    dW / dX GEMMs while the bias gradient is its fp32 column sum"""
    if not getattr(cfg, "emulate_bf16", False):
        return F.linear(x, w, b)
    return _qb(F.linear(x, _q(w, cfg)), cfg) + b


def wsddn_scores(p, x, num_per_image, prefix="roi_heads.box_predictor.", cfg=None):
    """fast_rcnn.py:493-527 WSDDNOutputLayers.forward."""
    outs = []
    for xx in x.split(num_per_image, dim=0):
        cls = _head_linear(xx, p[prefix + "cls.weight"], p[prefix + "cls.bias"], cfg)
        det = _head_linear(xx, p[prefix + "det.weight"], p[prefix + "det.bias"], cfg)
        outs.append(F.softmax(cls, dim=1) * F.softmax(det, dim=0))
    return torch.cat(outs, dim=0)


def predict_probs_img(scores, num_per_image):
    """fast_rcnn.py:331-343 / :689-700."""
    s = torch.cat([xx.sum(dim=0, keepdim=True) for xx in scores.split(num_per_image, dim=0)], dim=0)
    return torch.clamp(s, min=1e-6, max=1.0 - 1e-6)


def wsddn_loss(scores, num_per_image, gt_oh, mean_loss=True):
    """fast_rcnn.py:317-329 binary_cross_entropy_loss (mean over N*K, then / N again)."""
    img = predict_probs_img(scores, num_per_image)
    return F.binary_cross_entropy(img, gt_oh, reduction="mean" if mean_loss else "sum") / gt_oh.size(0)


# --------------------------------------------------------------------------------------------
# CSCROIHeads (roi_heads_csc.py; op in oracle/csc_ops.c)
# --------------------------------------------------------------------------------------------
def csc_forward(cpgs, labels, preds, rois5, cfg: "OracleCfg"):
    """wsl/layers/csc.py:9-47 + csrc/csc/csc_cuda.cu:352-552 (restated in csc_ops.c; the reference is CUDA-only).
    cpgs [B,K,H,W], labels / preds [B,K], rois5 [R,5] -> W [R,K], PL, NL."""
    cpgs, labels, preds = cpgs.contiguous().float(), labels.contiguous().float(), preds.contiguous().float()
    rois5 = rois5.contiguous().float()
    B, K, H, Wd = cpgs.shape
    R = rois5.shape[0]
    W = torch.empty((R, K), dtype=torch.float32)
    _lib().oracle_csc_forward(_fp(cpgs), _fp(labels), _fp(preds), _fp(rois5), B, K, H, Wd, R,
                              ctypes.c_float(cfg.csc_fg_threshold), int(cfg.csc_area_sqrt),
                              ctypes.c_float(cfg.csc_context_scale), _fp(W))
    return W, labels.clone(), torch.zeros_like(labels)


def csc_cpgs(scores, image_tensor, gt_oh, cfg: "OracleCfg"):
    """roi_heads_csc.py:423-471 _forward_cpg: per labelled class whose image score reaches tau, the gradient of the
    summed class score w.r.t. the normalised image; |.|, max over the colour channels, divided by its maximum."""
    K = cfg.num_classes
    img = scores.detach().sum(dim=0, keepdim=True)
    H, Wd = image_tensor.shape[-2:]
    cpgs = torch.zeros((1, K, H, Wd), dtype=torch.float32)
    for c in range(K):
        if gt_oh[0, c] < 0.5 or img[0, c] < cfg.csc_tau:
            continue
        go = torch.zeros_like(scores)
        go[:, c] = 1.0
        (g,) = torch.autograd.grad(scores, image_tensor, grad_outputs=go, retain_graph=True)
        g = g.detach().abs().max(dim=1)[0]
        cpgs[0, c] = (g / g.max())[0]
    return cpgs


def csc_weights(scores, image_tensor, gt_oh, rois5, cfg: "OracleCfg"):
    """roi_heads_csc.py:473-510 _forward_csc (both sides of CSC_MAX_ITER). Returns W_pos, W_neg, PL, NL, cpgs."""
    if cfg.csc_iter > cfg.csc_max_iter:
        return torch.ones_like(scores), torch.zeros_like(scores), gt_oh, torch.zeros_like(gt_oh), None
    cpgs = csc_cpgs(scores, image_tensor, gt_oh, cfg)
    W, PL, NL = csc_forward(cpgs, gt_oh, scores.detach().sum(dim=0, keepdim=True), rois5, cfg)
    return torch.clamp(W, min=0.0).abs(), torch.clamp(W, max=0.0).abs(), PL, NL, cpgs


def csc_losses(scores, W_pos, W_neg, PL, NL, mean_loss):
    """fast_rcnn.py:887-931 CSCOutputs.csc_loss (loss_weight 1, empty prefix)."""
    pos = torch.clamp((scores * W_pos).sum(dim=0, keepdim=True), min=1e-20, max=1.0 - 1e-20)
    neg = torch.clamp((scores * W_neg).sum(dim=0, keepdim=True), min=1e-20, max=1.0 - 1e-20)
    red = "mean" if mean_loss else "sum"
    return {"loss_cls_pos": F.binary_cross_entropy(pos, PL, reduction=red) / PL.size(0),
            "loss_cls_neg": F.binary_cross_entropy(neg, NL, reduction=red) / NL.size(0)}


def get_image_level_gt(gt_classes_list, num_classes):
    """roi_heads.py:137-153."""
    ints = [torch.unique(g, sorted=True).to(torch.int64) for g in gt_classes_list]
    oh = torch.cat([torch.zeros(1, num_classes).scatter_(1, g.unsqueeze(0), 1) for g in ints], dim=0)
    return ints, oh


def get_pgt(prev_boxes, prev_scores, gt_ints, img_scores, num_classes):
    """roi_heads_oicr.py:491-567. prev_boxes: list of [R,4] (Boxes path) or [R,4K] tensors;
    prev_scores: list of [R, K or K+1]. -> per image (pgt_boxes [G,4], pgt_classes [G], pgt_scores [G],
    pgt_weights [G], pgt_idx [G])."""
    out = []
    for i, (pb, ps, g) in enumerate(zip(prev_boxes, prev_scores, gt_ints)):
        S = torch.index_select(ps, 1, g)
        sc, idx = torch.max(S, dim=0)
        if pb.shape[1] == 4:
            boxes = pb[idx]
        else:
            b = pb.view(-1, num_classes, 4)
            b = torch.index_select(b, 1, g)
            b = torch.index_select(b, 0, idx).view(-1, 4)
            diag = torch.tensor([k * g.numel() + k for k in range(g.numel())], dtype=torch.int64)
            boxes = torch.index_select(b, 0, diag)
        w = torch.index_select(img_scores[i: i + 1], 1, g).reshape(-1)
        out.append((boxes, g, sc, w, idx))
    return out


def label_proposals(prop_boxes, tgt_boxes, tgt_classes, num_classes, cfg: OracleCfg):
    """roi_heads.py:255-353 label_and_sample_proposals + _sample_proposals :214-246 (sampling
    disabled by the early return). -> (gt_classes [R] i64, matched_idxs [R] i64, gt_boxes [R,4])."""
    iou = pairwise_iou(tgt_boxes, prop_boxes)
    matched, labels = matcher(iou, cfg.iou_thresholds, cfg.iou_labels)
    if tgt_classes.numel() > 0:
        gc = tgt_classes[matched].clone()
        gc[labels == 0] = num_classes
        gc[labels == -1] = -1
        gb = tgt_boxes[matched]
    else:
        gc = torch.zeros_like(matched) + num_classes
        gb = torch.zeros(len(matched), 4)
    return gc, matched, gb


def oicr_cls_loss(logits, gt_classes, weights):
    """fast_rcnn.py:1087-1096 + :1128-1144."""
    w = weights.clone()
    w[gt_classes == -1] = 0.0
    valid = (w > 1e-12).to(w.dtype)
    loss = F.cross_entropy(logits, gt_classes, reduction="none", ignore_index=-1)
    return (loss * w).sum() / valid.sum()


def oicr_box_reg_loss(deltas, gt_classes, prop_boxes, gt_boxes, num_classes, cfg):
    """fast_rcnn.py:1146-1211 (smooth_l1 beta=0 => L1, reduction sum, / R)."""
    fg = ((gt_classes >= 0) & (gt_classes < num_classes)).nonzero().view(-1)
    cols = 4 * gt_classes[fg][:, None] + torch.arange(4)
    tgt = get_deltas(prop_boxes, gt_boxes, cfg.bbox_weights)
    return torch.abs(deltas[fg[:, None], cols] - tgt[fg]).sum() / gt_classes.numel()


def roi_heads_train(p, feat, prop_boxes, objectness, gt_classes_list, cfg: OracleCfg, dropout_masks=None,
                    return_aux=False, image_tensor=None):
    """roi_heads_oicr.py:248-291 + :320-421 (training branch). Returns the loss dict.
    heads == "csc": roi_heads_csc.py:231-266 + :301-352 (image_tensor = the normalised images, requires_grad)."""
    K = cfg.num_classes
    nper = [len(b) for b in prop_boxes]
    gt_ints, gt_oh = get_image_level_gt(gt_classes_list, K)
    pooled = pool_features(feat, prop_boxes, cfg)
    obn = torch.cat([o + 1 for o in objectness], dim=0)
    pooled = _q(pooled * obn.view(-1, 1, 1, 1), cfg)
    x = dan_forward(p, pooled, cfg, True, dropout_masks)
    scores = wsddn_scores(p, x, nper, cfg=cfg)
    if cfg.heads == "csc":
        assert len(prop_boxes) == 1, "CSCROIHeads reads image_sizes[0] / gt_classes_img_oh[0]: one image per step"
        W_pos, W_neg, PL, NL, cpgs = csc_weights(scores, image_tensor, gt_oh, boxes_to_rois(prop_boxes), cfg)
        losses = csc_losses(scores, W_pos, W_neg, PL, NL, cfg.mean_loss)
        cfg.csc_iter += 1
        aux = {"pooled": pooled, "fc7": x, "scores": scores, "W_pos": W_pos, "W_neg": W_neg, "cpgs": cpgs}
        return (losses, aux) if return_aux else losses
    losses = {"loss_cls": wsddn_loss(scores, nper, gt_oh, cfg.mean_loss)}
    img_scores = predict_probs_img(scores, nper).detach()
    prev_scores = list(scores.detach().split(nper, dim=0))
    prev_boxes = [b for b in prop_boxes]
    props_cat = torch.cat(prop_boxes, dim=0)
    aux = {"pooled": pooled, "fc7": x, "scores": scores, "img_scores": img_scores, "pgt": [], "labels": [],
           "weights": [], "logits": []}
    if cfg.heads == "pcl":
        # roi_heads_pcl.py:311-334: branch k clusters on branch k-1's probabilities (one image per step)
        assert len(prop_boxes) == 1, "PCL asserts a batch of one image (third_party/pcl.py:94)"
        last = scores.detach().numpy()
        aux["pcl"] = []
        for k in range(cfg.refine_num):
            pre = "roi_heads.box_refinery_%d." % k
            logits = _head_linear(x, p[pre + "cls_score.weight"], p[pre + "cls_score.bias"], cfg)
            loss, tg, probs = _PclLossFn.apply(logits, prop_boxes[0].numpy(), last, gt_oh[0].numpy())
            losses["loss_cls_r%d" % k] = loss
            aux["logits"].append(logits)
            aux["pcl"].append(tg)
            last = probs
        return (losses, aux) if return_aux else losses
    for k in range(cfg.refine_num):
        pgt = get_pgt(prev_boxes, prev_scores, gt_ints, img_scores, K)
        gcs, ws, gbs = [], [], []
        for (pb, pc, _, pw, _), props in zip(pgt, prop_boxes):
            gc, matched, gb = label_proposals(props, pb, pc, K, cfg)
            gcs.append(gc)
            gbs.append(gb)
            ws.append(torch.index_select(pw, 0, matched))
        gt_classes = torch.cat(gcs)
        weights = torch.cat(ws)
        pre = "roi_heads.box_refinery_%d." % k
        logits = _head_linear(x, p[pre + "cls_score.weight"], p[pre + "cls_score.bias"], cfg)
        if cfg.refine_reg[k]:
            deltas = _head_linear(x, p[pre + "bbox_pred.weight"], p[pre + "bbox_pred.bias"], cfg)
        else:
            deltas = torch.zeros(logits.shape[0], 4 * K)
        losses["loss_cls_r%d" % k] = oicr_cls_loss(logits, gt_classes, weights)
        if cfg.refine_reg[k]:
            losses["loss_box_reg_r%d" % k] = oicr_box_reg_loss(deltas, gt_classes, props_cat, torch.cat(gbs), K, cfg)
        prev_scores = list(F.softmax(logits, dim=-1).detach().split(nper, dim=0))
        prev_boxes = list(apply_deltas(deltas.detach(), props_cat, cfg.bbox_weights).split(nper, dim=0))
        aux["pgt"].append(pgt)
        aux["labels"].append(gt_classes)
        aux["weights"].append(weights)
        aux["logits"].append(logits)
    return (losses, aux) if return_aux else losses


class _PclLossFn(torch.autograd.Function):
    """wsl/layers/pcl_loss.py:10-93 around fast_rcnn.py:1725-1745 (targets are constants; the gradient reaches the
    logits through the softmax of predict_probs)"""

    @staticmethod
    def forward(ctx, logits, boxes, last, im_labels):
        from . import pcl_oracle as PO

        loss, dl, probs, t = PO.pcl_refine_loss(logits.detach().numpy(), boxes, last, im_labels)
        ctx.save_for_backward(torch.from_numpy(dl))
        return torch.tensor(float(loss), dtype=torch.float32), t, probs

    @staticmethod
    def backward(ctx, g, _t, _p):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk):
    """fast_rcnn.py:88-141. -> (pred_boxes [n,4], scores [n], pred_classes [n], kept_row_idx [n])."""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not valid.all():
        boxes = boxes[valid]
        scores = scores[valid]
    scores = scores[:, :-1]
    nreg = boxes.shape[1] // 4
    b = boxes.reshape(-1, 4).clone()
    h, w = image_shape
    b[:, 0].clamp_(min=0, max=w)
    b[:, 1].clamp_(min=0, max=h)
    b[:, 2].clamp_(min=0, max=w)
    b[:, 3].clamp_(min=0, max=h)
    b = b.view(-1, nreg, 4)
    mask = scores > score_thresh
    inds = mask.nonzero()
    b = b[inds[:, 0], 0] if nreg == 1 else b[mask]
    s = scores[mask]
    keep = batched_nms(b, s, inds[:, 1], nms_thresh)
    if topk >= 0:
        keep = keep[:topk]
    return b[keep], s[keep], inds[keep, 1], inds[keep, 0]


def roi_heads_inference(p, feat, prop_boxes, objectness, image_sizes, cfg: OracleCfg):
    """roi_heads_oicr.py:407-421 + fast_rcnn.py:1444-1474, :1534-1594."""
    K = cfg.num_classes
    nper = [len(b) for b in prop_boxes]
    pooled = pool_features(feat, prop_boxes, cfg)
    obn = torch.cat([o + 1 for o in objectness], dim=0)
    x = dan_forward(p, pooled * obn.view(-1, 1, 1, 1), cfg, False)
    props_cat = torch.cat(prop_boxes, dim=0)
    if cfg.refine_num == 0:
        # WSDDNROIHeads (roi_heads_wsddn.py:305-309): WSDDNOutputLayers.inference, fast_rcnn.py:587-608 - the MIL scores
        # themselves with a zero background column (predict_probs :668-687), zero deltas (predict_boxes :645-666)
        scores = wsddn_scores(p, x, nper)
        probs = torch.cat((scores, torch.zeros(scores.shape[0], 1)), dim=1)
        deltas = torch.zeros(x.shape[0], 4 * K)
    elif cfg.refine_reg[-1]:
        pre = "roi_heads.box_refinery_%d." % (cfg.refine_num - 1)
        probs = F.softmax(F.linear(x, p[pre + "cls_score.weight"], p[pre + "cls_score.bias"]), dim=-1)
        deltas = _head_linear(x, p[pre + "bbox_pred.weight"], p[pre + "bbox_pred.bias"], cfg)
    else:
        probs = None
        deltas = torch.zeros(x.shape[0], 4 * K)
        for k in range(cfg.refine_num):
            pre = "roi_heads.box_refinery_%d." % k
            pk = F.softmax(F.linear(x, p[pre + "cls_score.weight"], p[pre + "cls_score.bias"]), dim=-1)
            probs = pk if probs is None else probs + pk
            # deltas of non-reg heads are zeros (fast_rcnn.py:1377-1386); the mean stays zero
        probs = probs / cfg.refine_num
        if cfg.heads == "pcl":  # pcl_bg, fast_rcnn.py:1463-1465: the branches keep the background in column 0
            probs = torch.cat((probs[:, 1:], probs[:, :1]), dim=1)
    boxes = apply_deltas(deltas, props_cat, cfg.bbox_weights)
    res = []
    for b, s, sz in zip(boxes.split(nper), probs.split(nper), image_sizes):
        res.append(fast_rcnn_inference_single_image(b, s, sz, cfg.score_thresh, cfg.nms_thresh, cfg.topk))
    return res, list(probs.split(nper)), list(boxes.split(nper))


# --------------------------------------------------------------------------------------------
# Whole model + trainer step
# --------------------------------------------------------------------------------------------
def model_train_losses(p, batch, cfg: OracleCfg, dropout_masks=None, return_aux=False):
    """rcnn.py:138-197 (training branch). batch: list of dicts with image [3,H,W], proposal_boxes
    [R,4], objectness_logits [R], gt_classes [G]."""
    x, _ = preprocess_image([b["image"] for b in batch], cfg)
    if cfg.heads == "csc":  # rcnn.py:170-171: images.tensor.requires_grad = True
        x = x.detach().requires_grad_(True)
    feat = backbone_forward(p, x, cfg)
    return roi_heads_train(p, feat, [b["proposal_boxes"] for b in batch], [b["objectness_logits"] for b in batch],
                           [b["gt_classes"] for b in batch], cfg, dropout_masks, return_aux,
                           image_tensor=x if cfg.heads == "csc" else None)


def model_inference(p, batch, cfg: OracleCfg):
    """rcnn.py:199-240 (do_postprocess=False)."""
    x, sizes = preprocess_image([b["image"] for b in batch], cfg)
    feat = backbone_forward(p, x, cfg)
    return roi_heads_inference(p, feat, [b["proposal_boxes"] for b in batch],
                               [b["objectness_logits"] for b in batch], sizes, cfg)


def trainable_names(p, cfg: OracleCfg, freeze_at=5):
    """resnet_ws.py:512-534 / vgg.py:154-206 freeze; FrozenBN buffers are never trainable."""
    names = []
    for n in p:
        if n.endswith(("running_mean", "running_var")) or ".norm." in n:
            continue
        if n.startswith("backbone."):
            parts = n.split(".")
            stage = parts[1]
            idx = 1 if stage == "stem" else int(stage[-1])
            if freeze_at >= idx:
                continue
        if "bbox_pred" in n:
            k = int(n.split("box_refinery_")[1].split(".")[0])
            if not cfg.refine_reg[k]:
                continue  # unused parameter: grad is None in the reference (F10)
        names.append(n)
    return names


class SGDState:
    """torch.optim.SGD as configured by detectron2/solver/build.py:93-137 (per-parameter groups:
    bias lr x BIAS_LR_FACTOR, bias wd = WEIGHT_DECAY_BIAS), momentum buffers created at first step."""

    def __init__(self, cfg: OracleCfg):
        self.cfg = cfg
        self.buf = {}

    def step(self, p, grads, lr_scale=1.0):
        c = self.cfg
        for n, g in grads.items():
            is_bias = n.endswith(".bias")
            lr = c.base_lr * (c.bias_lr_factor if is_bias else 1.0) * lr_scale
            wd = c.weight_decay_bias if is_bias else c.weight_decay
            d = g + wd * p[n] if wd != 0 else g.clone()
            if n not in self.buf:
                self.buf[n] = d.clone()
            else:
                self.buf[n].mul_(c.momentum).add_(d)
            p[n] = p[n] - lr * self.buf[n]


def train_step(p, batch, cfg: OracleCfg, opt: SGDState, dropout_masks=None, freeze_at=5, world_grads=None,
               return_aux=False):
    """projects/WSL/tools/train_net.py:65-117 with ITER_SIZE=1: fwd, sum of losses, bwd, SGD step.
    world_grads: optional hook(grads)->grads emulating the DDP mean all-reduce.
    return_aux: also return roi_heads_train's intermediate values (MIL scores, pseudo-GT, labels) of this step."""
    names = trainable_names(p, cfg, freeze_at)
    leaves = {n: p[n].detach().clone().requires_grad_(True) for n in names}
    q = dict(p)
    q.update(leaves)
    aux = None
    if return_aux:
        losses, aux = model_train_losses(q, batch, cfg, dropout_masks, True)
    else:
        losses = model_train_losses(q, batch, cfg, dropout_masks)
    total = sum(losses.values())
    gl = torch.autograd.grad(total, [leaves[n] for n in names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(p[n])) for n, g in zip(names, gl)}
    if world_grads is not None:
        grads = world_grads(grads)
    opt.step(p, grads)
    out = {k: float(v.detach()) for k, v in losses.items()}
    return (out, grads, aux) if return_aux else (out, grads)


# --------------------------------------------------------------------------------------------
# Parameter construction (names/shapes follow the reference constructors)
# --------------------------------------------------------------------------------------------
def param_shapes(cfg: OracleCfg) -> "Dict[str, Tuple[int, ...]]":
    """state_dict names -> shapes of GeneralizedRCNNWSL for this cfg (resnet_ws.py:357-416, :122-215,
    :616-703; vgg.py:125-231; box_head.py:55-60; fast_rcnn.py:453-461, :1316-1327)."""
    s = {}

    def conv(name, cin, cout, k, bias=False, norm=True):
        s[name + ".weight"] = (cout, cin, k, k)
        if bias:
            s[name + ".bias"] = (cout,)
        if norm:
            for t in ("weight", "bias", "running_mean", "running_var"):
                s[name + ".norm." + t] = (cout,)

    if cfg.arch == "vgg16":
        cprev = 3
        for si, (cout, n) in enumerate([(64, 2), (128, 2), (256, 3), (512, 3), (512, 3)]):
            for i in range(n):
                conv("backbone.plain%d.0.conv%d" % (si + 1, i + 1), cprev, cout, 3, bias=True, norm=False)
                cprev = cout
        cfeat = cprev
    else:
        so = cfg.stem_out
        conv("backbone.stem.conv1", 3, so, 3)
        conv("backbone.stem.conv2", so, so, 3)
        conv("backbone.stem.conv3", so, so, 3)
        cin, cout, bott = so, cfg.res2_out, cfg.width_per_group
        for name, nblk, _, _ in resnet_ws_stage_plan(cfg):
            for b in range(nblk):
                bp = "backbone.%s.%d." % (name, b)
                if cin != cout:
                    conv(bp + "shortcut", cin, cout, 1)
                if cfg.arch == "wsr18":
                    conv(bp + "conv1", cin, cout, 3)
                    conv(bp + "conv2", cout, cout, 3)
                else:
                    conv(bp + "conv1", cin, bott, 1)
                    conv(bp + "conv2", bott, bott, 3)
                    conv(bp + "conv3", bott, cout, 1)
                cin = cout
            cout *= 2
            bott *= 2
        cfeat = cin
    P = cfg.pooler_res
    d1, d2 = cfg.dan_dim
    K = cfg.num_classes
    s["roi_heads.box_head.fc1.weight"] = (d1, cfeat * P * P)
    s["roi_heads.box_head.fc1.bias"] = (d1,)
    s["roi_heads.box_head.fc2.weight"] = (d2, d1)
    s["roi_heads.box_head.fc2.bias"] = (d2,)
    for n in ("cls", "det"):
        s["roi_heads.box_predictor.%s.weight" % n] = (K, d2)
        s["roi_heads.box_predictor.%s.bias" % n] = (K,)
    for k in range(cfg.refine_num):
        pre = "roi_heads.box_refinery_%d." % k
        s[pre + "cls_score.weight"] = (K + 1, d2)
        s[pre + "cls_score.bias"] = (K + 1,)
        s[pre + "bbox_pred.weight"] = (4 * K, d2)
        s[pre + "bbox_pred.bias"] = (4 * K,)
    return s


def seeded_tensor(name: str, shape, seed: int) -> torch.Tensor:
    """Deterministic O(1)-activation weights keyed by NAME (numpy legacy RandomState: a frozen
    stream), so the reference model and this oracle are filled identically without sharing files.
    Scales chosen so the frozen backbone and the MIL head stay out of saturation."""
    import zlib

    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    n = int(np.prod(shape))
    z = rs.standard_normal(n).astype(np.float64)
    leaf = name.rsplit(".", 1)[1]
    if ".norm." in name:
        v = {"weight": 0.85 + 0.1 * np.tanh(z), "bias": 0.05 * z, "running_mean": 0.05 * z,
             "running_var": 1.0 + 0.2 * np.tanh(z)}[leaf]
        if ".conv3.norm.weight" in name:
            v = 0.35 * v  # damp the residual branch so 16-33 blocks stay O(1)
        elif ".conv2.norm.weight" in name and ".res" in name:
            v = 0.6 * v
    elif leaf == "bias":
        v = 0.1 + 0.02 * z if "box_head" in name else 0.02 * z
    elif len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        v = z * math.sqrt((1.7 if ".plain" in name else 2.0) / fan_in)
        if shape[1] == 3:
            v = v / 64.0  # first conv sees mean-subtracted 0..255 pixels (|x| ~ 70)
    else:
        fan_in = shape[1]
        gain = {"fc1": 1.4, "fc2": 1.4, "cls": 4.0, "det": 4.0, "cls_score": 3.0, "bbox_pred": 0.3}[name.split(".")[-2]]
        v = z * (gain / math.sqrt(fan_in))
    return torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape).copy())


def seeded_params(shapes, seed=0) -> Dict[str, torch.Tensor]:
    return {n: seeded_tensor(n, tuple(sh), seed) for n, sh in shapes.items()}


def init_params(cfg: OracleCfg, seed=0) -> Dict[str, torch.Tensor]:
    return seeded_params(param_shapes(cfg), seed)


def synthetic_batch(n_images, R, cfg: OracleCfg, seed=1234, H=224, W=224):
    """SURVEY §8(d) synthetic inputs: uint8-valued image, proposals x0,y0~U[0,W-40), w,h~U[20, W-x0],
    objectness U[0,1) sorted descending, 1..3 distinct GT classes."""
    g = torch.Generator().manual_seed(seed)
    batch = []
    for _ in range(n_images):
        img = torch.randint(0, 256, (3, H, W), generator=g).float()
        x0 = torch.rand(R, generator=g) * (W - 40)
        y0 = torch.rand(R, generator=g) * (H - 40)
        bw = 20 + torch.rand(R, generator=g) * (W - x0 - 20)
        bh = 20 + torch.rand(R, generator=g) * (H - y0 - 20)
        boxes = torch.stack([x0, y0, (x0 + bw).clamp(max=W), (y0 + bh).clamp(max=H)], dim=1)
        obj = torch.sort(torch.rand(R, generator=g), descending=True).values
        G = int(torch.randint(1, 4, (1,), generator=g))
        cls = torch.randperm(cfg.num_classes, generator=g)[:G].to(torch.int64)
        batch.append({"image": img, "proposal_boxes": boxes, "objectness_logits": obj, "gt_classes": cls})
    return batch


# --------------------------------------------------------------------------------------------
# Test-time augmentation.  Restates projects/WSL/wsl/modeling/test_time_augmentation_avg.py
# (DatasetMapperTTAAVG :68-137, transform_proposals :27-65, GeneralizedRCNNWithTTAAVG :139-321) together with the
# transforms it drives: ResizeShortestEdge.get_transform (detectron2/data/transforms/augmentation_impl.py:155-175),
# ResizeTransform (detectron2/data/transforms/transform.py:83-134) and fvcore's HFlipTransform / Transform.apply_box
# (fvcore is external and absent: its published semantics are restated).  Pinned by tests/golden/tta_r50c4_tiny.npz,
# produced by the reference's own GeneralizedRCNNWithTTAAVG.
# --------------------------------------------------------------------------------------------
def tta_resize_shape(h, w, size, max_size):
    """augmentation_impl.py:164-174"""
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh = newh * scale
        neww = neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def tta_resize_image(img_hwc, new_h, new_w):
    """transform.py:101-122: PIL bilinear for uint8, F.interpolate(bilinear, align_corners=False) otherwise"""
    import numpy as np

    if img_hwc.dtype == np.uint8:
        from PIL import Image

        return np.asarray(Image.fromarray(img_hwc).resize((new_w, new_h), Image.BILINEAR))
    t = torch.from_numpy(np.ascontiguousarray(img_hwc)).permute(2, 0, 1)[None]
    t = F.interpolate(t, (new_h, new_w), mode="bilinear", align_corners=False)
    return t[0].permute(1, 2, 0).numpy()


def pil_bilinear_coeffs(in_size, out_size):
    """Pillow's Resample.c (precompute_coeffs with the BILINEAR = triangle filter of support 1, normalize_coeffs_8bpc), restated
    loop for loop: -> (bounds [(first source position, count)] per output position, integer coefficients scaled by 2^22).
    Pillow is an un-vendored dependency of the reference (detectron2/data/transforms/transform.py:101-122 calls
    PIL.Image.resize(..., BILINEAR)); pinned against the installed Pillow and against the reference-generated augmented
    images of tests/golden/tta_r50c4_tiny.npz (tests/test_oracle_golden.py)."""
    import math

    scale = float(in_size) / out_size
    fscale = max(scale, 1.0)
    support = 1.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / fscale
    bounds, coef = [], []
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k, ww = [0.0] * ksize, 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            k[x] = w
            ww += w
        if ww != 0.0:
            for x in range(xmax):
                k[x] /= ww
        bounds.append((xmin, xmax))
        coef.append([int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22)) for v in k])
    return bounds, coef


def pil_bilinear_resize_u8(img_hwc, new_h, new_w):
    """PIL.Image.fromarray(img).resize((new_w, new_h), BILINEAR) for an 8-bit [H, W, C] image without Pillow: horizontal pass
    (result rounded and clipped to 8 bits), then vertical pass (ImagingResample, 8bpc paths)."""
    import numpy as np

    def one_pass(x, axis, n_out):
        n_in = x.shape[axis]
        if n_in == n_out:
            return x
        bounds, coef = pil_bilinear_coeffs(n_in, n_out)
        x = np.moveaxis(x, axis, 0)
        out = np.empty((n_out,) + x.shape[1:], dtype=np.int64)
        for i, ((lo, n), k) in enumerate(zip(bounds, coef)):
            acc = np.full(x.shape[1:], 1 << 21, dtype=np.int64)
            for j in range(n):
                acc += x[lo + j] * k[j]
            out[i] = np.clip(acc >> 22, 0, 255)
        return np.moveaxis(out, 0, axis)

    x = np.asarray(img_hwc).astype(np.int64)
    x = one_pass(x, 1, new_w)
    x = one_pass(x, 0, new_h)
    return x.astype(np.uint8)


def tta_apply_box(boxes, steps):
    """Transform.apply_box through a TransformList: the 4 corners go through every transform's apply_coords (float32
    array x python float, i.e. float32 arithmetic), the result is their axis-aligned bounding box.
    steps: ("scale", sx, sy) = ResizeTransform.apply_coords (transform.py:124-127), ("hflip", W) = HFlipTransform."""
    import numpy as np

    b = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    idx = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
    c = b[:, idx].reshape(-1, 2).copy()
    for st in steps:
        if st[0] == "scale":
            c[:, 0] = c[:, 0] * st[1]
            c[:, 1] = c[:, 1] * st[2]
        elif st[0] == "hflip":
            c[:, 0] = st[1] - c[:, 0]
        else:
            raise ValueError(st)
    c = c.reshape(-1, 4, 2)
    return np.concatenate((c.min(axis=1), c.max(axis=1)), axis=1)


def tta_augment(image_chw, proposal_boxes, objectness, min_sizes, max_size, flip, topk):
    """DatasetMapperTTAAVG.__call__ + transform_proposals: one dict per (size[, flip]) with the resized (flipped)
    image [3,h,w], the transformed / clipped / non-empty / top-k proposals, and the forward and inverse box steps."""
    import numpy as np

    img = image_chw.permute(1, 2, 0).numpy()
    h, w = img.shape[:2]
    out = []
    for size in min_sizes:
        nh, nw = tta_resize_shape(h, w, size, max_size)
        rimg = tta_resize_image(np.copy(img), nh, nw)
        fwd = [("scale", nw * 1.0 / w, nh * 1.0 / h)]
        inv = [("scale", w * 1.0 / nw, h * 1.0 / nh)]
        variants = [(rimg, fwd, inv)]
        if flip:
            variants.append((np.flip(rimg, axis=1), fwd + [("hflip", nw)], [("hflip", nw)] + inv))
        for im, f, iv in variants:
            bx = torch.from_numpy(tta_apply_box(proposal_boxes.numpy(), f))
            bx[:, 0].clamp_(min=0, max=nw)
            bx[:, 1].clamp_(min=0, max=nh)
            bx[:, 2].clamp_(min=0, max=nw)
            bx[:, 3].clamp_(min=0, max=nh)
            keep = ((bx[:, 2] - bx[:, 0]) > 0) & ((bx[:, 3] - bx[:, 1]) > 0)
            out.append({"image": torch.from_numpy(np.ascontiguousarray(im.transpose(2, 0, 1))),
                        "proposal_boxes": bx[keep][:topk], "objectness_logits": objectness[keep][:topk], "inverse": iv})
    return out


def tta_inference(p, image_chw, proposal_boxes, objectness, orig_hw, cfg: OracleCfg, min_sizes, max_size, flip=True,
                  topk=1000, return_aux=False):
    """GeneralizedRCNNWithTTAAVG._inference_one_image (box branch): per-augmentation inference, boxes mapped back to the
    original image and averaged, scores averaged, then fast_rcnn_inference_single_image on the averages."""
    augs = tta_augment(image_chw, proposal_boxes, objectness, min_sizes, max_size, flip, topk)
    all_boxes, all_scores = [], []
    for a in augs:
        _, scores, boxes = model_inference(p, [a], cfg)
        nb = boxes[0]
        back = tta_apply_box(nb.reshape(-1, 4).numpy(), a["inverse"])
        all_boxes.append(torch.from_numpy(back).reshape(1, *nb.shape))
        all_scores.append(scores[0][None])
    avg_boxes = torch.mean(torch.cat(all_boxes, dim=0), dim=0)
    avg_scores = torch.mean(torch.cat(all_scores, dim=0), dim=0)
    det = fast_rcnn_inference_single_image(avg_boxes, avg_scores, orig_hw, cfg.score_thresh, cfg.nms_thresh, cfg.topk)
    return (det, augs, avg_boxes, avg_scores) if return_aux else det
