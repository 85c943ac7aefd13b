"""ROI heads of the DRN-WSOD / OICR path behind the reference's names and constructor arguments:

  ROIPooler                    detectron2/modeling/poolers.py:99-246
  Matcher                      detectron2/modeling/matcher.py:8-126
  Box2BoxTransform             detectron2/modeling/box_regression.py:17-110
  DiscriminativeAdaptionNeck   projects/WSL/wsl/modeling/roi_heads/box_head.py:14-103
  WSDDNOutputLayers            projects/WSL/wsl/modeling/roi_heads/fast_rcnn.py:396-700
  OICROutputLayers             projects/WSL/wsl/modeling/roi_heads/fast_rcnn.py:1267-1594
  ROIHeads / OICRROIHeads / WSDDNROIHeads
                               projects/WSL/wsl/modeling/roi_heads/{roi_heads.py:156-353, roi_heads_oicr.py:33-567,
                               roi_heads_wsddn.py}

Same module tree => same state_dict keys (roi_heads.box_head.fc1.weight, roi_heads.box_predictor.cls.weight,
roi_heads.box_refinery_0.cls_score.weight, ...).  The arithmetic does not run module by module: OICRROIHeads
drives one fused schedule of HIP kernels (`_HeadEngine`) — ROIPool fused with the objectness scaling writes
the fc6 operand, three MFMA GEMM groups (fc6, fc7, all predictor Linears concatenated), the MIL/OICR kernels,
and an explicit backward (dW/dX GEMMs) that writes gradients straight into a flat fp32 arena shared with the
fused SGD kernel and the gradient all-reduce.  No host synchronisation happens inside a step."""
import inspect
import math
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
from torch import nn

from .. import compute_dtype, ops
from .._cabi import DrnError
from ..config import configurable
from ..events import get_event_storage, has_event_storage
from ..layers import Linear, ROIAlign, RoIPool, ShapeSpec, to_nhwc
from ..registry import ROI_BOX_HEAD_REGISTRY, ROI_HEADS_REGISTRY
from ..structures import Boxes, Instances


# ------------------------------------------------------------------------------------------------
class Matcher:
    """Config holder with the reference's constructor checks (matcher.py:24-59).  The matching itself
    (column arg-max over GT + threshold labelling, no low-quality matches on this path) runs inside
    drn_oicr_targets."""

    def __init__(self, thresholds: List[float], labels: List[int], allow_low_quality_matches: bool = False):
        thresholds = thresholds[:]
        assert thresholds[0] > 0
        thresholds.insert(0, -float("inf"))
        thresholds.append(float("inf"))
        assert all(low <= high for (low, high) in zip(thresholds[:-1], thresholds[1:]))
        assert all(l in [-1, 0, 1] for l in labels)
        assert len(labels) == len(thresholds) - 1
        if allow_low_quality_matches:
            raise DrnError("allow_low_quality_matches is off the OICR path (roi_heads.py:207-211 passes False)")
        self.thresholds = thresholds
        self.labels = labels
        self.allow_low_quality_matches = allow_low_quality_matches


class Box2BoxTransform:
    """box_regression.py:17-110 (weights + scale clamp; apply_deltas executes drn_apply_deltas)."""

    def __init__(self, weights, scale_clamp: float = math.log(1000.0 / 16)):
        self.weights = tuple(weights)
        self.scale_clamp = scale_clamp

    def apply_deltas(self, deltas, boxes):
        k = deltas.shape[1] // 4
        return ops.apply_deltas(deltas.float().contiguous(), boxes.float().contiguous(), k, self.weights)


def convert_boxes_to_pooler_format(box_lists: List[Boxes]):
    """poolers.py:62-96 -> [M,5] (batch index, x0, y0, x1, y1)."""
    return torch.cat([torch.cat((torch.full((len(b), 1), float(i), dtype=b.tensor.dtype, device=b.tensor.device),
                                 b.tensor), dim=1) for i, b in enumerate(box_lists)], dim=0)


class ROIPooler(nn.Module):
    """poolers.py:99-246, single feature level (the only case on this path)."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224, canonical_level=4):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        assert len(output_size) == 2 and output_size[0] == output_size[1]
        self.output_size = output_size
        if len(scales) != 1:
            raise DrnError("multi-level (FPN) pooling is off the DRN-WSOD path")
        self.pooler_type, self.sampling_ratio, self.scales = pooler_type, sampling_ratio, tuple(scales)
        if pooler_type == "ROIAlign":
            self.level_poolers = nn.ModuleList(ROIAlign(output_size, s, sampling_ratio, aligned=False) for s in scales)
        elif pooler_type == "ROIAlignV2":
            self.level_poolers = nn.ModuleList(ROIAlign(output_size, s, sampling_ratio, aligned=True) for s in scales)
        elif pooler_type == "ROIPool":
            self.level_poolers = nn.ModuleList(RoIPool(output_size, spatial_scale=s) for s in scales)
        else:
            raise ValueError("Unknown pooler type: {}".format(pooler_type))
        min_level = -(math.log2(scales[0]))
        assert math.isclose(min_level, int(min_level)), "Featuremap stride is not power of 2!"

    def kernel_args(self):
        mode = 0 if self.pooler_type == "ROIPool" else 1
        return dict(P=self.output_size[0], scale=self.scales[0], mode=mode, sampling_ratio=self.sampling_ratio,
                    aligned=self.pooler_type == "ROIAlignV2")

    def forward(self, x: List[torch.Tensor], box_lists: List[Boxes]):
        assert isinstance(x, list) and isinstance(box_lists, list), "Arguments to pooler must be lists"
        assert len(x) == 1 and len(box_lists) == x[0].size(0)
        return self.level_poolers[0](x[0], convert_boxes_to_pooler_format(box_lists))


# ------------------------------------------------------------------------------------------------
@ROI_BOX_HEAD_REGISTRY.register()
class DiscriminativeAdaptionNeck(nn.Module):
    """box_head.py:14-103: flatten -> fc1 -> ReLU -> dropout(0.5) -> fc2 -> ReLU -> dropout(0.5)."""

    @configurable
    def __init__(self, input_shape: ShapeSpec, *, conv_dims: List[int], fc_dims: List[int], conv_norm=""):
        super().__init__()
        assert len(conv_dims) + len(fc_dims) > 0
        if len(conv_dims):
            raise DrnError("DAN conv layers (NUM_CONV > 0) are not used by any DRN-WSOD config")
        self._output_size = (input_shape.channels, input_shape.height, input_shape.width)
        self.conv_norm_relus = []
        self.fcs = []
        for k, fc_dim in enumerate(fc_dims):
            size = self._output_size if isinstance(self._output_size, int) else int(
                self._output_size[0] * self._output_size[1] * self._output_size[2])
            fc = Linear(size, fc_dim)
            self.add_module("fc{}".format(k + 1), fc)
            self.fcs.append(fc)
            self._output_size = fc_dim
        for layer in self.fcs:
            torch.nn.init.normal_(layer.weight, std=0.005)
            torch.nn.init.constant_(layer.bias, 0.1)
        self.dropout_p = 0.5          # box_head.py:90 hard-codes p=0.5
        self.dropout_masks = None     # test hook: explicit [M, fc_dim] multiplier masks

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {"input_shape": input_shape, "conv_dims": [cfg.MODEL.ROI_BOX_HEAD.CONV_DIM] * cfg.MODEL.ROI_BOX_HEAD.NUM_CONV,
                "fc_dims": cfg.MODEL.ROI_BOX_HEAD.DAN_DIM, "conv_norm": cfg.MODEL.ROI_BOX_HEAD.NORM}

    @property
    def output_shape(self):
        o = self._output_size
        return ShapeSpec(channels=o) if isinstance(o, int) else ShapeSpec(channels=o[0], height=o[1], width=o[2])

    def forward(self, x):
        raise DrnError("DiscriminativeAdaptionNeck runs inside the fused OICRROIHeads schedule; call the ROI heads")


def build_box_head(cfg, input_shape):
    return ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, input_shape)


class _OutputLayersBase(nn.Module):
    def _common(self, box2box_transform, test_score_thresh, test_nms_thresh, test_topk_per_image, smooth_l1_beta,
                box_reg_loss_type, loss_weight, mean_loss):
        self.box2box_transform = box2box_transform
        self.smooth_l1_beta = smooth_l1_beta
        self.test_score_thresh = test_score_thresh
        self.test_nms_thresh = test_nms_thresh
        self.test_topk_per_image = test_topk_per_image
        self.box_reg_loss_type = box_reg_loss_type
        if isinstance(loss_weight, float):
            loss_weight = {"loss_cls": loss_weight, "loss_box_reg": loss_weight}
        self.loss_weight = loss_weight
        self.mean_loss = mean_loss

    @staticmethod
    def _cfg_common(cfg):
        return {
            "box2box_transform": Box2BoxTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS),
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "cls_agnostic_bbox_reg": cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG,
            "smooth_l1_beta": cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA,
            "test_score_thresh": cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
            "test_nms_thresh": cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
            "test_topk_per_image": cfg.TEST.DETECTIONS_PER_IMAGE,
            "box_reg_loss_type": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE,
            "loss_weight": {"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT},
            "mean_loss": cfg.WSL.MEAN_LOSS,
        }


class WSDDNOutputLayers(_OutputLayersBase):
    """fast_rcnn.py:396-700: `cls` and `det` Linear(input, K), Xavier-uniform, zero bias."""

    @configurable
    def __init__(self, input_shape, *, box2box_transform, num_classes, test_score_thresh=0.0, test_nms_thresh=0.5,
                 test_topk_per_image=100, cls_agnostic_bbox_reg=False, smooth_l1_beta=0.0,
                 box_reg_loss_type="smooth_l1", loss_weight=1.0, mean_loss=True):
        super().__init__()
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.num_bbox_reg_classes = 1 if cls_agnostic_bbox_reg else num_classes
        self.box_dim = len(box2box_transform.weights)
        self.num_classes = num_classes
        self.cls = Linear(input_size, num_classes)
        self.det = Linear(input_size, num_classes)
        nn.init.xavier_uniform_(self.cls.weight)
        nn.init.xavier_uniform_(self.det.weight)
        for l in [self.cls, self.det]:
            nn.init.constant_(l.bias, 0)
        self._common(box2box_transform, test_score_thresh, test_nms_thresh, test_topk_per_image, smooth_l1_beta,
                     box_reg_loss_type, loss_weight, mean_loss)

    @classmethod
    def from_config(cls, cfg, input_shape):
        d = cls._cfg_common(cfg)
        d["input_shape"] = input_shape
        return d


class OICROutputLayers(_OutputLayersBase):
    """fast_rcnn.py:1267-1594: `cls_score` Linear(input, K+1) (std 0.01) and `bbox_pred` Linear(input, 4K)
    (std 0.001; only used when WSL.REFINE_REG[k])."""

    @configurable
    def __init__(self, input_shape, *, box2box_transform, num_classes, test_score_thresh=0.0, test_nms_thresh=0.5,
                 test_topk_per_image=100, cls_agnostic_bbox_reg=False, smooth_l1_beta=0.0,
                 box_reg_loss_type="smooth_l1", loss_weight=1.0, mean_loss=True, refine_k=None, refine_reg=False):
        super().__init__()
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        if cls_agnostic_bbox_reg:
            raise DrnError("class-agnostic box regression is not used by any DRN-WSOD config")
        self.num_classes = num_classes
        self.cls_score = Linear(input_size, num_classes + 1)
        self.num_bbox_reg_classes = num_classes
        self.box_dim = len(box2box_transform.weights)
        self.bbox_pred = Linear(input_size, self.num_bbox_reg_classes * self.box_dim)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in [self.cls_score, self.bbox_pred]:
            nn.init.constant_(l.bias, 0)
        self._common(box2box_transform, test_score_thresh, test_nms_thresh, test_topk_per_image, smooth_l1_beta,
                     box_reg_loss_type, loss_weight, mean_loss)
        self.refine_k = refine_k
        self.refine_reg = refine_reg

    @classmethod
    def from_config(cls, cfg, input_shape, refine_k):
        d = cls._cfg_common(cfg)
        d.update(input_shape=input_shape, refine_reg=cfg.WSL.REFINE_REG, refine_k=refine_k)
        return d


# ------------------------------------------------------------------------------------------------
class _TrainFn(torch.autograd.Function):
    """Hooks the fused schedule into autograd: `sum(loss_dict.values()).backward()` (the reference trainer,
    projects/WSL/tools/train_net.py:92-107) lands here with one upstream gradient per loss."""

    @staticmethod
    def forward(ctx, anchor, engine, state):
        ctx.engine, ctx.state = engine, state
        return tuple(state["loss_list"])

    @staticmethod
    def backward(ctx, *gouts):
        ctx.engine.backward(ctx.state, gouts)
        return None, None, None


class _HeadEngine:
    """Flat parameter arena + the fused fwd/bwd kernel schedule of the DAN / WSDDN / OICR heads."""

    def __init__(self, heads):
        self.h = heads
        self.arena_w = None
        self.arena_g = None
        self.segments = []       # (name, param, offset, numel, used)
        self._shadow_key = None
        self._dirty = True
        self._ws = {}
        self._grads_valid = False
        self._drop_iter = 0
        self.anchor = None

    # ---- layout --------------------------------------------------------------------------------
    def _layout(self):
        h = self.h
        K = h.num_classes
        cols, off = [], 0
        cols.append(("cls", h.box_predictor.cls, off, K)); off += K
        cols.append(("det", h.box_predictor.det, off, K)); off += K
        for k in range(h.refine_K):
            cols.append(("r%d" % k, h.box_refinery[k].cls_score, off, K + 1)); off += K + 1
        for k in range(h.refine_K):
            if h.refine_reg[k]:
                cols.append(("b%d" % k, h.box_refinery[k].bbox_pred, off, 4 * K)); off += 4 * K
        return cols, off

    def ensure(self, device):
        h = self.h
        fc1 = h.box_head.fc1
        if (self.arena_w is not None and self.arena_w.device == device and
                fc1.weight.data_ptr() == self.arena_w.data_ptr() + 4 * self._fc1_off):
            return
        cols, NH = self._layout()
        self.cols, self.NH = cols, NH
        order = [(n + ".weight", m.weight, True) for n, m, _, _ in cols]
        order += [(n + ".bias", m.bias, True) for n, m, _, _ in cols]
        order += [("fc2.weight", h.box_head.fc2.weight, True), ("fc2.bias", h.box_head.fc2.bias, True),
                  ("fc1.bias", fc1.bias, True), ("fc1.weight", fc1.weight, True)]
        for k in range(h.refine_K):
            if not h.refine_reg[k]:  # unused parameters: kept in the arena, never touched
                order += [("u%d.weight" % k, h.box_refinery[k].bbox_pred.weight, False),
                          ("u%d.bias" % k, h.box_refinery[k].bbox_pred.bias, False)]
        # the concatenated head weights / biases must stay contiguous (they are ONE GEMM operand); every other
        # segment starts on a 64-element boundary = one 128-byte line of the bf16 shadow arena (256 B of the fp32 one):
        # the GEMMs fetch operands in 128-byte K-slabs per row, and a row that starts mid-line makes every slab straddle
        # two lines (measured on the fc6 forward GEMM: 392 us with a 16-byte aligned W1 vs 340 us line-aligned)
        contiguous = {n + ".weight" for n, _, _, _ in cols[1:]} | {n + ".bias" for n, _, _, _ in cols[1:]}
        offs, off = [], 0
        for name, p, used in order:
            if name not in contiguous:
                off = (off + 63) // 64 * 64
            offs.append(off)
            off += p.numel()
        total = (off + 63) // 64 * 64
        w = torch.zeros((total,), dtype=torch.float32, device=device)
        g = torch.zeros((total,), dtype=torch.float32, device=device)
        self.segments = []
        for (name, p, used), off in zip(order, offs):
            n = p.numel()
            w[off: off + n].copy_(p.detach().reshape(-1).to(device))
            p.data = w[off: off + n].view(p.shape)
            p.grad = None
            self.segments.append((name, p, off, n, used))
            if name == "fc1.weight":
                self._fc1_off = off
        self.arena_s = None  # bf16 shadow of the arena (same flat layout), refreshed by the fused SGD kernel
        self._shadow_from_sgd = False
        self.n_used = sum(n for _, _, _, n, u in self.segments if u)
        self.arena_w, self.arena_g = w, g
        self._seg = {name: (o, n) for name, _, o, n, _ in self.segments}
        self._dirty = True
        self._ws = {}
        self._grads_valid = False
        self.anchor = torch.zeros((), device=device, requires_grad=True)

    def _gview(self, name, shape=None):
        o, n = self._seg[name]
        v = self.arena_g[o: o + n]
        return v.view(shape) if shape is not None else v

    def _wview(self, name, shape=None):
        o, n = self._seg[name]
        v = self.arena_w[o: o + n]
        return v.view(shape) if shape is not None else v

    def mark_dirty(self, shadow_fresh=False):
        """weights changed; shadow_fresh: the fused SGD kernel already rewrote the flat bf16 shadow"""
        self._dirty = True
        self._shadow_from_sgd = shadow_fresh

    def refresh_transposes(self):
        """K-major twins of fc7 / predictor weights for the dX GEMMs.  The pipelined optimizer calls this on its own
        stream right behind the small-tensor SGD bucket (under the fc6 dW GEMMs) and sets `_transposes_fresh`, which
        takes the two launches off the front of the next forward."""
        h, sh = self.h, self.sh
        fc1, fc2 = h.box_head.fc1, h.box_head.fc2
        D1, D2, NH = fc1.weight.shape[0], fc2.weight.shape[0], self.NH
        o, _ = self._seg[self.cols[0][0] + ".weight"]
        # a shadow that is a view of the flat compute-dtype arena is refreshed by the SGD kernel itself (or by
        # refresh_shadows just before this call): transposing IT is a bf16 -> bf16 copy with 16-B accesses (~10 us)
        # instead of a scalar fp32 -> bf16 pass over the master weights (100 us beside the dW GEMM, on the optimizer
        # stream's chain small SGD -> transposes -> fc6 SGD slabs that ends the step)
        src2 = fc2.weight.data if sh.get("W2_own", True) else sh["W2"]
        srch = self.arena_w[o: o + NH * D2].view(NH, D2) if sh.get("Wh_own", True) else sh["Wh"]
        ops.transpose2d(src2, D2, D1, out=sh["W2T"])
        ops.transpose2d(srch, NH, D2, out=sh["WhT"])

    # ---- compute-dtype shadows of the weights -----------------------------------------------------
    def refresh_shadows(self, dtype):
        key = (dtype,) + tuple(p._version for _, p, _, _, u in self.segments if u)
        if not self._dirty and key == self._shadow_key:
            return
        h = self.h
        dev = self.arena_w.device
        fc1, fc2 = h.box_head.fc1, h.box_head.fc2
        D1, K1 = fc1.weight.shape
        D2 = fc2.weight.shape[0]
        NH = self.NH
        kp = lambda k: ops.kpad(k, dtype)
        o, _ = self._seg[self.cols[0][0] + ".weight"]
        wh = self.arena_w[o: o + NH * D2].view(NH, D2)
        if not hasattr(self, "sh") or self.sh.get("dtype") != dtype:
            z = lambda r, c: torch.zeros((r, c), dtype=dtype, device=dev)
            self.sh = dict(dtype=dtype, W2T=z(D1, kp(D2)), WhT=z(D2, kp(NH)))
            # operands whose padded layout equals the master layout live in the flat shadow arena (bf16) or ARE the
            # master (fp32); only padded ones get their own buffer
            self.arena_s = torch.zeros_like(self.arena_w, dtype=torch.bfloat16) if dtype == torch.bfloat16 else None
            self._shadow_from_sgd = False
            flat = self.arena_s if dtype == torch.bfloat16 else self.arena_w
            for nm, pnt, rows, cols_ in (("W1", fc1.weight, D1, K1), ("W2", fc2.weight, D2, D1), ("Wh", wh, NH, D2)):
                if kp(cols_) == cols_:
                    so = (pnt.data_ptr() - self.arena_w.data_ptr()) // 4
                    self.sh[nm], self.sh[nm + "_own"] = flat[so: so + rows * cols_].view(rows, cols_), False
                else:
                    self.sh[nm], self.sh[nm + "_own"] = z(rows, kp(cols_)), True
        sh = self.sh
        for nm, src, rows, cols_ in (("W1", fc1.weight.data, D1, K1), ("W2", fc2.weight.data, D2, D1), ("Wh", wh, NH, D2)):
            if sh[nm + "_own"] or (dtype == torch.bfloat16 and not self._shadow_from_sgd):
                ops.cast2d(src, rows, cols_, sh[nm])
        sh["W1v"] = sh["W1"]
        if not getattr(self, "_transposes_fresh", False):
            self.refresh_transposes()
        self._transposes_fresh = False
        self._shadow_key, self._dirty, self._shadow_from_sgd = key, False, False

    # ---- workspaces ---------------------------------------------------------------------------------
    def ws(self, M, dtype, training):
        """Activation / gradient workspaces for M proposals.  The buffers are allocated for a CAPACITY (the largest M seen,
        rounded up to 64; grow-only) and handed out as views for the M of this batch: real data has another proposal count
        every step, and re-allocating (and zero-filling) ~150 MB of workspaces per step made the eager step allocator-
        bound.  K-role buffers - the transposed twins, whose columns M .. kpad(M) are read by a GEMM as zero padding -
        get that pad re-zeroed whenever M changes (a narrow strided fill), so a view never shows a previous batch's
        columns."""
        key = (dtype, training)
        base = self._ws.get(key)
        kp = lambda k: ops.kpad(k, dtype)
        if base is None or M > base["cap"]:
            h = self.h
            dev = self.arena_w.device
            D1, K1 = h.box_head.fc1.weight.shape
            D2 = h.box_head.fc2.weight.shape[0]
            cap = (M + 63) // 64 * 64
            z = lambda r, c, dt=dtype: torch.zeros((r, c), dtype=dt, device=dev)
            NHp = kp(self.NH)
            b = dict(H1=z(cap, kp(D1)), H2=z(cap, kp(D2)), logits=z(cap, NHp, torch.float32))
            if training:
                Cp = kp(cap)
                b.update(H1T=z(D1, Cp), H2T=z(D2, Cp), dlogits=z(cap, NHp, torch.float32), dS=z(cap, NHp),
                         dST=z(self.NH, Cp), dH2=z(cap, D2, torch.float32), dP2=z(cap, kp(D2)), dP2T=z(D2, Cp),
                         dH1=torch.zeros((1, cap, D1), dtype=torch.float32, device=dev), dP1T=z(D1, Cp),
                         colpart=z((cap + 63) // 64, max(D1, D2, self.NH), torch.float32),
                         # one scratch per layer for the deferred mode (the three reductions then run later, together)
                         colpart3=[z((cap + 63) // 64, n_, torch.float32) for n_ in (self.NH, D2, D1)])
            base = self._ws[key] = dict(cap=cap, bufs=b, M=None, views=None)
        if base["M"] != M:
            b, Mp = base["bufs"], kp(M)
            v = {}
            for name, t in b.items():
                if name in self._WS_T:          # [X, kpad(cap)] transposed twins: columns of this batch + zero pad
                    v[name] = t[:, :Mp]
                    if Mp > M and base["M"] is not None:
                        t[:, M:Mp].zero_()
                elif name == "dH1":             # [splits, cap, D1]
                    v[name] = t[:, :M]
                elif name == "colpart":
                    v[name] = t[: (M + 63) // 64]
                elif name == "colpart3":
                    v[name] = [c[: (M + 63) // 64] for c in t]
                else:                           # [cap, X] row-major
                    v[name] = t[:M]
            base["M"], base["views"] = M, v
        return base["views"]

    _WS_T = ("H1T", "H2T", "dST", "dP2T", "dP1T")

    def pool(self, feat_nhwc, rois, objectness, training, slot=None, want_argmax=False, prefetch=False):
        """ROIPool/ROIAlign fused with the objectness scaling -> fc6 operand A [M, C*P*P] (+ A^T for the dW GEMM when
        training), into one of two buffer sets so that the NEXT batch can be pooled on a side stream (prefetch=True)
        while the current batch's forward/backward still reads its own set.  A set is 'pending' from its prefetch until
        a forward consumes it, 'current' from then until the next forward; a new call never takes a pending set, and
        takes the current one only when the other is pending AND the current batch's backward has been issued (the
        trainer's order: prefetch of batch t+1 in front of forward t).  `slot` pins the set (GraphedTrainStep)."""
        h = self.h
        dev, dtype = feat_nhwc.device, feat_nhwc.dtype
        self.ensure(dev)
        M = rois.shape[0]
        K1 = h.box_head.fc1.weight.shape[1]
        key = (dtype, training)
        if getattr(self, "_pool_key", None) != key:
            # one pair of operand sets per (dtype, training), like ws(): an inference pass between two training steps
            # (EvalHook) must never replace - i.e. free - the training sets, whose addresses a captured heads graph and
            # the eagerly issued pooling piece of GraphedTrainStep both hold
            by_key = self.__dict__.setdefault("_pool_by_key", {})
            if getattr(self, "_pool_key", None) is not None:
                by_key[self._pool_key] = (self._pool_sets, self._pool_cap, self._pool_current_done)
            self._pool_sets, self._pool_cap, self._pool_current_done = by_key.get(key, (None, 0, True))
            self._pool_key = key
        cap = self._pool_cap
        if self._pool_sets is None or M > cap:
            # capacity-based like ws(): the operand pair of the largest batch seen, views for this batch
            own = getattr(self, "pool_sets_pin_owner", None)
            if getattr(self, "pool_sets_pinned", None) == key and self._pool_sets is not None and (own is None or own() is not None):
                raise DrnError("the fc6 operand sets of %s are pinned by a captured step (GraphedTrainStep) and cannot "
                               "grow from %d to %d proposals" % (key, cap, M))
            cap = (M + 63) // 64 * 64
            z = lambda r_, c_: torch.zeros((r_, c_), dtype=dtype, device=dev)
            self._pool_sets = [dict(A_buf=z(cap, ops.kpad(K1, dtype)), AT_buf=z(K1, ops.kpad(cap, dtype)) if training else None,
                                    state="free", M=None) for _ in range(2)]
            self._pool_cap = cap
            self._pool_current_done = True
        if slot is None:
            free = [i for i, q in enumerate(self._pool_sets) if q["state"] == "free"]
            cur = [i for i, q in enumerate(self._pool_sets) if q["state"] == "current"]
            if free:
                slot = free[0]
            elif cur and (not training or self._pool_current_done):
                slot = cur[0]
            else:
                raise DrnError("no free fc6-operand buffer set: at most one batch can be pooled ahead of the one in flight")
        s = self._pool_sets[slot]
        if s["M"] != M:
            Mp = ops.kpad(M, dtype)
            s["A"] = s["A_buf"][:M]
            s["AT"] = s["AT_buf"][:, :Mp] if s["AT_buf"] is not None else None
            if s["AT"] is not None and Mp > M and s["M"] is not None:
                s["AT_buf"][:, M:Mp].zero_()  # K-role pad of the dW GEMM: never a previous batch's columns
            s["M"] = M
        if prefetch:
            s["state"] = "pending"
        else:
            self._mark_current(s)
        ka = h.box_pooler.kernel_args()
        # round 3: in the bf16 mode the fc6 weight gradient reads A itself (drn_gemm_tn); of A^T only the rows of the columns
        # its tail-balancing launch peels off are still needed (run_fc1_tail) - the pooling launch skips the rest
        s["t_row0"] = self._fc1_tail_row0(dtype, h.box_head.fc1.weight.shape[0], K1) if s["AT"] is not None else 0
        pp = ka["P"] * ka["P"] if "P" in ka else 49
        res = ops.roi_pool_nhwc(feat_nhwc, rois, objectness, out=s["A"], out_t=s["AT"],
                                want_argmax=want_argmax and ka["mode"] == 0, t_first_channel=s["t_row0"] // pp, **ka)
        s["argmax"] = res[1] if (want_argmax and ka["mode"] == 0) else None
        return s

    # ---- fc6 weight gradient: row slabs, tail balancing, TN operand -------------------------------------------------
    fc1_tn = True  # bf16: dW = dP1^T . A through drn_gemm_tn (no materialised A^T); False keeps the NT form on A^T (A/B)

    def _fc1_slabs(self, D1):
        ends = getattr(self, "fc1_slab_ends", None)
        if ends is None:
            nslab = getattr(self, "fc1_grad_slabs", 1)
            rows = (D1 + nslab - 1) // nslab
            ends = [min(D1, (s_ + 1) * rows) for s_ in range(nslab)]
        return [(a, b) for a, b in zip([0] + list(ends[:-1]), ends) if a < b]

    def _fc1_use_tn(self, dtype):
        return (self.fc1_tn and dtype == torch.bfloat16
                and self._tn_selfcheck())

    _tn_ok = None

    @classmethod
    def _tn_selfcheck(cls):
        """First use, once per process: drn_gemm_tn's transposing LDS reads are inline asm with hand-counted
        `lgkmcnt` waits that the compiler does not see; another hipcc version or register allocation could read a fragment
        before it has landed and produce silently wrong gradients.  One small TN launch (the persistent ping-pong kernel:
        288 tiles, ragged K rows) against drn_gemm_nt on the materialised transpose - bit-identical by construction -
        decides whether the TN form is used; on a mismatch the NT form on A^T takes over, loudly."""
        if cls._tn_ok is None:
            if torch.cuda.is_current_stream_capturing():
                return True  # decided by the eager priming step that precedes every capture
            g = torch.Generator(device="cpu").manual_seed(11)
            M, N, K, kb = 2304, 2048, 192, 150
            a = torch.randn((M, K), generator=g).to("cuda", torch.bfloat16)
            bt = torch.randn((kb, N), generator=g).to("cuda", torch.bfloat16)
            b = torch.zeros((N, K), dtype=torch.bfloat16, device="cuda")
            b[:, :kb] = bt.t()
            a[:, kb:] = 1.0  # the K padding of Bt reads as zeros whatever A holds there
            ok = torch.equal(ops.gemm_tn(a, bt, M, N, K, kb), ops.gemm_nt(a, b, M, N, K))
            if not ok:
                import warnings

                warnings.warn("drn_gemm_tn failed its first-use self-check against drn_gemm_nt on this toolchain: the fc6 "
                              "weight gradient falls back to the NT form on a materialised A^T (fc1_tn = False)")
            cls._tn_ok = bool(ok)
        return cls._tn_ok

    def _fc1_col_plan(self, dtype, D1, K1):
        """(n_main, slab width in columns) of the fused dW + SGD launch (drn_gemm_tn_sgd), or None when it does not apply: TN
        operand form, a bf16 gradient bucket, whole 256-row tiles.  (Round 4 also ran the unfused dW in column slabs of exact
        rounds with a block update per slab - measured neutral twice, profiles/r4_01_col_slabs_ab.txt - removed in round 5.)"""
        if getattr(self, "fc1_fused_tn", None) is not None and D1 % 256 == 0:
            # ONE slab of all exact rounds, the rest are the trailing columns
            tiles_m = D1 // 256
            ncu = getattr(self, "_ncu", None)
            if ncu is None:
                ncu = self._ncu = torch.cuda.get_device_properties(self.arena_w.device).multi_processor_count
            step = ncu // math.gcd(ncu, tiles_m)
            nt = (K1 // 256) // step * step
            if getattr(self, "fc1_fused_all", 1) and K1 % 256 == 0:
                # the trailing tile columns ride in the same launch as a partial last round - it runs while the other
                # workgroups drain - instead of a small-tile launch + block update of their own: no A^T rows at all, the
                # pooling launch loses its 64-ROI tail launch (+2 % same box, profiles/r4_17; fc1_fused_all = 0: A/B)
                nt = K1 // 256
            if nt > 0 and self._fc1_use_tn(dtype):
                return nt * 256, nt * 256
        return None

    def _fc1_tail_row0(self, dtype, D1, K1):
        """first row of A^T the fc6 dW still reads: the smallest main-column count over the row slabs (the columns from
        there on are peeled into the small-tile NT launch, which takes A^T); 0 when the NT form is used throughout"""
        if not self._fc1_use_tn(dtype):
            return 0
        plan = self._fc1_col_plan(dtype, D1, K1)
        if plan is not None:
            return plan[0]
        return min(ops.gemm_nt_main_cols(b - a, K1) for a, b in self._fc1_slabs(D1))

    def _mark_current(self, s):
        for q in getattr(self, "_pool_sets", ()):
            if q["state"] == "current" and q is not s:
                q["state"] = "free"
        s["state"] = "current"
        self._pool_current_done = False

    @staticmethod
    def _splits(M, N, K, dtype):
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        nslab = K * ops.esize(dtype) // 128
        return max(1, min(1024 // max(tiles, 1), max(1, nslab // 8), 16))

    def _linear_fwd(self, A, Wt, M, N, K, bias, relu, out, outT, mask, seed, drop_p, seed_dev=None, fused=False):
        # fused (fc7, bf16): GEMM + bias + ReLU + dropout + the transposed copy as ONE launch of the eight-wave 128x128 kernel
        # (drn_linear_act_fwd) - no split-K partials, no second pass; outside that kernel's class the two-launch form below
        if (fused and A.dtype == torch.bfloat16 and getattr(self, "fused_fc7_fwd", True) and out is not None and
                out.dtype == torch.bfloat16 and
                ops.linear_act_fwd(A, Wt, M, N, K, bias, relu, mask, seed, drop_p, out=out, outT=outT, seed_dev=seed_dev)):
            return
        s = self._splits(M, N, K, A.dtype)
        part = ops.gemm_nt(A, Wt, M, N, K, splits=s)
        ops.bias_act_fwd(part, M, N, bias, relu, mask, seed, drop_p, out=out, outT=outT, seed_dev=seed_dev)

    def fc6_partials(self, pooled, M, training):
        """The fc6 forward GEMM on its own (split-K partial sums into a static workspace).  GraphedTrainStep issues this
        ONE launch eagerly in front of the captured heads graph, so that the step's dominant kernel can be bracketed
        by HIP events on its stream inside the timed region (a node of a replayed hipGraph cannot); forward() then takes
        the partials through `fc6_part`."""
        h = self.h
        dev, dtype = pooled["A"].device, pooled["A"].dtype
        self.ensure(dev)
        if pooled.get("kshard"):
            return self._fc6_partials_kshard(pooled, M)
        self.refresh_shadows(dtype)
        D1, K1 = h.box_head.fc1.weight.shape
        w = self.ws(M, dtype, training)
        K1p = ops.kpad(K1, dtype)
        s = self._splits(M, D1, K1p, dtype)
        if w.get("part1") is None or w["part1"].shape[0] != s:
            w["part1"] = torch.empty((s, M, D1), dtype=torch.float32, device=dev)
        ops.gemm_nt(pooled["A"], self.sh["W1v"], M, D1, K1p, out=w["part1"], splits=s)
        return w["part1"]

    # ---- fc6 sharded along K over the data-parallel ranks (round 4; FusedSGD.enable_pipelined(exchange="fc6_kshard")) ------
    # Data parallelism exchanges the fc6 weight gradient - 205 MB in bf16 per step and GPU, against 1.3 ms of compute with one
    # image per GPU - and has every rank run the 2-GB optimizer pass (or, sharded, gather 205 MB of updated weights).  fc6
    # is H1 = A . W^T with K = C*49 = 50176: rank k keeps the COLUMNS k of W (channels k*C/N .. (k+1)*C/N of the pooled
    # features - `c*49 + bin` makes a channel range a column range), pools that channel slice of ALL N ranks' images (the
    # ranks all-gather their 0.4-MB feature maps and 40-KB proposal lists, not their 200-MB pooled matrices), multiplies
    # [N*R x K/N] . [D1 x K/N]^T - the same FLOPs as its own [R x K] . [D1 x K]^T - and a reduce-scatter over row blocks hands
    # every rank the complete H1 of its own image.  Backward: all-gather of dP1 (8 MB per rank), dW[:, columns k] =
    # dP1_all^T . A_k - again the same FLOPs - and the optimizer updates only those columns: no gradient exchange and no
    # weight gather for fc6 at all, 1/N of the optimizer traffic.  Wire bytes per step and GPU at N = 8: 2 x 57 MB
    # (fp32 partial sums of H1 out, bf16 dP1 in) instead of 2 x 180 MB.  The arithmetic is the mean-gradient step of DDP
    # (detectron2/engine/defaults.py:279-282) up to the fp32 summation order of the K blocks.
    kshard = None  # dict(group, world, rank) once enabled

    def _ks_gather(self, t):
        ks = self.kshard
        out = torch.empty((ks["world"],) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=ks["group"])
        return out

    def pool_kshard(self, feat_nhwc, rois, objectness):
        """pooled operand of the K-sharded fc6: this rank's channel slice of EVERY rank's image -> A_k [N*M, (C/N)*P*P]
        (collectives on the current stream: all-gather of the feature maps, proposals and objectness)"""
        h, ks = self.h, self.kshard
        N, rank = ks["world"], ks["rank"]
        dtype = feat_nhwc.dtype
        M = rois.shape[0]
        n, H, W, C = feat_nhwc.shape
        ka = h.box_pooler.kernel_args()
        pp = ka["P"] * ka["P"] if "P" in ka else 49
        if C % N or ((C // N) * pp * ops.esize(dtype)) % 128:
            raise DrnError("fc6 K-sharding needs the channel count %d to split over %d ranks into whole 128-byte K slabs" % (C, N))
        # equal shapes on all ranks (fixed-size batches; ragged real data keeps the sharded gradient exchange)
        key = (n, H, W, C, M)
        if ks.get("checked") != key:
            shp = torch.tensor(key, dtype=torch.int64, device=feat_nhwc.device)
            allshp = self._ks_gather(shp)
            if not bool((allshp == shp).all()):
                raise DrnError("fc6 K-sharding needs the same image / proposal shapes on every rank (got %s)" % allshp.tolist())
            ks["checked"] = key
        c0, c1 = rank * (C // N), (rank + 1) * (C // N)
        # ONE all-gather: [feature map | proposals | objectness] of every rank as bytes (three small collectives were three
        # launch + rendezvous latencies on the step's dependent chain)
        fb, rb = feat_nhwc.numel() * feat_nhwc.element_size(), rois.numel() * 4
        ob = objectness.numel() * 4 if objectness is not None else 0
        pack = ks.get("pack")
        if pack is None or pack.numel() != fb + rb + ob:
            pack = ks["pack"] = torch.empty((fb + rb + ob,), dtype=torch.uint8, device=feat_nhwc.device)
            ks["pack_all"] = torch.empty((N, fb + rb + ob), dtype=torch.uint8, device=feat_nhwc.device)
        pack[:fb].view(feat_nhwc.dtype).copy_(feat_nhwc.reshape(-1))
        pack[fb: fb + rb].view(torch.float32).copy_(rois.reshape(-1))
        if ob:
            pack[fb + rb:].view(torch.float32).copy_(objectness.reshape(-1))
        allp = ks["pack_all"]
        dist.all_gather_into_tensor(allp.view(-1), pack, group=ks["group"])
        feats = allp[:, :fb].view(feat_nhwc.dtype).view(N, n, H, W, C)[..., c0:c1].reshape(N * n, H, W, c1 - c0).contiguous()
        rois_all = allp[:, fb: fb + rb].view(torch.float32).view(N, M, 5).clone()
        rois_all[:, :, 0] += (torch.arange(N, device=rois.device, dtype=rois.dtype) * n).view(N, 1)
        rois_all = rois_all.view(N * M, 5)
        obj_all = allp[:, fb + rb:].view(torch.float32).reshape(N * M).contiguous() if ob else None
        cols = (c1 - c0) * pp
        MA = N * M
        D1 = h.box_head.fc1.weight.shape[0]
        buf = ks.get("A")
        if buf is None or buf.shape != (MA, cols) or buf.dtype != dtype:
            buf = ks["A"] = torch.zeros((MA, cols), dtype=dtype, device=feat_nhwc.device)
            ks["part"] = torch.empty((1, MA, D1), dtype=torch.float32, device=feat_nhwc.device)
            ks["h1"] = torch.empty((1, M, D1), dtype=torch.float32, device=feat_nhwc.device)
        ops.roi_pool_nhwc(feats, rois_all, obj_all, out=buf, **ka)
        ks["cols"] = (c0 * pp, c1 * pp, MA)
        return dict(A=buf, AT=None, t_row0=0, kshard=True, state="current", M=M)

    def _fc6_partials_kshard(self, pooled, M):
        """[N*M x K/N] . [D1 x K/N]^T on this rank's columns of the compute copy, then the reduce-scatter over row blocks:
        -> the complete fp32 pre-activation of THIS rank's proposals, [1, M, D1]"""
        ks = self.kshard
        k0, k1, MA = ks["cols"]
        dtype = pooled["A"].dtype
        self.refresh_shadows(dtype)
        D1 = self.h.box_head.fc1.weight.shape[0]
        W1k = self.sh["W1v"][:, k0:k1]  # (row pitch K1)
        # few ranks leave too few 256x256 tiles for the CUs (N = 2: 16 x 8): split K like the replicated forward does and
        # add the splits up in split order before the wire
        tiles = ((MA + 255) // 256) * ((D1 + 255) // 256)
        nslab = (k1 - k0) * ops.esize(dtype) // 128
        sp = max(1, min(256 // max(tiles, 1), nslab // 16, 8))
        if sp > 1:
            if ks.get("parts") is None or ks["parts"].shape[0] != sp:
                ks["parts"] = torch.empty((sp, MA, D1), dtype=torch.float32, device=pooled["A"].device)
            ops.gemm_nt(pooled["A"], W1k, MA, D1, k1 - k0, out=ks["parts"], splits=sp)
            torch.sum(ks["parts"], dim=0, keepdim=True, out=ks["part"])
        else:
            ops.gemm_nt(pooled["A"], W1k, MA, D1, k1 - k0, out=ks["part"])
        if ks.get("wire") == torch.bfloat16:
            # optional: the partial sums cross the wire in bf16 (half the bytes of the step's largest collective; RCCL adds
            # them in bf16 - N roundings of 2^-9 in front of an activation that is stored in bf16 anyway)
            if ks.get("part16") is None or ks["part16"].shape != ks["part"].shape:
                ks["part16"] = torch.empty_like(ks["part"], dtype=torch.bfloat16)
                ks["h16"] = torch.empty_like(ks["h1"], dtype=torch.bfloat16)
            ks["part16"].copy_(ks["part"])
            dist.reduce_scatter_tensor(ks["h16"].view(-1), ks["part16"].view(-1), group=ks["group"])
            ks["h1"].copy_(ks["h16"])
        else:
            dist.reduce_scatter_tensor(ks["h1"].view(-1), ks["part"].view(-1), group=ks["group"])
        return ks["h1"]

    def _fc6_tail_kshard(self, w, M, D1, K1, dtype, gw, hook):
        """dW[:, own columns] = dP1_all^T . A_k (run_fc1_tail, K-sharded fc6)"""
        ks = self.kshard
        k0, k1, MA = ks["cols"]
        dP1_all = self._ks_gather(w["dP1"][:M, :D1].contiguous()).view(MA, D1)
        dPT = ops.transpose2d(dP1_all, MA, D1)            # [D1, kpad(N*M)]: the K-major operand of the dW GEMM
        Kp = dPT.shape[1]
        out = gw[:, k0:k1].unsqueeze(0)
        fused = getattr(self, "fc1_fused_cols", None)
        if dtype == torch.bfloat16 and fused is not None and fused(dPT, ks["A"], D1, k0, k1, Kp, MA, gw):
            return  # gradient and update of the owned columns in one launch
        if dtype == torch.bfloat16:
            ops.gemm_tn(dPT, ks["A"], D1, k1 - k0, Kp, MA, out=out)
        else:
            AT = ops.transpose2d(ks["A"], MA, k1 - k0)    # fp32 parity mode: the NT form on a transposed copy
            ops.gemm_nt(dPT, AT, D1, k1 - k0, Kp, out=out)
        if hook is not None:
            hook(("fc1b", 0, D1, k0, k1))

    # ---- forward -------------------------------------------------------------------------------------
    def forward(self, feat_nhwc, rois, objectness, training, img_off=None, n_img=1, gt=None, pooled=None,
                fc6_part=None):
        h = self.h
        fg_hook = getattr(self, "feature_grad_hook", None) if training else None
        csc = training and getattr(h, "csc_head", False)
        kshard = training and self.kshard is not None
        if self.kshard is not None and not training:
            # an evaluation forward reads ALL K columns of fc1.weight, and every rank has only updated its own block since
            # the last gather: gather the owners' columns first.  That is a collective - every rank has to run the
            # evaluation, like every rank has to checkpoint - guarded by the optimizer's fail-fast rendezvous.
            sync = self.kshard.get("sync")
            if sync is not None:
                sync()
        if kshard:
            if fg_hook is not None or csc:
                raise DrnError("fc6 K-sharding needs a frozen trunk (FREEZE_AT = 5) and the OICR / WSDDN / PCL heads")
            if pooled is None:
                pooled = self.pool_kshard(feat_nhwc, rois, objectness)
            elif not pooled.get("kshard"):
                raise DrnError("fc6 K-sharding: the prefetched operand was pooled for the replicated fc6")
        elif pooled is None:
            pooled = self.pool(feat_nhwc, rois, objectness, training, want_argmax=fg_hook is not None or csc)
        elif fg_hook is not None or csc:
            raise DrnError("a trainable backbone / the CSC head cannot use a prefetched pooled operand")
        else:
            self._mark_current(pooled)  # a prefetched set is consumed by this forward
        dev, dtype = pooled["A"].device, pooled["A"].dtype
        self.ensure(dev)
        self.refresh_shadows(dtype)
        M = rois.shape[0]
        w = dict(self.ws(M, dtype, training))
        w["A"], w["AT"], w["AT_row0"] = pooled["A"], pooled["AT"], pooled.get("t_row0", 0)
        sh = self.sh
        fc1, fc2 = h.box_head.fc1, h.box_head.fc2
        D1, K1 = fc1.weight.shape
        D2 = fc2.weight.shape[0]
        K, NH = h.num_classes, self.NH
        kp = lambda k: ops.kpad(k, dtype)
        drop_p = h.box_head.dropout_p if training else 0.0
        masks = h.box_head.dropout_masks if training else None
        seed, seed_dev = 0, None
        if training and masks is None and drop_p > 0:
            # counter-based masks keyed by (torch seed, device-side step counter, layer): the counter lives on the
            # device and is advanced on the stream, so a replayed hipGraph draws fresh masks every step
            if getattr(self, "seed_dev", None) is None or self.seed_dev.device != dev:
                self.seed_dev = torch.zeros((1,), dtype=torch.int64, device=dev)
            seed, seed_dev = torch.initial_seed() & 0xFFFFFFFFFFFF, self.seed_dev
        if kshard and fc6_part is None:
            fc6_part = self._fc6_partials_kshard(pooled, M)
        if fc6_part is not None:  # the GEMM was issued by fc6_partials() (eagerly, in front of a captured graph)
            ops.bias_act_fwd(fc6_part, M, D1, fc1.bias.data, True, masks[0] if masks else None, seed, drop_p,
                             out=w["H1"], outT=w["H1T"] if training else None, seed_dev=seed_dev)
        else:
            self._linear_fwd(w["A"], sh["W1v"], M, D1, kp(K1), fc1.bias.data, True, w["H1"],
                             w["H1T"] if training else None, masks[0] if masks else None, seed, drop_p, seed_dev)
        self._linear_fwd(w["H1"], sh["W2"], M, D2, kp(D1), fc2.bias.data, True, w["H2"], w["H2T"] if training else None,
                         masks[1] if masks else None, seed + 0x9E3779B1, drop_p, seed_dev, fused=True)
        bo, _ = self._seg[self.cols[0][0] + ".bias"]
        # (the logits pass has no dropout: given the counter it advances it - behind both dropout layers - instead of a
        # counter_add launch of its own on the heads' dependent chain)
        logit_seed = 2654435761 if seed_dev is not None else 0
        # drn_mil_oicr_losses: the predictor's split-K reduce + bias, WSDDN and the refinement cascade in six launches
        # (nine as separate calls); taken when the heads are exactly cls / det / non-regressing OICR branches
        fuse_tail = (training and not csc and getattr(self, "fused_loss_tail", True) and h.refine_K > 0 and
                     getattr(h, "refine_mode", "oicr") != "pcl" and not any(h.refine_reg[: h.refine_K]) and
                     NH == 2 * K + h.refine_K * (K + 1))
        if fuse_tail:
            logit_part = ops.gemm_nt(w["H2"], sh["Wh"], M, NH, kp(D2), splits=self._splits(M, NH, kp(D2), dtype))
        else:
            self._linear_fwd(w["H2"], sh["Wh"], M, NH, kp(D2), self.arena_w[bo: bo + NH], False, w["logits"], None, None,
                             logit_seed, 0.0, seed_dev)
        col = {n: c for n, _, c, _ in self.cols}
        if not training:
            return w, col
        if csc:
            return self._forward_csc(w, col, M, dtype, rois, objectness, feat_nhwc, pooled, gt, img_off, n_img, masks,
                                     drop_p, fg_hook)
        # ---- losses (fused with their dlogits) ----
        dl = w["dlogits"]
        thr = h.proposal_matcher.thresholds[1:-1]
        chain = None
        if fuse_tail:
            scores, img_scores, loss_part, chain = ops.mil_oicr_losses(
                logit_part, self.arena_w[bo: bo + NH], w["logits"], col["cls"], col["det"], K, img_off, n_img, gt["onehot"],
                [col["r%d" % k] for k in range(h.refine_K)], gt["props"], gt["classes"], gt["count"], thr,
                h.proposal_matcher.labels, dlogits=dl, mean_loss=h.box_predictor.mean_loss, max_rows=gt["max_rows"],
                seed_inc=logit_seed, seed_dev=seed_dev)
        else:
            scores, img_scores, loss_part = ops.wsddn_fwd_bwd(w["logits"], col["cls"], col["det"], K, img_off, n_img,
                                                             gt["onehot"], dlogits=dl, mean_loss=h.box_predictor.mean_loss,
                                                             max_rows=gt["max_rows"])
        for m in [h.box_predictor] + list(h.box_refinery[: h.refine_K]):
            assert m.loss_weight.get("loss_cls", 1.0) == 1.0, "loss_cls weight != 1 is not used by any config"
        loss_names = ["loss_cls"]
        loss_list = [(loss_part if n_img == 1 else ops.sum_small(loss_part)).view(())]
        head_cols = [("cls", 0), ("det", 0)]
        prev_scores, prev_boxes, prev_zero = scores, gt["props"], False
        aux = dict(scores=scores, img_scores=img_scores, targets=[])
        pcl = getattr(h, "refine_mode", "oicr") == "pcl"
        if pcl and h.refine_K > 0:
            # PCLROIHeads (roi_heads_pcl.py:311-334): every branch's targets come from the previous branch's
            # probabilities through proposal clustering; targets, loss and dlogits of all branches in three launches
            if n_img != 1:
                raise DrnError("PCL clusters the proposals of ONE image per step (third_party/pcl.py:94 asserts it)")
            if any(h.refine_reg[: h.refine_K]):
                raise DrnError("PCL with box regression branches is not on the built path (no reference config uses it)")
            adj = ops.pcl_adjacency(gt["props"], 0.4)
            res = ops.pcl_refine(w["logits"], [col["r%d" % k] for k in range(h.refine_K)], K, scores, gt["props"], adj,
                                 gt["onehot"].view(-1), dl)
            for k in range(h.refine_K):
                loss_names.append("loss_cls_r%d" % k)
                loss_list.append(res[k]["loss"].view(()))
                head_cols.append(("r%d" % k, len(loss_list) - 1))
                aux["targets"].append(res[k])
        elif chain is None and h.refine_K > 0 and not any(h.refine_reg[: h.refine_K]):
            # no branch regresses boxes: every branch's targets depend only on the previous branch's softmax of logits
            # that already exist, so the whole cascade is four launches (drn_oicr_refine_chain)
            chain = ops.oicr_refine_chain(w["logits"], [col["r%d" % k] for k in range(h.refine_K)], K, scores,
                                          gt["props"], img_off, n_img, gt["classes"], gt["count"], img_scores, thr,
                                          h.proposal_matcher.labels, dlogits=dl)
        for k in range(0 if pcl else h.refine_K):
            if chain is not None:
                tg, probs, loss = chain[k]
            else:
                tg = ops.oicr_targets(prev_scores, prev_boxes, gt["props"], img_off, n_img, gt["classes"], gt["count"],
                                      img_scores, K, thr, h.proposal_matcher.labels, zero_delta_decode=prev_zero)
                probs, loss = ops.softmax_ce(w["logits"], col["r%d" % k], K + 1, tg["labels"], tg["weights"], dlogits=dl)
            loss_names.append("loss_cls_r%d" % k)
            loss_list.append(loss.view(()))
            head_cols.append(("r%d" % k, len(loss_list) - 1))
            aux["targets"].append(tg)
            prev_scores = probs
            bw = h.box_refinery[k].box2box_transform.weights
            if h.refine_reg[k]:  # fast_rcnn.py:1146-1211 + :1227-1240: loss_box_reg_r{k}
                lreg = ops.box_reg_loss(w["logits"], col["b%d" % k], K, tg["labels"], gt["props"], tg["gt_boxes"], bw,
                                        dlogits=dl)
                loss_names.append("loss_box_reg_r%d" % k)
                loss_list.append(lreg.view(()))
                head_cols.append(("b%d" % k, len(loss_list) - 1))
                prev_boxes, prev_zero = ops.apply_deltas(w["logits"], gt["props"], K, bw, col0=col["b%d" % k]), False
            else:
                # a non-regressing head passes apply_deltas(0, proposals) on: only the G mined boxes are ever read,
                # so the decode happens inside the next targets kernel instead of materialising [M, 4K] boxes
                prev_boxes, prev_zero = gt["props"], True
        state = dict(w=w, M=M, dtype=dtype, loss_list=loss_list, head_cols=head_cols, masks=masks, drop_p=drop_p,
                     aux=aux)
        if fg_hook is not None:  # what the gradient of the feature map needs (trainable backbone)
            state["fg"] = dict(hook=fg_hook, rois=rois, obj=objectness, feat_shape=tuple(feat_nhwc.shape),
                               argmax=pooled.get("argmax"))
        self._last_state = state
        outs = _TrainFn.apply(self.anchor, self, state)
        return dict(zip(loss_names, outs)), state

    # ---- CSCROIHeads -----------------------------------------------------------------------------------
    def _forward_csc(self, w, col, M, dtype, rois, objectness, feat_nhwc, pooled, gt, img_off, n_img, masks, drop_p,
                     fg_hook):
        """roi_heads_csc.py:301-352 (training branch): MIL scores, the CSC weights (image-gradient maps through
        input_gradient(), csc.hip), the two weighted image-level losses fused with their dlogits."""
        h = self.h
        K = h.num_classes
        if n_img != 1:
            raise DrnError("CSCROIHeads works on ONE image per step (roi_heads_csc.py:433,442 read image_sizes[0] / "
                           "gt_classes_img_oh[0])")
        scores, img_scores, _, rowsm = ops.wsddn_fwd_bwd(w["logits"], col["cls"], col["det"], K, img_off, 1, gt["onehot"],
                                                         dlogits=None, mean_loss=h.box_predictor.mean_loss,
                                                         max_rows=gt["max_rows"], return_rowsm=True)
        fg = dict(rois=rois, obj=objectness, feat_shape=tuple(feat_nhwc.shape), argmax=pooled.get("argmax"))
        W, cpgs = h._csc_weights(self, w, col, scores, rowsm, M, dtype, fg, masks, drop_p)
        loss = ops.csc_loss(w["logits"], col["cls"], col["det"], K, scores, rowsm, W, gt["onehot"].view(-1),
                            h.box_predictor.mean_loss, dlogits=w["dlogits"])
        state = dict(w=w, M=M, dtype=dtype, loss_list=[loss[0].view(()), loss[1].view(())],
                     head_cols=[("cls", 0), ("det", 0)], masks=masks, drop_p=drop_p, csc=True,
                     aux=dict(scores=scores, img_scores=img_scores, targets=[], W=W, cpgs=cpgs))
        if fg_hook is not None:
            state["fg"] = dict(fg, hook=fg_hook)
        self._last_state = state
        outs = _TrainFn.apply(self.anchor, self, state)
        return dict(zip(["loss_cls_pos", "loss_cls_neg"], outs)), state

    def input_gradient(self, w, M, dtype, fg, masks, drop_p, refresh_w1t=True):
        """gradient of the feature map for the logits gradient sitting in w["dlogits"]: the d/dx half of backward()
        alone (predictor -> fc7 -> fc6 -> RoIPool / ROIAlign with the objectness scaling); no weight or bias gradient
        is written, nothing the training backward needs is overwritten except the scratch it rewrites itself."""
        h = self.h
        sh = self.sh
        fc1, fc2 = h.box_head.fc1, h.box_head.fc2
        D1, K1 = fc1.weight.shape
        D2 = fc2.weight.shape[0]
        NH = self.NH
        kp = lambda k: ops.kpad(k, dtype)
        dev = self.arena_w.device
        ops.bias_act_bwd(w["dlogits"], M, NH, dpre=w["dS"])
        ops.gemm_nt(w["dS"], sh["WhT"], M, D2, kp(NH), out=w["dH2"].view(1, M, D2))
        ops.bias_act_bwd(w["dH2"], M, D2, saved=w["H2"], mask=masks[1] if masks else None, drop_p=drop_p, dpre=w["dP2"])
        s1 = w["dH1"].shape[0]
        ops.gemm_nt(w["dP2"], sh["W2T"], M, D1, kp(D2), out=w["dH1"], splits=s1)
        if "dP1" not in w or w["dP1"].shape != w["H1"].shape:
            w["dP1"] = torch.zeros_like(w["H1"])
        ops.bias_act_bwd(w["dH1"], M, D1, saved=w["H1"], mask=masks[0] if masks else None, drop_p=drop_p, dpre=w["dP1"])
        return self._feature_gradient(w, fg, M, D1, K1, dtype, refresh_w1t)

    # ---- backward ------------------------------------------------------------------------------------
    def backward(self, st, gouts):
        h = self.h
        w, M, dtype = st["w"], st["M"], st["dtype"]
        sh = self.sh
        fc1, fc2 = h.box_head.fc1, h.box_head.fc2
        D1, K1 = fc1.weight.shape
        D2 = fc2.weight.shape[0]
        NH = self.NH
        kp = lambda k: ops.kpad(k, dtype)
        Mp = kp(M)
        dev = self.arena_w.device
        # per-loss upstream gradients -> per-column scale of dlogits (stays on the device)
        if gouts is None or isinstance(gouts, float):
            # d(scale * sum of losses): every loss has the same upstream gradient (1: GraphedTrainStep; 1 / ITER_SIZE:
            # Trainer through backward_losses) - a cached device vector, no autograd pass
            nl, val = len(st["loss_list"]), 1.0 if gouts is None else float(gouts)
            cache = self.__dict__.setdefault("_scale_vecs", {})
            colscale = cache.get((nl, val, dev))
            if colscale is None:
                if len(cache) > 8:
                    cache.clear()
                colscale = cache[(nl, val, dev)] = torch.full((nl,), val, dtype=torch.float32, device=dev)
        else:
            g = [torch.zeros((), device=dev) if x is None else x.float().reshape(()) for x in gouts]
            if st.get("csc") and not (gouts[0] is not None and gouts[1] is not None and torch.equal(g[0], g[1])):
                # both CSC losses act on the same cls / det columns and their dlogits were written as one sum
                raise DrnError("loss_cls_pos and loss_cls_neg must be back-propagated with one common weight (the "
                               "reference trainer sums the loss dict)")
            colscale = torch.stack(g)  # one entry per loss; columns find theirs through the static index table
        key = tuple(st["head_cols"])
        if getattr(self, "_colidx_key", None) != key:
            trained = dict(st["head_cols"])
            idx = torch.full((NH,), -1, dtype=torch.int32)
            for n, _, o, c in self.cols:
                if n in trained:
                    idx[o: o + c] = trained[n]
            self._colidx, self._colidx_key = idx.to(dev), key
        colidx = self._colidx
        acc = self._grads_valid and fc2.weight.grad is not None  # fc1.weight.grad is not materialised in bucket modes
        bo, _ = self._seg[self.cols[0][0] + ".bias"]
        wo, _ = self._seg[self.cols[0][0] + ".weight"]
        # bias gradients = column sums of the three pre-activation gradients, in two stages.  With the pipelined optimizer
        # the second stage (three 10-us launches) leaves the critical path: the partials stay in per-layer scratch and
        # `flush_colsums` adds them up on the optimizer stream right in front of the small-tensor SGD bucket
        defer = getattr(self, "defer_colsum", False) and getattr(self, "grad_ready_hook", None) is not None
        nparts = (M + 63) // 64
        self._pending_colsum = []

        def colsum_args(layer, N_, view):
            if not defer:
                return dict(colsum=view, accumulate_colsum=acc, colpart=w["colpart"])
            self._pending_colsum.append((w["colpart3"][layer], nparts, N_, view, acc))
            return dict(colsum=None, colpart=w["colpart3"][layer])

        # heads: dS, dS^T, bias grads
        ops.bias_act_bwd(w["dlogits"], M, NH, colscale=colscale, colidx=colidx, dpre=w["dS"], dpreT=w["dST"],
                         **colsum_args(0, NH, self.arena_g[bo: bo + NH]))
        ops.gemm_nt(w["dST"], w["H2T"], NH, D2, Mp, out=self.arena_g[wo: wo + NH * D2].view(1, NH, D2), accumulate=acc)
        # predictor dX + fc7's activation backward: one launch when the contraction is skinny (NH <= 256 columns) - the
        # fp32 dH2 [M, D2] never goes to memory (drn_gemm_nt_act_bwd, bit-identical to the two calls)
        m2 = st["masks"][1] if st["masks"] else None
        cs2 = colsum_args(1, D2, self._gview("fc2.bias"))
        if not (dtype == torch.bfloat16 and getattr(self, "fused_pred_dx", True) and
                ops.gemm_nt_act_bwd(w["dS"], sh["WhT"], M, D2, kp(NH), saved=w["H2"], mask=m2, drop_p=st["drop_p"],
                                    dpre=w["dP2"], dpreT=w["dP2T"], **cs2)):
            ops.gemm_nt(w["dS"], sh["WhT"], M, D2, kp(NH), out=w["dH2"].view(1, M, D2))
            ops.bias_act_bwd(w["dH2"], M, D2, saved=w["H2"], mask=m2, drop_p=st["drop_p"], dpre=w["dP2"], dpreT=w["dP2T"],
                             **cs2)
        # fc7 dX: [M, D1] over K = D2 is too few 256x256 tiles for one pass (64 at R = 2000) - split K like the forward
        # GEMMs and let the activation backward behind it sum the partials (63 -> ~45 us at the bench shape)
        s1 = self._splits(M, D1, kp(D2), dtype) if getattr(self, "fc7_dx_split", True) else 1
        pair = dtype == torch.bfloat16 and getattr(self, "fc7_bwd_pair", True)
        if pair:
            # in the paired launch the dX has about half the CUs to fill: as many (tile, K-split) items as that, not as the
            # whole chip - half the partial sums to write and for the activation backward to read (R50-C4: 4 -> 2 splits,
            # +1.3 % same box, profiles/r3_27)
            t256 = ((M + 255) // 256) * ((D1 + 255) // 256)
            s1 = getattr(self, "fc7_pair_dx_splits", 0) or max(1, min(s1, 128 // max(t256, 1)))
        if w["dH1"].shape[0] != s1:
            base = self._ws[(dtype, True)]
            base["bufs"]["dH1"] = torch.zeros((s1, base["cap"], D1), dtype=torch.float32, device=dev)
            w["dH1"] = base["views"]["dH1"] = base["bufs"]["dH1"][:, :M]
        g_w = dict(A=w["dP2T"], B=w["H1T"], M=D2, N=D1, K=Mp, out=self._gview("fc2.weight", (1, D2, D1)), accumulate=acc)
        g_x = dict(A=w["dP2"], B=sh["W2T"], M=M, N=D1, K=kp(D2), out=w["dH1"], splits=s1)
        if pair:
            # fc7 weight gradient and fc7 dX are independent (both read dP2) and each leaves CUs idle on its own (128 tiles;
            # 64 tiles x 4 short K-splits): ONE persistent launch whose workgroups are divided between the two (round 3)
            ops.gemm_nt_pair(g_w, g_x)
        else:
            ops.gemm_nt(g_w["A"], g_w["B"], D2, D1, Mp, out=g_w["out"], accumulate=acc)
            ops.gemm_nt(g_x["A"], g_x["B"], M, D1, kp(D2), out=w["dH1"], splits=s1)
        # fc6 (the backbone is frozen: no dX)
        fg = st.get("fg")
        need_dp1 = fg is not None or self.kshard is not None  # (K-sharded fc6: the ranks all-gather dP1 row-major)
        if need_dp1 and ("dP1" not in w or w["dP1"].shape != w["H1"].shape):
            w["dP1"] = torch.zeros_like(w["H1"])
        ops.bias_act_bwd(w["dH1"], M, D1, saved=w["H1"], mask=st["masks"][0] if st["masks"] else None,
                         drop_p=st["drop_p"], dpre=w["dP1"] if need_dp1 else None, dpreT=w["dP1T"],
                         **colsum_args(2, D1, self._gview("fc1.bias")))
        self._tail = (w["dP1T"], w["AT"], D1, K1, Mp, acc, w["A"], M, w.get("AT_row0", 0))
        self._tail_w = w
        if not getattr(self, "defer_fc1_tail", False):
            self.run_fc1_tail()
        if fg is not None:
            self._feature_backward(w, fg, M, D1, K1, dtype, acc)

    def _feature_gradient(self, w, fg, M, D1, K1, dtype, refresh_w1t=True):
        """MODEL.BACKBONE.FREEZE_AT < 5 (and the CSC image-gradient passes): fc6 dX = dP1 . W1 (NT GEMM on a K-major copy
        of W1), RoIPool / ROIAlign backward (with the objectness scaling) -> gradient of the feature map."""
        h = self.h
        kp = lambda k: ops.kpad(k, dtype)
        fc1 = h.box_head.fc1
        if getattr(self, "_w1t_key", None) != (dtype, K1, D1):
            self._w1t = torch.zeros((K1, kp(D1)), dtype=dtype, device=self.arena_w.device)
            self._dA = torch.zeros((1, M, kp(K1)), dtype=dtype, device=self.arena_w.device)
            self._w1t_key = (dtype, K1, D1)
        if self._dA.shape[1] != M:
            self._dA = torch.zeros((1, M, kp(K1)), dtype=dtype, device=self.arena_w.device)
        if refresh_w1t:  # the K-major copy of W1 (several CSC passes of one step share it)
            ops.transpose2d(fc1.weight.data, D1, K1, out=self._w1t)
        ops.gemm_nt(w["dP1"], self._w1t, M, K1, kp(D1), out=self._dA[:, :, :K1])
        ka = h.box_pooler.kernel_args()
        return ops.roi_pool_backward_nhwc(self._dA[0], fg["rois"], fg["obj"], fg["feat_shape"], argmax=fg["argmax"], **ka)

    def _feature_backward(self, w, fg, M, D1, K1, dtype, acc):
        dfeat = self._feature_gradient(w, fg, M, D1, K1, dtype)
        fg["hook"](dfeat, acc)
        hook = getattr(self, "grad_ready_hook", None)
        if hook is not None:
            hook("backbone")

    def flush_colsums(self):
        """second stage of the deferred bias-gradient column sums (current stream; see backward)"""
        # (the list describes the LAST backward that ran in Python; a replayed hipGraph re-executes that backward's
        # kernels into the same scratch, so the list is kept, not consumed)
        for colpart, nparts, N_, view, acc in getattr(self, "_pending_colsum", ()):
            ops.colsum_reduce(colpart, nparts, N_, view, acc)

    def run_fc1_tail(self):
        """Last piece of the explicit backward: announce the small gradients, then the fc6 weight gradient in row
        slabs (each announced as soon as its GEMM is queued).  Normally called by backward() itself; the multi-GPU
        graphed step sets `defer_fc1_tail` and calls it eagerly after replaying the captured part, so the RCCL calls
        issued from the hooks are ordinary stream work and never part of a hipGraph."""
        dP1T, AT, D1, K1, Mp, acc, A, M, at_row0 = self._tail
        hook = getattr(self, "grad_ready_hook", None)
        if hook is not None:
            hook("small")  # everything except fc1.weight is final: the DP engine starts reducing it now
        bucket = getattr(self, "fc1_grad_bucket", None)  # [D1, K1] exchange buffer (bf16 or fp32) instead of the arena
        if bucket is not None and acc:
            raise DrnError("a bucketed fc6 gradient cannot be combined with gradient accumulation")
        gw = bucket if bucket is not None else self._gview("fc1.weight", (D1, K1))
        if self.kshard is not None:
            if acc:
                raise DrnError("fc6 K-sharding cannot be combined with gradient accumulation")
            self._fc6_tail_kshard(self._tail_w, M, D1, K1, dP1T.dtype, gw, hook)
            self._grads_valid = True
            self._pool_current_done = True
            for name, p, o, n, used in self.segments:
                if used and p.grad is None and name != "fc1.weight":
                    p.grad = self.arena_g[o: o + n].view(p.shape)
            return
        # Tail balancing peels the same trailing columns off every equal-height slab (drn_gemm_nt: a small-tile launch
        # in front of each persistent one - 14 + 22 us and two launch gaps per step for two slabs).  When all slabs
        # agree on the split, ONE launch computes the peeled columns of all rows first and the slabs' main columns -
        # exact rounds - follow; same kernels' arithmetic per element (tile size does not change the summation order)
        n0 = K1
        slabs = self._fc1_slabs(D1)
        tn = self._fc1_use_tn(dP1T.dtype)
        if at_row0 > 0 and not tn:
            raise DrnError("this batch was pooled for the TN form of the fc6 weight gradient (A^T rows below %d were "
                           "not written) but the backward runs the NT form: fc1_tn / the fused update changed in "
                           "between" % at_row0)
        plan = None if acc else self._fc1_col_plan(dP1T.dtype, D1, K1)
        if plan is not None:
            # column slabs: the trailing columns that do not fill a slab first (small-tile NT launch on the A^T tail rows,
            # all fc6 rows), then slabs of exact rounds straight from A; every piece is announced as a block
            # ("fc1b", r0, r1, c0, c1) the moment its GEMM is queued
            n_main, wcols = plan
            if n_main < at_row0:
                raise DrnError("A^T rows %d.. are needed, the pooling launch wrote them from %d on" % (n_main, at_row0))
            if n_main < K1:
                ops.gemm_nt(dP1T, AT[n_main:], D1, K1 - n_main, Mp, out=gw[:, n_main:].unsqueeze(0))
                if hook is not None:
                    hook(("fc1b", 0, D1, n_main, K1))
            ftn = getattr(self, "fc1_fused_tn", None)
            for c0 in range(0, n_main, wcols):
                c1 = min(n_main, c0 + wcols)
                if ftn is not None and c0 == 0 and c1 == n_main and bucket is not None and \
                        ftn(dP1T, A, D1, n_main, Mp, M, gw):
                    continue  # gradient AND update done by the one launch (drn_gemm_tn_sgd)
                ops.gemm_tn(dP1T, A[:, c0:c1], D1, c1 - c0, Mp, M, out=gw[:, c0:c1].unsqueeze(0))
                if hook is not None:
                    hook(("fc1b", 0, D1, c0, c1))
            slabs = []
        if getattr(self, "fc1_joint_peel", 1) and len(slabs) > 1 and not acc:
            cols = {ops.gemm_nt_main_cols(b - a, K1) for a, b in slabs}
            if len(cols) == 1 and 0 < min(cols) < K1:
                n0 = cols.pop()
                if n0 < at_row0:
                    raise DrnError("A^T rows %d.. are needed, the pooling launch wrote them from %d on" % (n0, at_row0))
                ops.gemm_nt(dP1T, AT[n0:], D1, K1 - n0, Mp, out=gw[:, n0:].unsqueeze(0))
        for r0, r1 in slabs:
            if tn:
                # main columns (exact rounds of the persistent kernel) straight from A; a slab whose peel was not part
                # of a joint launch peels its own trailing columns through the NT small-tile kernel first
                m0 = min(n0, ops.gemm_nt_main_cols(r1 - r0, K1))
                if m0 < n0:
                    if m0 < at_row0:
                        raise DrnError("A^T rows %d.. are needed, the pooling launch wrote them from %d on" % (m0, at_row0))
                    ops.gemm_nt(dP1T[r0:r1], AT[m0:n0], r1 - r0, n0 - m0, Mp, out=gw[r0:r1, m0:n0].unsqueeze(0),
                                accumulate=acc)
                if m0 > 0:
                    ops.gemm_tn(dP1T[r0:r1], A[:, :m0], r1 - r0, m0, Mp, M, out=gw[r0:r1, :m0].unsqueeze(0),
                                accumulate=acc)
            else:
                ops.gemm_nt(dP1T[r0:r1], AT[:n0], r1 - r0, n0, Mp, out=gw[r0:r1, :n0].unsqueeze(0), accumulate=acc)
            if hook is not None:
                hook(("fc1", r0, r1))
        self._grads_valid = True
        self._pool_current_done = True  # the last reader of this batch's A^T has been issued
        skip = bucket is not None
        for name, p, o, n, used in self.segments:
            if used and p.grad is None and not (skip and name == "fc1.weight"):
                p.grad = self.arena_g[o: o + n].view(p.shape)


# ------------------------------------------------------------------------------------------------
class ROIHeads(nn.Module):
    """roi_heads.py:156-212 (constructor / from_config)."""

    @configurable
    def __init__(self, *, num_classes, batch_size_per_image, positive_fraction, proposal_matcher,
                 proposal_append_gt=True):
        super().__init__()
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.num_classes = num_classes
        self.proposal_matcher = proposal_matcher
        self.proposal_append_gt = proposal_append_gt

    @classmethod
    def from_config(cls, cfg):
        return {
            "batch_size_per_image": cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION,
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "proposal_append_gt": cfg.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT,
            "proposal_matcher": Matcher(cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS, cfg.MODEL.ROI_HEADS.IOU_LABELS,
                                        allow_low_quality_matches=False),
        }


def get_image_level_gt(targets, num_classes):
    """roi_heads.py:137-153.  `unique(sorted=True)` of a handful of ints: done on the host copy of the labels
    (they arrive from the data loader on the host), so no device sync is introduced."""
    if targets is None:
        return None, None, None
    ints = [torch.unique(t.gt_classes.detach().cpu(), sorted=True).to(torch.int64) for t in targets]
    oh = torch.zeros((len(ints), num_classes), dtype=torch.float32)
    for i, g in enumerate(ints):
        oh[i, g] = 1
    return ints, ints, oh


@ROI_HEADS_REGISTRY.register()
class OICRROIHeads(ROIHeads):
    """roi_heads_oicr.py:33-567."""

    @configurable
    def __init__(self, *, box_in_features, box_pooler, box_head, box_predictor, mask_in_features=None,
                 mask_pooler=None, mask_head=None, keypoint_in_features=None, keypoint_pooler=None,
                 keypoint_head=None, train_on_pred_boxes=False, output_dir=None, vis_test=False, vis_period=0,
                 refine_K=4, refine_reg=(False, False, False, False), box_refinery=(None, None, None, None),
                 cls_agnostic_bbox_reg=False, **kwargs):
        super().__init__(**kwargs)
        if mask_in_features is not None or keypoint_in_features is not None or train_on_pred_boxes:
            raise DrnError("mask / keypoint heads and train_on_pred_boxes are off the DRN-WSOD path")
        self.in_features = self.box_in_features = box_in_features
        self.box_pooler = box_pooler
        self.box_head = box_head
        self.box_predictor = box_predictor
        self.mask_on = self.keypoint_on = False
        self.train_on_pred_boxes = train_on_pred_boxes
        self.iter = self.iter_test = self.epoch_test = 0
        self.output_dir, self.vis_test, self.vis_period = output_dir, vis_test, vis_period
        self.refine_K = refine_K
        self.refine_reg = list(refine_reg)
        self.box_refinery = list(box_refinery)
        for k in range(self.refine_K):
            self.add_module("box_refinery_{}".format(k), self.box_refinery[k])
        self.cls_agnostic_bbox_reg = cls_agnostic_bbox_reg
        self._engine = _HeadEngine(self)

    @classmethod
    def from_config(cls, cfg, input_shape):
        ret = super().from_config(cfg)
        ret["train_on_pred_boxes"] = cfg.MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES
        if inspect.ismethod(cls._init_box_head):
            ret.update(cls._init_box_head(cfg, input_shape))
        return ret

    _refine_from_cfg = True

    @classmethod
    def _init_box_head(cls, cfg, input_shape):
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        pooler_resolution = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        pooler_scales = tuple(1.0 / input_shape[k].stride for k in in_features)
        in_channels = [input_shape[f].channels for f in in_features]
        assert len(set(in_channels)) == 1, in_channels
        in_channels = in_channels[0]
        box_pooler = ROIPooler(output_size=pooler_resolution, scales=pooler_scales,
                               sampling_ratio=cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO,
                               pooler_type=cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE)
        box_head = build_box_head(cfg, ShapeSpec(channels=in_channels, height=pooler_resolution, width=pooler_resolution))
        box_predictor = WSDDNOutputLayers(cfg, box_head.output_shape)
        refine_K = cfg.WSL.REFINE_NUM if cls._refine_from_cfg else 0
        box_refinery = [OICROutputLayers(cfg, box_head.output_shape, k) for k in range(refine_K)]
        return {"box_in_features": in_features, "box_pooler": box_pooler, "box_head": box_head,
                "box_predictor": box_predictor, "output_dir": cfg.OUTPUT_DIR, "vis_test": cfg.WSL.VIS_TEST,
                "vis_period": cfg.VIS_PERIOD, "refine_K": refine_K, "refine_reg": cfg.WSL.REFINE_REG,
                "box_refinery": box_refinery, "cls_agnostic_bbox_reg": cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG}

    # ---- helpers --------------------------------------------------------------------------------------
    def _gather_inputs(self, features, proposals):
        feat = features[self.box_in_features[0]]
        nhwc = feat.permute(0, 2, 3, 1)
        if not nhwc.is_contiguous() or nhwc.dtype != compute_dtype():
            nhwc = to_nhwc(feat, compute_dtype())
        self._props = None
        if len(proposals) == 1:
            # one image (every shipped config trains and tests with one image per GPU): rois, logits and the boxes' contiguous
            # copy in ONE launch instead of torch.full + two torch.cat + three copies in front of every forward
            b, l = proposals[0].proposal_boxes.tensor, proposals[0].objectness_logits
            if (b.is_cuda and b.dtype == torch.float32 and b.dim() == 2 and b.is_contiguous() and l is not None and l.is_cuda and
                    l.device == b.device and l.dtype == torch.float32 and l.dim() == 1 and l.shape[0] == b.shape[0] and
                    l.is_contiguous()):
                rois, obj, props = ops.stage_rois(b, l, 0.0)
                self._props = (rois, props)  # (tied to THIS rois tensor: a prefetch of a later batch may gather in between)
                return nhwc, rois, obj
        rois = convert_boxes_to_pooler_format([p.proposal_boxes for p in proposals]).float().contiguous()
        obj = torch.cat([p.objectness_logits for p in proposals], dim=0).float().contiguous()
        return nhwc, rois, obj

    def _take_props(self, rois):
        """the proposal boxes as a contiguous [M, 4] tensor: the copy _gather_inputs made in its staging launch, else a slice copy"""
        p, self._props = getattr(self, "_props", None), None
        return p[1] if p is not None and p[0] is rois and p[1] is not None else rois[:, 1:].contiguous()

    def _zero_onehot(self, n_img, K, dev):
        z = getattr(self, "_zero_oh", None)
        if z is None or z.shape != (n_img, K) or z.device != dev:
            z = self._zero_oh = torch.zeros((n_img, K), dtype=torch.float32, device=dev)
        return z

    def prefetch_pooled(self, features, proposals):
        """pool a FUTURE batch's proposals (on whatever stream is current) into the engine's spare buffer set"""
        if getattr(self._engine, "kshard", None) is not None:
            return None  # K-sharded fc6: the operand is built from every rank's features inside the forward
        nhwc, rois, obj = self._gather_inputs(features, proposals)
        return dict(rois=rois, obj=obj, props=self._take_props(rois),
                    pooled=self._engine.pool(nhwc, rois, obj, self.training, prefetch=True))

    def forward(self, images, features, proposals, targets=None, prefetched=None):
        """roi_heads_oicr.py:248-291."""
        self.images = images
        self._prefetched = prefetched
        if self.training:
            assert targets
            losses = self._forward_box(features, proposals, targets)
            self.iter += 1
            if self.iter_test > 0:
                self.epoch_test += 1
            self.iter_test = 0
            return proposals, losses
        pred_instances, all_scores, all_boxes = self._forward_box(features, proposals, None)
        self.iter_test += 1
        return pred_instances, {}, all_scores, all_boxes

    def _forward_box(self, features, proposals, targets):
        """roi_heads_oicr.py:320-421."""
        pre = getattr(self, "_prefetched", None)
        self._prefetched = None
        if pre is not None:
            nhwc, rois, obj, pooled = None, pre["rois"], pre["obj"], pre["pooled"]
            self._props = (rois, pre.get("props"))
        else:
            nhwc, rois, obj = self._gather_inputs(features, proposals)
            pooled = None
        dev = rois.device
        nper = [len(p) for p in proposals]
        n_img = len(proposals)
        K = self.num_classes
        if self.training:
            ints, _, oh = get_image_level_gt(targets, K)
            self.gt_classes_img_int, self.gt_classes_img_oh = ints, oh
            gmax = max(1, max(len(g) for g in ints))
            gcl = torch.zeros((n_img, gmax), dtype=torch.int32)
            for i, g in enumerate(ints):
                gcl[i, : len(g)] = g.to(torch.int32)
            off = torch.tensor([0] + list(torch.tensor(nper).cumsum(0).tolist()), dtype=torch.int32)
            gt = dict(onehot=oh.to(dev, non_blocking=True), classes=gcl.to(dev, non_blocking=True),
                      count=torch.tensor([len(g) for g in ints], dtype=torch.int32).to(dev, non_blocking=True),
                      props=self._take_props(rois), max_rows=max(nper))
            losses, state = self._engine.forward(nhwc, rois, obj, True, off.to(dev, non_blocking=True), n_img, gt,
                                                 pooled=pooled)
            self.pred_class_img_logits = state["aux"]["img_scores"]
            self._last_state = state
            if has_event_storage():  # kept as device scalars: no sync (the reference syncs 3x here)
                st = get_event_storage()
                o1 = obj + 1
                st.put_scalar("proposals/objectness_logits+1 mean", o1.mean())
                st.put_scalar("proposals/objectness_logits+1 max", o1.max())
                st.put_scalar("proposals/objectness_logits+1 min", o1.min())
            return losses
        w, col = self._engine.forward(nhwc, rois, obj, False, pooled=pooled)
        heads = [k for k in range(self.refine_K)]
        props = self._take_props(rois)
        last = self.box_refinery[-1] if self.refine_K else self.box_predictor
        if self.refine_K == 0:
            # WSDDNROIHeads (roi_heads_wsddn.py:305-309 -> WSDDNOutputLayers.inference, fast_rcnn.py:587-608): the MIL
            # scores with a zero background column, zero deltas
            off = torch.tensor([0] + list(torch.tensor(nper).cumsum(0).tolist()), dtype=torch.int32).to(dev)
            zeros = self._zero_onehot(n_img, K, dev)
            sc, _, _ = ops.wsddn_fwd_bwd(w["logits"], col["cls"], col["det"], K, off, n_img, zeros, max_rows=max(nper))
            probs = torch.zeros((sc.shape[0], K + 1), dtype=torch.float32, device=dev)
            probs[:, :K].copy_(sc)
            boxes = ops.apply_deltas(None, props, K, last.box2box_transform.weights)
        elif self.refine_reg[-1]:
            probs, _ = ops.softmax_ce(w["logits"], col["r%d" % heads[-1]], K + 1)
            boxes = ops.apply_deltas(w["logits"], props, K, last.box2box_transform.weights, col0=col["b%d" % heads[-1]])
        else:
            probs = ops.mean_softmax(w["logits"], [col["r%d" % k] for k in heads], K + 1,
                                     bg_first=getattr(self, "refine_mode", "oicr") == "pcl")
            boxes = ops.apply_deltas(None, props, K, last.box2box_transform.weights)
        results, all_scores, all_boxes = [], [], []
        if getattr(self, "scores_only", False):
            # GeneralizedRCNNWithTTAAVG's per-augmentation passes use all_scores / all_boxes alone
            # (test_time_augmentation_avg.py:269-294 drops the first return value): no threshold / sort / NMS per pass
            for s, b in zip(probs.split(nper), boxes.split(nper)):
                all_scores.append(s.unsqueeze(0))
                all_boxes.append(b.unsqueeze(0))
            return None, all_scores, all_boxes
        for p, s, b in zip(proposals, probs.split(nper), boxes.split(nper)):
            ob, os_, oc, _ = ops.detect_topk(b, s, p.image_size, last.test_score_thresh, last.test_nms_thresh,
                                             last.test_topk_per_image)
            r = Instances(p.image_size)
            r.pred_boxes = Boxes(ob)
            r.scores = os_
            r.pred_classes = oc
            results.append(r)
            all_scores.append(s.unsqueeze(0))
            all_boxes.append(b.unsqueeze(0))
        return results, all_scores, all_boxes


@ROI_HEADS_REGISTRY.register()
class PCLROIHeads(OICRROIHeads):
    """roi_heads_pcl.py:28-349: the same trunk, neck, MIL head and K+1-way refinement branches as OICR; the branches
    are trained with the proposal-cluster loss (fast_rcnn.py:1725-1745) instead of the pseudo-GT cross entropy, keep
    the background in column 0, and are averaged with `pcl_bg` at test time (roi_heads_pcl.py:341-349).  The reference
    runs the clustering in numpy / scikit-learn and the loss in C++ on the host; here both stay on the device
    (csrc/pcl.hip)."""

    refine_mode = "pcl"


@ROI_HEADS_REGISTRY.register()
class WSDDNROIHeads(OICRROIHeads):
    """roi_heads_wsddn.py: the MIL head alone (no refinement branches); inference scores are the MIL scores
    (WSDDNOutputLayers.inference, fast_rcnn.py:587-608)."""

    _refine_from_cfg = False


@ROI_HEADS_REGISTRY.register()
class CSCROIHeads(OICRROIHeads):
    """roi_heads_csc.py:40-551: the WSDDN MIL head trained with two image-level BCE losses on CSC-weighted score sums.
    Per labelled class whose image score reaches `tau`, the gradient of the summed class score w.r.t. the (normalised)
    input image gives a saliency map (`_forward_cpg`: here the d/dx half of the explicit backward through the predictor,
    fc7, fc6, RoIPool and EVERY trunk layer down to the pixels); the map, thresholded and summed into a table, scores
    each proposal by its frame / context contrast (CSCPool) and becomes the signed weight W (csc.hip).  After
    WSL.CSC_MAX_ITER iterations W_pos = 1, W_neg = 0.  Inference is WSDDN's.  Debug dumps (`_save_mask`) and the
    `Statistic` log writer are the reference's control plane and are not rebuilt."""

    _refine_from_cfg = False
    csc_head = True

    @configurable
    def __init__(self, *, csc_max_iter=35000, **kwargs):
        super().__init__(**kwargs)
        self.csc_max_iter = csc_max_iter
        self.tau, self.fg_threshold, self.bg_threshold = 0.7, 0.1, 0.005  # roi_heads_csc.py:107-109
        self.context_scale, self.area_sqrt = 1.8, True  # :110-118 (mass / density thresholds are unused by the op)
        self.image_grad_fn = None  # set by GeneralizedRCNNWSL: feature-map gradient -> image gradient (NHWC)

    @classmethod
    def from_config(cls, cfg, input_shape):
        ret = super().from_config(cfg, input_shape)
        ret["csc_max_iter"] = cfg.WSL.CSC_MAX_ITER
        return ret

    def prefetch_pooled(self, features, proposals):
        raise DrnError("the CSC head differentiates through RoIPool and the trunk: no prefetched operands")

    def _csc_weights(self, eng, w, col, scores, rowsm, M, dtype, fg, masks, drop_p):
        """roi_heads_csc.py:423-510 (_forward_cpg + _forward_csc). Returns (W [M, K] or None past CSC_MAX_ITER, cpgs)."""
        K = self.num_classes
        if self.iter > self.csc_max_iter:
            return None, None
        if self.image_grad_fn is None:
            raise DrnError("CSCROIHeads needs the meta-architecture's image-gradient pass (GeneralizedRCNNWSL with cpg)")
        H, Wd = self.images.image_sizes[0]
        if tuple(self.images.nhwc.shape[1:3]) != (H, Wd):
            raise DrnError("CSC maps are image-sized: the padded batch tensor must equal the image (one image per step)")
        dev = scores.device
        labelled = [int(c) for c in self.gt_classes_img_int[0].tolist()]
        # the reference branches on the image score per class on the host too (roi_heads_csc.py:445): one read of K floats
        img = scores.sum(dim=0).tolist()
        cpgs = torch.zeros((K, H, Wd), dtype=torch.float32, device=dev)
        W = torch.ones((M, K), dtype=torch.float32, device=dev)
        table = torch.empty((H, Wd), dtype=torch.float32, device=dev)
        first = True
        for c in labelled:
            if img[c] >= self.tau:
                ops.csc_loss(w["logits"], col["cls"], col["det"], K, scores, rowsm, None, None,
                             self.box_predictor.mean_loss, dlogits=w["dlogits"], seed_class=c)
                dfeat = eng.input_gradient(w, M, dtype, fg, masks, drop_p, refresh_w1t=first)
                first = False
                ops.csc_cpg(self.image_grad_fn(dfeat), 3, out=cpgs[c])
            # (a labelled class below tau keeps a zero map and still goes through the op, as in the reference)
            ops.csc_weights(cpgs[c], self.fg_threshold, fg["rois"], scores, c, self.area_sqrt, self.context_scale, W, table)
        return W, cpgs


def build_roi_heads(cfg, input_shape):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)
