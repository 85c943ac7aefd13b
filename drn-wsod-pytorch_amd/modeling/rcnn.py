"""GeneralizedRCNNWSL (projects/WSL/wsl/modeling/meta_arch/rcnn.py:23-325) — same registry name, constructor
and call contract: `model(batched_inputs) -> dict[str, Tensor]` when training, `list[{"instances": Instances}]`
in eval, `model.inference(batched_inputs, detected_instances=None, do_postprocess=True)`."""
from typing import Optional, Tuple

import torch
from torch import nn

from .. import compute_dtype, ops
from .._cabi import DrnError
from ..config import configurable
from ..registry import META_ARCH_REGISTRY
from ..structures import Boxes, ImageList, Instances
from .backbone import build_backbone
from .roi_heads import build_roi_heads

__all__ = ["GeneralizedRCNNWSL", "build_model", "detector_postprocess"]


def detector_postprocess(results: Instances, output_height: int, output_width: int):
    """projects/WSL/wsl/modeling/postprocessing.py: rescale boxes to the requested output size, clip, drop
    empty ones (<= 100 boxes: plumbing)."""
    sx, sy = output_width / results.image_size[1], output_height / results.image_size[0]
    out = Instances((output_height, output_width), **results.get_fields())
    boxes = out.pred_boxes.clone()
    boxes.scale(sx, sy)
    boxes.clip(out.image_size)
    out.pred_boxes = boxes
    return out[boxes.nonempty()]


@META_ARCH_REGISTRY.register()
class GeneralizedRCNNWSL(nn.Module):
    @configurable
    def __init__(self, *, backbone, proposal_generator, load_proposals: bool, roi_heads, pixel_mean: Tuple[float],
                 pixel_std: Tuple[float], input_format: Optional[str] = None, vis_period: int = 0, cpg: bool = False):
        super().__init__()
        if proposal_generator is not None:
            raise DrnError("learned proposal generators are off the DRN-WSOD path (precomputed proposals only)")
        if cpg and not hasattr(roi_heads, "image_grad_fn"):
            raise DrnError("input-image gradients are built for CSCROIHeads only (WSJDS / X heads are off this path)")
        self.backbone = backbone
        self.proposal_generator = None
        self.load_proposals = load_proposals
        self.roi_heads = roi_heads
        self.input_format = input_format
        self.vis_period = vis_period
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1))
        self._mean, self._std = tuple(float(v) for v in pixel_mean), tuple(float(v) for v in pixel_std)
        # rcnn.py:170-171,190-192: `images.tensor.requires_grad = True` around the forward.  Here: every trunk unit keeps
        # what its explicit backward needs and the head gets the d feature -> d image pass
        self.cpg = cpg
        if cpg:
            self.backbone.input_grad = True
            self.roi_heads.image_grad_fn = self.backbone.input_gradient_nhwc

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        name = cfg.MODEL.ROI_HEADS.NAME
        return {"backbone": backbone, "proposal_generator": None, "load_proposals": cfg.MODEL.LOAD_PROPOSALS,
                "roi_heads": build_roi_heads(cfg, backbone.output_shape()), "input_format": cfg.INPUT.FORMAT,
                "vis_period": cfg.VIS_PERIOD, "pixel_mean": cfg.MODEL.PIXEL_MEAN, "pixel_std": cfg.MODEL.PIXEL_STD,
                "cpg": ("CSC" in name or "WSJDS" in name or "XROIHeads" in name)}

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        """rcnn.py:242-249: one fused kernel per image: (x - mean) / std, zero pad to the batch max, NCHW -> NHWC,
        cast to the compute dtype."""
        dtype = compute_dtype()
        images = [x["image"].to(self.device, non_blocking=True) for x in batched_inputs]
        nhwc, sizes = ops.preprocess_nhwc(images, self._mean, self._std, dtype, 8 if dtype == torch.bfloat16 else 4)
        view = nhwc[..., : len(self._mean)].permute(0, 3, 1, 2)
        view._drn_nhwc = nhwc
        return ImageList(view, sizes, nhwc)

    def prefetch_features(self, batched_inputs):
        """Run preprocess + backbone of a FUTURE batch on a side stream.  The shipped configs freeze the whole
        backbone (FREEZE_AT=5), so its forward has no dependency on the head update: its ~45 small, latency-bound
        conv launches hide under the current batch's fc6/fc7 GEMMs instead of serialising in front of them."""
        if self.cpg or self.backbone.trainable():
            return  # only legal for a frozen backbone that nothing differentiates through
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        # the packed conv weights / folded FrozenBN affines are built lazily on first use: build (or re-validate) them
        # HERE, on the caller's stream, so the side stream never creates buffers that the caller's own forward of the
        # current batch reads concurrently (a first-iteration race when the prefetch precedes any forward)
        dtype = compute_dtype()
        for m in self.backbone.conv_modules():
            m.packed(dtype)
        # one batch ahead is the contract (the engine owns two fc6-operand sets: the batch in flight and ONE pooled ahead):
        # a prefetched batch that was never consumed - the caller changed its mind about what comes next - is dropped
        # here and its operand set handed back, instead of failing the new prefetch with "no free buffer set"
        for key_ in [k for k in getattr(self, "_prefetch_cache", {}) if k != id(batched_inputs)]:
            stale = self._prefetch_cache.pop(key_)
            if stale[3] is not None and stale[3]["pooled"]["state"] == "pending":
                stale[3]["pooled"]["state"] = "free"
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side), torch.no_grad():
            images = self.preprocess_image(batched_inputs)
            features = self.backbone(images.tensor)
            pooled = None
            if self.training and hasattr(self.roi_heads, "prefetch_pooled") and "proposals" in batched_inputs[0]:
                # ROIPool + the A^T copy of the next batch leave the critical path too
                pooled = self.roi_heads.prefetch_pooled(features, self._proposals(batched_inputs))
            ev = torch.cuda.Event()
            ev.record(self._side)
        if not hasattr(self, "_prefetch_cache"):
            self._prefetch_cache = {}
        if len(self._prefetch_cache) >= 2:
            stale = self._prefetch_cache.pop(next(iter(self._prefetch_cache)))
            if stale[3] is not None:
                stale[3]["pooled"]["state"] = "free"  # never consumed: give its fc6-operand buffer set back
        again = self._prefetch_cache.pop(id(batched_inputs), None)
        if again is not None and again[3] is not None and again[3]["pooled"] is not (pooled or {}).get("pooled"):
            again[3]["pooled"]["state"] = "free"
        self._prefetch_cache[id(batched_inputs)] = (images, features, ev, pooled)

    def _features(self, batched_inputs):
        pre = getattr(self, "_prefetch_cache", {}).pop(id(batched_inputs), None)
        if pre is not None:
            images, features, ev, pooled = pre
            main = torch.cuda.current_stream()
            main.wait_event(ev)
            for t in list(features.values()) + [images.nhwc]:
                t.record_stream(main)
            return images, features, pooled
        images = self.preprocess_image(batched_inputs)
        return images, self.backbone(images.tensor), None

    def backward_losses(self, scale=1.0):
        """= (scale * sum(loss_dict.values())).backward() for the loss dict the last training forward returned, without the
        autograd pass: the heads' explicit backward (and, through its hook, the trunk's) is called directly with one
        common upstream gradient.  Same kernels, same arithmetic; ~0.45 ms less host time per eager step (the autograd
        engine's thread hand-over and its scalar add / ones / stack launches).  Returns False when the model has no fused
        head engine state to run (the caller then uses autograd)."""
        eng = getattr(self.roi_heads, "_engine", None)
        st = getattr(eng, "_last_state", None) if eng is not None else None
        if st is None:
            return False
        eng._last_state = None
        eng.backward(st, float(scale))
        return True

    def _backbone_backward(self, dfeat_nhwc, accumulate):
        self.backbone.backward_nhwc(dfeat_nhwc, accumulate)

    def _proposals(self, batched_inputs):
        assert self.load_proposals and "proposals" in batched_inputs[0], "this path uses precomputed proposals"
        return [x["proposals"].to(self.device) for x in batched_inputs]

    def forward(self, batched_inputs):
        if not self.training:
            return self.inference(batched_inputs)
        images, features, pooled = self._features(batched_inputs)
        eng = getattr(self.roi_heads, "_engine", None)
        if eng is not None:
            # MODEL.BACKBONE.FREEZE_AT < 5: the heads' explicit backward hands the feature-map gradient to the trunk's
            eng.feature_grad_hook = self._backbone_backward if self.backbone.trainable() else None
        # image-level labels are read on the host (they come from the loader there): no device round trip
        gt_instances = [x["instances"] for x in batched_inputs] if "instances" in batched_inputs[0] else None
        proposals = self._proposals(batched_inputs)
        if pooled is not None:
            _, detector_losses = self.roi_heads(images, features, proposals, gt_instances, prefetched=pooled)
        else:
            _, detector_losses = self.roi_heads(images, features, proposals, gt_instances)
        losses = {}
        losses.update(detector_losses)
        return losses

    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True):
        assert not self.training
        assert detected_instances is None, "forward_with_given_boxes is off this path"
        with torch.no_grad():
            images = self.preprocess_image(batched_inputs)
            features = self.backbone(images.tensor)
            proposals = self._proposals(batched_inputs)
            results, _, all_scores, all_boxes = self.roi_heads(images, features, proposals, None)
        if do_postprocess:
            return GeneralizedRCNNWSL._postprocess(results, batched_inputs, images.image_sizes)
        return results, all_scores, all_boxes

    @staticmethod
    def _postprocess(instances, batched_inputs, image_sizes):
        processed = []
        for r, inp, size in zip(instances, batched_inputs, image_sizes):
            h, w = inp.get("height", size[0]), inp.get("width", size[1])
            processed.append({"instances": detector_postprocess(r, h, w)})
        return processed


def build_model(cfg):
    """detectron2/modeling/meta_arch/build.py:16-23."""
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
