"""Test-time augmentation of the WSL detector, behind the reference's names:
`DatasetMapperTTAAVG` and `GeneralizedRCNNWithTTAAVG` (projects/WSL/wsl/modeling/test_time_augmentation_avg.py:68-137,
:139-321) with the transforms they drive - ResizeShortestEdge (detectron2/data/transforms/augmentation_impl.py:155-175),
ResizeTransform (detectron2/data/transforms/transform.py:83-134), fvcore's HFlipTransform.

Split of work: the mapper (image resize through PIL for uint8 images exactly like the reference, flip, proposal boxes
scaled / flipped / clipped / filtered / top-k) is host-side data preparation, as in the reference.  Everything after it
stays on the GPU: one `model.inference` per augmentation, `drn_tta_accumulate` maps each augmentation's [R, 4K]
predictions back to the original image and folds them into the running box / score averages (the reference copies
them to the host and back for this), and `drn_detect_topk` runs the final score-threshold / NMS / top-k."""
import copy

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from .._cabi import DrnError
from ..structures import Boxes, Instances

__all__ = ["DatasetMapperTTAAVG", "GeneralizedRCNNWithTTAAVG", "resize_shortest_edge_shape"]

VECTORISED_MAPPER = True  # tools: False = the per-augmentation mapper (its proposals are uploaded inside each pass)


def resize_shortest_edge_shape(h, w, size, max_size):
    """ResizeShortestEdge.get_transform, augmentation_impl.py:164-174"""
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


def _resize_image(img_hwc, new_h, new_w):
    """ResizeTransform.apply_image, transform.py:101-122"""
    if img_hwc.dtype == np.uint8:
        try:
            from PIL import Image
        except ImportError as e:  # the reference has the same dependency for uint8 images
            raise DrnError("PIL is needed to resize uint8 images like the reference does") from e
        return np.asarray(Image.fromarray(img_hwc).resize((new_w, new_h), Image.BILINEAR))
    t = torch.from_numpy(np.ascontiguousarray(img_hwc)).permute(2, 0, 1)[None]
    t = F.interpolate(t, (new_h, new_w), mode="bilinear", align_corners=False)
    return t[0].permute(1, 2, 0).numpy()


def _apply_box(boxes, sx, sy, flip_w):
    """Transform.apply_box through [ResizeTransform, HFlipTransform?]: corners in float32, then their bounding box"""
    b = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    idx = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
    c = b[:, idx].reshape(-1, 2).copy()
    c[:, 0] = c[:, 0] * sx
    c[:, 1] = c[:, 1] * sy
    if flip_w is not None:
        c[:, 0] = flip_w - c[:, 0]
    c = c.reshape(-1, 4, 2)
    # (element-wise minima / maxima of the four corners: ndarray.min(axis=1) over the short middle axis took 3 ms per call on the
    # GPU box's host - 97 of the 101 ms of a 16-augmentation mapper call; same values, min / max are exact)
    lo = np.minimum(np.minimum(c[:, 0], c[:, 1]), np.minimum(c[:, 2], c[:, 3]))
    hi = np.maximum(np.maximum(c[:, 0], c[:, 1]), np.maximum(c[:, 2], c[:, 3]))
    return np.concatenate((lo, hi), axis=1)


class DatasetMapperTTAAVG:
    """test_time_augmentation_avg.py:68-137: dataset dict -> list of augmented dataset dicts (len(min_sizes) x
    (2 if flip else 1)); each carries "tta" = (sx, sy, flip_w) of the INVERSE box transform instead of a TransformList."""

    def __init__(self, cfg, device=None):
        # device (round 5): resize + flip of uint8 images run on that device (drn_resize_bilinear_u8: Pillow's integer
        # BILINEAR resample restated - bit-identical images) instead of 8 PIL resizes on the host per image, which took
        # ~0.4 s of a 0.41-s TTA call (tools/tta_bench.py, profiles/r5_20_tta.txt); None = the host path
        self.device = device
        self.min_sizes = cfg.TEST.AUG.MIN_SIZES
        self.max_size = cfg.TEST.AUG.MAX_SIZE
        self.flip = cfg.TEST.AUG.FLIP
        self.image_format = cfg.INPUT.FORMAT
        self.proposal_topk = cfg.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST if cfg.MODEL.LOAD_PROPOSALS else None

    def __call__(self, dataset_dict):
        img = dataset_dict["image"]
        if self.device is not None and img.dtype == torch.uint8 and VECTORISED_MAPPER:
            return self._augmented_device(dataset_dict)
        return list(self._augmented(dataset_dict))

    def _augmented_device(self, dataset_dict):
        """The device path in one sweep: the image goes up once and is resized / flipped there (drn_resize_bilinear_u8); the proposal
        transform of ALL augmentations (transform_proposals, :27-65: scale, flip, bounding box of the corners, clip, drop empty boxes,
        top-k) is ONE set of float32 numpy operations on an [A, R, 4] block - the same operations on the same values as _apply_box
        / Boxes.clip / Boxes.nonempty per augmentation, so the same bits - written into a pinned staging buffer and uploaded in ONE
        asynchronous copy; each augmentation's Instances holds views of it.  (Per augmentation this was ~30 small numpy / torch calls
        and a pageable H2D copy inside its pass, which waits for the stream's queued work: 16-40 ms of host time per image on the GPU
        boxes' hosts, more than the 16 device passes take - profiles/r5_56_tta_batch.txt.)"""
        img = dataset_dict["image"].detach().cpu().permute(1, 2, 0).numpy()
        h, w = img.shape[:2]
        if (dataset_dict["height"], dataset_dict["width"]) != (h, w):
            raise DrnError("TTA on an already resized input (pre_tfm of the reference) is off this path")
        src = torch.from_numpy(np.ascontiguousarray(img)).to(self.device, non_blocking=True)  # [H, W, 3] uint8, once
        augs = [(resize_shortest_edge_shape(h, w, size, self.max_size), fl) for size in self.min_sizes
                for fl in ([False, True] if self.flip else [False])]
        out = []
        for (nh, nw), fl in augs:
            dic = {k: v for k, v in dataset_dict.items() if k not in ("image", "proposals")}
            dic["image"] = ops.resize_bilinear_u8(src, nh, nw, flip=fl)  # fp32 [3, nh, nw] of the same bytes
            dic["tta"] = (w * 1.0 / nw, h * 1.0 / nh, float(nw) if fl else -1.0)
            out.append(dic)
        if self.proposal_topk is None:
            return out
        prop = dataset_dict["proposals"]
        b = prop.proposal_boxes.tensor.detach().cpu().numpy().astype(np.float32, copy=False).reshape(-1, 4)
        logit = prop.objectness_logits.detach().cpu().numpy()
        A, R = len(augs), b.shape[0]
        f32 = np.float32
        sx = np.array([nw * 1.0 / w for (nh, nw), _ in augs], dtype=np.float64).astype(f32).reshape(A, 1, 1)
        sy = np.array([nh * 1.0 / h for (nh, nw), _ in augs], dtype=np.float64).astype(f32).reshape(A, 1, 1)
        cx = b[:, [0, 2, 0, 2]][None] * sx  # [A, R, 4] corner xs (x0, x1, x0, x1), float32 x float32 like `c[:, 0] * sx`
        cy = b[:, [1, 1, 3, 3]][None] * sy
        flipped = np.array([fl for _, fl in augs])
        fw = np.array([nw for (nh, nw), _ in augs], dtype=f32).reshape(A, 1, 1)
        if flipped.any():
            cx[flipped] = fw[flipped] - cx[flipped]
        x0, x1 = cx.min(axis=2), cx.max(axis=2)  # (min / max are exact: any order)
        y0, y1 = cy.min(axis=2), cy.max(axis=2)
        fh = np.array([nh for (nh, nw), _ in augs], dtype=f32).reshape(A, 1)
        fw = fw.reshape(A, 1)
        x0, x1 = np.minimum(np.maximum(x0, f32(0)), fw), np.minimum(np.maximum(x1, f32(0)), fw)  # Boxes.clip
        y0, y1 = np.minimum(np.maximum(y0, f32(0)), fh), np.minimum(np.maximum(y1, f32(0)), fh)
        keep = ((x1 - x0) > 0) & ((y1 - y0) > 0)  # Boxes.nonempty(threshold=0)
        boxes = np.stack((x0, y0, x1, y1), axis=2)  # [A, R, 4]
        k = self.proposal_topk
        cnt = [min(int(keep[a].sum()), k) for a in range(A)]
        n = sum(cnt)
        if getattr(self, "_pin", None) is None or self._pin.shape[0] < 5 * n:
            self._pin = torch.empty((max(5 * n, 1),), dtype=torch.float32).pin_memory()
            self._pin_ev = None
        if self._pin_ev is not None:
            self._pin_ev.synchronize()  # the previous call's copy has left the staging buffer (normally long done)
        pin = self._pin.numpy()
        hb, hl = pin[: 4 * n].reshape(n, 4), pin[4 * n: 5 * n]  # [all boxes | all logits]: every device view below is contiguous
        o = 0
        for a in range(A):
            c = cnt[a]
            if keep[a].all():
                hb[o: o + c] = boxes[a, :c]
                hl[o: o + c] = logit[:c]
            else:
                hb[o: o + c] = boxes[a][keep[a]][:c]
                hl[o: o + c] = logit[keep[a]][:c]
            o += c
        dev = self._pin[: 5 * n].to(self.device, non_blocking=True)
        self._pin_ev = torch.cuda.Event()
        self._pin_ev.record()
        db, dl = dev[: 4 * n].view(n, 4), dev[4 * n: 5 * n]
        o = 0
        for a, ((nh, nw), _) in enumerate(augs):
            p = Instances((nh, nw))
            p.proposal_boxes = Boxes(db[o: o + cnt[a]])
            p.objectness_logits = dl[o: o + cnt[a]]
            out[a]["proposals"] = p
            o += cnt[a]
        return out

    def _augmented(self, dataset_dict):
        img = dataset_dict["image"].detach().cpu().permute(1, 2, 0).numpy()
        h, w = img.shape[:2]
        if (dataset_dict["height"], dataset_dict["width"]) != (h, w):
            raise DrnError("TTA on an already resized input (pre_tfm of the reference) is off this path")
        on_dev = self.device is not None and img.dtype == np.uint8
        if on_dev:
            src = torch.from_numpy(np.ascontiguousarray(img)).to(self.device, non_blocking=True)  # [H, W, 3] uint8, once
        for size in self.min_sizes:
            nh, nw = resize_shortest_edge_shape(h, w, size, self.max_size)
            rimg = None if on_dev else _resize_image(np.copy(img), nh, nw)
            for flipped in ([False, True] if self.flip else [False]):
                dic = {k: v for k, v in dataset_dict.items() if k not in ("image", "proposals")}
                if on_dev:
                    dic["image"] = ops.resize_bilinear_u8(src, nh, nw, flip=flipped)  # fp32 [3, nh, nw] of the same bytes
                else:
                    im = np.flip(rimg, axis=1) if flipped else rimg
                    dic["image"] = torch.from_numpy(np.ascontiguousarray(im.transpose(2, 0, 1)))
                dic["tta"] = (w * 1.0 / nw, h * 1.0 / nh, float(nw) if flipped else -1.0)
                if self.proposal_topk is not None:
                    # transform_proposals (:27-65)
                    prop = dataset_dict["proposals"]
                    boxes = _apply_box(prop.proposal_boxes.tensor.detach().cpu().numpy(), nw * 1.0 / w, nh * 1.0 / h,
                                       nw if flipped else None)
                    boxes = Boxes(torch.from_numpy(boxes))
                    logits = prop.objectness_logits.detach().cpu()
                    boxes.clip((nh, nw))
                    keep = boxes.nonempty(threshold=0)
                    boxes, logits = boxes[keep], logits[keep]
                    p = Instances((nh, nw))
                    p.proposal_boxes = boxes[: self.proposal_topk]
                    p.objectness_logits = logits[: self.proposal_topk]
                    dic["proposals"] = p
                yield dic


class GeneralizedRCNNWithTTAAVG(nn.Module):
    """test_time_augmentation_avg.py:139-321 (box branch; masks / keypoints are off the WSL path)."""

    def __init__(self, cfg, model, tta_mapper=None, batch_size=1):
        super().__init__()
        from .rcnn import GeneralizedRCNNWSL

        model = getattr(model, "module", model)
        assert isinstance(model, GeneralizedRCNNWSL), \
            "TTA is only supported on GeneralizedRCNNWSL. Got a model of type {}".format(type(model))
        self.cfg = cfg.clone()
        assert not self.cfg.MODEL.KEYPOINT_ON, "TTA for keypoint is not supported yet"
        if self.cfg.MODEL.MASK_ON:
            raise DrnError("mask heads are off the DRN-WSOD path")
        if batch_size < 1:
            raise DrnError("batch_size must be >= 1")
        self.model = model
        self.tta_mapper = tta_mapper if tta_mapper is not None else DatasetMapperTTAAVG(cfg, device=model.device)
        self.batch_size = batch_size

    def __call__(self, batched_inputs):
        out = []
        for x in batched_inputs:
            if "image" not in x:
                raise DrnError("reading images from file_name is the data loader's job (off this path)")
            ret = copy.copy(x)
            if "height" not in ret and "width" not in ret:
                ret["height"], ret["width"] = x["image"].shape[1], x["image"].shape[2]
            out.append(self._inference_one_image(ret))
        return out

    def _get_augmented_boxes(self, augmented_inputs):
        """:269-294, on the device: running means of the back-transformed boxes and of the scores"""
        acc_b = acc_s = None
        n = len(augmented_inputs)
        heads = self.model.roi_heads
        # _batch_inference (:200-225): consecutive augmentations in groups of `batch_size` through ONE model.inference (the mapper
        # emits each size's plain and flipped image next to each other, so batch_size = 2 batches same-size pairs: one conv chain
        # per pair); the averages are accumulated per augmentation in the mapper's order either way
        prev, heads.scores_only = getattr(heads, "scores_only", False), True
        try:
            for g0 in range(0, n, self.batch_size):
                group = augmented_inputs[g0: g0 + self.batch_size]
                # the per-pass detections are never used here (the reference computes and drops them): skip the inference tail
                _, scores, boxes = self.model.inference([{k: v for k, v in inp.items() if k != "tta"} for inp in group],
                                                        do_postprocess=False)
                for j, inp in enumerate(group):
                    i = g0 + j
                    sx, sy, flip_w = inp["tta"]
                    b, s = boxes[j][0].contiguous(), scores[j][0].contiguous()
                    if acc_b is None:
                        acc_b, acc_s = torch.empty_like(b), torch.empty_like(s)
                    elif acc_b.shape != b.shape:
                        raise DrnError("augmentations kept different numbers of proposals (%s vs %s): the averages are "
                                       "undefined (the reference fails in torch.cat here)" % (tuple(acc_b.shape), tuple(b.shape)))
                    ops.tta_accumulate(b, s, acc_b, acc_s, np.float32(sx), np.float32(sy), flip_w, i == 0, n if i == n - 1 else 0)
        finally:
            heads.scores_only = prev
        return acc_b, acc_s

    def _inference_one_image(self, input):
        orig_shape = (input["height"], input["width"])
        with torch.no_grad():
            # (queueing each augmentation's pass while the host prepares the next one was built and measured: 261 vs 86 ms per
            # image - the host work between the launches stretches the dependent chain; all augmentations first, then the passes)
            augmented_inputs = self.tta_mapper(input)
            all_boxes, all_scores = self._get_augmented_boxes(augmented_inputs)
            # _merge_detections (:296-309) = fast_rcnn_inference_single_image on the averages
            ob, os_, oc, _ = ops.detect_topk(all_boxes, all_scores, orig_shape, self.cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
                                             self.cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST, self.cfg.TEST.DETECTIONS_PER_IMAGE)
        r = Instances(orig_shape)
        r.pred_boxes = Boxes(ob)
        r.scores = os_
        r.pred_classes = oc
        return {"instances": r}
