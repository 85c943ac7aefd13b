"""PASCAL VOC detection evaluation (second half): turns the detector's / the TTA wrapper's outputs
into AP and CorLoc, behind the reference's names: `PascalVOCDetectionEvaluator`, `voc_ap`, `voc_eval`,
`voc_eval_corloc`, `parse_rec` (detectron2/evaluation/pascal_voc_evaluation.py:21-180, :182-234, :237-350, :353-447;
the CorLoc metric is this fork's addition for weakly supervised detection).

Same arithmetic, different plumbing: predictions stay in memory (the reference writes one text file per class and reads
it back); they still go through the reference's text quantisation (score %.3f, box %.1f after the +1 shift of
xmin / ymin, :58-66) because the ranking and the overlaps are computed on those rounded numbers.  Ground truth is read
from VOC XML files or handed over as {image_id: [(class name, difficult, [xmin, ymin, xmax, ymax])]}.
Pinned by tests/golden/voc_eval.npz (the reference's functions on a synthetic annotation set)."""
import os
import xml.etree.ElementTree as ET
from collections import OrderedDict, defaultdict

import numpy as np

__all__ = ["PascalVOCDetectionEvaluator", "parse_rec", "voc_ap", "voc_eval", "voc_eval_corloc", "format_prediction"]


def parse_rec(filename):
    """VOC XML -> [(name, difficult, [xmin, ymin, xmax, ymax])]"""
    out = []
    for obj in ET.parse(filename).findall("object"):
        bb = obj.find("bndbox")
        out.append((obj.find("name").text, int(obj.find("difficult").text),
                    [int(bb.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax")]))
    return out


def format_prediction(image_id, score, box):
    """the line PascalVOCDetectionEvaluator.process writes (:58-66): 1-based xmin / ymin, 3 / 1 decimals"""
    xmin, ymin, xmax, ymax = box
    xmin += 1
    ymin += 1
    return f"{image_id} {score:.3f} {xmin:.1f} {ymin:.1f} {xmax:.1f} {ymax:.1f}"


def voc_ap(rec, prec, use_07_metric=False):
    if use_07_metric:  # 11-point interpolation of VOC07
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def _class_gt(annos, classname):
    gt, npos, npos_im = {}, 0, 0
    for iid, objs in annos.items():
        R = [o for o in objs if o[0] == classname]
        difficult = np.array([o[1] for o in R]).astype(bool)
        gt[iid] = {"bbox": np.array([o[2] for o in R]), "difficult": difficult, "det": [False] * len(R)}
        npos += int(sum(~difficult))
        if len(R) > 0:
            npos_im += min(1, int(sum(~difficult)))
    return gt, npos, npos_im


def _ranked(lines):
    split = [x.strip().split(" ") for x in lines]
    ids = [x[0] for x in split]
    conf = np.array([float(x[1]) for x in split])
    BB = np.array([[float(z) for z in x[2:]] for x in split]).reshape(-1, 4)
    order = np.argsort(-conf)
    return [ids[k] for k in order], BB[order, :]


def _max_overlap(bb, BBGT):
    """VOC devkit IoU with the +1 pixel convention; (-inf, -1) when the image has no box of the class"""
    if BBGT.size == 0:
        return -np.inf, -1
    iw = np.maximum(np.minimum(BBGT[:, 2], bb[2]) - np.maximum(BBGT[:, 0], bb[0]) + 1.0, 0.0)
    ih = np.maximum(np.minimum(BBGT[:, 3], bb[3]) - np.maximum(BBGT[:, 1], bb[1]) + 1.0, 0.0)
    inters = iw * ih
    uni = ((bb[2] - bb[0] + 1.0) * (bb[3] - bb[1] + 1.0) + (BBGT[:, 2] - BBGT[:, 0] + 1.0) * (BBGT[:, 3] - BBGT[:, 1] + 1.0)
           - inters)
    ov = inters / uni
    return np.max(ov), int(np.argmax(ov))


def voc_eval(lines, annos, classname, ovthresh=0.5, use_07_metric=False):
    """lines: prediction lines of this class (format_prediction); annos: {image_id: [(name, difficult, bbox)]}.
    Returns (rec, prec, ap) like the reference's voc_eval (:237-350)."""
    gt, npos, _ = _class_gt(annos, classname)
    ids, BB = _ranked(lines)
    nd = len(ids)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d in range(nd):
        R = gt[ids[d]]
        ovmax, jmax = _max_overlap(BB[d, :].astype(float), R["bbox"].astype(float))
        if ovmax > ovthresh:
            if not R["difficult"][jmax]:
                if not R["det"][jmax]:
                    tp[d] = 1.0
                    R["det"][jmax] = 1
                else:
                    fp[d] = 1.0
        else:
            fp[d] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos) if npos > 0 else np.zeros_like(tp)  # no object of this class anywhere: recall 0, AP 0
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def voc_eval_corloc(lines, annos, classname, ovthresh=0.5, use_07_metric=False):
    """CorLoc (:353-447): fraction of images containing the class whose top-ranked detection of that class hits"""
    gt, _, npos_im = _class_gt(annos, classname)
    if len(lines) == 0 or npos_im == 0:  # (the reference divides by zero for a class that no image contains)
        return 0.0
    ids, BB = _ranked(lines)
    hit, miss = [], []
    for d in range(len(ids)):
        if ids[d] in hit or ids[d] in miss:
            continue
        R = gt[ids[d]]
        if all(R["difficult"]):
            continue
        ovmax, _ = _max_overlap(BB[d, :].astype(float), R["bbox"].astype(float))
        (hit if ovmax > ovthresh else miss).append(ids[d])
    return 1.0 * len(hit) / npos_im


class PascalVOCDetectionEvaluator:
    """reset() / process(inputs, outputs) / evaluate() like the reference's evaluator.  `annotations` is either the
    dict described above or None, in which case `dirname/Annotations/{id}.xml` of the ids listed in
    `dirname/ImageSets/Main/{split}.txt` are parsed."""

    def __init__(self, class_names, dirname=None, split="test", year=2007, annotations=None, gather=None):
        assert year in (2007, 2012), year
        self._class_names, self._is_2007 = list(class_names), year == 2007
        if annotations is None:
            with open(os.path.join(dirname, "ImageSets", "Main", split + ".txt")) as f:
                ids = [x.strip() for x in f.readlines()]
            annotations = {i: parse_rec(os.path.join(dirname, "Annotations", i + ".xml")) for i in ids}
        self._annos = annotations
        self._gather = gather  # callable(list-per-class dict) -> list of such dicts (one per rank); None = single process
        self.reset()

    def reset(self):
        self._predictions = defaultdict(list)

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            inst = out["instances"]
            boxes = inst.pred_boxes.tensor.detach().cpu().numpy()
            for box, score, cls in zip(boxes, inst.scores.tolist(), inst.pred_classes.tolist()):
                self._predictions[cls].append(format_prediction(inp["image_id"], score, box))

    def evaluate(self):
        parts = self._gather(self._predictions) if self._gather is not None else [self._predictions]
        if parts is None:
            return None  # not the main process
        preds = defaultdict(list)
        for part in parts:
            for c, lines in part.items():
                preds[c].extend(lines)
        aps, cls_ = defaultdict(list), defaultdict(list)
        for ci, name in enumerate(self._class_names):
            lines = preds.get(ci, [])
            for thr in range(50, 100, 5):
                ap = voc_eval(lines, self._annos, name, thr / 100.0, self._is_2007)[2] if lines else 0.0
                aps[thr].append(ap * 100)
                cls_[thr].append(voc_eval_corloc(lines, self._annos, name, thr / 100.0, self._is_2007) * 100)
        ret = OrderedDict()
        m = {t: np.mean(x) for t, x in aps.items()}
        ret["bbox"] = {"AP": np.mean(list(m.values())), "AP50": m[50], "AP75": m[75]}
        m = {t: np.mean(x) for t, x in cls_.items()}
        ret["bbox CorLoc"] = {"CL": np.mean(list(m.values())), "CL50": m[50], "CL75": m[75]}
        ret["per_class"] = {"AP50": dict(zip(self._class_names, aps[50])), "CL50": dict(zip(self._class_names, cls_[50]))}
        return ret
