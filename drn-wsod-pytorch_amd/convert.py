"""Offline converters on either side of the path: the file formats the training path is fed
with.  Host-side numpy / pickle work, like the reference's scripts; nothing here runs per step.

Proposal files (what `data.load_proposals_into_dataset` reads; reference `projects/WSL/tools/proposal_convert.py`):
  * `proposals_from_selective_search` - :16-49.  One MATLAB cell of `[n_i, 4]` boxes per image, 1-indexed and in
    (y1, x1, y2, x2) order -> 0-indexed (x1, y1, x2, y2) int16, every score 1.0 (Selective Search has none).
  * `proposals_from_mcg` - :52-95.  One `<image id>.mat` per image holding `boxes` / `scores` (`bboxes` /
    `bboxes_scores` in the Flickr dumps), same index / order convention, scores kept as float32.
  * `write_proposal_file` - the pickle both end with: `dict(boxes=[...], scores=[...], indexes=[...])`
    (`load_proposals_into_dataset` renames `indexes` -> `ids`, `scores` -> `objectness_logits` on load).

Checkpoint key renaming in front of `checkpoint.DetectionCheckpointer` (which then applies the Caffe2 -> Detectron2
heuristics):
  * `rename_ws_pth_keys` - `projects/WSL/tools/convert_resnet_ws_pth.py:10-30`: the authors' ImageNet training
    checkpoint (`module.backbone.*`, `module.neck.fc*`) -> this model's state_dict names.
  * `rename_ws_c2_blobs` - `convert_resnet_ws_c2.py:68-84`: three-conv stem `conv1_N*` -> `stem_convN*`, `fc6*` / `fc7*`
    -> `fc1*` / `fc2*`.
  * `rename_vgg_blobs` - `convert_vgg.py:68-88`: `convP_C*` -> `plainP_0_convC*`, `fc6*` / `fc7*` -> `fc1*` / `fc2*`.
The two blob renamers keep the reference scripts' substring tests (a key containing "fc" is read as `fc<digit>...`), so
they accept exactly the files those scripts accept."""
import os
import pickle

import numpy as np

__all__ = ["proposals_from_selective_search", "proposals_from_mcg", "write_proposal_file", "rename_ws_pth_keys",
           "rename_ws_c2_blobs", "rename_vgg_blobs", "convert_checkpoint_file"]

_YXYX_TO_XYXY = (1, 0, 3, 2)


def _xyxy0(raw):
    """1-indexed (y1, x1, y2, x2) rows -> 0-indexed (x1, y1, x2, y2), stored as int16 like the reference's files"""
    raw = np.asarray(raw)
    if raw.ndim != 2 or raw.shape[1] != 4:
        raise ValueError("proposal boxes must be [n, 4], got %r" % (raw.shape,))
    return (raw[:, _YXYX_TO_XYXY] - 1).astype(np.int16)


def proposals_from_selective_search(raw_boxes, image_ids):
    """raw_boxes: the `boxes` cell array of the Selective Search .mat (`scipy.io.loadmat(f)["boxes"].ravel()`), or its
    path; image_ids: the dataset's image ids in the SAME order (the reference asserts equal lengths, :23)."""
    if isinstance(raw_boxes, (str, os.PathLike)):
        import scipy.io

        raw_boxes = scipy.io.loadmat(raw_boxes)["boxes"].ravel()
    if len(raw_boxes) != len(image_ids):
        raise ValueError("%d box arrays for %d images" % (len(raw_boxes), len(image_ids)))
    boxes = [_xyxy0(b) for b in raw_boxes]
    scores = [np.ones((len(b),), dtype=np.float32) for b in boxes]
    return dict(boxes=boxes, scores=scores, indexes=list(image_ids))


def proposals_from_mcg(mat_dir, image_ids, file_stems=None, flickr=False):
    """One `<stem>.mat` per image under mat_dir.  file_stems: the names the .mat files go by when they differ from
    the image ids (the reference derives them from the image file name for COCO / Flickr, :66-71); flickr selects that
    dump's variable names (:76-81)."""
    import scipy.io

    bkey, skey = ("bboxes", "bboxes_scores") if flickr else ("boxes", "scores")
    stems = image_ids if file_stems is None else file_stems
    if len(stems) != len(image_ids):
        raise ValueError("%d file stems for %d images" % (len(stems), len(image_ids)))
    boxes, scores = [], []
    for stem in stems:
        m = scipy.io.loadmat(os.path.join(mat_dir, "%s.mat" % (stem,)))
        b = _xyxy0(m[bkey])
        s = np.squeeze(np.asarray(m[skey]).astype(np.float32))
        if s.ndim == 0:  # a single proposal squeezes to a scalar: the loader sorts and indexes per image
            s = s.reshape(1)
        if len(s) != len(b):
            raise ValueError("%s.mat: %d boxes, %d scores" % (stem, len(b), len(s)))
        boxes.append(b)
        scores.append(s)
    return dict(boxes=boxes, scores=scores, indexes=list(image_ids))


def write_proposal_file(path, proposals):
    with open(path, "wb") as f:
        pickle.dump(dict(boxes=proposals["boxes"], scores=proposals["scores"], indexes=proposals["indexes"]), f,
                    pickle.HIGHEST_PROTOCOL)


# ---- checkpoint keys ------------------------------------------------------------------------------------------------
def rename_ws_pth_keys(state_dict):
    """`module.neck.fc*` -> `roi_heads.box_head.fc*`, `module.backbone.*` -> `backbone.*`, other `module.neck.*` ->
    `roi_heads.box_head.*`; anything else is kept under its name (the reference prints a warning and keeps it)."""
    table = (("module.neck.fc", "roi_heads.box_head.fc"), ("module.backbone.", "backbone."),
             ("module.neck.", "roi_heads.box_head."))
    out = {}
    for k, v in state_dict.items():
        for old, new in table:
            if old in k:
                k = k.replace(old, new)
                break
        out[k] = v
    return out


def _fc_shift(k):
    # "fc6_w" -> "fc1_w", "fc7_b" -> "fc2_b": the digit after "fc" minus 5
    return "fc%d%s" % (int(k[2]) - 5, k[3:])


def rename_ws_c2_blobs(blobs):
    out = {}
    for k, v in blobs.items():
        nk = k
        if "conv1_" in k and "res" not in k:  # the three-conv stem of the WS-ResNets: conv1_1_w, conv1_2_bn_s, ...
            nk = "stem_conv%s%s" % (k[6], k[7:])
        if "fc" in k:
            nk = _fc_shift(k)
        out[nk] = v
    return out


def rename_vgg_blobs(blobs):
    out = {}
    for k, v in blobs.items():
        nk = k
        if "conv" in k:  # conv3_2_w -> plain3_0_conv2_w
            nk = "plain%s_0_conv%s%s" % (k[4], k[6], k[7:])
        elif "fc" in k:
            nk = _fc_shift(k)
        out[nk] = v
    return out


def _blobs_of(data):
    """the `_load_file` of both blob scripts (convert_resnet_ws_c2.py:44-59): a Detectron2 zoo file passes its "model"
    through; a Caffe2 file is its "blobs" (detection models) or itself (ImageNet models) without the momentum blobs"""
    if "model" in data and "__author__" in data:
        return data["model"]
    if "blobs" in data:
        data = data["blobs"]
    return {k: v for k, v in data.items() if not k.endswith("_momentum")}


def convert_checkpoint_file(src, dst, kind):
    """kind: "ws_c2" / "vgg" (pickled blob dicts in, pickle protocol 2 out, like the scripts' save_object) or "ws_pth"
    (torch checkpoint with a "state_dict" in, torch.save out)."""
    if kind == "ws_pth":
        import torch

        torch.save(rename_ws_pth_keys(torch.load(src, map_location="cpu")["state_dict"]), dst)
        return
    if kind not in ("ws_c2", "vgg"):
        raise ValueError("kind must be 'ws_c2', 'vgg' or 'ws_pth'")
    with open(src, "rb") as f:
        blobs = _blobs_of(pickle.load(f, encoding="latin1"))
    out = rename_ws_c2_blobs(blobs) if kind == "ws_c2" else rename_vgg_blobs(blobs)
    tmp = "%s.tmp.%d" % (dst, os.getpid())
    with open(tmp, "wb") as f:
        pickle.dump(out, f, 2)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, dst)
