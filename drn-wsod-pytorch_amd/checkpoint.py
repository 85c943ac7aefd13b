"""Weight I/O: load the released WSL / Detectron checkpoints into the model.

Mirrors `DetectionCheckpointer` (detectron2/checkpoint/detection_checkpoint.py:11-73) over fvcore's `Checkpointer`
(external; its published load / save / resume behaviour is restated), the Caffe2 -> Detectron2 blob renaming of
`convert_basic_c2_names` / `convert_c2_detectron_names` (detectron2/checkpoint/c2_model_loading.py:12-197) for the
families the WSL path can meet (trunk, box head, predictors; RPN / FPN / mask / keypoint blobs are left under their
dotted names and simply do not match), and the longest-suffix alignment of `align_and_update_state_dicts` (:202-313).
The released `resnet{18,50,101}_ws_model_120_d2.pkl` files are Caffe2-style blob dicts whose stem / fc6 / fc7 blobs were
renamed by projects/WSL/tools/convert_resnet_ws_c2.py (`stem_convN_*`, `fc1_*`, `fc2_*`).

Pinned by tests/golden/ckpt_r50c4_tiny.npz: the reference's own functions run on a synthetic WSL-style checkpoint."""
import logging
import os
import pickle
import re

import numpy as np
import torch

from ._cabi import DrnError

__all__ = ["DetectionCheckpointer", "align_and_update_state_dicts", "convert_c2_detectron_names", "load_checkpoint_file"]

log = logging.getLogger(__name__)

# (kind, pattern, replacement): "sub" = regex substitution, "rep" = plain substring replacement; applied in order to
# one key after '_' -> '.'.  Facts of the two naming schemes, c2_model_loading.py:22-66 and :122-127, :165.
_RULES = (
    ("sub", r"\.b$", ".bias"), ("sub", r"\.w$", ".weight"),
    ("sub", r"bn\.s$", "norm.weight"), ("sub", r"bn\.bias$", "norm.bias"), ("sub", r"bn\.rm", "norm.running_mean"),
    ("sub", r"bn\.running.mean$", "norm.running_mean"), ("sub", r"bn\.riv$", "norm.running_var"),
    ("sub", r"bn\.running.var$", "norm.running_var"), ("sub", r"bn\.gamma$", "norm.weight"),
    ("sub", r"bn\.beta$", "norm.bias"), ("sub", r"gn\.s$", "norm.weight"), ("sub", r"gn\.bias$", "norm.bias"),
    ("sub", r"^res\.conv1\.norm\.", "conv1.norm."), ("sub", r"^conv1\.", "stem.conv1."),
    ("rep", ".branch1.", ".shortcut."), ("rep", ".branch2a.", ".conv1."), ("rep", ".branch2b.", ".conv2."),
    ("rep", ".branch2c.", ".conv3."),
    ("sub", r"^bbox\.pred", "bbox_pred"), ("sub", r"^cls\.score", "cls_score"), ("sub", r"^fc6\.", "box_head.fc1."),
    ("sub", r"^fc7\.", "box_head.fc2."), ("sub", r"^head\.conv", "box_head.conv"),
    ("rep", "conv5.mask", "mask_head.deconv"),
)
_HARD = {"pred_b": "linear_b", "pred_w": "linear_w"}


def _rename(key):
    k = _HARD.get(key, key).replace("_", ".")
    for kind, pat, rep in _RULES:
        k = re.sub(pat, rep, k) if kind == "sub" else k.replace(pat, rep)
    return k


def convert_c2_detectron_names(weights):
    """Caffe2 blob dict -> (dict under Detectron2 names, {new name: original name}).  Background-class handling as in
    c2_model_loading.py:178-196: `bbox_pred.*` drops the 4 background rows, `cls_score.*` moves row 0 to the end."""
    new_w, back = {}, {}
    for orig in sorted(weights):
        k = _rename(orig)
        if k in back:
            raise DrnError("two checkpoint blobs map to %s (%s, %s)" % (k, back[k], orig))
        v = weights[orig]
        if k.startswith("bbox_pred."):
            v = v[4:]
        elif k.startswith("cls_score."):
            v = torch.cat([v[1:], v[:1]])
        new_w[k], back[k] = v, orig
    return new_w, back


def align_and_update_state_dicts(model_state_dict, ckpt_state_dict, c2_conversion=True):
    """Every model key takes the checkpoint key that equals it or is its longest '.'-delimited suffix
    (c2_model_loading.py:202-261); shape mismatches are skipped with a warning (:263-275); one checkpoint key matched
    by two model keys is an error (:278-285).  Updates model_state_dict in place with clones; returns
    (matched {ckpt key: model key}, unmatched model keys, unmatched checkpoint keys under their original names)."""
    if c2_conversion:
        ckpt, orig = convert_c2_detectron_names(ckpt_state_dict)
    else:
        ckpt, orig = dict(ckpt_state_dict), {k: k for k in ckpt_state_dict}
    by_leaf = {}
    for ck in ckpt:
        by_leaf.setdefault(ck.rsplit(".", 1)[-1], []).append(ck)
    matched = {}
    for mk in sorted(model_state_dict):
        best = None
        for ck in by_leaf.get(mk.rsplit(".", 1)[-1], ()):
            if (mk == ck or mk.endswith("." + ck)) and (best is None or len(ck) > len(best)):
                best = ck
        if best is None:
            continue
        v = ckpt[best]
        if tuple(model_state_dict[mk].shape) != tuple(v.shape):
            log.warning("Shape of %s in checkpoint is %s, while shape of %s in model is %s: not loaded", best,
                        tuple(v.shape), mk, tuple(model_state_dict[mk].shape))
            continue
        if best in matched:
            raise ValueError("Cannot match one checkpoint key to multiple keys in the model: %s -> %s and %s"
                             % (best, matched[best], mk))
        model_state_dict[mk] = v.clone()
        matched[best] = mk
    loaded = set(matched.values())
    return matched, [k for k in sorted(model_state_dict) if k not in loaded], \
        [orig[k] for k in sorted(ckpt) if k not in matched]


def load_checkpoint_file(filename):
    """DetectionCheckpointer._load_file (detection_checkpoint.py:27-49)"""
    if filename.endswith(".pkl"):
        with open(filename, "rb") as f:
            data = pickle.load(f, encoding="latin1")
        if "model" in data and "__author__" in data:
            return data  # Detectron2 model-zoo format
        if "blobs" in data:  # Caffe2 detection models; ImageNet ones are a flat blob dict
            data = data["blobs"]
        data = {k: v for k, v in data.items() if not k.endswith("_momentum")}
        return {"model": data, "__author__": "Caffe2", "matching_heuristics": True}
    loaded = torch.load(filename, map_location="cpu")
    return loaded if "model" in loaded else {"model": loaded}


class DetectionCheckpointer:
    """`DetectionCheckpointer(model, save_dir, **checkpointables)`: load() / save() / resume_or_load() with the
    reference's file formats; checkpointables are objects with state_dict() / load_state_dict() (optimizer, scheduler)."""

    def __init__(self, model, save_dir="", *, save_to_disk=None, **checkpointables):
        self.model = getattr(model, "module", model)
        self.save_dir = save_dir
        self.save_to_disk = True if save_to_disk is None else save_to_disk
        self.checkpointables = dict(checkpointables)

    # ---- load ---------------------------------------------------------------------------------------
    def load(self, path, checkpointables=None):
        if not path:
            log.info("No checkpoint found. Initializing model from scratch")
            return {}
        if not os.path.isfile(path):
            raise DrnError("Checkpoint %s not found!" % path)
        ckpt = load_checkpoint_file(path)
        self.last_incompatible = self._load_model(ckpt)
        for key in (self.checkpointables if checkpointables is None else checkpointables):
            if key in ckpt:
                self.checkpointables[key].load_state_dict(ckpt.pop(key))
        return {k: v for k, v in ckpt.items() if k != "model"}

    def _load_model(self, ckpt):
        sd = ckpt.pop("model")
        sd = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sd.items()}
        for k, v in sd.items():
            if not isinstance(v, torch.Tensor):
                raise ValueError("Unsupported type found in checkpoint! {}: {}".format(k, type(v)))
        if ckpt.get("matching_heuristics", False):
            msd = self.model.state_dict()
            align_and_update_state_dicts(msd, sd, c2_conversion=ckpt.get("__author__", None) == "Caffe2")
            sd = msd
        else:
            sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
            msd = self.model.state_dict()
            for k in list(sd):  # fvcore drops tensors whose shape differs instead of failing the whole load
                if k in msd and tuple(msd[k].shape) != tuple(sd[k].shape):
                    log.warning("Skip loading parameter %s: shape %s in the checkpoint, %s in the model", k,
                                tuple(sd[k].shape), tuple(msd[k].shape))
                    sd.pop(k)
        inc = self.model.load_state_dict(sd, strict=False)
        missing = [k for k in inc.missing_keys if k not in ("pixel_mean", "pixel_std")]  # set from the config anyway
        return missing, list(inc.unexpected_keys)

    # ---- save / resume ------------------------------------------------------------------------------
    def save(self, name, **kwargs):
        for obj in self.checkpointables.values():
            if hasattr(obj, "sync_master"):
                # sharded optimizer state: every rank gathers the rows the others own - a collective, so it runs on
                # every rank, also on those that do not write the file
                obj.sync_master()
        if not self.save_dir or not self.save_to_disk:
            return
        data = {"model": self.model.state_dict()}
        for key, obj in self.checkpointables.items():
            data[key] = obj.state_dict()
        data.update(kwargs)
        basename = "{}.pth".format(name)
        os.makedirs(self.save_dir, exist_ok=True)
        torch.save(data, os.path.join(self.save_dir, basename))
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(basename)

    def has_checkpoint(self):
        return os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, "last_checkpoint"), "r") as f:
                return os.path.join(self.save_dir, f.read().strip())
        except IOError:
            return ""

    def resume_or_load(self, path, *, resume=True):
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        return self.load(path, checkpointables=[])
