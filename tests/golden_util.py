"""Helpers shared by the tests: load golden fixtures, rebuild their configs/inputs for the oracle
and for the HIP path. (Fixtures were produced by tests/golden/gen_golden.py from the reference.)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle import wsod_oracle as O  # noqa: E402

TINY = dict(stem_out=8, res2_out=32, width_per_group=8, dan_dim=(48, 64), num_classes=5)

MODEL_CASES = {
    "model_r50dc5_tiny": O.OracleCfg(arch="wsr50", out_feature="res5", res5_dilation=2, **TINY),
    "model_r50c4_tiny": O.OracleCfg(arch="wsr50", out_feature="res4", res5_dilation=1, **TINY),
    "model_r50c4_align_tiny": O.OracleCfg(arch="wsr50", out_feature="res4", res5_dilation=1,
                                          pooler_type="ROIAlignV2", **TINY),
    "model_r18dc5_tiny": O.OracleCfg(arch="wsr18", out_feature="res5", res5_dilation=2, stem_out=8, res2_out=64,
                                     dan_dim=(48, 64), num_classes=5),
    "model_vgg16_small": O.OracleCfg(arch="vgg16", out_feature="plain5", res5_dilation=2, dan_dim=(64, 64),
                                     num_classes=5, pixel_mean=(103.939, 116.779, 123.68), base_lr=0.001),
    "model_r50c4_dropmask_tiny": O.OracleCfg(arch="wsr50", out_feature="res4", res5_dilation=1, **TINY),
    "model_r50c4_reg_tiny": O.OracleCfg(arch="wsr50", out_feature="res4", res5_dilation=1, refine_num=4,
                                        refine_reg=(False, False, False, True), **TINY),
}
FREEZE_AT = {"model_r50c4_align_tiny": 3}


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def batch_from(d):
    batch = []
    for i in range(int(d["n_img"])):
        batch.append({k: torch.from_numpy(d["in%d_%s" % (i, k)]) for k in
                      ("image", "proposal_boxes", "objectness_logits", "gt_classes", "gt_boxes")})
    return batch


def dropmasks_from(d):
    if "dropmask0" in d:
        return [torch.from_numpy(d["dropmask0"]), torch.from_numpy(d["dropmask1"])]
    return None
