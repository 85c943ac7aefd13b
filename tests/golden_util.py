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
    "model_wsddn_r50c4_tiny": O.OracleCfg(arch="wsr50", out_feature="res4", res5_dilation=1, heads="wsddn", refine_num=0,
                                          refine_reg=(), score_thresh=1e-9, nms_thresh=0.5, mean_loss=False, **TINY),
    "model_pcl_r50c4_tiny": O.OracleCfg(arch="wsr50", out_feature="res4", res5_dilation=1, heads="pcl", **TINY),
}
FREEZE_AT = {"model_r50c4_align_tiny": 3}


def csc_case():
    """model_csc_r18dc5_tiny.npz: CSCROIHeads on the tiny WSR-18 DC5 trunk (csc_WSR_18_DC5_1x.yaml + overrides); a fresh
    config each call because `csc_iter` is state"""
    return O.OracleCfg(arch="wsr18", out_feature="res5", res5_dilation=1, stem_out=8, res2_out=64, dan_dim=(48, 64),
                       num_classes=4, heads="csc", refine_num=0, refine_reg=(), score_thresh=1e-9, nms_thresh=0.5,
                       mean_loss=False, base_lr=0.00002, csc_max_iter=2, csc_iter=1, csc_tau=0.15, dropout=0.0)


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


# --------------------------------------------------------------------------------------------
# Product-side helpers (HIP path)
# --------------------------------------------------------------------------------------------
_YAML = {"wsr50": "PascalVOC-Detection/oicr_WSR_50_DC5_1x.yaml", "wsr18": "PascalVOC-Detection/oicr_WSR_18_DC5_1x.yaml",
         "wsr101": "PascalVOC-Detection/oicr_WSR_101_DC5_1x.yaml", "vgg16": "PascalVOC-Detection/oicr_V_16_DC5_1x.yaml"}


def drn_cfg(ocfg, device="cuda", freeze_at=5):
    """Product config equal to what the reference yaml + the fixture's overrides produce.  The yaml files are
    not available on the GPU box, so the values the shipped oicr_* yamls set are spelled out here;
    tests/test_surface_cpu.py checks (in the build container) that loading the unmodified yaml gives the same."""
    from __graft_entry__ import load_package

    load_package()
    from drn_wsod_pytorch_amd.config import add_wsl_config, get_cfg

    cfg = get_cfg()
    add_wsl_config(cfg)
    vgg = ocfg.arch == "vgg16"
    feat = ocfg.out_feature
    L = ["MODEL.META_ARCHITECTURE", "GeneralizedRCNNWSL", "MODEL.DEVICE", device, "MODEL.LOAD_PROPOSALS", "True",
         "MODEL.PIXEL_MEAN", str(list(ocfg.pixel_mean)), "MODEL.BACKBONE.FREEZE_AT", str(freeze_at),
         "MODEL.BACKBONE.NAME", "build_vgg_backbone" if vgg else "build_ws_resnet_backbone",
         "MODEL.ROI_HEADS.NAME", {"pcl": "PCLROIHeads", "wsddn": "WSDDNROIHeads", "csc": "CSCROIHeads"}.get(ocfg.heads, "OICRROIHeads"), "MODEL.ROI_HEADS.NUM_CLASSES", str(ocfg.num_classes),
         "MODEL.ROI_HEADS.IN_FEATURES", str([feat]), "MODEL.ROI_HEADS.SCORE_THRESH_TEST", repr(ocfg.score_thresh),
         "MODEL.ROI_HEADS.NMS_THRESH_TEST", repr(ocfg.nms_thresh), "MODEL.ROI_HEADS.PROPOSAL_APPEND_GT", "False",
         "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", "4096", "MODEL.ROI_HEADS.POSITIVE_FRACTION", "1.0",
         "MODEL.ROI_BOX_HEAD.NAME", "DiscriminativeAdaptionNeck", "MODEL.ROI_BOX_HEAD.POOLER_TYPE", ocfg.pooler_type,
         "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", str(ocfg.pooler_res), "MODEL.ROI_BOX_HEAD.NUM_CONV", "0",
         "MODEL.ROI_BOX_HEAD.NUM_FC", "2", "MODEL.ROI_BOX_HEAD.DAN_DIM", str(list(ocfg.dan_dim)),
         "WSL.MEAN_LOSS", str(bool(ocfg.mean_loss)), "WSL.REFINE_NUM", str(ocfg.refine_num), "WSL.REFINE_REG", str(list(ocfg.refine_reg)),
         "SOLVER.BASE_LR", str(ocfg.base_lr), "SOLVER.WEIGHT_DECAY", "0.0005", "SOLVER.BIAS_LR_FACTOR", "2.0",
         "SOLVER.WEIGHT_DECAY_BIAS", "0.0", "SOLVER.WARMUP_ITERS", "0", "SOLVER.STEPS", "(35000, 50000)",
         "SOLVER.MAX_ITER", "50000", "SOLVER.IMS_PER_BATCH", "4", "WSL.CSC_MAX_ITER", str(ocfg.csc_max_iter)]
    if vgg:
        L += ["MODEL.VGG.DEPTH", "16", "MODEL.VGG.CONV5_DILATION", str(ocfg.res5_dilation)]
    else:
        L += ["MODEL.RESNETS.DEPTH", ocfg.arch[3:], "MODEL.RESNETS.OUT_FEATURES", str([feat]),
              "MODEL.RESNETS.RES5_DILATION", str(ocfg.res5_dilation), "MODEL.RESNETS.STEM_OUT_CHANNELS",
              str(ocfg.stem_out), "MODEL.RESNETS.RES2_OUT_CHANNELS", str(ocfg.res2_out),
              "MODEL.RESNETS.WIDTH_PER_GROUP", str(ocfg.width_per_group)]
    cfg.merge_from_list(L)
    return cfg


def drn_model(ocfg, seed, device="cuda", freeze_at=5, precision="fp32"):
    """Build the product model and fill it with the same name-seeded weights as the reference / oracle."""
    from __graft_entry__ import load_package

    pkg = load_package()
    pkg.set_precision(precision)
    from drn_wsod_pytorch_amd.modeling import build_model

    cfg = drn_cfg(ocfg, device, freeze_at)
    model = build_model(cfg)
    sd = model.state_dict()
    new = {n: (t if n in ("pixel_mean", "pixel_std") else O.seeded_tensor(n, tuple(t.shape), seed)) for n, t in sd.items()}
    model.load_state_dict(new)
    return cfg, model


def drn_inputs(batch, with_gt=True):
    from drn_wsod_pytorch_amd.structures import Boxes, Instances

    out = []
    for b in batch:
        h, w = b["image"].shape[1:]
        prop = Instances((h, w))
        prop.proposal_boxes = Boxes(b["proposal_boxes"])
        prop.objectness_logits = b["objectness_logits"]
        d = {"image": b["image"], "proposals": prop, "height": h, "width": w}
        if with_gt:
            inst = Instances((h, w))
            inst.gt_boxes = Boxes(b["gt_boxes"]) if "gt_boxes" in b else Boxes(torch.zeros(len(b["gt_classes"]), 4))
            inst.gt_classes = b["gt_classes"]
            d["instances"] = inst
        out.append(d)
    return out


def voc_fixture(seed, n_img=7, classes=("cat", "dog", "bus")):
    """synthetic VOC-style ground truth + detections shared by the generator and the tests (deterministic)"""
    rs = np.random.RandomState(seed)
    annos, dets = {}, []
    for i in range(n_img):
        iid = "%06d" % (i + 1)
        objs = []
        for _ in range(rs.randint(0, 4)):
            x0, y0 = rs.randint(1, 200), rs.randint(1, 150)
            objs.append((classes[rs.randint(len(classes))], int(rs.rand() < 0.25),
                         [int(x0), int(y0), int(x0 + rs.randint(20, 120)), int(y0 + rs.randint(20, 100))]))
        annos[iid] = objs
        for name, diff, bb in objs:  # detections near the objects (some good, some shifted, some duplicated)
            for _ in range(rs.randint(0, 3)):
                j = rs.randn(4) * rs.choice([2.0, 25.0])
                dets.append((classes.index(name) if rs.rand() < 0.85 else rs.randint(len(classes)), iid,
                             float(rs.rand()), [bb[0] - 1 + j[0], bb[1] - 1 + j[1], bb[2] + j[2], bb[3] + j[3]]))
        for _ in range(rs.randint(0, 3)):  # pure false positives
            x0, y0 = rs.rand() * 200, rs.rand() * 150
            dets.append((rs.randint(len(classes)), iid, float(rs.rand()), [x0, y0, x0 + 50, y0 + 40]))
    return list(classes), annos, dets
