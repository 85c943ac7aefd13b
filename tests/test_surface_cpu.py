"""CPU tests of the host-side operator surface: registries, config loading of the UNMODIFIED reference yaml
files (when /root/reference is present), state_dict keys/shapes equal to the reference's (through the
reference-pinned oracle table), optimizer parameter groups, LR schedule, and loud failure off the path."""
import os

import pytest
import torch

import golden_util as G
from __graft_entry__ import load_package

O = G.O
REF_CFG = "/root/reference/projects/WSL/configs"


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def _build(ocfg, freeze_at=5):
    from drn_wsod_pytorch_amd.modeling import build_model

    return build_model(G.drn_cfg(ocfg, "cpu", freeze_at))


@pytest.mark.parametrize("name", sorted(G.MODEL_CASES))
def test_state_dict_matches_reference_names(pkg, name):
    ocfg = G.MODEL_CASES[name]
    model = _build(ocfg)
    sd = {k: tuple(v.shape) for k, v in model.state_dict().items() if k not in ("pixel_mean", "pixel_std")}
    assert sd == {k: tuple(v) for k, v in O.param_shapes(ocfg).items()}
    trainable = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    d = G.load(name)
    if G.FREEZE_AT.get(name, 5) == 5:
        assert trainable == sorted(d["trainable"].tolist())


def test_full_size_r50_c4_shapes(pkg):
    """BASELINE configs[1]: R50-WS truncated at res4 -> fc6 = Linear(50176, 2048); 112,560,471 trainable."""
    ocfg = O.OracleCfg(arch="wsr50", out_feature="res4", res5_dilation=1)
    model = _build(ocfg)
    assert tuple(model.roi_heads.box_head.fc1.weight.shape) == (2048, 50176)
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 112560471
    assert model.backbone.output_shape()["res4"].stride == 16 and model.backbone.output_shape()["res4"].channels == 1024


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference yaml files only exist in the build container")
@pytest.mark.parametrize("arch,yaml_rel", [("wsr50", "PascalVOC-Detection/oicr_WSR_50_DC5_1x.yaml"),
                                            ("wsr18", "PascalVOC-Detection/oicr_WSR_18_DC5_1x.yaml"),
                                            ("wsr101", "PascalVOC-Detection/oicr_WSR_101_DC5_1x.yaml"),
                                            ("vgg16", "PascalVOC-Detection/oicr_V_16_DC5_1x.yaml"),
                                            ("wsr50", "COCO-Detection/oicr_WSR_50_DC5_1x.yaml"),
                                            ("wsr50", "PascalVOC-Detection/reg/oicr_WSR_50_DC5_1x.yaml"),
                                            ("wsr50", "PascalVOC-Detection/pcl_WSR_50_DC5_1x.yaml"),
                                            ("wsr50", "PascalVOC-Detection/wsddn_WSR_50_DC5_1x.yaml")])
def test_unmodified_reference_yaml_loads(pkg, arch, yaml_rel):
    from drn_wsod_pytorch_amd.config import add_wsl_config, get_cfg
    from drn_wsod_pytorch_amd.modeling import build_model

    path = os.path.join(REF_CFG, yaml_rel)
    if not os.path.exists(path):
        pytest.skip(yaml_rel)
    cfg = get_cfg()
    add_wsl_config(cfg)
    cfg.merge_from_file(path)
    cfg.merge_from_list(["MODEL.DEVICE", "cpu"])
    assert cfg.MODEL.META_ARCHITECTURE == "GeneralizedRCNNWSL"
    base = os.path.basename(yaml_rel)
    assert cfg.MODEL.ROI_HEADS.NAME == {"pcl": "PCLROIHeads", "wsddn": "WSDDNROIHeads"}.get(base.split("_")[0], "OICRROIHeads")
    assert cfg.MODEL.BACKBONE.FREEZE_AT == 5 and cfg.SOLVER.BIAS_LR_FACTOR == 2.0 and cfg.SOLVER.WEIGHT_DECAY_BIAS == 0.0
    assert tuple(cfg.SOLVER.STEPS) == (35000, 50000) or "COCO" in yaml_rel or base.startswith("wsddn")
    if base.startswith("wsddn"):
        assert cfg.WSL.MEAN_LOSS is False and cfg.WSL.ITER_SIZE == 32 and cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST == 0.5
        m = build_model(cfg)
        assert type(m.roi_heads).__name__ == "WSDDNROIHeads" and not any("box_refinery" in k for k in m.state_dict())
        return
    if "COCO" not in yaml_rel and "reg/" not in yaml_rel:
        K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        feat = "plain5" if arch == "vgg16" else "res5"
        ocfg = O.OracleCfg(arch=arch, out_feature=feat, res5_dilation=2, num_classes=K,
                           heads="pcl" if "pcl_" in yaml_rel else "oicr", dan_dim=tuple(cfg.MODEL.ROI_BOX_HEAD.DAN_DIM), res2_out=cfg.MODEL.RESNETS.RES2_OUT_CHANNELS,
                           pixel_mean=tuple(cfg.MODEL.PIXEL_MEAN), base_lr=cfg.SOLVER.BASE_LR)
        ref = G.drn_cfg(ocfg, "cpu")
        for key in ("MODEL.ROI_HEADS", "MODEL.ROI_BOX_HEAD", "MODEL.BACKBONE", "WSL"):
            a, b = cfg, ref
            for part in key.split("."):
                a, b = a[part], b[part]
            for k in b:
                assert a[k] == b[k] or list(a[k]) == list(b[k]), (key, k, a[k], b[k])
    if arch in ("wsr18",):
        m = build_model(cfg)
        assert "backbone.res5.1.conv2.weight" in m.state_dict()


def test_optimizer_groups_and_schedule(pkg):
    from drn_wsod_pytorch_amd.engine import WarmupMultiStepLR

    class _Opt:
        param_groups = [{"lr": 0.01, "initial_lr": 0.01}, {"lr": 0.02, "initial_lr": 0.02}]

    sched = WarmupMultiStepLR(_Opt, (3, 5), 0.1, warmup_factor=0.001, warmup_iters=2)
    lrs = [_Opt.param_groups[0]["lr"]]
    for _ in range(6):
        sched.step()
        lrs.append(_Opt.param_groups[0]["lr"])
    exp = [0.01 * 0.001, 0.01 * (0.001 * 0.5 + 0.5), 0.01, 0.001, 0.001, 0.0001, 0.0001]
    assert all(abs(a - b) < 1e-12 for a, b in zip(lrs, exp)), lrs


def test_off_path_fails_loudly(pkg):
    from drn_wsod_pytorch_amd._cabi import DrnError
    from drn_wsod_pytorch_amd.config import add_wsl_config, get_cfg
    from drn_wsod_pytorch_amd.modeling import build_model

    ocfg = G.MODEL_CASES["model_r50c4_tiny"]
    cfg = G.drn_cfg(ocfg, "cpu")
    cfg.merge_from_list(["MODEL.ROI_HEADS.NAME", "CSCOICRROIHeads"])  # named by csc_oicr_V_16_DC5_1x.yaml, absent from the
    with pytest.raises(KeyError):                                     # reference's own wsl/modeling/roi_heads too
        build_model(cfg)
    cfg.merge_from_list(["MODEL.ROI_HEADS.NAME", "CSCROIHeads", "WSL.REFINE_NUM", "0"])  # built
    m = build_model(cfg)
    assert type(m.roi_heads).__name__ == "CSCROIHeads" and m.cpg and m.backbone.input_grad
    assert m.roi_heads.image_grad_fn == m.backbone.input_gradient_nhwc and m.roi_heads.refine_K == 0
    cfg.merge_from_list(["WSL.REFINE_NUM", str(ocfg.refine_num)])
    cfg.merge_from_list(["MODEL.ROI_HEADS.NAME", "PCLROIHeads"])  # built: same module tree as OICR
    m = build_model(cfg)
    assert type(m.roi_heads).__name__ == "PCLROIHeads" and m.roi_heads.refine_mode == "pcl"
    assert set(m.state_dict()) == set(_build(ocfg).state_dict())
    model = _build(ocfg)
    model.train()
    batch = G.drn_inputs(G.batch_from(G.load("model_r50c4_tiny")))
    with pytest.raises((DrnError, AssertionError)):  # CPU tensors: the product has no CPU path
        model(batch)


def test_instances_boxes_api(pkg):
    from drn_wsod_pytorch_amd.structures import Boxes, Instances

    b = Boxes(torch.tensor([[0.0, 0, 10, 10], [5, 5, 5, 9]]))
    assert b.area().tolist() == [100.0, 0.0] and b.nonempty().tolist() == [True, False]
    i = Instances((20, 30), pred_boxes=b, scores=torch.tensor([0.5, 0.2]))
    assert len(i) == 2 and len(i[i.scores > 0.3]) == 1 and i.image_size == (20, 30)
    with pytest.raises(AssertionError):
        i.bad = torch.zeros(3)
    assert len(Instances.cat([i, i])) == 4


def test_tta_mapper_matches_reference_golden():
    """DatasetMapperTTAAVG (host-side data preparation of the TTA path) vs the reference's own mapper: augmented
    images (PIL resize + flip) and transformed / clipped / top-k proposals bit for bit"""
    import numpy as np
    import torch

    import golden_util as G
    from drn_wsod_pytorch_amd.modeling import DatasetMapperTTAAVG
    from drn_wsod_pytorch_amd.structures import Boxes, Instances

    d = G.load("tta_r50c4_tiny")
    cfg = G.drn_cfg(G.MODEL_CASES["model_r50c4_tiny"], "cpu")
    cfg.merge_from_list(["TEST.AUG.MIN_SIZES", str(tuple(int(x) for x in d["min_sizes"])), "TEST.AUG.MAX_SIZE",
                         str(int(d["max_size"])), "TEST.AUG.FLIP", "True", "DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST",
                         str(int(d["topk"]))])
    img = torch.from_numpy(d["image_u8"])
    H, W = img.shape[1:]
    prop = Instances((H, W))
    prop.proposal_boxes = Boxes(torch.from_numpy(d["proposal_boxes"]))
    prop.objectness_logits = torch.from_numpy(d["objectness_logits"])
    augs = DatasetMapperTTAAVG(cfg)({"image": img, "proposals": prop, "height": H, "width": W})
    assert len(augs) == int(d["n_aug"])
    for i, a in enumerate(augs):
        assert np.array_equal(a["image"].numpy(), d["aug%d_image" % i]), i
        assert np.array_equal(a["proposals"].proposal_boxes.tensor.numpy(), d["aug%d_boxes" % i]), i
        assert np.array_equal(a["proposals"].objectness_logits.numpy(), d["aug%d_obj" % i]), i


def test_checkpoint_alignment_matches_reference_golden(tmp_path):
    """checkpoint.py vs the reference's own convert_c2_detectron_names / align_and_update_state_dicts on the synthetic
    WSL-style Caffe2 checkpoint of tests/golden/gen_golden.py `ckpt`: identical renaming and identical
    model-key <- blob assignment (Caffe2 mode incl. the skipped shape mismatch and the ignored decoys; plain mode),
    then the file-level paths: Caffe2 .pkl, Detectron2-zoo .pkl, .pth save / resume."""
    import pickle

    import numpy as np
    import torch

    import golden_util as G
    from __graft_entry__ import load_package

    load_package()
    from drn_wsod_pytorch_amd import checkpoint as C
    from drn_wsod_pytorch_amd.modeling import build_model

    d = G.load("ckpt_r50c4_tiny")
    model = build_model(G.drn_cfg(G.MODEL_CASES["model_r50c4_tiny"], "cpu"))
    sd = model.state_dict()
    assert list(sd.keys()) == list(d["model_keys"])
    keys = [str(k) for k in d["ckpt_keys"]]
    ckpt = {k: torch.full(tuple(int(x) for x in d["ckpt_shape%d" % i]), float(i + 1)) for i, k in enumerate(keys)}
    blobs = {k: v for k, v in ckpt.items() if not k.endswith("_momentum")}
    new_w, back = C.convert_c2_detectron_names(blobs)
    assert sorted(new_w) == [str(x) for x in d["renamed"]]
    assert [back[k] for k in sorted(new_w)] == [str(x) for x in d["renamed_orig"]]
    msd = {k: torch.zeros_like(v) for k, v in sd.items()}
    C.align_and_update_state_dicts(msd, blobs, c2_conversion=True)
    assert [int(v.reshape(-1)[0]) for v in msd.values()] == d["map_c2"].tolist()
    src = {k[len("backbone."):]: torch.full(tuple(v.shape), float(j + 1)) for j, (k, v) in enumerate(sd.items())
           if k.startswith("backbone.")}  # same construction as the generator: value = 1-based index among ALL model keys
    assert list(src.keys()) == [str(x) for x in d["d2_keys"]]
    msd = {k: torch.zeros_like(v) for k, v in sd.items()}
    C.align_and_update_state_dicts(msd, src, c2_conversion=False)
    assert [int(v.reshape(-1)[0]) for v in msd.values()] == d["map_d2"].tolist()
    # file level: Caffe2-style pkl (numpy blobs + a momentum blob) -> model
    f = tmp_path / "wsl_c2.pkl"
    with open(f, "wb") as fh:
        pickle.dump({"blobs": {k: v.numpy() for k, v in ckpt.items()}}, fh)
    ck = C.DetectionCheckpointer(model, str(tmp_path))
    ck.load(str(f))
    got = model.state_dict()
    for mk, idx in zip(sd.keys(), d["map_c2"].tolist()):
        if idx > 0:
            assert float(got[mk].reshape(-1)[0]) == float(idx), mk
    # .pth round trip + resume
    ck.save("model_0000001", iteration=1)
    model2 = build_model(G.drn_cfg(G.MODEL_CASES["model_r50c4_tiny"], "cpu"))
    extra = C.DetectionCheckpointer(model2, str(tmp_path)).resume_or_load("", resume=True)
    assert extra["iteration"] == 1
    for k, v in model.state_dict().items():
        assert torch.equal(v, model2.state_dict()[k]), k
    # Detectron2-zoo pkl: exact names, no heuristics
    f2 = tmp_path / "zoo.pkl"
    with open(f2, "wb") as fh:
        pickle.dump({"model": {k: v.numpy() for k, v in model.state_dict().items()}, "__author__": "test"}, fh)
    model3 = build_model(G.drn_cfg(G.MODEL_CASES["model_r50c4_tiny"], "cpu"))
    C.DetectionCheckpointer(model3).load(str(f2))
    for k, v in model.state_dict().items():
        assert torch.equal(v, model3.state_dict()[k]), k


def test_data_path_matches_reference_golden(tmp_path):
    """data.py vs the reference's own load_proposals_into_dataset + DatasetMapper (tests/golden/gen_golden.py `data`):
    proposal file with Detectron1 key names, then four TRAIN draws under the same numpy seed (random crop, multi-scale
    resize through PIL, flip, brightness / saturation blends, box annotations, proposals transformed / clipped /
    de-duplicated / top-k) and one TEST pass - images, boxes, logits and classes bit for bit."""
    import pickle

    import numpy as np
    from PIL import Image

    import golden_util as G
    from __graft_entry__ import load_package

    load_package()
    from drn_wsod_pytorch_amd import data as D

    d = G.load("data_mapper")
    cfg = G.drn_cfg(G.MODEL_CASES["model_r50c4_tiny"], "cpu")
    cfg.merge_from_list(["INPUT.MIN_SIZE_TRAIN", "(48, 64, 80)", "INPUT.MAX_SIZE_TRAIN", "120", "INPUT.MIN_SIZE_TEST", "64",
                         "INPUT.MAX_SIZE_TEST", "100", "INPUT.CROP.ENABLED", "True",
                         "DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TRAIN", "30", "DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST", "25"])
    rgb = d["rgb"]
    H, W = rgb.shape[:2]
    fn = str(tmp_path / "000123.png")
    Image.fromarray(rgb).save(fn)
    pf = str(tmp_path / "props.pkl")
    with open(pf, "wb") as f:
        pickle.dump({"indexes": [7, 123], "boxes": [np.zeros((3, 4), np.float32), d["boxes"]],
                     "scores": [np.zeros(3, np.float32), d["scores"]]}, f)
    annos = [{"bbox": [10.0, 8.0, 50.0, 40.0], "bbox_mode": 0, "category_id": 3},
             {"bbox": [30.5, 20.25, 80.0, 58.0], "bbox_mode": 0, "category_id": 1},
             {"bbox": [5.0, 5.0, 20.0, 20.0], "bbox_mode": 0, "category_id": 2, "iscrowd": 1}]
    rec = {"file_name": fn, "height": H, "width": W, "image_id": 123, "annotations": annos}
    recs = D.load_proposals_into_dataset([dict(rec)], pf)
    assert np.array_equal(recs[0]["proposal_boxes"], d["loaded_boxes"])
    assert np.array_equal(recs[0]["proposal_objectness_logits"], d["loaded_logits"])
    for tag, is_train, nrep in (("train", True, 4), ("test", False, 1)):
        mapper = D.DatasetMapper(cfg, is_train)
        np.random.seed(int(d["seed"]))
        for rep in range(nrep):
            out = mapper(recs[0])
            k = "%s%d_" % (tag, rep)
            assert np.array_equal(out["image"].numpy(), d[k + "image"]), k
            assert np.array_equal(out["proposals"].proposal_boxes.tensor.numpy(), d[k + "prop_boxes"]), k
            assert np.array_equal(out["proposals"].objectness_logits.numpy(), d[k + "prop_logits"]), k
            if is_train:
                assert np.array_equal(out["instances"].gt_boxes.tensor.numpy(), d[k + "gt_boxes"]), k
                assert np.array_equal(out["instances"].gt_classes.numpy(), d[k + "gt_classes"]), k


def test_voc_evaluation_matches_reference_golden():
    """evaluation.py vs the reference's voc_eval / voc_eval_corloc on the shared synthetic VOC fixture: recall and
    precision curves, AP (VOC07 11-point and area) and CorLoc for every class and IoU threshold, and the evaluator's
    aggregated dict through process() / evaluate()"""
    import numpy as np
    import torch

    import golden_util as G
    from __graft_entry__ import load_package

    load_package()
    from drn_wsod_pytorch_amd import evaluation as E
    from drn_wsod_pytorch_amd.structures import Boxes, Instances

    d = G.load("voc_eval")
    classes, annos, dets = G.voc_fixture(int(d["seed"]))
    lines = {c: [] for c in range(len(classes))}
    for c, iid, score, box in dets:
        lines[c].append(E.format_prediction(iid, score, np.array(box, dtype=np.float32)))
    for year07, tag in ((True, "y07"), (False, "y12")):
        for ci, name in enumerate(classes):
            for ti, thr in enumerate(range(50, 100, 5)):
                ln = lines[ci] or [""]
                if lines[ci]:
                    rec, prec, ap = E.voc_eval(lines[ci], annos, name, thr / 100.0, year07)
                    assert abs(ap * 100 - d["ap_" + tag][ti, ci]) < 1e-9, (tag, name, thr)
                    if thr == 50:
                        assert np.array_equal(rec, d["rec_%d_%s" % (year07, name)])
                        assert np.array_equal(prec, d["prec_%d_%s" % (year07, name)])
                    cl = E.voc_eval_corloc(lines[ci], annos, name, thr / 100.0, year07)
                    assert abs(cl * 100 - d["corloc_" + tag][ti, ci]) < 1e-9, (tag, name, thr)
    # the evaluator object: same numbers through Instances
    ev = E.PascalVOCDetectionEvaluator(classes, annotations=annos, year=2007)
    by_img = {}
    for c, iid, score, box in dets:
        by_img.setdefault(iid, []).append((c, score, box))
    for iid, items in by_img.items():
        inst = Instances((300, 300))
        inst.pred_boxes = Boxes(torch.tensor([b for _, _, b in items], dtype=torch.float32))
        inst.scores = torch.tensor([s for _, s, _ in items], dtype=torch.float64)
        inst.pred_classes = torch.tensor([c for c, _, _ in items])
        ev.process([{"image_id": iid}], [{"instances": inst}])
    res = ev.evaluate()
    has = [ci for ci in range(len(classes)) if lines[ci]]
    ap50 = [d["ap_y07"][0, ci] if ci in has else 0.0 for ci in range(len(classes))]
    assert abs(res["bbox"]["AP50"] - np.mean(ap50)) < 1e-9
    cl50 = [d["corloc_y07"][0, ci] for ci in range(len(classes))]
    assert abs(res["bbox CorLoc"]["CL50"] - np.mean(cl50)) < 1e-9


def test_samplers_and_batch_loader_match_reference_golden(pkg):
    """SURVEY 8(e) partition: TrainingSampler (rank g takes elements g, g+W, ... of one shared shuffled stream),
    InferenceSampler shards, AspectRatioGroupedDataset batches and MapDataset's fallback draws, index for index against
    the reference's own classes (tests/golden/samplers.npz); then the batch loader built on them"""
    import itertools

    import numpy as np

    from drn_wsod_pytorch_amd import data as D

    d = G.load("samplers")
    for tag in "abc":
        size, seed, world, shuffle, n = (int(x) for x in d["train_%s_cfg" % tag])
        streams = []
        for r in range(world):
            got = [int(x) for x in itertools.islice(iter(D.TrainingSampler(size, bool(shuffle), seed, r, world)), n)]
            assert got == d["train_%s_r%d" % (tag, r)].tolist(), (tag, r)
            streams.append(got)
        # the ranks interleave back into ONE stream of whole permutations
        merged = [streams[i % world][i // world] for i in range(world * n)]
        assert sorted(merged[:size]) == list(range(size))
    for tag in "abcde":
        size, world = (int(x) for x in d["infer_%s_cfg" % tag])
        allidx = []
        for r in range(world):
            s = D.InferenceSampler(size, r, world)
            assert list(s) == d["infer_%s_r%d" % (tag, r)].tolist() and len(s) == len(d["infer_%s_r%d" % (tag, r)])
            allidx += list(s)
        assert allidx == list(range(size))  # every sample exactly once
    items = [{"width": int(w), "height": int(h), "id": i} for i, (w, h) in enumerate(d["group_wh"])]
    got = [[x["id"] for x in b] for b in D.AspectRatioGroupedDataset(items, 3)]
    assert got == d["group_batches"].tolist()
    md = D.MapDataset(list(range(10)), lambda x: None if x % 3 == 0 else x * 10)
    assert [md[i] for i in range(10)] + [md[i] for i in range(10)] == d["map_out"].tolist()
    # loader: 2 ranks, IMS_PER_BATCH 4 -> 2 images per rank and step; grouped and plain
    ds = [{"width": 200 + 10 * (i % 3), "height": 210, "id": i} for i in range(11)]
    for grouping in (False, True):
        per_rank = []
        for r in range(2):
            loader = D.build_batch_data_loader(D.MapDataset(ds, lambda x: x), D.TrainingSampler(len(ds), True, 5, r, 2), 4,
                                               aspect_ratio_grouping=grouping, num_workers=0, world_size=2)
            per_rank.append([[x["id"] for x in b] for b in itertools.islice(iter(loader), 4)])
            assert all(len(b) == 2 for b in per_rank[-1])
        if not grouping:
            ref = [[int(x) for x in itertools.islice(iter(D.TrainingSampler(len(ds), True, 5, r, 2)), 8)] for r in range(2)]
            assert [sum(b, []) for b in per_rank] == ref
    with pytest.raises(Exception):
        D.build_batch_data_loader(ds, D.TrainingSampler(len(ds), True, 5, 0, 3), 4, world_size=3)


def test_alias_layer_reference_import_names(tmp_path):
    """the reference's import lines for this path resolve to this package after aliases.install(); a model
    builds from the reference's UNMODIFIED yaml through those names; names outside the path fail loudly."""
    import subprocess
    import sys

    code = r'''
import sys
sys.path.insert(0, %r)
from __graft_entry__ import load_package
load_package()
import drn_wsod_pytorch_amd.aliases as A
names = A.install()
from detectron2.config import get_cfg
from detectron2.checkpoint import DetectionCheckpointer
from detectron2.layers import Conv2d, FrozenBatchNorm2d, ROIAlign, ShapeSpec, cat
from detectron2.structures import Boxes, Instances, ImageList
from detectron2.modeling import build_model, META_ARCH_REGISTRY, ROI_HEADS_REGISTRY, BACKBONE_REGISTRY
from detectron2.modeling.poolers import ROIPooler
from detectron2.utils.events import EventStorage, get_event_storage
from detectron2.evaluation import PascalVOCDetectionEvaluator
import detectron2.data.detection_utils as utils
from wsl.config import add_wsl_config
from wsl.modeling import GeneralizedRCNNWithTTAAVG
import detectron2, wsl
assert detectron2.layers.Conv2d is Conv2d and wsl.config.add_wsl_config is add_wsl_config
assert "OICRROIHeads" in ROI_HEADS_REGISTRY and "GeneralizedRCNNWSL" in META_ARCH_REGISTRY
assert "build_ws_resnet_backbone" in BACKBONE_REGISTRY
import os
yaml = "/root/reference/projects/WSL/configs/PascalVOC-Detection/oicr_WSR_18_DC5_1x.yaml"
if os.path.exists(yaml):
    cfg = get_cfg(); add_wsl_config(cfg); cfg.merge_from_file(yaml)
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.WEIGHTS", ""])
    model = build_model(cfg)
    assert type(model).__name__ == "GeneralizedRCNNWSL" and type(model.roi_heads).__name__ == "OICRROIHeads"
from drn_wsod_pytorch_amd._cabi import DrnError
import detectron2.evaluation as ev
try:
    ev.COCOEvaluator
    raise SystemExit("control-plane name resolved")
except DrnError as e:  # loud, with the reason
    assert "COCOEvaluator" in str(e) and isinstance(e, AttributeError)
# ... and Python's attribute / import protocols keep working (ADVICE r2)
assert not hasattr(ev, "COCOEvaluator") and getattr(ev, "COCOEvaluator", 7) == 7
try:
    from detectron2.evaluation import COCOEvaluator
    raise SystemExit("control-plane name resolved")
except ImportError as e:
    assert "COCOEvaluator" in str(e)
try:
    from detectron2.utils import comm
    raise SystemExit("control-plane module resolved")
except ImportError:
    pass
A.uninstall()
assert "detectron2" not in sys.modules
print("ALIAS_OK", len(names))
''' % (G.ROOT,)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "ALIAS_OK" in out.stdout, out.stdout + out.stderr
