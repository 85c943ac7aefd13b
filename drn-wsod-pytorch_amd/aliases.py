"""`detectron2.*` / `wsl.*` import names for this package ("our own namespace with identical class /
registry / key names + a thin alias layer").

    import drn_wsod_pytorch_amd.aliases as A; A.install()
    from detectron2.config import get_cfg                    # projects/WSL/tools/train_net.py:26
    from detectron2.checkpoint import DetectionCheckpointer   # :25
    from wsl.config import add_wsl_config                      # :45
    from wsl.modeling import GeneralizedRCNNWithTTAAVG         # :46
    from detectron2.layers import Conv2d, FrozenBatchNorm2d, ROIAlign, ShapeSpec, cat
    from detectron2.modeling import build_model, META_ARCH_REGISTRY, ROI_HEADS_REGISTRY

Only the hot path and the §8(f) rows that were built are behind these names.  Everything else of the reference
(DefaultTrainer / launch / hooks, MetadataCatalog, the COCO / LVIS / Cityscapes evaluators, model zoo, export, ...) is
the control plane this repo does not rebuild: asking an alias module for such a name raises DrnError naming it, it
never returns a stand-in.  install() refuses to shadow a real detectron2 that is already imported."""
import sys
import types

from ._cabi import DrnError

_PREFIX = "drn_wsod_pytorch_amd"


def _table():
    from . import checkpoint, config, data, engine, evaluation, events, layers, registry, structures
    from .modeling import backbone, rcnn, roi_heads, tta

    pick = lambda mod, *names: {n: getattr(mod, n) for n in names}
    t = {
        "detectron2": {"__version__": "0.2"},  # detectron2/__init__.py:10 of the reference fork
        "detectron2.layers": pick(layers, "Conv2d", "Linear", "FrozenBatchNorm2d", "get_norm", "ROIAlign", "ShapeSpec", "cat",
                                  "nonzero_tuple", "CNNBlockBase"),
        "detectron2.structures": pick(structures, "Boxes", "Instances", "ImageList"),
        "detectron2.config": pick(config, "CfgNode", "get_cfg", "configurable"),
        "detectron2.checkpoint": pick(checkpoint, "DetectionCheckpointer"),
        "detectron2.checkpoint.c2_model_loading": pick(checkpoint, "convert_c2_detectron_names", "align_and_update_state_dicts"),
        "detectron2.solver": pick(engine, "build_optimizer", "build_lr_scheduler", "WarmupMultiStepLR"),
        "detectron2.utils": {},
        "detectron2.utils.events": pick(events, "EventStorage", "get_event_storage"),
        "detectron2.utils.registry": pick(registry, "Registry"),
        "detectron2.evaluation": pick(evaluation, "PascalVOCDetectionEvaluator"),
        "detectron2.evaluation.pascal_voc_evaluation": pick(evaluation, "PascalVOCDetectionEvaluator", "voc_eval", "voc_ap",
                                                            "voc_eval_corloc", "parse_rec"),
        "detectron2.data": pick(data, "DatasetMapper", "build_detection_train_loader", "build_batch_data_loader", "MapDataset"),
        "detectron2.data.detection_utils": pick(data, "read_image", "transform_proposals", "transform_instance_annotations",
                                                "annotations_to_instances", "filter_empty_instances", "build_augmentation"),
        "detectron2.data.samplers": pick(data, "TrainingSampler", "InferenceSampler"),
        "detectron2.data.transforms": pick(data, "ResizeShortestEdge", "RandomFlip", "RandomCrop", "RandomBrightness",
                                           "RandomSaturation", "ResizeTransform", "HFlipTransform", "TransformList"),
        "detectron2.modeling": dict(pick(registry, "META_ARCH_REGISTRY", "BACKBONE_REGISTRY", "ROI_HEADS_REGISTRY",
                                         "ROI_BOX_HEAD_REGISTRY"),
                                    build_model=rcnn.build_model, build_backbone=backbone.build_backbone,
                                    build_roi_heads=roi_heads.build_roi_heads, build_box_head=roi_heads.build_box_head,
                                    Backbone=backbone.Backbone, detector_postprocess=rcnn.detector_postprocess),
        "detectron2.modeling.poolers": pick(roi_heads, "ROIPooler", "convert_boxes_to_pooler_format"),
        "detectron2.modeling.matcher": pick(roi_heads, "Matcher"),
        "detectron2.modeling.box_regression": pick(roi_heads, "Box2BoxTransform"),
        "wsl": {},
        "wsl.config": pick(config, "add_wsl_config"),
        "wsl.modeling": dict(pick(tta, "GeneralizedRCNNWithTTAAVG", "DatasetMapperTTAAVG"),
                             GeneralizedRCNNWSL=rcnn.GeneralizedRCNNWSL),
        "wsl.modeling.meta_arch": {"GeneralizedRCNNWSL": rcnn.GeneralizedRCNNWSL},
        "wsl.modeling.backbone": pick(backbone, "build_ws_resnet_backbone", "build_vgg_backbone", "ResNet", "VGG16", "BasicStem",
                                      "BasicBlock", "BottleneckBlock", "PlainBlock"),
        "wsl.modeling.roi_heads": pick(roi_heads, "OICRROIHeads", "WSDDNROIHeads", "PCLROIHeads", "CSCROIHeads", "DiscriminativeAdaptionNeck",
                                       "WSDDNOutputLayers", "OICROutputLayers"),
    }
    return t


class OffPathName(DrnError, AttributeError):
    """An off-path name was asked of an alias module.  It is a DrnError (loud, with the reason) AND an AttributeError,
    so Python's attribute and import protocols keep working: hasattr() / getattr(mod, x, default) return
    False / the default, and `from detectron2.utils import comm` becomes the interpreter's own ImportError, which try/except guards
    catch."""


class _AliasModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        raise OffPathName("%s.%s is not behind the DRN-WSOD hot path this package rebuilds (control plane / other model "
                       "families of the reference are out of scope; see INTEGRATION.md)" % (self.__name__, name))


def install(force=False):
    """register the alias modules in sys.modules; returns the list of module names installed"""
    real = sys.modules.get("detectron2")
    if real is not None and not isinstance(real, _AliasModule) and not force:
        raise DrnError("a real `detectron2` is already imported: the alias layer will not shadow it")
    table = _table()
    for name in sorted(table):
        mod = _AliasModule(name)
        mod.__dict__.update(table[name])
        mod.__dict__["__all__"] = sorted(k for k in table[name] if not k.startswith("__"))
        mod.__path__ = []  # a package: `import detectron2.layers` resolves through sys.modules
        sys.modules[name] = mod
    for name in table:  # parent.child attributes, like a real package tree
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[name])
    return sorted(table)


def uninstall():
    for name in [n for n, m in sys.modules.items() if isinstance(m, _AliasModule)]:
        del sys.modules[name]
