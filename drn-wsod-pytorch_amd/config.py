"""Config surface the reference's projects/WSL yaml files are written against: a yacs-compatible
CfgNode (`_BASE_` inheritance, KEY VALUE overrides), `get_cfg()` with the detectron2 defaults this
path reads (detectron2/config/defaults.py), `add_wsl_config` (projects/WSL/wsl/config/defaults.py:7-43)
and `@configurable` (detectron2/config/config.py:109-163).  yacs/fvcore are not installed here, so
this is a small re-implementation; unmodified reference yaml files load through it."""
import ast
import copy
import functools
import inspect
import os

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    def __init__(self, init_dict=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self._frozen:
            raise AttributeError("Attempted to set {} on an immutable CfgNode".format(name))
        self[name] = value

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        return out

    def clone(self):
        return copy.deepcopy(self)

    def _set_frozen(self, flag):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return self._frozen

    def dump(self, **kw):
        def conv(n):
            if isinstance(n, dict):
                return {k: conv(v) for k, v in n.items()}
            return list(n) if isinstance(n, tuple) else n

        return yaml.safe_dump(conv(self), **kw)

    @staticmethod
    def load_yaml_with_base(filename):
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}

        def merge(a, b):
            for k, v in a.items():
                if isinstance(v, dict) and isinstance(b.get(k), dict):
                    merge(v, b[k])
                else:
                    b[k] = v

        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if base.startswith("~"):
                base = os.path.expanduser(base)
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            base_cfg = CfgNode.load_yaml_with_base(base)
            merge(cfg, base_cfg)
            return base_cfg
        return cfg

    def merge_from_file(self, cfg_filename, allow_unsafe=False):
        self.merge_from_other_cfg(CfgNode(CfgNode.load_yaml_with_base(cfg_filename)))

    def merge_from_other_cfg(self, other):
        _merge(other, self)

    def merge_from_list(self, cfg_list):
        assert len(cfg_list) % 2 == 0, "Override list has odd length: {}".format(cfg_list)
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            d = self
            parts = full_key.split(".")
            for s in parts[:-1]:
                if s not in d:
                    raise KeyError("Non-existent config key: {}".format(full_key))
                d = d[s]
            d[parts[-1]] = _coerce(_decode(v), d.get(parts[-1]), full_key)


def _decode(v):
    if isinstance(v, dict) and not isinstance(v, CfgNode):
        return CfgNode(v)
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old, key=""):
    if old is None or type(new) is type(old):
        return new
    for a, b in ((tuple, list), (list, tuple)):
        if isinstance(old, a) and isinstance(new, b):
            return a(new)
    if isinstance(old, float) and isinstance(new, int):
        return float(new)
    if isinstance(old, (CfgNode, dict)) != isinstance(new, (CfgNode, dict)):
        raise ValueError("Type mismatch for config key {}: {} vs {}".format(key, type(old), type(new)))
    return new


def _merge(a, b, path=""):
    for k, v in a.items():
        v = _decode(v)
        if k in b and isinstance(v, CfgNode) and isinstance(b[k], CfgNode):
            _merge(v, b[k], path + k + ".")
        elif k in b:
            b[k] = _coerce(v, b[k], path + k)
        else:
            b[k] = v  # keys of off-path subsystems (RPN, mask heads, ...) are carried, not interpreted


CN = CfgNode


def get_cfg():
    """Defaults of the keys this path reads (detectron2/config/defaults.py:245-315,461-493,500-546)."""
    _C = CN()
    _C.VERSION = 2
    _C.MODEL = CN()
    _C.MODEL.LOAD_PROPOSALS = False
    _C.MODEL.MASK_ON = False
    _C.MODEL.KEYPOINT_ON = False
    _C.MODEL.DEVICE = "cuda"
    _C.MODEL.META_ARCHITECTURE = "GeneralizedRCNN"
    _C.MODEL.WEIGHTS = ""
    _C.MODEL.PIXEL_MEAN = [103.530, 116.280, 123.675]
    _C.MODEL.PIXEL_STD = [1.0, 1.0, 1.0]
    _C.INPUT = CN({"MIN_SIZE_TRAIN": (800,), "MIN_SIZE_TRAIN_SAMPLING": "choice", "MAX_SIZE_TRAIN": 1333,
                   "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333, "FORMAT": "BGR", "MASK_FORMAT": "polygon"})
    _C.INPUT.CROP = CN({"ENABLED": False, "TYPE": "relative_range", "SIZE": [0.9, 0.9]})
    _C.DATASETS = CN({"TRAIN": (), "PROPOSAL_FILES_TRAIN": (), "PRECOMPUTED_PROPOSAL_TOPK_TRAIN": 2000, "TEST": (),
                      "PROPOSAL_FILES_TEST": (), "PRECOMPUTED_PROPOSAL_TOPK_TEST": 1000})
    _C.DATALOADER = CN({"NUM_WORKERS": 4, "ASPECT_RATIO_GROUPING": True, "SAMPLER_TRAIN": "TrainingSampler",
                        "REPEAT_THRESHOLD": 0.0, "FILTER_EMPTY_ANNOTATIONS": True})
    _C.MODEL.BACKBONE = CN({"NAME": "build_resnet_backbone", "FREEZE_AT": 2})
    _C.MODEL.PROPOSAL_GENERATOR = CN({"NAME": "RPN", "MIN_SIZE": 0})
    _C.MODEL.RPN = CN({"IN_FEATURES": ["res4"], "PRE_NMS_TOPK_TEST": 6000, "POST_NMS_TOPK_TEST": 1000})
    _C.MODEL.ROI_HEADS = CN({"NAME": "Res5ROIHeads", "NUM_CLASSES": 80, "IN_FEATURES": ["res4"], "IOU_THRESHOLDS": [0.5],
                             "IOU_LABELS": [0, 1], "BATCH_SIZE_PER_IMAGE": 512, "POSITIVE_FRACTION": 0.25,
                             "SCORE_THRESH_TEST": 0.05, "NMS_THRESH_TEST": 0.5, "PROPOSAL_APPEND_GT": True})
    _C.MODEL.ROI_BOX_HEAD = CN({"NAME": "", "BBOX_REG_LOSS_TYPE": "smooth_l1", "BBOX_REG_LOSS_WEIGHT": 1.0,
                                "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0), "SMOOTH_L1_BETA": 0.0,
                                "POOLER_RESOLUTION": 14, "POOLER_SAMPLING_RATIO": 0, "POOLER_TYPE": "ROIAlignV2",
                                "NUM_FC": 0, "FC_DIM": 1024, "NUM_CONV": 0, "CONV_DIM": 256, "NORM": "",
                                "CLS_AGNOSTIC_BBOX_REG": False, "TRAIN_ON_PRED_BOXES": False})
    _C.MODEL.ROI_MASK_HEAD = CN({"NAME": "MaskRCNNConvUpsampleHead", "POOLER_RESOLUTION": 14, "NUM_CONV": 0})
    _C.MODEL.SEM_SEG_HEAD = CN({"NAME": "SemSegFPNHead"})
    _C.MODEL.RESNETS = CN({"DEPTH": 50, "OUT_FEATURES": ["res4"], "NUM_GROUPS": 1, "NORM": "FrozenBN",
                           "WIDTH_PER_GROUP": 64, "STRIDE_IN_1X1": True, "RES5_DILATION": 1, "RES2_OUT_CHANNELS": 256,
                           "STEM_OUT_CHANNELS": 64, "DEFORM_ON_PER_STAGE": [False, False, False, False],
                           "DEFORM_MODULATED": False, "DEFORM_NUM_GROUPS": 1})
    _C.SOLVER = CN({"LR_SCHEDULER_NAME": "WarmupMultiStepLR", "MAX_ITER": 40000, "BASE_LR": 0.001, "MOMENTUM": 0.9,
                    "NESTEROV": False, "WEIGHT_DECAY": 0.0001, "WEIGHT_DECAY_NORM": 0.0, "GAMMA": 0.1,
                    "STEPS": (30000,), "WARMUP_FACTOR": 1.0 / 1000, "WARMUP_ITERS": 1000, "WARMUP_METHOD": "linear",
                    "CHECKPOINT_PERIOD": 5000, "IMS_PER_BATCH": 16, "REFERENCE_WORLD_SIZE": 0, "BIAS_LR_FACTOR": 1.0,
                    "WEIGHT_DECAY_BIAS": 0.0001})
    _C.SOLVER.CLIP_GRADIENTS = CN({"ENABLED": False, "CLIP_TYPE": "value", "CLIP_VALUE": 1.0, "NORM_TYPE": 2.0})
    _C.TEST = CN({"EXPECTED_RESULTS": [], "EVAL_PERIOD": 0, "DETECTIONS_PER_IMAGE": 100})
    _C.TEST.AUG = CN({"ENABLED": False, "MIN_SIZES": (400, 500, 600, 700, 800, 900, 1000, 1100, 1200), "MAX_SIZE": 4000,
                      "FLIP": True})
    _C.OUTPUT_DIR = "./output"
    _C.SEED = -1
    _C.CUDNN_BENCHMARK = False
    _C.VIS_PERIOD = 0
    return _C


def add_wsl_config(cfg):
    """projects/WSL/wsl/config/defaults.py:7-43."""
    _C = cfg
    _C.MODEL.VGG = CN({"DEPTH": 16, "OUT_FEATURES": ["plain5"], "CONV5_DILATION": 1})
    _C.WSL = CN({"VIS_TEST": False, "ITER_SIZE": 1, "MEAN_LOSS": True, "USE_OBN": True, "CSC_MAX_ITER": 35000,
                 "REFINE_NUM": 3, "REFINE_REG": [False, False, False]})
    _C.MODEL.ROI_BOX_HEAD.DAN_DIM = [4096, 4096]
    _C.DATASETS.VAL = ()
    _C.DATASETS.PROPOSAL_FILES_VAL = ()
    _C.MODEL.SEM_SEG_HEAD.ASSP_CONVS_DIM = [1024, 1024]
    _C.MODEL.SEM_SEG_HEAD.MASK_SOFTMAX = False
    _C.MODEL.SEM_SEG_HEAD.CONSTRAINT = False
    _C.TEST.EVAL_TRAIN = True


def configurable(init_func):
    """detectron2/config/config.py:109-163: `Cls(cfg, *a, **k)` -> `Cls(**Cls.from_config(cfg, *a, **k))`;
    explicit keyword construction passes straight through."""
    assert init_func.__name__ == "__init__", "@configurable should only be used for __init__!"

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        try:
            from_config_func = type(self).from_config
        except AttributeError:
            raise AttributeError("Class with @configurable must have a 'from_config' classmethod.")
        if not inspect.ismethod(from_config_func):
            raise TypeError("Class with @configurable must have a 'from_config' classmethod.")
        if _called_with_cfg(*args, **kwargs):
            init_func(self, **_get_args_from_config(from_config_func, *args, **kwargs))
        else:
            init_func(self, *args, **kwargs)

    return wrapped


def _called_with_cfg(*args, **kwargs):
    if len(args) and isinstance(args[0], CfgNode):
        return True
    return isinstance(kwargs.pop("cfg", None), CfgNode)


def _get_args_from_config(from_config_func, *args, **kwargs):
    sig = inspect.signature(from_config_func)
    if list(sig.parameters.keys())[0] != "cfg":
        raise TypeError("{}.from_config must take 'cfg' as the first argument!".format(from_config_func.__self__))
    support_var = any(p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD) for p in sig.parameters.values())
    if support_var:
        return from_config_func(*args, **kwargs)
    names = set(sig.parameters.keys())
    extra = {k: kwargs.pop(k) for k in list(kwargs) if k not in names}
    ret = from_config_func(*args, **kwargs)
    ret.update(extra)
    return ret
