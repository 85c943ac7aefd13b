"""Import harness for the REFERENCE (build container only).

Used solely by tests/golden/gen_golden.py to run the unmodified reference Python from
/root/reference and dump golden input/output vectors.  Nothing here travels into the product and
nothing in `-m gpu` tests / smoke / bench imports it (there is no /root/reference on the GPU box).
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
STUBS = os.path.join(HERE, "ref_stubs")
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))

AUTO_STUB = (
    "cv2", "pycocotools", "lvis", "caffe2", "onnx", "mock", "panopticapi", "cityscapesscripts", "tensorboard",
    "pydensecrf", "wsl._C", "detectron2.data", "detectron2.evaluation", "detectron2.export",
    "detectron2.checkpoint", "detectron2.utils.visualizer", "detectron2.utils.video_visualizer",
    "detectron2.modeling.test_time_augmentation",
    "wsl.modeling.test_time_augmentation_avg", "wsl.modeling.test_time_augmentation_union",
    "fvcore.common.checkpoint", "fvcore.common.timer", "fvcore.nn.precise_bn", "skimage", "shapely", "imagesize",
    "wsl.modeling.seg_heads", "wsl.data", "matplotlib", "pycocotools.mask",
)


# real reference modules that live under an auto-stubbed package (TTA golden: the reference's own resize / flip
# transforms and its GeneralizedRCNNWithTTAAVG are imported for real; their parents stay stubs)
REAL_UNDER_STUB = ("detectron2.data.transforms", "wsl.modeling.test_time_augmentation_avg", "detectron2.data.build",
                   "detectron2.data.detection_utils", "detectron2.data.dataset_mapper",
                   "detectron2.evaluation.pascal_voc_evaluation", "detectron2.data.samplers", "detectron2.data.common")
REAL_PATHS = {"detectron2.data": os.path.join(REF, "detectron2", "data"),
              "detectron2.evaluation": os.path.join(REF, "detectron2", "evaluation")}


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, n):
        if n.startswith("__") and n.endswith("__"):
            raise AttributeError(n)
        return _Dummy()

    def __mro_entries__(self, bases):
        return (object,)

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in REAL_UNDER_STUB:  # `from . import transforms`: hand out the real sub-module, not a dummy class
            import importlib

            mod = importlib.import_module(full)
            setattr(self, name, mod)
            return mod
        cls = type(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        for s in REAL_UNDER_STUB:
            if fullname == s or fullname.startswith(s + "."):
                return None
        for s in AUTO_STUB:
            if fullname == s or fullname.startswith(s + "."):
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = [REAL_PATHS[spec.name]] if spec.name in REAL_PATHS else []
        if spec.name == "cv2":
            m.__version__ = "2.0.0"
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    global _installed
    if _installed:
        return
    _installed = True
    for pth in (os.path.join(REF, "projects", "WSL"), REF, STUBS, REPO):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    import numpy as np
    import PIL.Image

    if not hasattr(PIL.Image, "LINEAR"):
        PIL.Image.LINEAR = PIL.Image.BILINEAR
    for n, t in (("int", int), ("bool", bool), ("float", float)):
        if not hasattr(np, n):
            setattr(np, n, t)
    sys.meta_path.insert(0, _Finder())
    # detectron2._C := the reference ROIAlign CPU source compiled in place (oracle/_ref)
    import importlib.util

    spec = importlib.util.spec_from_file_location("_obuild", os.path.join(REPO, "oracle", "build.py"))
    ob = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ob)
    so = ob.build_ref()
    import torch  # noqa: F401  (libtorch must be loaded first)

    spec = importlib.util.spec_from_file_location("d2_roialign_ref", so)
    refmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(refmod)
    sys.modules["detectron2._C"] = refmod
    import detectron2

    detectron2._C = refmod
    # wsl._C stays an auto-stub except for the PCL loss: the reference's pcl_loss_cpu.cpp compiled in place
    spec = importlib.util.spec_from_file_location("wsl_pcl_ref", ob.build_ref_pcl())
    pclmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pclmod)
    import wsl._C as wsl_c

    wsl_c.pcl_loss_forward = pclmod.pcl_loss_forward
    wsl_c.pcl_loss_backward = pclmod.pcl_loss_backward
    import wsl.modeling  # noqa: F401
    import wsl.modeling.meta_arch  # noqa: F401


def build_reference_model(yaml_rel, opts=()):
    """Build the unmodified reference model from an unmodified projects/WSL yaml."""
    install()
    from detectron2.config import get_cfg
    from detectron2.modeling import build_model
    from wsl.config import add_wsl_config

    cfg = get_cfg()
    add_wsl_config(cfg)
    cfg.merge_from_file(os.path.join(REF, "projects", "WSL", "configs", yaml_rel))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.WEIGHTS", ""] + list(opts))
    model = build_model(cfg)
    return cfg, model
