"""Stand-in for absent torchvision: __version__ 0.6.0 forces the reference's own _ROIAlign path
(detectron2/layers/roi_align.py:9-15). ops.* are the ORACLE restatements (parity unpinned)."""
__version__ = "0.6.0"
from . import ops  # noqa
