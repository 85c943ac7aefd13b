from .transform import *  # noqa: F401,F403
from .transform import BlendTransform, CropTransform, HFlipTransform, NoOpTransform, Transform, TransformList  # noqa: F401
