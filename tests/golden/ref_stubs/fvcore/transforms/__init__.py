from .transform import *  # noqa: F401,F403
from .transform import HFlipTransform, NoOpTransform, Transform, TransformList  # noqa: F401
