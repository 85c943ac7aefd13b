class Transform:
    def _set_attributes(self, params=None):
        if params:
            for k, v in params.items():
                if k != "self" and not k.startswith("_"):
                    setattr(self, k, v)

    @classmethod
    def register_type(cls, data_type, func=None):
        if func is None:
            def deco(f):
                return f
            return deco


class TransformList(Transform):
    def __init__(self, transforms):
        self.transforms = transforms


class _T(Transform):
    def __init__(self, *a, **k):
        pass


HFlipTransform = VFlipTransform = NoOpTransform = BlendTransform = CropTransform = GridSampleTransform = ScaleTransform = _T
