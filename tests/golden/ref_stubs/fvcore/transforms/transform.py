"""Stand-in for fvcore.transforms.transform (fvcore is an external, absent dependency of the reference: setup.py
requires fvcore>=0.1.1).  Only used in the build container to import the unmodified reference for golden generation.
The classes the TTA path touches (Transform.apply_box, TransformList, HFlipTransform, NoOpTransform) restate fvcore's
published semantics; everything else is a placeholder."""
import numpy as np


class Transform:
    def _set_attributes(self, params=None):
        if params:
            for k, v in params.items():
                if k != "self" and not k.startswith("_"):
                    setattr(self, k, v)

    def apply_box(self, box):
        # fvcore: the 4 corners go through apply_coords, the result is their axis-aligned bounding box
        idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
        coords = np.asarray(box).reshape(-1, 4)[:, idxs].reshape(-1, 2)
        coords = self.apply_coords(coords).reshape((-1, 4, 2))
        minxy = coords.min(axis=1)
        maxxy = coords.max(axis=1)
        return np.concatenate((minxy, maxxy), axis=1)

    def inverse(self):
        raise NotImplementedError

    @classmethod
    def register_type(cls, data_type, func=None):
        if func is None:
            def deco(f):
                return f
            return deco


class TransformList(Transform):
    def __init__(self, transforms):
        super().__init__()
        tfms_flatten = []
        for t in transforms:
            assert isinstance(t, Transform), t
            if isinstance(t, TransformList):
                tfms_flatten.extend(t.transforms)
            else:
                tfms_flatten.append(t)
        self.transforms = tfms_flatten

    def _apply(self, x, meth):
        for t in self.transforms:
            x = getattr(t, meth)(x)
        return x

    def __getattribute__(self, name):
        if name.startswith("apply_"):
            return lambda x: self._apply(x, name)
        return super().__getattribute__(name)

    def __add__(self, other):
        others = other.transforms if isinstance(other, TransformList) else [other]
        return TransformList(self.transforms + others)

    def __radd__(self, other):
        others = other.transforms if isinstance(other, TransformList) else [other]
        return TransformList(others + self.transforms)

    def __len__(self):
        return len(self.transforms)

    def __getitem__(self, idx):
        return self.transforms[idx]

    def inverse(self):
        return TransformList([x.inverse() for x in self.transforms[::-1]])


class HFlipTransform(Transform):
    def __init__(self, width):
        super().__init__()
        self._set_attributes(locals())

    def apply_image(self, img):
        if img.ndim <= 3:
            return np.flip(img, axis=1)
        return np.flip(img, axis=-2)

    def apply_coords(self, coords):
        coords[:, 0] = self.width - coords[:, 0]
        return coords

    def inverse(self):
        return self


class NoOpTransform(Transform):
    def apply_image(self, img):
        return img

    def apply_coords(self, coords):
        return coords

    def inverse(self):
        return self

    def __getattr__(self, name):
        if name.startswith("apply_"):
            return lambda x: x
        raise AttributeError(name)


class BlendTransform(Transform):
    """fvcore: img' = src_weight * src_image + dst_weight * img; uint8 inputs are computed in float32, clipped to
    [0, 255] and cast back; coordinates are untouched"""

    def __init__(self, src_image, src_weight, dst_weight):
        super().__init__()
        self._set_attributes(locals())

    def apply_image(self, img, interp=None):
        if img.dtype == np.uint8:
            img = img.astype(np.float32)
            img = self.src_weight * self.src_image + self.dst_weight * img
            return np.clip(img, 0, 255).astype(np.uint8)
        return self.src_weight * self.src_image + self.dst_weight * img

    def apply_coords(self, coords):
        return coords

    def inverse(self):
        return NoOpTransform()


class CropTransform(Transform):
    """fvcore: crop the window [y0, y0+h) x [x0, x0+w); coordinates shift by (-x0, -y0)"""

    def __init__(self, x0, y0, w, h, orig_w=None, orig_h=None):
        super().__init__()
        self._set_attributes(locals())

    def apply_image(self, img):
        if len(img.shape) <= 3:
            return img[self.y0: self.y0 + self.h, self.x0: self.x0 + self.w]
        return img[..., self.y0: self.y0 + self.h, self.x0: self.x0 + self.w, :]

    def apply_coords(self, coords):
        coords[:, 0] -= self.x0
        coords[:, 1] -= self.y0
        return coords


class _T(Transform):
    def __init__(self, *a, **k):
        pass


VFlipTransform = GridSampleTransform = ScaleTransform = _T
