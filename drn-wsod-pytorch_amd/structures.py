"""Boxes / Instances / ImageList holders with the reference's interface
(detectron2/structures/{boxes,instances,image_list}.py).  Plain Python field bags over device
tensors: plumbing, no arithmetic on the hot path happens here (the IoU / matching / clipping the
reference does with these classes is done inside the HIP head and inference-tail kernels)."""
import itertools
from typing import Any, Dict, List, Tuple, Union

import torch


class Boxes:
    """Nx4 XYXY absolute boxes (detectron2/structures/boxes.py:130-300)."""

    def __init__(self, tensor: torch.Tensor):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device("cpu")
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, 4)).to(dtype=torch.float32, device=device)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs):
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size: Tuple[int, int]) -> None:
        h, w = box_size
        self.tensor[:, 0].clamp_(min=0, max=w)
        self.tensor[:, 1].clamp_(min=0, max=h)
        self.tensor[:, 2].clamp_(min=0, max=w)
        self.tensor[:, 3].clamp_(min=0, max=h)

    def nonempty(self, threshold: float = 0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, "Indexing on Boxes with {} failed to return a matrix!".format(item)
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    def scale(self, scale_x: float, scale_y: float) -> None:
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    @classmethod
    def cat(cls, boxes_list):
        assert isinstance(boxes_list, (list, tuple))
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self):
        return self.tensor.device

    def __iter__(self):
        yield from self.tensor


class Instances:
    """The per-image record that travels between the loader, the model and the evaluator
    (detectron2/structures/instances.py defines the contract): an image size plus named columns that all have one
    entry per instance - tensors, `Boxes`, lists.  Columns are reached as attributes (`inst.gt_classes = t`,
    `inst.scores`), rows by indexing, and records of one image are merged with `Instances.cat`.

    Stored as one ordered {name: column} table; every operation below is a map over that table."""

    def __init__(self, image_size: Tuple[int, int], **columns: Any):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for name, col in columns.items():
            self.set(name, col)

    # -- the table -------------------------------------------------------------------------------------------------
    @property
    def image_size(self):
        return self._image_size

    def set(self, name, value):
        n = len(value)
        if self._fields and n != len(self):
            raise AssertionError("Adding a field of length {} to a Instances of length {}".format(n, len(self)))
        self._fields[name] = value

    def get(self, name):
        return self._fields[name]

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        self._fields.pop(name)

    def get_fields(self):
        return self._fields

    def __len__(self):
        if not self._fields:
            raise NotImplementedError("Empty Instances does not support __len__!")
        return len(next(iter(self._fields.values())))

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    # -- attribute access = column access -----------------------------------------------------------------------------
    def __setattr__(self, name, value):
        if name[:1] == "_":
            object.__setattr__(self, name, value)
        else:
            self.set(name, value)

    def __getattr__(self, name):  # only reached when normal lookup fails, i.e. for column names
        cols = self.__dict__.get("_fields")
        if cols is None or name not in cols:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return cols[name]

    # -- maps over the table ------------------------------------------------------------------------------------------
    def _mapped(self, fn):
        out = Instances(self._image_size)
        for name, col in self._fields.items():
            out.set(name, fn(col))
        return out

    def to(self, *args, **kwargs):
        return self._mapped(lambda col: col.to(*args, **kwargs) if hasattr(col, "to") else col)

    def __getitem__(self, rows):
        if isinstance(rows, int) and not isinstance(rows, bool):
            n = len(self)
            if not -n <= rows < n:
                raise IndexError("Instances index out of range!")
            rows = slice(rows, None, n)  # one row, kept as a length-1 record
        return self._mapped(lambda col: col[rows])

    @staticmethod
    def _join(cols):
        head = cols[0]
        if isinstance(head, torch.Tensor):
            return torch.cat(cols, dim=0)
        if isinstance(head, list):
            return list(itertools.chain.from_iterable(cols))
        joiner = getattr(type(head), "cat", None)
        if joiner is None:
            raise ValueError("Unsupported type {} for concatenation".format(type(head)))
        return joiner(cols)

    @staticmethod
    def cat(instance_lists):
        if not instance_lists or not all(isinstance(r, Instances) for r in instance_lists):
            raise AssertionError("Instances.cat takes a non-empty list of Instances")
        first = instance_lists[0]
        if len(instance_lists) == 1:
            return first
        out = Instances(first.image_size)
        for name in first._fields:
            out.set(name, Instances._join([r.get(name) for r in instance_lists]))
        return out

    def __repr__(self):
        h, w = self._image_size
        cols = ", ".join("{}: {}".format(k, v) for k, v in self._fields.items())
        return "Instances(num_instances={}, image_height={}, image_width={}, fields=[{}])".format(len(self), h, w, cols)


class ImageList:
    """Batched, zero-padded images + true sizes (detectron2/structures/image_list.py:11-119).  `.tensor`
    is the [N,C,H,W] view the reference exposes; `.nhwc` is the padded NHWC buffer the HIP backbone reads."""

    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]], nhwc=None):
        self.tensor = tensor
        self.image_sizes = image_sizes
        self.nhwc = nhwc

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    def to(self, *args, **kwargs):
        return ImageList(self.tensor.to(*args, **kwargs), self.image_sizes, self.nhwc)

    @property
    def device(self):
        return self.tensor.device
