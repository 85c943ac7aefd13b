"""The data formats either side of the model: precomputed-proposal files and the dataset mapper.

Mirrors, under the reference's names and argument meanings:
  load_proposals_into_dataset      detectron2/data/build.py:102-153 (pickle with ids/boxes/objectness_logits, Detectron1
                                   key names indexes/scores accepted, proposals sorted by descending score)
  transform_proposals              detectron2/data/detection_utils.py:209-254 (+ Boxes.unique_boxes, boxes.py:214-226)
  transform_instance_annotations,  detection_utils.py:257-318, :363-426, :486-512
  annotations_to_instances, filter_empty_instances
  read_image                       detection_utils.py:59-89, :162-181 (PIL -> HWC uint8, BGR by channel flip)
  build_augmentation               detection_utils.py:568-592 (this fork adds RandomBrightness / RandomSaturation)
  ResizeShortestEdge, RandomFlip,  detectron2/data/transforms/augmentation_impl.py:73-102, :125-175, :232-281, :403-455
  RandomCrop, RandomBrightness, RandomSaturation
  ResizeTransform                  detectron2/data/transforms/transform.py:83-134
  DatasetMapper                    detectron2/data/dataset_mapper.py:36-185 (INPUT.CROP on, as every WSL yaml sets)
fvcore's HFlip / Crop / Blend transforms and Transform.apply_box are external to the reference; their published
semantics are restated.  Augmentations draw from np.random in the reference's order, so a seeded run reproduces the
reference's crops / scales / flips / colour factors draw for draw (tests/golden/data_mapper.npz).

  TrainingSampler, InferenceSampler  detectron2/data/samplers/distributed_sampler.py:12-56, :172-200 (the data-parallel
                                   partition of SURVEY 8(e): rank g takes elements g, g+W, ... of ONE shared shuffled stream)
  AspectRatioGroupedDataset,       detectron2/data/common.py:115-149, :17-73; detectron2/data/build.py:249-296, :299-354
  MapDataset, build_batch_data_loader, build_detection_train_loader
This is host-side preparation by design (as in the reference: loader workers); everything it emits is what
GeneralizedRCNNWSL.forward consumes."""
import copy
import pickle

import numpy as np
import torch
import torch.nn.functional as F

from ._cabi import DrnError
from .structures import Boxes, Instances

__all__ = ["BoxMode", "DatasetMapper", "build_augmentation", "load_proposals_into_dataset", "read_image",
           "transform_proposals", "transform_instance_annotations", "annotations_to_instances", "filter_empty_instances",
           "ResizeShortestEdge", "RandomFlip", "RandomCrop", "RandomBrightness", "RandomSaturation", "TransformList",
           "TrainingSampler", "InferenceSampler", "AspectRatioGroupedDataset", "MapDataset", "build_batch_data_loader",
           "build_detection_train_loader"]


class BoxMode:
    XYXY_ABS, XYWH_ABS = 0, 1

    @staticmethod
    def convert(box, from_mode, to_mode):
        from_mode, to_mode = int(from_mode), int(to_mode)
        if from_mode == to_mode:
            return box
        if (from_mode, to_mode) != (BoxMode.XYWH_ABS, BoxMode.XYXY_ABS):
            raise DrnError("box mode conversion %d -> %d is off the WSL path" % (from_mode, to_mode))
        single = isinstance(box, (list, tuple))
        arr = np.array(box, dtype=np.float64 if single else None).reshape(-1, 4).copy()
        arr[:, 2] += arr[:, 0]
        arr[:, 3] += arr[:, 1]
        return type(box)(arr.flatten().tolist()) if single else arr


# ---- transforms (deterministic; produced by the augmentations below) ------------------------------------------
class _Transform:
    def apply_coords(self, coords):
        return coords

    def apply_image(self, img):
        return img

    def apply_box(self, box):
        idx = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
        c = np.asarray(box).reshape(-1, 4)[:, idx].reshape(-1, 2)
        c = self.apply_coords(c).reshape((-1, 4, 2))
        return np.concatenate((c.min(axis=1), c.max(axis=1)), axis=1)


class NoOpTransform(_Transform):
    pass


class ResizeTransform(_Transform):
    def __init__(self, h, w, new_h, new_w):
        self.h, self.w, self.new_h, self.new_w = h, w, new_h, new_w

    def apply_image(self, img):
        assert img.shape[:2] == (self.h, self.w)
        if img.dtype == np.uint8:
            from PIL import Image

            return np.asarray(Image.fromarray(img).resize((self.new_w, self.new_h), Image.BILINEAR))
        t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1)[None]
        t = F.interpolate(t, (self.new_h, self.new_w), mode="bilinear", align_corners=False)
        return t[0].permute(1, 2, 0).numpy()

    def apply_coords(self, coords):
        coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
        coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
        return coords


class HFlipTransform(_Transform):
    def __init__(self, width):
        self.width = width

    def apply_image(self, img):
        return np.flip(img, axis=1)

    def apply_coords(self, coords):
        coords[:, 0] = self.width - coords[:, 0]
        return coords


class CropTransform(_Transform):
    def __init__(self, x0, y0, w, h):
        self.x0, self.y0, self.w, self.h = x0, y0, w, h

    def apply_image(self, img):
        return img[self.y0: self.y0 + self.h, self.x0: self.x0 + self.w]

    def apply_coords(self, coords):
        coords[:, 0] -= self.x0
        coords[:, 1] -= self.y0
        return coords


class BlendTransform(_Transform):
    def __init__(self, src_image, src_weight, dst_weight):
        self.src_image, self.src_weight, self.dst_weight = src_image, src_weight, dst_weight

    def apply_image(self, img):
        if img.dtype == np.uint8:
            img = img.astype(np.float32)
            img = self.src_weight * self.src_image + self.dst_weight * img
            return np.clip(img, 0, 255).astype(np.uint8)
        return self.src_weight * self.src_image + self.dst_weight * img


class TransformList(_Transform):
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def apply_image(self, img):
        for t in self.transforms:
            img = t.apply_image(img)
        return img

    def apply_coords(self, coords):
        for t in self.transforms:
            coords = t.apply_coords(coords)
        return coords


# ---- augmentations (draw from np.random exactly where the reference does) ----------------------------------------
class ResizeShortestEdge:
    def __init__(self, short_edge_length, max_size=2 ** 31 - 1, sample_style="range"):
        assert sample_style in ("range", "choice"), sample_style
        if isinstance(short_edge_length, int):
            short_edge_length = (short_edge_length, short_edge_length)
        self.short_edge_length, self.max_size, self.is_range = short_edge_length, max_size, sample_style == "range"

    def get_transform(self, img):
        from .modeling.tta import resize_shortest_edge_shape

        h, w = img.shape[:2]
        if self.is_range:
            size = np.random.randint(self.short_edge_length[0], self.short_edge_length[1] + 1)
        else:
            size = np.random.choice(self.short_edge_length)
        if size == 0:
            return NoOpTransform()
        nh, nw = resize_shortest_edge_shape(h, w, size, self.max_size)
        return ResizeTransform(h, w, nh, nw)


class RandomFlip:
    def __init__(self, prob=0.5):
        self.prob = prob

    def get_transform(self, img):
        w = img.shape[1]
        return HFlipTransform(w) if np.random.uniform(0, 1.0) < self.prob else NoOpTransform()


class RandomCrop:
    def __init__(self, crop_type, crop_size):
        assert crop_type in ("relative_range", "relative", "absolute", "absolute_range")
        self.crop_type, self.crop_size = crop_type, crop_size

    def get_crop_size(self, image_size):
        h, w = image_size
        if self.crop_type == "relative":
            ch, cw = self.crop_size
            return int(h * ch + 0.5), int(w * cw + 0.5)
        if self.crop_type == "relative_range":
            cs = np.asarray(self.crop_size, dtype=np.float32)
            ch, cw = cs + np.random.rand(2) * (1 - cs)
            return int(h * ch + 0.5), int(w * cw + 0.5)
        if self.crop_type == "absolute":
            return min(self.crop_size[0], h), min(self.crop_size[1], w)
        ch = np.random.randint(min(h, self.crop_size[0]), min(h, self.crop_size[1]) + 1)
        cw = np.random.randint(min(w, self.crop_size[0]), min(w, self.crop_size[1]) + 1)
        return ch, cw

    def get_transform(self, img):
        h, w = img.shape[:2]
        croph, cropw = self.get_crop_size((h, w))
        h0 = np.random.randint(h - croph + 1)
        w0 = np.random.randint(w - cropw + 1)
        return CropTransform(w0, h0, cropw, croph)


class RandomBrightness:
    def __init__(self, intensity_min, intensity_max):
        self.lo, self.hi = intensity_min, intensity_max

    def get_transform(self, img):
        w = np.random.uniform(self.lo, self.hi)
        return BlendTransform(src_image=0, src_weight=1 - w, dst_weight=w)


class RandomSaturation:
    def __init__(self, intensity_min, intensity_max):
        self.lo, self.hi = intensity_min, intensity_max

    def get_transform(self, img):
        assert img.shape[-1] == 3, "RandomSaturation only works on RGB images"
        w = np.random.uniform(self.lo, self.hi)
        grayscale = img.dot([0.299, 0.587, 0.114])[:, :, np.newaxis]
        return BlendTransform(src_image=grayscale, src_weight=1 - w, dst_weight=w)


def build_augmentation(cfg, is_train):
    if is_train:
        augs = [ResizeShortestEdge(cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN, cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING),
                RandomFlip(), RandomBrightness(1.0 / 1.5, 1.5), RandomSaturation(1.0 / 1.5, 1.5)]
    else:
        augs = [ResizeShortestEdge(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, "choice")]
    return augs


def apply_augmentations(augs, image):
    tfms = []
    for a in augs:
        t = a.get_transform(image)
        image = t.apply_image(image)
        tfms.append(t)
    return image, TransformList(tfms)


# ---- files -----------------------------------------------------------------------------------------------------
def read_image(file_name, format=None):
    from PIL import Image

    with open(file_name, "rb") as f:
        image = Image.open(f)
        if format is not None:
            image = image.convert("RGB" if format == "BGR" else format)
        image = np.asarray(image)
    if format == "L":
        image = np.expand_dims(image, -1)
    elif format == "BGR":
        image = image[:, :, ::-1]
    return image


def load_proposals_into_dataset(dataset_dicts, proposal_file):
    with open(proposal_file, "rb") as f:
        proposals = pickle.load(f, encoding="latin1")
    for old, new in (("indexes", "ids"), ("scores", "objectness_logits")):
        if old in proposals:
            proposals[new] = proposals.pop(old)
    wanted = {str(r["image_id"]) for r in dataset_dicts}
    index_of = {str(i): k for k, i in enumerate(proposals["ids"]) if str(i) in wanted}
    mode = int(proposals["bbox_mode"]) if "bbox_mode" in proposals else BoxMode.XYXY_ABS
    for r in dataset_dicts:
        k = index_of[str(r["image_id"])]
        boxes, logits = proposals["boxes"][k], proposals["objectness_logits"][k]
        order = logits.argsort()[::-1]
        r["proposal_boxes"], r["proposal_objectness_logits"], r["proposal_bbox_mode"] = boxes[order], logits[order], mode
    return dataset_dicts


# ---- per-record transforms -----------------------------------------------------------------------------------------
def unique_boxes(boxes, scale=1.0):
    """Boxes.unique_boxes, boxes.py:214-226: first index of every distinct rounded box, ascending"""
    v = np.array([1, 1e3, 1e6, 1e9])
    hashes = np.round(boxes.tensor.numpy() * scale).dot(v).astype(int)
    _, index = np.unique(hashes, return_index=True)
    return np.sort(index)


def transform_proposals(dataset_dict, image_shape, transforms, *, proposal_topk, min_box_size=0):
    if "proposal_boxes" not in dataset_dict:
        return
    boxes = transforms.apply_box(BoxMode.convert(dataset_dict.pop("proposal_boxes"), dataset_dict.pop("proposal_bbox_mode"),
                                                 BoxMode.XYXY_ABS))
    boxes = Boxes(torch.as_tensor(boxes, dtype=torch.float32))
    logits = torch.as_tensor(dataset_dict.pop("proposal_objectness_logits").astype("float32"))
    boxes.clip(image_shape)
    keep = torch.as_tensor(unique_boxes(boxes))
    boxes, logits = boxes[keep], logits[keep]
    keep = boxes.nonempty(threshold=min_box_size)
    boxes, logits = boxes[keep], logits[keep]
    p = Instances(image_shape)
    p.proposal_boxes = boxes[:proposal_topk]
    p.objectness_logits = logits[:proposal_topk]
    dataset_dict["proposals"] = p


def transform_instance_annotations(annotation, transforms, image_size):
    bbox = BoxMode.convert(annotation["bbox"], annotation["bbox_mode"], BoxMode.XYXY_ABS)
    bbox = transforms.apply_box(np.array([bbox]))[0].clip(min=0)
    annotation["bbox"] = np.minimum(bbox, list(image_size + image_size)[::-1])
    annotation["bbox_mode"] = BoxMode.XYXY_ABS
    return annotation


def annotations_to_instances(annos, image_size):
    target = Instances(image_size)
    boxes = [BoxMode.convert(o["bbox"], o["bbox_mode"], BoxMode.XYXY_ABS) for o in annos]
    target.gt_boxes = Boxes(torch.as_tensor(np.array(boxes, dtype=np.float64).reshape(-1, 4), dtype=torch.float32))
    target.gt_classes = torch.tensor([o["category_id"] for o in annos], dtype=torch.int64)
    return target


def filter_empty_instances(instances):
    keep = instances.gt_boxes.nonempty(threshold=1e-5)
    return instances[keep]


class DatasetMapper:
    """dataset dict (file_name or an HWC uint8 "image_array", height, width, annotations, proposal_*) -> model input"""

    def __init__(self, cfg, is_train=True):
        self.is_train = is_train
        self.augmentations = build_augmentation(cfg, is_train)
        if cfg.INPUT.CROP.ENABLED and is_train:
            self.augmentations.insert(0, RandomCrop(cfg.INPUT.CROP.TYPE, cfg.INPUT.CROP.SIZE))
        self.image_format = cfg.INPUT.FORMAT
        self.proposal_topk = None
        if cfg.MODEL.LOAD_PROPOSALS:
            self.proposal_topk = (cfg.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TRAIN if is_train
                                  else cfg.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST)

    def __call__(self, dataset_dict):
        d = copy.deepcopy(dataset_dict)
        image = d.pop("image_array") if "image_array" in d else read_image(d["file_name"], format=self.image_format)
        if "width" in d or "height" in d:
            if (image.shape[1], image.shape[0]) != (d["width"], d["height"]):
                raise DrnError("Mismatched (W,H): got %s, expect %s" % ((image.shape[1], image.shape[0]),
                                                                        (d["width"], d["height"])))
        d.setdefault("width", image.shape[1])
        d.setdefault("height", image.shape[0])
        image, transforms = apply_augmentations(self.augmentations, image)
        shape = image.shape[:2]
        d["image"] = torch.as_tensor(np.ascontiguousarray(image.transpose(2, 0, 1)))
        if self.proposal_topk is not None:
            transform_proposals(d, shape, transforms, proposal_topk=self.proposal_topk)
        if not self.is_train:
            d.pop("annotations", None)
            return d
        if "annotations" in d:
            annos = [transform_instance_annotations(o, transforms, shape) for o in d.pop("annotations")
                     if o.get("iscrowd", 0) == 0]
            d["instances"] = filter_empty_instances(annotations_to_instances(annos, shape))
        return d


# ---------------------------------------------------------------------------------------------------------------------
# samplers and the batch loader (how the images are partitioned over the ranks)
# ---------------------------------------------------------------------------------------------------------------------
def _rank_world(rank, world_size):
    import torch.distributed as dist

    if rank is None or world_size is None:
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1
    return int(rank), int(world_size)


class TrainingSampler:
    """distributed_sampler.py:12-56: an infinite stream shuffle(range(size)) + shuffle(range(size)) + ... drawn from ONE
    generator seeded identically on every rank; rank g yields elements g, g+W, g+2W, ... of it.  `seed=None` asks the
    ranks to agree on one (comm.shared_random_seed: rank 0's draw broadcast to all)."""

    def __init__(self, size, shuffle=True, seed=None, rank=None, world_size=None):
        assert size > 0
        self._size, self._shuffle = size, shuffle
        self._rank, self._world_size = _rank_world(rank, world_size)
        if seed is None:
            seed = _shared_random_seed()
        self._seed = int(seed)

    def __iter__(self):
        import itertools

        yield from itertools.islice(self._infinite_indices(), self._rank, None, self._world_size)

    def _infinite_indices(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            if self._shuffle:
                yield from torch.randperm(self._size, generator=g)
            else:
                yield from torch.arange(self._size)


def _shared_random_seed():
    """detectron2/utils/comm.py shared_random_seed: every rank draws, rank 0's value wins"""
    import torch.distributed as dist

    seed = int(np.random.randint(2 ** 31))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        box = [seed]
        dist.broadcast_object_list(box, src=0)
        seed = int(box[0])
    return seed


class InferenceSampler:
    """distributed_sampler.py:172-200: contiguous shards of ceil(size / W) indices; the last ranks may get fewer / none"""

    def __init__(self, size, rank=None, world_size=None):
        assert size > 0
        self._size = size
        self._rank, self._world_size = _rank_world(rank, world_size)
        shard = (size - 1) // self._world_size + 1
        self._local_indices = range(shard * self._rank, min(shard * (self._rank + 1), size))

    def __iter__(self):
        yield from self._local_indices

    def __len__(self):
        return len(self._local_indices)


class MapDataset:
    """detectron2/data/common.py:17-73: dataset[idx] -> map_func(dataset[idx]); when the mapper returns None (an image
    it cannot use) another index is drawn from random.Random(42) until one maps"""

    def __init__(self, dataset, map_func):
        import random

        self._dataset, self._map_func = dataset, map_func
        self._rng = random.Random(42)
        self._fallback_candidates = set(range(len(dataset)))

    def __len__(self):
        return len(self._dataset)

    def __getitem__(self, idx):
        retry, cur = 0, int(idx)
        while True:
            data = self._map_func(self._dataset[cur])
            if data is not None:
                self._fallback_candidates.add(cur)
                return data
            retry += 1
            self._fallback_candidates.discard(cur)
            cur = self._rng.sample(sorted(self._fallback_candidates), k=1)[0]
            if retry >= 3 and retry % 3 == 0:
                import logging

                logging.getLogger(__name__).warning("Failed to apply `_map_func` for idx: %d, retry count: %d", idx, retry)


class AspectRatioGroupedDataset:
    """detectron2/data/common.py:115-149: two buckets (w > h, else); a bucket is emitted when it reaches batch_size"""

    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size = dataset, batch_size
        self._buckets = [[] for _ in range(2)]

    def __iter__(self):
        for d in self.dataset:
            bucket = self._buckets[0 if d["width"] > d["height"] else 1]
            bucket.append(d)
            if len(bucket) == self.batch_size:
                yield bucket[:]
                del bucket[:]


def build_batch_data_loader(dataset, sampler, total_batch_size, *, aspect_ratio_grouping=False, num_workers=0,
                            world_size=None):
    """detectron2/data/build.py:249-296: per-rank batch = total_batch_size / W (must divide); with grouping the elements
    flow one by one into AspectRatioGroupedDataset, otherwise consecutive sampler indices form a batch (incomplete
    batches dropped).  torch's DataLoader carries the worker processes exactly as in the reference."""
    import operator

    world = _rank_world(None, None)[1] if world_size is None else int(world_size)
    if total_batch_size <= 0 or total_batch_size % world != 0:
        raise DrnError("Total batch size (%d) must be divisible by the number of gpus (%d)." % (total_batch_size, world))
    batch_size = total_batch_size // world
    if aspect_ratio_grouping:
        loader = torch.utils.data.DataLoader(dataset, sampler=sampler, num_workers=num_workers, batch_sampler=None,
                                             collate_fn=operator.itemgetter(0))
        return AspectRatioGroupedDataset(loader, batch_size)
    batch_sampler = torch.utils.data.sampler.BatchSampler(sampler, batch_size, drop_last=True)
    return torch.utils.data.DataLoader(dataset, num_workers=num_workers, batch_sampler=batch_sampler,
                                       collate_fn=lambda batch: batch)


def build_detection_train_loader(cfg, dataset_dicts, mapper=None, *, rank=None, world_size=None):
    """detectron2/data/build.py:299-354 without the dataset registry (out of scope: the caller passes the list of dataset
    dicts, e.g. after load_proposals_into_dataset): DatasetMapper(cfg, True) by default, TrainingSampler, batches of
    SOLVER.IMS_PER_BATCH / W images grouped by aspect ratio (DATALOADER.ASPECT_RATIO_GROUPING)."""
    if mapper is None:
        mapper = DatasetMapper(cfg, True)
    dataset = MapDataset(dataset_dicts, mapper)
    name = cfg.DATALOADER.SAMPLER_TRAIN
    if name != "TrainingSampler":
        raise DrnError("training sampler %r is not on the WSL path (every projects/WSL yaml uses TrainingSampler)" % name)
    r, w = _rank_world(rank, world_size)
    sampler = TrainingSampler(len(dataset), rank=r, world_size=w)
    return build_batch_data_loader(dataset, sampler, cfg.SOLVER.IMS_PER_BATCH,
                                   aspect_ratio_grouping=cfg.DATALOADER.ASPECT_RATIO_GROUPING,
                                   num_workers=cfg.DATALOADER.NUM_WORKERS, world_size=w)
