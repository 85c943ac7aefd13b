import torch
from . import _o


def nms(boxes, scores, thr):
    return _o.nms(boxes, scores, thr)


def batched_nms(boxes, scores, idxs, thr):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    n = boxes.shape[0]
    import ctypes
    keep = torch.empty(n, dtype=torch.int64)
    b = boxes.contiguous().float(); s = scores.contiguous().float(); i = idxs.contiguous().to(torch.int64)
    k = _o._lib().oracle_batched_nms(_o._fp(b), _o._fp(s), _o._fp(i), ctypes.c_int64(n), ctypes.c_float(thr), _o._fp(keep))
    return keep[:k].clone()
