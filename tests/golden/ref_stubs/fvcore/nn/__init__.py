import torch
from . import weight_init  # noqa


def smooth_l1_loss(input, target, beta, reduction="none"):
    if beta < 1e-5:
        loss = torch.abs(input - target)
    else:
        n = torch.abs(input - target)
        loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    if reduction == "mean":
        loss = loss.mean()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


def giou_loss(*a, **k):
    raise NotImplementedError


def sigmoid_focal_loss_jit(*a, **k):
    raise NotImplementedError


sigmoid_focal_loss = sigmoid_focal_loss_jit
sigmoid_focal_loss_star = sigmoid_focal_loss_jit
sigmoid_focal_loss_star_jit = sigmoid_focal_loss_jit
