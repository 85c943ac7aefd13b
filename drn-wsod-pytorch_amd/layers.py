"""The `detectron2.layers` operator surface this path uses (detectron2/layers/__init__.py:2-11):
Conv2d (+norm +activation), FrozenBatchNorm2d, Linear, ROIAlign, ShapeSpec, CNNBlockBase, get_norm,
cat, nonzero_tuple — same names, constructor arguments and state_dict keys, executing on the HIP
kernels.  Feature maps travel NHWC in the compute dtype; the [N,C,H,W] tensors handed across module
boundaries are channels-last views of those buffers (shape-compatible with the reference)."""
from collections import namedtuple

import torch
import torch.nn.functional as F
from torch import nn

from . import compute_dtype, ops
from ._cabi import DrnError


# CSCROIHeads' image-gradient passes walk the trunk's explicit backward for d/dx only: no weight / bias gradients are
# written and the blocks keep their saved activations (several passes per step, then the real backward)
_DX_ONLY = [False]


class dx_only:
    def __enter__(self):
        self._was, _DX_ONLY[0] = _DX_ONLY[0], True

    def __exit__(self, *a):
        _DX_ONLY[0] = self._was


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    """detectron2/layers/shape_spec.py."""

    def __new__(cls, *, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)


def cat(tensors, dim=0):
    """detectron2/layers/wrappers.py:14-22."""
    assert isinstance(tensors, (list, tuple))
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


def nonzero_tuple(x):
    if x.dim() == 0:
        return x.unsqueeze(0).nonzero().unbind(1)
    return x.nonzero().unbind(1)


def to_nhwc(x, dtype=None, cpad_to=None):
    """[N,C,H,W] (any memory format) -> contiguous [N,H,W,Cp] in `dtype` (plumbing copy; free when x
    already is a channels-last view of an NHWC buffer)."""
    dtype = dtype or compute_dtype()
    y = x.permute(0, 2, 3, 1)
    c = y.shape[-1]
    if cpad_to and c % cpad_to:
        cp = (c + cpad_to - 1) // cpad_to * cpad_to
        out = torch.zeros(y.shape[:-1] + (cp,), dtype=dtype, device=x.device)
        out[..., :c] = y
        return out
    return y.to(dtype).contiguous()


def from_nhwc(y):
    """contiguous [N,H,W,C] -> [N,C,H,W] channels-last view (no copy)."""
    return y.permute(0, 3, 1, 2)


class FrozenBatchNorm2d(nn.Module):
    """detectron2/layers/batch_norm.py:14-124: fixed statistics + affine as BUFFERS named weight, bias,
    running_mean, running_var.  On the hot path it is folded into the preceding Conv2d's epilogue."""

    _version = 3

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def folded(self):
        """(scale, bias) of y = x*scale + bias, batch_norm.py:46-50."""
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale.float().contiguous(), (self.bias - self.running_mean * scale).float().contiguous()

    def forward(self, x):
        scale, bias = self.folded()
        return x * scale.reshape(1, -1, 1, 1).to(x.dtype) + bias.reshape(1, -1, 1, 1).to(x.dtype)

    def __repr__(self):
        return "FrozenBatchNorm2d(num_features={}, eps={})".format(self.num_features, self.eps)

    @classmethod
    def convert_frozen_batchnorm(cls, module):
        """Replace every BatchNorm2d / SyncBatchNorm below `module` (or `module` itself) by a FrozenBatchNorm2d holding the same
        statistics and affine; returns the converted module (detectron2/layers/batch_norm.py:92-124 describes the contract).
        Written as one pass over named_modules() with in-place re-parenting instead of a recursive rebuild."""
        bn_types = (nn.BatchNorm2d, nn.SyncBatchNorm)

        def frozen_from(bn):
            out = cls(bn.num_features, eps=bn.eps)
            with torch.no_grad():
                if bn.affine:
                    out.weight.copy_(bn.weight)
                    out.bias.copy_(bn.bias)
                out.running_mean.copy_(bn.running_mean)
                out.running_var.copy_(bn.running_var)
            return out

        if isinstance(module, bn_types):
            return frozen_from(module)
        todo = [(parent, name, child) for parent in module.modules() for name, child in parent.named_children()
                if isinstance(child, bn_types)]
        for parent, name, child in todo:
            setattr(parent, name, frozen_from(child))
        return module


def get_norm(norm, out_channels):
    """detectron2/layers/batch_norm.py:127-149 (the WSL path only ever asks for "FrozenBN" or none)."""
    if isinstance(norm, str):
        if len(norm) == 0:
            return None
        if norm != "FrozenBN":
            raise DrnError("norm '%s' is off the DRN-WSOD hot path (only FrozenBN / none are built)" % norm)
        return FrozenBatchNorm2d(out_channels)
    return norm(out_channels)


class Conv2d(nn.Conv2d):
    """detectron2/layers/wrappers.py:41-99: torch.nn.Conv2d + `norm` + `activation`; forward = conv -> norm ->
    activation.  Runs as ONE implicit-GEMM MFMA kernel with the frozen-BN affine, an optional residual add and
    the ReLU in the epilogue."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation
        self._pack_key = None
        self._pack = None
        assert self.groups == 1 and self.kernel_size[0] == self.kernel_size[1], "off the DRN-WSOD path"
        assert self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1]

    def cin_pad(self, dtype):
        q = 16 if dtype == ops.FP8 else 8 if dtype == torch.bfloat16 else 4
        return (self.in_channels + q - 1) // q * q

    # ---- quantised trunk (BASELINE configs[4]: "fp8 MFMA conv path"; no reference counterpart) ------------
    _fp8 = None          # dict(out_scale, out_dtype) once enable_fp8() ran
    _calib_amax = None   # float while Backbone.calibrate_fp8() records this conv's output range

    def enable_fp8(self, out_scale, out_dtype=None):
        """Run this (frozen) conv on the fp8 MFMA path: weights quantised per output channel to OCP e4m3fn, the output
        stored as fp8 with the per-tensor scale `out_scale` (y_q = fp8(y * out_scale)), or as `out_dtype` (bf16 for the
        feature map that leaves the trunk) unscaled."""
        if self.weight.requires_grad:
            raise DrnError("the fp8 conv path is for the frozen trunk (MODEL.BACKBONE.FREEZE_AT = 5)")
        out_dtype = out_dtype or ops.FP8
        self._fp8 = dict(out_scale=float(out_scale) if out_dtype == ops.FP8 else 1.0, out_dtype=out_dtype)
        self._packq_key = None

    def disable_fp8(self):
        self._fp8 = None

    def packed_fp8(self, in_dtype, in_scale):
        """(w, alpha, beta) of drn_conv2d_nhwc_q for an input stored as x * in_scale in `in_dtype`:
        alpha[c] = bn_scale[c] * s_y / (s_x * s_w[c]), beta[c] = bn_bias[c] * s_y, w[c] = fp8(w[c] * s_w[c]) with
        s_w[c] = 448 / max|w[c]| (a bf16 input - the image, whose +-150 range and 3-channel reduction fp8 would ruin -
        keeps bf16 weights, s_w = 1)."""
        q = self._fp8
        key = (in_dtype, float(in_scale), q["out_scale"], q["out_dtype"], self.weight.data_ptr(), self.weight._version,
               getattr(self, "_pack_gen", 0))
        if key != getattr(self, "_packq_key", None):
            with torch.no_grad():
                cout, cin, kh, kw = self.weight.shape
                w = self.weight.detach().float()
                if self.norm is not None:
                    bn_scale, bn_bias = self.norm.folded()
                    if self.bias is not None:
                        bn_bias = bn_bias + self.bias.detach().float() * bn_scale
                else:
                    bn_scale = torch.ones(cout, device=w.device)
                    bn_bias = self.bias.detach().float() if self.bias is not None else torch.zeros(cout, device=w.device)
                if in_dtype == ops.FP8:
                    s_w = ops.FP8_MAX / w.abs().amax(dim=(1, 2, 3)).clamp_min(1e-12)
                    wq = (w * s_w.view(-1, 1, 1, 1)).clamp(-ops.FP8_MAX, ops.FP8_MAX).to(ops.FP8)
                else:
                    s_w = torch.ones(cout, device=w.device)
                    wq = w.to(in_dtype)
                cp = self.cin_pad(in_dtype)
                wp = torch.zeros((cout, kh, kw, cp), dtype=torch.float32, device=w.device)
                wp[..., :cin] = wq.float().permute(0, 2, 3, 1)  # exact: every fp8 / bf16 value is an fp32 value
                k = kh * kw * cp
                packed = torch.zeros((cout, ops.kpad(k, in_dtype)), dtype=torch.float32, device=w.device)
                packed[:, :k] = wp.reshape(cout, k)
                packed = packed.to(in_dtype)
                alpha = (bn_scale * q["out_scale"] / (float(in_scale) * s_w)).float().contiguous()
                beta = (bn_bias * q["out_scale"]).float().contiguous()
                self._packq, self._packq_key = (packed, alpha, beta, s_w), key
        return self._packq

    def _run_fp8(self, x, residual, relu):
        q = self._fp8
        s_x = getattr(x, "_drn_scale", 1.0)
        wq, alpha, beta, _ = self.packed_fp8(x.dtype, s_x)
        assert x.shape[-1] == self.cin_pad(x.dtype), (x.shape, self.in_channels)
        res_mult = q["out_scale"] / getattr(residual, "_drn_scale", 1.0) if residual is not None else 1.0
        y = ops.conv2d_nhwc_q(x, wq, self.out_channels, self.kernel_size[0], self.kernel_size[1], self.stride[0],
                              self.padding[0], self.dilation[0], alpha, beta, q["out_dtype"], residual, res_mult, relu)
        y._drn_scale = q["out_scale"]
        return y

    def _source_tensors(self):
        """the tensors the packed compute copies are derived from (a launch plan re-checks their `_version`s)"""
        pr = self._parameters
        tens = [pr["weight"]] + ([pr["bias"]] if pr.get("bias") is not None else [])
        n_ = self._modules.get("norm")
        if n_ is not None:
            tens += list(n_.parameters()) + list(n_.buffers())
        return tens

    def packed(self, dtype):
        """(w [Cout, ldw] K-major with k = (kh*KW + kw)*Cin_pad + ci, scale [Cout], bias [Cout]) cached until
        a parameter / buffer changes (load_state_dict bumps _version)."""
        # (runs once per conv launch on the eager path: plain dict lookups instead of nn.Module.__getattr__ - 93 calls per
        # step were 0.8 ms of host time.  Buffers are re-read from the dicts: Module.to() replaces buffer objects)
        pr = self._parameters
        w_, b_, n_ = pr["weight"], pr.get("bias"), self._modules.get("norm")
        tens = [w_] if b_ is None else [w_, b_]
        if n_ is not None:
            nb = n_._buffers
            tens += ([nb["weight"], nb["bias"], nb["running_mean"], nb["running_var"]] if "running_var" in nb
                     else list(n_.parameters()) + list(n_.buffers()))
        key = (dtype,) + tuple([(t.data_ptr(), t._version) for t in tens])
        if key != self._pack_key:
            with torch.no_grad():
                cout, cin, kh, kw = self.weight.shape
                cp = self.cin_pad(dtype)
                w = torch.zeros((cout, kh, kw, cp), dtype=torch.float32, device=self.weight.device)
                w[..., :cin] = self.weight.detach().float().permute(0, 2, 3, 1)
                k = kh * kw * cp
                wp = torch.zeros((cout, ops.kpad(k, dtype)), dtype=dtype, device=w.device)
                wp[:, :k] = w.reshape(cout, k).to(dtype)
                if self.norm is not None:
                    assert isinstance(self.norm, FrozenBatchNorm2d), "only FrozenBN is built on this path"
                    scale, bias = self.norm.folded()
                    if self.bias is not None:
                        bias = bias + self.bias.detach().float() * scale
                else:
                    scale = None
                    bias = self.bias.detach().float().contiguous() if self.bias is not None else None
                self._pack = (wp, scale, bias)
                self._pack_key = key
        return self._pack

    def packed_dgrad(self, dtype):
        """weights of the data-gradient pass, which for a stride-1 conv (see _dgrad for stride > 1) is itself a conv over the output gradient with
        the taps flipped and the channel roles swapped: wd[ci][(kh'*KW + kw')*Cout + co] = w[co, ci, KH-1-kh', KW-1-kw'],
        padding dil*(K-1) - pad.  Rows beyond Cin (the channel padding of x) are zero."""
        key = (dtype, self.weight.device, self.weight.data_ptr(), self.weight._version, getattr(self, "_pack_gen", 0))
        if key != getattr(self, "_packd_key", None):
            with torch.no_grad():
                cout, cin, kh, kw = self.weight.shape
                w = self.weight.detach().float().flip(2, 3).permute(1, 2, 3, 0).reshape(cin, kh * kw * cout)
                wd = torch.zeros((self.cin_pad(dtype), ops.kpad(kh * kw * cout, dtype)), dtype=dtype, device=w.device)
                wd[:cin, : kh * kw * cout] = w.to(dtype)
                self._packd, self._packd_key = wd, key
        return self._packd

    def invalidate_packs(self):
        """the optimizer updated the weights in place (no _version bump): drop the packed compute copies"""
        self._pack_key = None
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1

    def _dgrad(self, g, x_shape, dtype):
        """d loss / d x from the pre-activation gradient g [N,Ho,Wo,Cout]: a stride-1 conv over g with flipped taps
        (packed_dgrad).  A strided conv (only stem.conv1 here) first spreads g over the stride-1 output grid - zeros
        between the taken positions and after the last one, up to H + 2p - d(K-1) rows - which is conv_transpose."""
        n, ho, wo, cout = g.shape
        k, s, d_, pd = self.kernel_size[0], self.stride[0], self.dilation[0], self.padding[0]
        if s > 1:
            hs, ws = x_shape[1] + 2 * pd - d_ * (k - 1), x_shape[2] + 2 * pd - d_ * (k - 1)
            gz = torch.zeros((n, hs, ws, cout), dtype=g.dtype, device=g.device)
            gz[:, ::s, ::s][:, :ho, :wo] = g
            g = gz
        return ops.conv2d_nhwc(g, self.packed_dgrad(dtype), self.cin_pad(dtype), k, k, 1, d_ * (k - 1) - pd, d_)

    def backward_nhwc(self, x, y, dy, relu, need_dx, residual, accumulate):
        """Explicit backward of run_nhwc (torch.autograd of F.conv2d + FrozenBatchNorm2d + relu_ + residual add):
        x [N,H,W,Cin_pad] input, y [N,Ho,Wo,Cout] output (saved for the ReLU mask), dy gradient of y (compute dtype or
        fp32).  Weight / bias gradients are written (or accumulated) into self.weight.grad / self.bias.grad, which the
        optimizer keeps as views of its fp32 gradient arena in the state_dict layout.
        Returns (dx or None, gradient of the residual input or None)."""
        dtype = x.dtype
        n, ho, wo, cout = y.shape
        P = n * ho * wo
        k, cin = self.kernel_size[0], self.in_channels
        _, scale, _ = self.packed(dtype)
        dy2 = dy.reshape(P, cout)
        saved = y.reshape(P, cout) if relu else None
        d_res = None
        if residual:
            d_res = torch.empty((P, cout), dtype=dtype, device=x.device)
            ops.bias_act_bwd(dy2, P, cout, saved=saved, dpre=d_res)
        key = (P, cout, dtype)
        if getattr(self, "_bw_key", None) != key:  # g^T keeps zero padding columns: allocate once per shape
            self._bw_gT = torch.zeros((cout, ops.kpad(P, dtype)), dtype=dtype, device=x.device)
            self._bw_key = key
        g = torch.empty((P, cout), dtype=dtype, device=x.device)
        want_w = self.weight.requires_grad and not _DX_ONLY[0]
        bgrad = self.bias.grad if (self.bias is not None and self.bias.requires_grad and not _DX_ONLY[0]) else None
        ops.bias_act_bwd(dy2, P, cout, saved=saved, colscale=scale, dpre=g, dpreT=self._bw_gT if want_w else None,
                         colsum=bgrad, accumulate_colsum=accumulate)
        if want_w:
            col = ops.im2col_t(x, cin, k, k, self.stride[0], self.padding[0], self.dilation[0])
            gw = self.weight.grad.view(1, cout, cin * k * k)
            ops.gemm_nt(self._bw_gT, col, cout, cin * k * k, ops.kpad(P, dtype), out=gw, accumulate=accumulate)
        dx = self._dgrad(g.view(n, ho, wo, cout), x.shape, dtype) if need_dx else None
        return dx, (d_res.view(n, ho, wo, cout) if d_res is not None else None)

    def run_nhwc(self, x, residual=None, relu=False, explicit_backward=False):
        if not explicit_backward and torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad):
            raise DrnError("a stand-alone Conv2d has no autograd on this path: trainable convs run inside the backbone, "
                           "whose blocks keep what their explicit backward needs (backbone.backward_nhwc)")
        if self._fp8 is not None:
            if explicit_backward:
                raise DrnError("the fp8 conv path has no backward: it serves the frozen trunk")
            return self._run_fp8(x, residual, relu)
        wp, scale, bias = self.packed(x.dtype)
        assert x.shape[-1] == self.cin_pad(x.dtype), (x.shape, self.in_channels)
        y = ops.conv2d_nhwc(x, wp, self.out_channels, self.kernel_size[0], self.kernel_size[1], self.stride[0],
                            self.padding[0], self.dilation[0], scale, bias, residual, relu)
        if self._calib_amax is not None:  # Backbone.calibrate_fp8(): range of this conv's stored output
            self._calib_amax = max(self._calib_amax, float(y.float().abs().max()))
        return y

    def forward(self, x):
        dtype = compute_dtype()
        fuse_relu = self.activation in (F.relu, F.relu_)
        y = self.run_nhwc(to_nhwc(x, dtype, 8 if dtype == torch.bfloat16 else 4), None, fuse_relu)
        y = from_nhwc(y)
        if self.activation is not None and not fuse_relu:
            y = self.activation(y)
        return y


class Linear(nn.Linear):
    """torch.nn.Linear (detectron2/layers/wrappers.py re-exports it).  Standalone forward runs the MFMA GEMM +
    bias epilogue; inside the ROI heads the fused head engine reads the parameters directly."""

    def forward(self, x):
        if torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad):
            raise DrnError("stand-alone Linear has no autograd on this path; train through OICRROIHeads")
        dtype = compute_dtype()
        m, k = x.shape
        kp = ops.kpad(k, dtype)
        a = torch.zeros((m, kp), dtype=dtype, device=x.device)
        a[:, :k] = x
        w = torch.zeros((self.out_features, kp), dtype=dtype, device=x.device)
        w[:, :k] = self.weight.detach()
        out = torch.zeros((m, self.out_features), dtype=torch.float32, device=x.device)
        ops.bias_act_fwd(ops.gemm_nt(a, w, m, self.out_features, kp), m, self.out_features,
                         self.bias.detach().float().contiguous() if self.bias is not None else None, False, out=out)
        return out


class _ROIOp(torch.autograd.Function):
    """detectron2/layers/roi_align.py:22-59 (`_ROIAlign`) and torchvision's RoIPool function: forward through
    drn_roi_pool_nhwc, backward w.r.t. the feature map through drn_roi_pool_backward_nhwc (rois get no gradient)."""

    @staticmethod
    def forward(ctx, input, rois, p, scale, mode, sampling_ratio, aligned):
        x = to_nhwc(input, input.dtype if input.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        rois = rois.float().contiguous()
        need = input.requires_grad
        res = ops.roi_pool_nhwc(x, rois, None, p, scale, mode=mode, sampling_ratio=sampling_ratio, aligned=aligned,
                                want_argmax=(need and mode == 0))
        out, arg = res if (need and mode == 0) else (res, None)
        ctx.args = (tuple(x.shape), p, scale, mode, sampling_ratio, aligned, input.dtype)
        ctx.save_for_backward(rois, arg)
        c = input.shape[1]
        return out[:, : c * p * p].reshape(rois.shape[0], c, p, p)

    @staticmethod
    def backward(ctx, grad):
        rois, arg = ctx.saved_tensors
        shape, p, scale, mode, sampling_ratio, aligned, dtype = ctx.args
        g = grad.reshape(grad.shape[0], -1)
        g = (g if g.dtype in (torch.float32, torch.bfloat16) else g.float()).contiguous()
        d = ops.roi_pool_backward_nhwc(g, rois, None, shape, p, scale, mode=mode, sampling_ratio=sampling_ratio,
                                       aligned=aligned, argmax=arg)
        return d.permute(0, 3, 1, 2).to(dtype), None, None, None, None, None, None


class ROIAlign(nn.Module):
    """detectron2/layers/roi_align.py:63-117."""

    def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=True):
        super().__init__()
        self.output_size = output_size if isinstance(output_size, (tuple, list)) else (output_size, output_size)
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.aligned = aligned

    def forward(self, input, rois):
        assert rois.dim() == 2 and rois.size(1) == 5
        assert rois.dim() == 2 and rois.size(1) == 5
        return _ROIOp.apply(input, rois, self.output_size[0], self.spatial_scale, 1, self.sampling_ratio, self.aligned)

    def __repr__(self):
        return "ROIAlign(output_size={}, spatial_scale={}, sampling_ratio={}, aligned={})".format(
            self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)


class RoIPool(nn.Module):
    """torchvision.ops.RoIPool interface (constructed at detectron2/modeling/poolers.py:162-165)."""

    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size = output_size if isinstance(output_size, (tuple, list)) else (output_size, output_size)
        self.spatial_scale = spatial_scale

    def forward(self, input, rois):
        return _ROIOp.apply(input, rois, self.output_size[0], self.spatial_scale, 0, 0, False)

    def __repr__(self):
        return "RoIPool(output_size={}, spatial_scale={})".format(self.output_size, self.spatial_scale)


class CNNBlockBase(nn.Module):
    """detectron2/layers/blocks.py:12-48."""

    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.stride = stride

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        FrozenBatchNorm2d.convert_frozen_batchnorm(self)
        return self
