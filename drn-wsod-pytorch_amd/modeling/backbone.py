"""DRN-WSOD backbones behind the reference's registry names:
`build_ws_resnet_backbone` (projects/WSL/wsl/modeling/backbone/resnet_ws.py:616-703; BasicStem :357-416,
BasicBlock :32-112, BottleneckBlock :115-237, ResNet :419-600) and `build_vgg_backbone`
(projects/WSL/wsl/modeling/backbone/vgg.py:125-244).  Same module tree => same state_dict keys
(`backbone.stem.conv1.norm.weight`, `backbone.res4.2.conv2.weight`, `backbone.plain5.0.conv3.bias`, ...).

Execution is MI355X-first: every block runs NHWC, each conv is one implicit-GEMM MFMA launch with the
FrozenBN affine + residual add + ReLU fused in its epilogue, pools are vectorised NHWC kernels; the
[N,C,H,W] tensors returned by forward() are channels-last views of those buffers."""
import ctypes
import os

import numpy as np
import torch
from torch import nn

from .. import _cabi as C
from .. import compute_dtype, ops
from .._cabi import DrnError
from ..layers import _DX_ONLY, dx_only, CNNBlockBase, Conv2d, FrozenBatchNorm2d, ShapeSpec, from_nhwc, get_norm, to_nhwc
from ..registry import BACKBONE_REGISTRY

__all__ = ["Backbone", "BasicStem", "BasicBlock", "BottleneckBlock", "ResNet", "PlainBlock", "VGG16",
           "build_ws_resnet_backbone", "build_vgg_backbone", "build_backbone"]


def _as(t, dtype):
    """gradient tensors travel through the trunk in the compute dtype (contiguous NHWC)"""
    t = t if t.dtype == dtype else ops.cast2d(t.reshape(-1, t.shape[-1]), t.numel() // t.shape[-1], t.shape[-1],
                                               torch.empty(t.shape, dtype=dtype, device=t.device).view(-1, t.shape[-1])
                                               ).view(t.shape)
    return t.contiguous()


def _pool(x, stride):
    """2x2 max-pool that keeps the per-tensor quantisation scale of an fp8 activation"""
    y = ops.maxpool2x2_nhwc(x, stride)
    if hasattr(x, "_drn_scale"):
        y._drn_scale = x._drn_scale
    return y


def c2_msra_fill(module):
    """fvcore.nn.weight_init.c2_msra_fill."""
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


class _PlanBuilder:
    """Records the trunk's layer sequence as `DrnTrunkOp`s (include/drn_wsod.h) over VALUES (one per layer output), then
    maps the values to a handful of reusable activation slots by their last use.  The entries mirror what
    Conv2d.run_nhwc / _run_fp8 and _pool pass to the per-layer entry points, so a plan issues the same launches."""

    # fused groups (3x3 + 1x1 tail, max pool in the conv's epilogue) flagged in the plans; DRN_FUSE_TAILS=0: A/B runs
    fuse_tails = os.environ.get("DRN_FUSE_TAILS", "1") != "0"

    def __init__(self, in_dtype, in_channels):
        self.vals = [dict(dtype=in_dtype, scale=1.0, c=in_channels)]  # value 0: the (normalised, padded) image
        self.ops, self.keep, self.srcs = [], [], []

    def conv(self, m, src, res=None, relu=False):
        v = self.vals[src]
        if m._fp8 is not None:
            q = m._fp8
            w, scale, bias, _ = m.packed_fp8(v["dtype"], v["scale"])
            out_dtype, out_scale = q["out_dtype"], q["out_scale"]
            res_mult = q["out_scale"] / self.vals[res]["scale"] if res is not None else 1.0
        else:
            w, scale, bias = m.packed(v["dtype"])
            out_dtype, out_scale, res_mult = v["dtype"], 1.0, 1.0
        if v["c"] != m.cin_pad(v["dtype"]):
            raise DrnError("trunk plan: %d stored input channels, the conv expects %d" % (v["c"], m.cin_pad(v["dtype"])))
        self.keep += [w, scale, bias]
        self.srcs += m._source_tensors()
        self.ops.append(dict(kind=0, src=src, res=-1 if res is None else res, w=C.ptr(w), scale=C.ptr(scale),
                             bias=C.ptr(bias), cin=v["c"], cout=m.out_channels, ksize=m.kernel_size[0],
                             stride=m.stride[0], pad=m.padding[0], dil=m.dilation[0], relu=int(bool(relu)),
                             ldw=w.stride(0), dtype=C.dt(v["dtype"]), out_dtype=C.dt(out_dtype),
                             res_dtype=C.dt(self.vals[res]["dtype"]) if res is not None else C.dt(out_dtype),
                             res_mult=float(res_mult)))
        self.vals.append(dict(dtype=out_dtype, scale=out_scale, c=m.out_channels))
        return len(self.vals) - 1

    def pool(self, src, stride):
        v = self.vals[src]
        self.ops.append(dict(kind=1, src=src, res=-1, w=None, scale=None, bias=None, cin=v["c"], cout=v["c"], ksize=2,
                             stride=int(stride), pad=0, dil=1, relu=0, ldw=0, dtype=C.dt(v["dtype"]),
                             out_dtype=C.dt(v["dtype"]), res_dtype=C.dt(v["dtype"]), res_mult=1.0))
        self.vals.append(dict(v))
        return len(self.vals) - 1

    def finish(self, outputs):
        """outputs: {feature name: value}.  Linear scan: a value's slot returns to the free list behind its last reader;
        the image and the output features are never overwritten."""
        last = {}
        for i, o in enumerate(self.ops):
            last[o["src"]] = i
            if o["res"] >= 0:
                last[o["res"]] = i
        pinned = {0} | set(outputs.values())
        # a 3x3 / 64 -> 64 conv whose output only the next op reads, a 1x1 conv to 256 channels (conv2 -> conv3 of a res2
        # bottleneck, resnet_ws.py:217-237): flagged, the executor runs the pair as one launch on large maps
        # (drn_conv3x3_pw_nhwc - the intermediate stays in LDS; bit-identical) and as two anywhere else
        readers = {}
        for o in self.ops:
            for v in {o["src"], o["res"]} - {-1}:
                readers[v] = readers.get(v, 0) + 1
        bf = C.dt(torch.bfloat16)
        for i in range(len(self.ops) - 1 if getattr(self, "fuse_tails", True) else 0):
            o, n = self.ops[i], self.ops[i + 1]
            if (o["kind"] == 0 and n["kind"] == 0 and o["ksize"] == 3 and o["cin"] == 64 and o["cout"] == 64 and o["stride"] == 1
                    and o["pad"] == 1 and o["dil"] == 1 and o["res"] < 0 and n["ksize"] == 1 and n["cin"] == 64 and n["cout"] == 256
                    and n["stride"] == 1 and n["pad"] == 0 and n["src"] == i + 1 and n["res"] != i + 1 and readers.get(i + 1) == 1
                    and (i + 1) not in pinned and o["dtype"] == bf and o["out_dtype"] == bf and n["dtype"] == bf
                    and n["out_dtype"] == bf and (n["res"] < 0 or n["res_dtype"] == bf)):
                o["kind"] |= 0x100
        # ... and a conv whose (ReLU'd, bf16) output only the next op reads, a 2x2 / stride-2 max pool: the deep stem's last 3x3
        # (64 -> 64) and the 1x1 of a flagged pair - the pool then runs in that launch's epilogue (DRN_TRUNK_FUSE_POOL)
        for i in range(len(self.ops) - 1 if getattr(self, "fuse_tails", True) else 0):
            o, n = self.ops[i], self.ops[i + 1]
            if (o["kind"] & 0xff) != 0 or n["kind"] != 1 or n["stride"] != 2 or n["src"] != i + 1 or readers.get(i + 1) != 1 \
                    or (i + 1) in pinned or not o["relu"] or o["dtype"] != bf or o["out_dtype"] != bf:
                continue
            stem3 = (o["ksize"] == 3 and o["cin"] == 64 and o["cout"] == 64 and o["stride"] == 1 and o["pad"] == 1
                     and o["dil"] == 1 and not (o["kind"] & 0x100) and (o["res"] < 0 or o["res_dtype"] == bf))
            tail = i > 0 and (self.ops[i - 1]["kind"] & 0x100) != 0
            if stem3 or tail:
                o["kind"] |= 0x200
        # a fused group is ONE launch: everything its ops read must stay alive until its LAST op has its output slot (the
        # two-launch form may reuse conv2's input slot for conv3's output, or the shortcut's slot for the pooled map - the
        # one-launch form reads those while it writes)
        i = 0
        while i < len(self.ops):
            end = i
            if self.ops[i]["kind"] & 0x100:
                end = i + 1
            if self.ops[end]["kind"] & 0x200:
                end += 1
            if end > i:
                for q in range(i, end + 1):
                    for v in {self.ops[q]["src"], self.ops[q]["res"]} - {-1}:
                        if v <= i:  # (values defined inside the group are its private intermediates)
                            last[v] = max(last.get(v, 0), end)
            i = end + 1
        free_at = {}
        for v, i_last in last.items():
            free_at.setdefault(i_last, []).append(v)
        slot_of, free, n_slots = {0: 0}, [], 1
        arr = (C.DrnTrunkOp * max(len(self.ops), 1))()
        for i, o in enumerate(self.ops):
            dst_val = i + 1  # op i defines value i + 1
            if free and dst_val not in pinned:  # an output feature owns its slot: it is sized for that feature alone
                slot = free.pop()
            else:
                slot, n_slots = n_slots, n_slots + 1
            slot_of[dst_val] = slot
            a = arr[i]
            for k, val in o.items():
                if k in ("src", "res"):
                    val = slot_of[val] if val >= 0 else -1
                setattr(a, k, val)
            a.dst = slot
            for val in sorted(free_at.get(i, ())):  # values whose last reader (or whose fused group) ends here
                if val not in pinned:
                    free.append(slot_of[val])
            if dst_val not in last and dst_val not in pinned:  # never read (cannot happen in the built trunks)
                free.append(slot)
        if n_slots > C.TRUNK_MAX_SLOTS:
            raise DrnError("trunk plan needs %d activation slots (> %d)" % (n_slots, C.TRUNK_MAX_SLOTS))
        return dict(ops=arr, n_ops=len(self.ops), n_slots=n_slots, keep=self.keep, srcs=self.srcs,
                    outputs={name: (slot_of[v], self.vals[v]["dtype"], self.vals[v]["scale"]) for name, v in outputs.items()},
                    out_slots={slot_of[v] for v in outputs.values()})


class Backbone(nn.Module):
    """detectron2/modeling/backbone/backbone.py."""

    # ---- launch plan of the frozen trunk (csrc/executor.hip, round 3) ------------------------------------------------
    # forward() of a trunk that keeps nothing for a backward pass is ONE drn_trunk_forward call instead of ~50 per-layer
    # Python walks (the eager step on changing image shapes was bound by that host time).  `use_plan = False` keeps the
    # per-layer path (A/B, and the test that pins plan == per-layer bit for bit).
    use_plan = True

    def _plan_emit(self, b, x):
        """-> {feature name: value} (ResNet / VGG16 walk their units through b)"""
        raise NotImplementedError

    def _apply(self, fn, *a, **k):  # .to() / .cuda() / .float(): parameters and buffers may move
        self.__dict__.pop("_plans", None)
        return super()._apply(fn, *a, **k)

    def _plan_for(self, dtype, cin):
        convs = self.conv_modules()
        # (fp8 configuration, pack generation) of every conv: Conv2d.invalidate_packs() - DataParallel.broadcast_parameters
        # and the trunk optimizer write `.data` in place, which bumps no `_version` - must drop the recorded pack pointers too
        # keyed on VALUES: `id(m._fp8)` collides when disable_fp8() + enable_fp8(new scale) hands out a new dict at
        # the old address, and a re-assigned `weight.data` / parameter moves data_ptr without touching `_version`
        cfg = tuple([(None if m._fp8 is None else (m._fp8["out_scale"], m._fp8["out_dtype"]), m.__dict__.get("_pack_gen", 0))
                     for m in convs])
        plans = self.__dict__.setdefault("_plans", {})
        p = plans.get((dtype, cin))
        if p is not None and p["cfg"] == cfg and p["versions"] == [(t._version, t.data_ptr()) for t in p["srcs"]]:
            return p
        b = _PlanBuilder(dtype, cin)
        p = b.finish(self._plan_emit(b, 0))
        p["cfg"], p["versions"] = cfg, [(t._version, t.data_ptr()) for t in p["srcs"]]
        p["bytes"], p["hwc"] = (ctypes.c_long * p["n_slots"])(), (ctypes.c_int * (3 * p["n_slots"]))()
        p["ptrs"], p["scratch"] = (ctypes.c_void_p * p["n_slots"])(), {}
        plans[(dtype, cin)] = p
        return p

    def _run_plan(self, x_nhwc):
        """x_nhwc: [N, H, W, Cpad] contiguous in the compute dtype -> {feature name: NHWC tensor}"""
        n, h, w, c = x_nhwc.shape
        p = self._plan_for(x_nhwc.dtype, c)
        dt_in = C.dt(x_nhwc.dtype)
        C.call("drn_trunk_shapes", p["ops"], p["n_ops"], p["n_slots"], 0, n, h, w, c, dt_in, p["bytes"], p["hwc"])
        st = C.stream()
        capturing = torch.cuda.is_current_stream_capturing()
        # scratch slots: grow-only per stream (launches of one stream are ordered, so the next forward may overwrite
        # them); under stream capture they are allocated inside the capture, where the graph's pool keeps them
        scratch = None if capturing else p["scratch"].setdefault(st, {})
        ptrs, outs, hold = p["ptrs"], {}, []
        ptrs[0] = x_nhwc.data_ptr()
        by_slot = {slot: (name, dtype, scale) for name, (slot, dtype, scale) in p["outputs"].items()}
        for s_ in range(1, p["n_slots"]):
            nb = p["bytes"][s_]
            if s_ in by_slot:
                name, dtype, scale = by_slot[s_]
                y = torch.empty((n, p["hwc"][3 * s_], p["hwc"][3 * s_ + 1], p["hwc"][3 * s_ + 2]), dtype=dtype,
                                device=x_nhwc.device)
                if dtype == ops.FP8:
                    y._drn_scale = scale
                outs[name] = y
                ptrs[s_] = y.data_ptr()
            elif capturing:
                hold.append(torch.empty((max(nb, 16),), dtype=torch.uint8, device=x_nhwc.device))
                ptrs[s_] = hold[-1].data_ptr()
            else:
                buf = scratch.get(s_)
                if buf is None or buf.numel() < nb:
                    buf = scratch[s_] = torch.empty((max(nb, 16),), dtype=torch.uint8, device=x_nhwc.device)
                ptrs[s_] = buf.data_ptr()
        C.call("drn_trunk_forward", p["ops"], p["n_ops"], p["n_slots"], 0, ptrs, n, h, w, c, dt_in, st)
        return outs

    @property
    def size_divisibility(self):
        return 0

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}

    # CSCROIHeads (rcnn.py:170-171 `images.tensor.requires_grad = True`): set by the meta-architecture; every unit then
    # keeps what its explicit backward needs, frozen or not
    input_grad = False

    def trainable(self, unit=None):
        """does `unit` (default: the whole trunk) hold a parameter with requires_grad?  Asked several times per step
        (which blocks keep activations, whether the trunk may run ahead on a side stream): the parameter lists are
        collected once - walking the module tree every time was ~1 ms of host time per eager step - and only the
        requires_grad flags are re-read, so freeze() / requires_grad_() are followed."""
        cache = self.__dict__.setdefault("_param_lists", {})
        key = id(unit) if unit is not None else 0
        ps = cache.get(key)
        if ps is None:
            ps = cache[key] = list((unit if unit is not None else self).parameters())
        for p_ in ps:
            if p_.requires_grad:
                return True
        return False

    def conv_modules(self):
        """every module with packed compute copies (Conv2d), collected once"""
        ms = self.__dict__.get("_conv_list")
        if ms is None:
            ms = self.__dict__["_conv_list"] = [m for m in self.modules() if hasattr(m, "packed")]
        return ms

    def input_gradient_nhwc(self, dfeat):
        """d (scalar) / d normalised image [N, H, W, Cin_pad] given its gradient w.r.t. the (single) output feature map:
        the explicit backward of EVERY unit for d/dx only - no weight / bias gradient is touched and the saved
        activations stay for further passes and for the training backward."""
        units = getattr(self, "_all_units", None)
        if units is None:
            raise DrnError("input_gradient_nhwc() needs a training-mode forward with `input_grad` set")
        d = dfeat
        with dx_only():
            for j in range(len(units) - 1, -1, -1):
                d = units[j].backward_nhwc(d, need_dx=True, accumulate=False)
        return d

    def _input_nhwc(self, x):
        """accept an ImageList-produced padded NHWC buffer (fast path) or any [N,3,H,W] tensor"""
        dtype = compute_dtype()
        nhwc = getattr(x, "_drn_nhwc", None)
        if nhwc is not None and nhwc.dtype == dtype:
            return nhwc
        return to_nhwc(x, dtype, 8 if dtype == torch.bfloat16 else 4)


class BasicStem(CNNBlockBase):
    def __init__(self, in_channels=3, out_channels=64, norm="BN"):
        super().__init__(in_channels, out_channels, 4)
        self.in_channels = in_channels
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=2, padding=1, bias=False,
                            norm=get_norm(norm, out_channels))
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False,
                            norm=get_norm(norm, out_channels))
        self.conv3 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False,
                            norm=get_norm(norm, out_channels))
        for l in (self.conv1, self.conv2, self.conv3):
            c2_msra_fill(l)
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2, padding=0)

    def forward_nhwc(self, x, save=False):
        o1 = self.conv1.run_nhwc(x, relu=True, explicit_backward=save)
        o2 = self.conv2.run_nhwc(o1, relu=True, explicit_backward=save)
        o3 = self.conv3.run_nhwc(o2, relu=True, explicit_backward=save)
        self._sv = (x, o1, o2, o3) if save else None
        return _pool(o3, 2)

    def plan(self, b, x):
        o3 = b.conv(self.conv3, b.conv(self.conv2, b.conv(self.conv1, x, relu=True), relu=True), relu=True)
        return b.pool(o3, 2)

    def backward_nhwc(self, dy, need_dx, accumulate):
        """explicit backward of forward_nhwc(save=True); training needs no image gradient (need_dx False: conv1 has no
        dgrad), CSCROIHeads' class maps do (Backbone.input_gradient_nhwc)"""
        x, o1, o2, o3 = self._sv
        d = ops.maxpool2x2_bwd_nhwc(o3, _as(dy, o3.dtype), 2)
        d, _ = self.conv3.backward_nhwc(o2, o3, d, True, True, False, accumulate)
        d, _ = self.conv2.backward_nhwc(o1, o2, d, True, True, False, accumulate)
        d, _ = self.conv1.backward_nhwc(x, o1, d, True, need_dx, False, accumulate)  # d/d image: CSC passes only
        if not _DX_ONLY[0]:
            self._sv = None
        return d

    def forward(self, x):
        return from_nhwc(self.forward_nhwc(to_nhwc(x, compute_dtype(), 8 if compute_dtype() == torch.bfloat16 else 4)))


class BasicBlock(CNNBlockBase):
    def __init__(self, in_channels, out_channels, *, stride=1, norm="BN", dilation=1, has_pool=False):
        super().__init__(in_channels, out_channels, stride)
        self.has_pool = has_pool
        self.pool_stride = stride
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, bias=False,
                                   norm=get_norm(norm, out_channels))
        else:
            self.shortcut = None
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=dilation, bias=False,
                            dilation=dilation, norm=get_norm(norm, out_channels))
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=dilation, bias=False,
                            dilation=dilation, norm=get_norm(norm, out_channels))
        for layer in (self.conv1, self.conv2, self.shortcut):
            if layer is not None:
                c2_msra_fill(layer)
        if self.has_pool:
            self.pool = nn.MaxPool2d(kernel_size=2, stride=self.pool_stride, padding=0)

    def forward_nhwc(self, x, save=False):
        o1 = self.conv1.run_nhwc(x, relu=True, explicit_backward=save)
        sc = self.shortcut.run_nhwc(x, explicit_backward=save) if self.shortcut is not None else x
        out = self.conv2.run_nhwc(o1, residual=sc, relu=True, explicit_backward=save)  # out += shortcut; relu_
        self._sv = (x, o1, sc, out) if save else None
        if self.has_pool:
            out = _pool(out, self.pool_stride)
        return out

    def plan(self, b, x):
        o1 = b.conv(self.conv1, x, relu=True)
        sc = b.conv(self.shortcut, x) if self.shortcut is not None else x
        out = b.conv(self.conv2, o1, res=sc, relu=True)
        return b.pool(out, self.pool_stride) if self.has_pool else out

    def backward_nhwc(self, dy, need_dx, accumulate):
        x, o1, sc, out = self._sv
        d = _as(dy, out.dtype)
        if self.has_pool:
            d = ops.maxpool2x2_bwd_nhwc(out, d, self.pool_stride)
        d1, d_sc = self.conv2.backward_nhwc(o1, out, d, True, True, True, accumulate)
        dx, _ = self.conv1.backward_nhwc(x, o1, d1, True, need_dx, False, accumulate)
        if self.shortcut is not None:
            dxs, _ = self.shortcut.backward_nhwc(x, sc, d_sc, False, need_dx, False, accumulate)
        else:
            dxs = d_sc
        if not _DX_ONLY[0]:
            self._sv = None
        return ops.add(dx, dxs) if need_dx else None

    def forward(self, x):
        return from_nhwc(self.forward_nhwc(to_nhwc(x)))


class BottleneckBlock(CNNBlockBase):
    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="BN",
                 stride_in_1x1=False, dilation=1, has_pool=False):
        super().__init__(in_channels, out_channels, stride)
        assert num_groups == 1, "grouped convs are off the DRN-WSOD path"
        self.has_pool = has_pool
        self.pool_stride = stride
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, bias=False,
                                   norm=get_norm(norm, out_channels))
        else:
            self.shortcut = None
        # every conv stride is forced to 1 (resnet_ws.py:148-150); down-sampling is the 2x2 pool below
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=1, bias=False,
                            norm=get_norm(norm, bottleneck_channels))
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=1, padding=dilation,
                            bias=False, groups=num_groups, dilation=dilation, norm=get_norm(norm, bottleneck_channels))
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False,
                            norm=get_norm(norm, out_channels))
        for layer in (self.conv1, self.conv2, self.conv3, self.shortcut):
            if layer is not None:
                c2_msra_fill(layer)
        if self.has_pool:
            self.pool = nn.MaxPool2d(kernel_size=2, stride=self.pool_stride, padding=0)

    def forward_nhwc(self, x, save=False):
        o1 = self.conv1.run_nhwc(x, relu=True, explicit_backward=save)
        o2 = self.conv2.run_nhwc(o1, relu=True, explicit_backward=save)
        sc = self.shortcut.run_nhwc(x, explicit_backward=save) if self.shortcut is not None else x
        out = self.conv3.run_nhwc(o2, residual=sc, relu=True, explicit_backward=save)
        self._sv = (x, o1, o2, sc, out) if save else None
        if self.has_pool:
            out = _pool(out, self.pool_stride)
        return out

    def plan(self, b, x):
        # (the projection shortcut first: conv3 then follows conv2 directly, which is what the executor's fused tail - conv2 +
        # conv3 as one launch on large maps - looks for; same ops, same results)
        sc = b.conv(self.shortcut, x) if self.shortcut is not None else x
        o2 = b.conv(self.conv2, b.conv(self.conv1, x, relu=True), relu=True)
        out = b.conv(self.conv3, o2, res=sc, relu=True)
        return b.pool(out, self.pool_stride) if self.has_pool else out

    def backward_nhwc(self, dy, need_dx, accumulate):
        """torch.autograd of resnet_ws.py:217-237: pool -> relu(conv3 + shortcut) -> conv2 -> conv1, with the gradient
        of the block input = conv1 path + shortcut path"""
        x, o1, o2, sc, out = self._sv
        d = _as(dy, out.dtype)
        if self.has_pool:
            d = ops.maxpool2x2_bwd_nhwc(out, d, self.pool_stride)
        d2, d_sc = self.conv3.backward_nhwc(o2, out, d, True, True, True, accumulate)
        d1, _ = self.conv2.backward_nhwc(o1, o2, d2, True, True, False, accumulate)
        dx, _ = self.conv1.backward_nhwc(x, o1, d1, True, need_dx, False, accumulate)
        if self.shortcut is not None:
            dxs, _ = self.shortcut.backward_nhwc(x, sc, d_sc, False, need_dx, False, accumulate)
        else:
            dxs = d_sc
        if not _DX_ONLY[0]:
            self._sv = None
        return ops.add(dx, dxs) if need_dx else None

    def forward(self, x):
        return from_nhwc(self.forward_nhwc(to_nhwc(x)))


class ResNet(Backbone):
    def __init__(self, stem, stages, num_classes=None, out_features=None):
        super().__init__()
        assert num_classes is None, "classification head is off the DRN-WSOD path"
        self.stem = stem
        self.num_classes = num_classes
        current_stride = self.stem.stride
        self._out_feature_strides = {"stem": current_stride}
        self._out_feature_channels = {"stem": self.stem.out_channels}
        self.stages_and_names = []
        for i, blocks in enumerate(stages):
            assert len(blocks) > 0, len(blocks)
            name = "res" + str(i + 2)
            stage = nn.Sequential(*blocks)
            self.add_module(name, stage)
            self.stages_and_names.append((stage, name))
            self._out_feature_strides[name] = current_stride = int(current_stride * np.prod([k.stride for k in blocks]))
            self._out_feature_channels[name] = blocks[-1].out_channels
        if out_features is None:
            out_features = [name]
        self._out_features = out_features
        children = [x[0] for x in self.named_children()]
        for f in self._out_features:
            assert f in children, "Available children: {}".format(", ".join(children))

    def forward(self, x):
        assert x.dim() == 4, "ResNet takes an input of shape (N, C, H, W). Got {} instead!".format(x.shape)
        outputs = {}
        # trainable blocks (MODEL.BACKBONE.FREEZE_AT < 5) keep their activations for backward_nhwc(); the gradient
        # stops at the first trainable block, so everything in front of it runs as pure inference
        save = self.training and torch.is_grad_enabled()
        units = [self.stem] + [b for stage, _ in self.stages_and_names for b in stage]
        first = next((i for i, u in enumerate(units) if self.trainable(u)), None)
        self._bw_units = units[first:] if (save and first is not None) else None
        keep_all = save and self.input_grad
        self._all_units = units if keep_all else None
        assert len(self._out_features) == 1 or (self._bw_units is None and not keep_all), \
            "trainable trunk / image gradients: one output feature (as configured)"
        if self.use_plan and self._bw_units is None and not keep_all and not getattr(self, "_calibrating", False):
            with torch.no_grad():
                return {k: from_nhwc(v) for k, v in self._run_plan(self._input_nhwc(x)).items()}
        with torch.no_grad():
            i = 0
            y = self.stem.forward_nhwc(self._input_nhwc(x), save=keep_all or (self._bw_units is not None and first == 0))
            if "stem" in self._out_features:
                outputs["stem"] = from_nhwc(y)
            for stage, name in self.stages_and_names:
                for block in stage:
                    i += 1
                    y = block.forward_nhwc(y, save=keep_all or (self._bw_units is not None and i >= first))
                if name in self._out_features:
                    outputs[name] = from_nhwc(y)
        return outputs

    def _plan_emit(self, b, x):
        outs = {}
        y = self.stem.plan(b, x)
        if "stem" in self._out_features:
            outs["stem"] = y
        for stage, name in self.stages_and_names:
            for block in stage:
                y = block.plan(b, y)
            if name in self._out_features:
                outs[name] = y
        return outs

    # ---- fp8 MFMA conv path (BASELINE configs[4]) -------------------------------------------------------------------
    def calibrate_fp8(self, images, margin=1.0):
        """Put the (frozen) trunk on the fp8 path.  `images`: iterable of preprocessed [N,3,H,W] inputs (ImageList.tensor)
        run ONCE in the bf16 mode to record the range of every conv's stored output; each activation tensor then gets
        the per-tensor scale 448 / (margin * max|y|) (OCP e4m3fn: max 448), weights are quantised per output channel.
        The image stays bf16 (so stem.conv1 runs bf16 x bf16 and only its OUTPUT is fp8) and the feature map that leaves
        the trunk is written in bf16 by the last conv.  Returns {conv module name: output scale}."""
        if compute_dtype() != torch.bfloat16:
            raise RuntimeError("the fp8 conv path extends the bf16 mode: set_precision('bf16') first")
        if any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("the fp8 conv path is for the frozen trunk (MODEL.BACKBONE.FREEZE_AT = 5)")
        if len(self._out_features) != 1 or self._out_features[0] != self.stages_and_names[-1][1]:
            raise RuntimeError("fp8 trunk: the single output feature must be the last built stage")
        convs = {n: m for n, m in self.named_modules() if isinstance(m, Conv2d)}
        self.disable_fp8()
        for m in convs.values():
            m._calib_amax = 0.0
        self._calibrating = True  # the per-layer walk records every conv's output range
        try:
            with torch.no_grad():
                for x in images:
                    self.forward(x)
        finally:
            self._calibrating = False
        last_block = self.stages_and_names[-1][0][-1]
        final = last_block.conv3 if hasattr(last_block, "conv3") else last_block.conv2
        scales = {}
        for n, m in convs.items():
            amax, m._calib_amax = m._calib_amax, None
            if m is final:
                m.enable_fp8(1.0, out_dtype=torch.bfloat16)
                scales[n] = 1.0
            else:
                scales[n] = ops.FP8_MAX / max(amax * margin, 1e-12)
                m.enable_fp8(scales[n])
        self._fp8_scales = scales
        return scales

    def disable_fp8(self):
        for m in self.modules():
            if isinstance(m, Conv2d):
                m.disable_fp8()
                m._calib_amax = None
        self._fp8_scales = None

    def backward_nhwc(self, dfeat, accumulate=False):
        """dfeat: gradient of the (single) output feature, NHWC; walks the trainable units in reverse"""
        units = self._bw_units
        assert units is not None, "backward_nhwc() needs a training-mode forward with trainable parameters"
        d = dfeat
        for j in range(len(units) - 1, -1, -1):
            d = units[j].backward_nhwc(d, need_dx=j > 0, accumulate=accumulate)
        self._bw_units = None

    def freeze(self, freeze_at=0):
        if freeze_at >= 1:
            self.stem.freeze()
        for idx, (stage, _) in enumerate(self.stages_and_names, start=2):
            if freeze_at >= idx:
                for block in stage.children():
                    block.freeze()
        return self

    @staticmethod
    def make_stage(block_class, num_blocks, *, in_channels, out_channels, **kwargs):
        blocks = []
        for i in range(num_blocks):
            curr = {}
            for k, v in kwargs.items():
                if k.endswith("_per_block"):
                    assert len(v) == num_blocks
                    curr[k[: -len("_per_block")]] = v[i]
                else:
                    curr[k] = v
            blocks.append(block_class(in_channels=in_channels, out_channels=out_channels, **curr))
            in_channels = out_channels
        return blocks


@BACKBONE_REGISTRY.register()
def build_ws_resnet_backbone(cfg, input_shape):
    norm = cfg.MODEL.RESNETS.NORM
    stem = BasicStem(in_channels=input_shape.channels, out_channels=cfg.MODEL.RESNETS.STEM_OUT_CHANNELS, norm=norm)
    freeze_at = cfg.MODEL.BACKBONE.FREEZE_AT
    out_features = cfg.MODEL.RESNETS.OUT_FEATURES
    depth = cfg.MODEL.RESNETS.DEPTH
    num_groups = cfg.MODEL.RESNETS.NUM_GROUPS
    bottleneck_channels = num_groups * cfg.MODEL.RESNETS.WIDTH_PER_GROUP
    in_channels = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
    out_channels = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS
    stride_in_1x1 = cfg.MODEL.RESNETS.STRIDE_IN_1X1
    res5_dilation = cfg.MODEL.RESNETS.RES5_DILATION
    assert not any(cfg.MODEL.RESNETS.DEFORM_ON_PER_STAGE), "deformable convs are off the DRN-WSOD path"
    assert res5_dilation in {1, 2}, "res5_dilation cannot be {}.".format(res5_dilation)
    num_blocks_per_stage = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3],
                            152: [3, 8, 36, 3]}[depth]
    if depth in [18, 34]:
        assert out_channels == 64, "Must set MODEL.RESNETS.RES2_OUT_CHANNELS = 64 for R18/R34"
        assert num_groups == 1, "Must set MODEL.RESNETS.NUM_GROUPS = 1 for R18/R34"
    stages = []
    max_stage_idx = max({"res2": 2, "res3": 3, "res4": 4, "res5": 5}[f] for f in out_features)
    for idx, stage_idx in enumerate(range(2, max_stage_idx + 1)):
        dilation = res5_dilation if stage_idx in (4, 5) else 1
        first_stride = 2 if idx == 0 or (stage_idx == 3 and res5_dilation == 1) else 1
        has_pool = stage_idx in (2, 3)
        n = num_blocks_per_stage[idx]
        kargs = {"num_blocks": n, "stride_per_block": [1] * (n - 1) + [first_stride],
                 "has_pool_per_block": [False] * (n - 1) + [has_pool], "in_channels": in_channels,
                 "out_channels": out_channels, "norm": norm, "dilation": dilation}
        if depth in [18, 34]:
            kargs["block_class"] = BasicBlock
        else:
            kargs.update(block_class=BottleneckBlock, bottleneck_channels=bottleneck_channels,
                         stride_in_1x1=stride_in_1x1, num_groups=num_groups)
        stages.append(ResNet.make_stage(**kargs))
        in_channels = out_channels
        out_channels *= 2
        bottleneck_channels *= 2
    return ResNet(stem, stages, out_features=out_features).freeze(freeze_at)


class PlainBlock(nn.Module):
    """vgg.py:34-122."""

    def __init__(self, in_channels, out_channels, num_conv=3, dilation=1, stride=1, has_pool=False):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.num_conv, self.dilation, self.has_pool, self.pool_stride = num_conv, dilation, has_pool, stride
        assert num_conv < 5
        cin = in_channels
        for i in range(num_conv):
            conv = Conv2d(cin, out_channels, kernel_size=3, stride=1, padding=dilation, bias=True, groups=1,
                          dilation=dilation, norm=None)
            c2_msra_fill(conv)
            setattr(self, "conv%d" % (i + 1), conv)
            cin = out_channels
        if has_pool:
            self.pool = nn.MaxPool2d(kernel_size=2, stride=self.pool_stride, padding=0)

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        FrozenBatchNorm2d.convert_frozen_batchnorm(self)
        return self

    def forward_nhwc(self, x, save=False):
        acts = [x]
        for i in range(self.num_conv):
            x = getattr(self, "conv%d" % (i + 1)).run_nhwc(x, relu=True, explicit_backward=save)
            acts.append(x)
        self._sv = acts if save else None
        if self.has_pool:
            x = ops.maxpool2x2_nhwc(x, self.pool_stride)
        return x

    def plan(self, b, x):
        for i in range(self.num_conv):
            x = b.conv(getattr(self, "conv%d" % (i + 1)), x, relu=True)
        return b.pool(x, self.pool_stride) if self.has_pool else x

    def backward_nhwc(self, dy, need_dx, accumulate):
        acts = self._sv
        d = _as(dy, acts[-1].dtype)
        if self.has_pool:
            d = ops.maxpool2x2_bwd_nhwc(acts[-1], d, self.pool_stride)
        for i in range(self.num_conv - 1, -1, -1):
            d, _ = getattr(self, "conv%d" % (i + 1)).backward_nhwc(acts[i], acts[i + 1], d, True, need_dx or i > 0, False,
                                                                   accumulate)
        if not _DX_ONLY[0]:
            self._sv = None
        return d

    def forward(self, x):
        return from_nhwc(self.forward_nhwc(to_nhwc(x, compute_dtype(), 8 if compute_dtype() == torch.bfloat16 else 4)))


class VGG16(Backbone):
    """vgg.py:125-231."""

    def __init__(self, conv5_dilation, freeze_at, num_classes=None, out_features=None):
        super().__init__()
        self.num_classes = num_classes
        self._out_feature_strides, self._out_feature_channels = {}, {}
        self.stages_and_names = []
        d2 = conv5_dilation == 2
        spec = [("plain1", PlainBlock(3, 64, num_conv=2, stride=2, has_pool=True), 2),
                ("plain2", PlainBlock(64, 128, num_conv=2, stride=2, has_pool=True), 4),
                ("plain3", PlainBlock(128, 256, num_conv=3, stride=2, has_pool=True), 8),
                ("plain4", PlainBlock(256, 512, num_conv=3, stride=1 if d2 else 2, has_pool=True), 8 if d2 else 16),
                ("plain5", PlainBlock(512, 512, num_conv=3, stride=1, dilation=conv5_dilation, has_pool=False),
                 8 if d2 else 16)]
        for i, (name, block, stride) in enumerate(spec):
            stage = nn.Sequential(block)
            self.add_module(name, stage)
            self.stages_and_names.append((stage, name))
            self._out_feature_strides[name] = stride
            self._out_feature_channels[name] = block.out_channels
            if freeze_at >= i + 1:
                block.freeze()
        self._out_features = out_features or [name]

    def forward(self, x):
        outputs = {}
        save = self.training and torch.is_grad_enabled()
        units = [b for stage, _ in self.stages_and_names for b in stage]
        first = next((i for i, u in enumerate(units) if self.trainable(u)), None)
        self._bw_units = units[first:] if (save and first is not None) else None
        keep_all = save and self.input_grad
        self._all_units = units if keep_all else None
        if self.use_plan and self._bw_units is None and not keep_all and not getattr(self, "_calibrating", False):
            with torch.no_grad():
                return {k: from_nhwc(v) for k, v in self._run_plan(self._input_nhwc(x)).items()}
        with torch.no_grad():
            y = self._input_nhwc(x)
            i = -1
            for stage, name in self.stages_and_names:
                for block in stage:
                    i += 1
                    y = block.forward_nhwc(y, save=keep_all or (self._bw_units is not None and i >= first))
                if name in self._out_features:
                    outputs[name] = from_nhwc(y)
        return outputs

    def _plan_emit(self, b, x):
        outs = {}
        for stage, name in self.stages_and_names:
            for block in stage:
                x = block.plan(b, x)
            if name in self._out_features:
                outs[name] = x
        return outs

    backward_nhwc = ResNet.backward_nhwc


@BACKBONE_REGISTRY.register()
def build_vgg_backbone(cfg, input_shape):
    if cfg.MODEL.VGG.DEPTH == 16:
        return VGG16(cfg.MODEL.VGG.CONV5_DILATION, cfg.MODEL.BACKBONE.FREEZE_AT)
    raise ValueError("VGG depth {} not supported".format(cfg.MODEL.VGG.DEPTH))


def build_backbone(cfg, input_shape=None):
    """detectron2/modeling/backbone/build.py."""
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)
    assert isinstance(backbone, Backbone)
    return backbone
