"""Host-only half of the trunk launch plan (csrc/executor.hip): drn_trunk_shapes derives every layer's geometry from the
image size and sizes the activation slots; a malformed plan is an argument error.  No kernel is launched."""
import ctypes

import pytest

from __graft_entry__ import build


@pytest.fixture(scope="module")
def cabi():
    return build()._cabi


def _op(C, kind, src, dst, res=-1, cin=8, cout=8, k=3, stride=1, pad=1, dil=1, dtype=1, out_dtype=None):
    o = C.DrnTrunkOp()
    o.kind, o.src, o.dst, o.res = kind, src, dst, res
    o.cin, o.cout, o.ksize, o.stride, o.pad, o.dil, o.relu, o.ldw = cin, cout, k, stride, pad, dil, 1, 64
    o.dtype, o.out_dtype, o.res_dtype, o.res_mult = dtype, dtype if out_dtype is None else out_dtype, dtype, 1.0
    return o


def test_trunk_shapes_follow_the_layers(cabi):
    C = cabi
    # stem-like: conv 3x3 / 2 (8 -> 64), conv 3x3 (64 -> 64), 2x2 pool / 2, 1x1 with a residual, pool / 1
    ops = (C.DrnTrunkOp * 5)(_op(C, 0, 0, 1, cin=8, cout=64, stride=2), _op(C, 0, 1, 2, cin=64, cout=64),
                             _op(C, 1, 2, 1, cin=64, cout=64, stride=2),
                             _op(C, 0, 1, 3, res=1, cin=64, cout=64, k=1, pad=0), _op(C, 1, 3, 2, cin=64, cout=64, stride=1))
    nbytes, hwc = (ctypes.c_long * 4)(), (ctypes.c_int * 12)()
    C.call("drn_trunk_shapes", ops, 5, 4, 0, 2, 37, 50, 8, 1, nbytes, hwc)
    h1, w1 = (37 + 2 - 3) // 2 + 1, (50 + 2 - 3) // 2 + 1  # 19 x 25
    hp, wp = (h1 - 2) // 2 + 1, (w1 - 2) // 2 + 1           # 9 x 12
    assert list(nbytes) == [0, 2 * h1 * w1 * 64 * 2, 2 * h1 * w1 * 64 * 2, 2 * hp * wp * 64 * 2]
    assert list(hwc)[3:] == [hp, wp, 64, hp - 1, wp - 1, 64, hp, wp, 64]
    # fp8 output halves the bytes of its slot
    ops[1].out_dtype = 2
    ops[2].dtype = ops[2].out_dtype = 2
    ops[3].dtype, ops[3].res_dtype = 2, 2
    C.call("drn_trunk_shapes", ops, 5, 4, 0, 2, 37, 50, 8, 1, nbytes, hwc)
    assert nbytes[2] == 2 * h1 * w1 * 64


@pytest.mark.parametrize("bad", ["channels", "residual_shape", "src_unset", "in_place", "slots"])
def test_trunk_shapes_reject_malformed_plans(cabi, bad):
    C = cabi
    ops = (C.DrnTrunkOp * 2)(_op(C, 0, 0, 1, cin=8, cout=64), _op(C, 0, 1, 2, res=1, cin=64, cout=64))
    n_slots = 3
    if bad == "channels":
        ops[1].cin = 32
    elif bad == "residual_shape":
        ops[1].stride = 2
    elif bad == "src_unset":
        ops[1].src = 2
        ops[1].dst = 1
    elif bad == "in_place":
        ops[1].dst = 1
    else:
        n_slots = C.TRUNK_MAX_SLOTS + 1
    nbytes = (ctypes.c_long * 32)()
    with pytest.raises(C.DrnError):
        C.call("drn_trunk_shapes", ops, 2, n_slots, 0, 1, 16, 16, 8, 1, nbytes, None)
