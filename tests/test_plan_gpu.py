"""The trunk launch plan (Backbone._run_plan -> drn_trunk_forward, csrc/executor.hip) against the per-layer walk it
replaces: the same entry points with the same arguments, so the feature maps must be bit-identical - for every trunk
family, both precisions, the fp8 trunk, changing image sizes (grow-only scratch slots) and under hipGraph capture."""
import numpy as np
import pytest
import torch

import golden_util as G
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
load_package()


def _images(model, sizes, seed=0):
    rs = np.random.RandomState(seed)
    out = []
    for n, h, w in sizes:
        batch = [{"image": torch.from_numpy(rs.randint(0, 256, (3, h, w)).astype(np.float32)).cuda()} for _ in range(n)]
        out.append(model.preprocess_image(batch).tensor)
    return out


def _both(model, x):
    bb = model.backbone
    with torch.no_grad():
        bb.use_plan = True
        a = bb(x)
        bb.use_plan = False
        b = bb(x)
        bb.use_plan = True
    return a, b


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["model_r50c4_tiny", "model_r50dc5_tiny", "model_r18dc5_tiny", "model_vgg16_small"])
def test_plan_equals_per_layer_walk(name, precision):
    if name not in G.MODEL_CASES:
        pytest.skip("no such fixture")
    ocfg = G.MODEL_CASES[name]
    cfg, model = G.drn_model(ocfg, 3, "cuda", 5, precision)
    model.eval()
    # growing, shrinking, batched: the scratch slots are grow-only and shared by consecutive calls
    for x in _images(model, [(1, 64, 64), (2, 97, 131), (1, 40, 56), (1, 160, 120)]):
        a, b = _both(model, x)
        assert a.keys() == b.keys()
        for k in a:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype
            assert torch.equal(a[k], b[k]), (name, precision, k)
    p = next(iter(model.backbone._plans.values()))
    assert p["n_slots"] <= 7, p["n_slots"]  # image + output + a handful of reusable activations


def test_plan_fused_bottleneck_tail_real_size():
    """Round 4: on a real-size image the plan runs conv2 (3x3, 64 -> 64) + conv3 (1x1 to 256, + shortcut, ReLU) of every res2
    bottleneck as ONE launch (DRN_TRUNK_FUSE_NEXT -> drn_conv3x3_pw_nhwc; the 3x3's output stays in LDS).  Against the
    per-layer walk - two launches per pair - on the full-width WS-ResNet50-C4 trunk, bf16, one and two images: bit-identical
    feature maps; three flagged ops in the plan; a small image takes the two-launch path of the same plan."""
    from oracle import wsod_oracle as O

    ocfg = O.OracleCfg(arch="wsr50", out_feature="res4", res5_dilation=1, num_classes=20)
    cfg, model = G.drn_model(ocfg, 3, "cuda", 5, "bf16")
    model.eval()
    for x in _images(model, [(1, 800, 1216), (2, 608, 800), (1, 224, 224)], seed=3):
        a, b = _both(model, x)
        for k in a:
            assert torch.equal(a[k], b[k]), (tuple(x.shape), k)
            assert float(a[k].float().abs().max()) > 0
    p = next(iter(model.backbone._plans.values()))
    assert sum(1 for i in range(p["n_ops"]) if p["ops"][i].kind & 0x100) == 3
    assert sum(1 for i in range(p["n_ops"]) if p["ops"][i].kind & 0x200) >= 1  # a max pool folded into the launch in front of it
    load_package().set_precision("fp32")


def test_plan_follows_weight_updates():
    ocfg = G.MODEL_CASES["model_r50c4_tiny"]
    cfg, model = G.drn_model(ocfg, 3, "cuda", 5, "bf16")
    model.eval()
    bb = model.backbone
    xs = _images(model, [(1, 96, 96), (1, 64, 80)])
    a0, _ = _both(model, xs[0])
    with torch.no_grad():  # an in-place update of a frozen weight (checkpoint load) must invalidate the recorded plan
        bb.res3[0].conv2.weight.mul_(1.5)
    a1, b1 = _both(model, xs[0])
    k = next(iter(a0))
    assert torch.equal(a1[k], b1[k]) and not torch.equal(a0[k], a1[k])
    # a write through `.data` bumps no version counter (DataParallel.broadcast_parameters, the trunk optimizer): those
    # callers drop the packs with Conv2d.invalidate_packs(), which must drop the plan's recorded pack pointers as well
    with torch.no_grad():
        bb.res3[0].conv2.weight.data.mul_(0.5)
    bb.res3[0].conv2.invalidate_packs()
    a1b, b1b = _both(model, xs[0])
    assert torch.equal(a1b[k], b1b[k]) and not torch.equal(a1b[k], a1[k])
    # a re-assigned `.data` (another tensor, same version counter): the plan's key holds data_ptr as well
    with torch.no_grad():
        bb.res3[0].conv2.weight.data = bb.res3[0].conv2.weight.data.clone() * 2.0
    a1c, b1c = _both(model, xs[0])
    assert torch.equal(a1c[k], b1c[k]) and not torch.equal(a1c[k], a1b[k])
    model.float()  # Module._apply: parameters / buffers may have moved - the plans are dropped
    assert "_plans" not in bb.__dict__
    a2, b2 = _both(model, xs[1])
    assert torch.equal(a2[k], b2[k])
    # (the fp8 trunk's plan is pinned against its per-layer walk in tests/test_fp8_gpu.py: it needs real channel widths)


def test_plan_under_graph_capture():
    ocfg = G.MODEL_CASES["model_r50c4_tiny"]
    cfg, model = G.drn_model(ocfg, 3, "cuda", 5, "bf16")
    model.eval()
    bb = model.backbone
    x = _images(model, [(1, 80, 112)])[0]
    with torch.no_grad():
        ref = bb(x)
        k = next(iter(ref))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            bb(x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = bb(x)
        big = _images(model, [(2, 200, 240)])[0]
        bb(big)  # a larger eager call re-allocates the eager scratch slots; the graph keeps its own
        out[k].zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out[k], ref[k])
