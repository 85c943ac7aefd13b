"""End-to-end parity at a REAL image size.

The tiny goldens pin the semantics against the reference and the full-size cases pin the bench shapes (224 x 224); the kernels
a real VOC image takes - conv_ring / pp8 / conv1x1_pp on 60 x 80 .. 120 x 160 maps, the walking / 4-channel-cell RoIPool
kernels, the eager variable-shape step - were only pinned as units.  Here the whole model runs at 480 x 640 with 500
proposals for the shipped recipe's trunk (WS-R50 dilated C5, stride 8: a 59 x 79 x 2048 map) and for the constructed C4 trunk
(stride 16: 30 x 40 x 1024) against the CPU oracle on the same seeded weights and SURVEY 8(d) inputs:
  fp32 parity mode: the trunk's output map and every loss within 1e-4 relative (north-star bound), the MIL image scores, the
    fc7 bias / predictor gradients, the per-proposal inference scores and the detections;
  bf16 (the benchmarked dtype): every loss within the bench-mode bound 3 x |A - B| + 1 % of both oracles (A = the oracle that
    rounds what the product stores in bf16, B = the fp32 oracle; tests/test_bench_mode_gpu.py derives the bound), finite
    gradients - and neither saturated nor exactly zero refinement losses (the failure VERDICT r5 found in a timing tool).
The oracle needs ~20 s of host time per case at this size."""
import copy
import os

import numpy as np
import pytest
import torch

import golden_util as G
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
O = G.O
load_package()

H, W, R = 480, 640, 500
CASES = {
    "r50dc5": dict(arch="wsr50", out_feature="res5", res5_dilation=2, num_classes=20),  # oicr_WSR_50_DC5_1x.yaml's trunk
    "r50c4": dict(arch="wsr50", out_feature="res4", res5_dilation=1, num_classes=20),   # BASELINE configs[1]'s trunk
}


def _relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _inputs(batch, cuda=False):
    ins = G.drn_inputs([dict(b, gt_boxes=torch.zeros(len(b["gt_classes"]), 4)) for b in batch])
    if cuda:
        for x in ins:
            x["image"] = x["image"].cuda()
            x["proposals"].proposal_boxes.tensor = x["proposals"].proposal_boxes.tensor.cuda()
            x["proposals"].objectness_logits = x["proposals"].objectness_logits.cuda()
    return ins


@pytest.mark.parametrize("case", list(CASES))
def test_real_size_fp32_train_step_and_inference_match_oracle(case):
    from drn_wsod_pytorch_amd.engine import build_optimizer

    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ocfg = O.OracleCfg(dropout=0.0, **CASES[case])
    p = O.init_params(ocfg, seed=3)
    batch = O.synthetic_batch(1, R, ocfg, seed=977, H=H, W=W)
    # ---- oracle: trunk map, inference on the step-0 weights, then one train step (which updates p)
    with torch.no_grad():
        x, _ = O.preprocess_image([b["image"] for b in batch], ocfg)
        ref_feat = O.backbone_forward(p, x, ocfg).numpy()
        ref_det, ref_scores, ref_boxes = O.model_inference(p, batch, ocfg)
    names = ["roi_heads.box_head.fc2.bias", "roi_heads.box_refinery_2.cls_score.weight", "roi_heads.box_predictor.det.weight"]
    ref_losses, ref_grads, aux = O.train_step(p, batch, ocfg, O.SGDState(ocfg), return_aux=True)

    cfg, model = G.drn_model(ocfg, 3, "cuda", 5, "fp32")
    model.roi_heads.box_head.dropout_p = 0.0
    # ---- trunk output map at the real size
    model.eval()
    with torch.no_grad():
        images = model.preprocess_image(_inputs(batch))
        feats = model.backbone(images.tensor)
    f = feats[ocfg.out_feature].float().cpu().numpy()
    assert f.shape == ref_feat.shape and f.shape[2:] == ((59, 79) if case == "r50dc5" else (30, 40))
    assert _relerr(f, ref_feat) < 1e-4, _relerr(f, ref_feat)
    # ---- inference: per-proposal scores, boxes, detections
    res, all_scores, all_boxes = model.inference(_inputs(batch), do_postprocess=False)
    assert _relerr(all_scores[0][0].cpu().numpy(), ref_scores[0].numpy()) < 1e-4
    assert torch.equal(all_boxes[0][0].cpu(), ref_boxes[0])  # zero-delta decode: bit-exact
    rb, rs, rc, rr = ref_det[0]
    n = min(len(rs), len(res[0]))
    assert abs(len(rs) - len(res[0])) <= 2 and n > 0
    assert torch.allclose(res[0].scores.cpu()[:n], rs[:n], rtol=1e-3, atol=1e-6)
    gaps = (rs[:-1] - rs[1:]).abs() if len(rs) > 1 else torch.zeros(0)
    stable = torch.ones(n, dtype=torch.bool)
    if n > 1:
        small = gaps[: n - 1] < 1e-5 * rs[: n - 1].abs()
        stable[:-1] &= ~small
        stable[1:] &= ~small
    assert torch.equal(res[0].pred_classes.cpu()[stable], rc[:n][stable])
    # ---- one train step (eager, variable-shape path)
    model.train()
    opt = build_optimizer(cfg, model)
    opt.zero_grad()
    losses = model(_inputs(batch))
    sum(losses.values()).backward()
    got = {k: float(v.detach()) for k, v in losses.items()}
    assert set(got) == set(ref_losses)
    for k in got:
        assert abs(got[k] - ref_losses[k]) <= 1e-4 * max(abs(ref_losses[k]), 1e-3), (k, got[k], ref_losses[k])
    st = model.roi_heads._last_state
    assert _relerr(st["aux"]["img_scores"].cpu().numpy(), aux["img_scores"].numpy()) < 1e-4
    sd = dict(model.named_parameters())
    for nme in names:
        assert _relerr(sd[nme].grad.detach().cpu().numpy(), ref_grads[nme].numpy()) < 4e-3, nme
    opt.step()
    torch.cuda.synchronize()
    load_package().set_precision("fp32")


@pytest.mark.parametrize("case", list(CASES))
def test_real_size_bf16_step_within_the_bench_mode_bound(case):
    from drn_wsod_pytorch_amd.engine import build_optimizer

    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ocfg = O.OracleCfg(dropout=0.0, **CASES[case])
    batch = O.synthetic_batch(1, R, ocfg, seed=977, H=H, W=W)
    ref = {}
    for emulate in (True, False):
        c = copy.deepcopy(ocfg)
        c.emulate_bf16 = emulate
        p = O.init_params(c, seed=3)
        ref[emulate], _ = O.train_step(p, batch, c, O.SGDState(c))
    cfg, model = G.drn_model(ocfg, 3, "cuda", 5, "bf16")
    model.roi_heads.box_head.dropout_p = 0.0
    model.train()
    opt = build_optimizer(cfg, model)
    opt.zero_grad()
    losses = model(_inputs(batch, cuda=True))
    sum(losses.values()).backward()
    got = {k: float(v.detach()) for k, v in losses.items()}
    assert set(got) == set(ref[True])
    for k in got:
        a, b = ref[True][k], ref[False][k]
        tol = 3.0 * abs(a - b) + 0.01 * max(abs(a), abs(b), 1e-3)
        assert np.isfinite(got[k]) and abs(got[k] - a) <= tol and abs(got[k] - b) <= tol, (k, got[k], a, b)
        if k.startswith("loss_cls_r"):  # a diverged run shows refinement losses of exactly 0 or in the hundreds
            assert 1e-4 < got[k] < 20.0, (k, got[k])
    for nme, q in model.named_parameters():
        if q.requires_grad and q.grad is not None:
            assert torch.isfinite(q.grad).all(), nme
    opt.step()
    torch.cuda.synchronize()
    load_package().set_precision("fp32")
