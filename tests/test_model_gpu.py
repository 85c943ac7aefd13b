"""GPU parity of the whole path (GeneralizedRCNNWSL -> backbone -> OICRROIHeads -> losses -> backward ->
fused SGD -> inference) against golden vectors of the unmodified reference and against the oracle.

fp32 parity mode: losses within 1e-4 (BASELINE north star), gradients within 2e-3 of their scale,
feature maps within 1e-4.  bf16 fast mode: compared with the same goldens at the looser tolerance
stated at the check (bf16 operands carry 2^-9 relative rounding per element)."""
import copy
import os

import numpy as np
import pytest
import torch

import golden_util as G
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
O = G.O
load_package()  # registers the hyphen-named package directory as drn_wsod_pytorch_amd
FROZEN_CASES = [n for n in sorted(G.MODEL_CASES) if G.FREEZE_AT.get(n, 5) == 5]
# the PCL golden's training numbers follow the reference's scikit-learn draw / numpy tie order (not functions of the
# inputs; the oracle test replays them): the product is compared with the oracle's fixed definitions instead
REF_PINNED_TRAIN = [n for n in FROZEN_CASES if G.MODEL_CASES[n].heads != "pcl"]


def _relerr(a, b, floor=1e-6):
    """max |a-b| relative to the scale of b; `floor` keeps tensors whose true value is exactly zero (e.g. the
    det-branch bias gradient: a column softmax is shift invariant) from dividing rounding noise by noise"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + floor))


def _record(key, value):
    """DRN_RECORD_SPREAD=<file>: append the measured error behind a tolerance, so that bounds are set from measurements
    (profiles/r4_04_parity_spreads.txt) instead of guessed"""
    path = os.environ.get("DRN_RECORD_SPREAD")
    if path:
        with open(path, "a") as fh:
            fh.write("%s %.3e\n" % (key, value))


def _setup(name, precision):
    assert torch.cuda.is_available()
    ocfg = G.MODEL_CASES[name]
    d = G.load(name)
    cfg, model = G.drn_model(ocfg, int(d["seed"]), "cuda", 5, precision)
    masks = G.dropmasks_from(d)
    model.roi_heads.box_head.dropout_masks = [m.cuda() for m in masks] if masks else None
    if masks is None:
        model.roi_heads.box_head.dropout_p = 0.0  # fixtures were generated with dropout patched to identity
    return ocfg, d, cfg, model


# measured (profiles/r4_04_parity_spreads.txt): step-1 losses of every reference-pinned fixture sit within 7.8e-6 of the
# reference's (step 0: 5.4e-6) - the 2e-2 of round 3 was a guess about "an ill-conditioned toy net"; the north-star bound holds
STEP1_LOSS_TOL = 1e-4


@pytest.mark.parametrize("name", REF_PINNED_TRAIN)
def test_train_two_steps_fp32(name):
    from drn_wsod_pytorch_amd.engine import build_optimizer

    ocfg, d, cfg, model = _setup(name, "fp32")
    batch = G.drn_inputs(G.batch_from(d))
    model.train()
    opt = build_optimizer(cfg, model)
    before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    for step in range(2):
        opt.zero_grad()
        losses = model(batch)
        sum(losses.values()).backward()
        got = {k: float(v.detach()) for k, v in losses.items()}
        for k, v in got.items():
            ref = float(d["step%d_%s" % (step, k)])
            tol = 1e-4 if step == 0 else STEP1_LOSS_TOL  # step 1 sits behind one lr=0.01 SGD step on an ill-conditioned toy net
            _record("two_steps.%s.step%d.%s" % (name, step, k), abs(v - ref) / max(abs(ref), 1e-3))
            assert abs(v - ref) <= tol * max(abs(ref), 1e-3), (step, k, v, ref)
        if step == 0:
            for n, p in model.named_parameters():
                if not p.requires_grad:
                    continue
                if "gradnone0." + n in d:
                    assert p.grad is None  # unused bbox_pred
                    continue
                g = p.grad.detach().cpu().numpy()
                if "grad0." + n in d:
                    ref_g = d["grad0." + n]
                    if np.abs(ref_g).max() < 1e-6:  # analytically zero (det bias): both sides are rounding noise
                        assert np.abs(g).max() < 1e-5, n
                    else:
                        assert _relerr(g, ref_g) < 2e-3, n
                else:
                    assert _relerr(g.reshape(-1)[:4096], d["gradhead0." + n]) < 2e-3, n
                    ref_abs = float(d["gradabs0." + n])
                    assert abs(float(np.abs(g.astype(np.float64)).sum()) - ref_abs) < 2e-3 * ref_abs, n
        opt.step()
        if step == 0:  # parameter delta of the first step = -lr * (g + wd * p): checks the fused SGD end to end
            for n, p in model.named_parameters():
                if p.requires_grad and p.grad is not None:
                    lr = cfg.SOLVER.BASE_LR * (2.0 if n.endswith("bias") else 1.0)
                    wd = 0.0 if n.endswith("bias") else cfg.SOLVER.WEIGHT_DECAY
                    exp = before[n] - lr * (p.grad + wd * before[n])
                    assert torch.allclose(p.detach(), exp, rtol=1e-5, atol=1e-7), n
    for n, p in model.named_parameters():
        if p.requires_grad and ("after2.head." + n) in d:
            got = p.detach().reshape(-1)[:2048].cpu().numpy()
            assert _relerr(got, d["after2.head." + n]) < 5e-3, n


@pytest.mark.parametrize("name", FROZEN_CASES)
def test_backbone_features_and_inference_fp32(name):
    ocfg, d, cfg, model = _setup(name, "fp32")
    batch = G.batch_from(d)
    model.eval()
    with torch.no_grad():
        images = model.preprocess_image(G.drn_inputs(batch, False))
        feats = model.backbone(images.tensor)
    f = feats[str(d["feat_name"])].float().cpu().numpy()
    assert f.shape == d["feat"].shape
    assert _relerr(f, d["feat"]) < 1e-4
    # inference with the step-0 weights against the oracle on the same weights (the fixture's detections were
    # taken after two SGD steps; those are covered by test_train_two_steps + the bit-exact tail tests)
    p = O.seeded_params(O.param_shapes(ocfg), int(d["seed"]))
    ocfg.dropout = 0.0
    ref, ref_scores, ref_boxes = O.model_inference(p, batch, ocfg)
    res, all_scores, all_boxes = model.inference(G.drn_inputs(batch, False), do_postprocess=False)
    for i in range(len(batch)):
        assert _relerr(all_scores[i][0].cpu().numpy(), ref_scores[i].numpy()) < 1e-4
        if any(ocfg.refine_reg):  # real deltas go through exp(): GEMM + expf rounding, not bit-exact
            assert torch.allclose(all_boxes[i][0].cpu(), ref_boxes[i], rtol=1e-4, atol=1e-3)
        else:
            assert torch.equal(all_boxes[i][0].cpu(), ref_boxes[i])  # zero-delta decode: bit-exact
        rb, rs, rc, rr = ref[i]
        # detections: identical classes / order wherever the oracle's score gaps exceed the fp32 noise
        n = min(len(rs), len(res[i]))
        gs = res[i].scores.cpu()
        assert abs(len(rs) - len(res[i])) <= 2
        assert torch.allclose(gs[:n], rs[:n], rtol=1e-3, atol=1e-6)
        gaps = (rs[:-1] - rs[1:]).abs() if len(rs) > 1 else torch.zeros(0)
        stable = torch.ones(n, dtype=torch.bool)
        if n > 1:
            small = gaps[: n - 1] < 1e-5 * rs[: n - 1].abs()
            stable[:-1] &= ~small
            stable[1:] &= ~small
        assert torch.equal(res[i].pred_classes.cpu()[stable], rc[:n][stable])
    out = model(G.drn_inputs(batch, False))
    assert len(out) == len(batch) and "instances" in out[0]


def test_roialign_pooler_model_fp32():
    """ROIAlignV2 pooling inside the full model (frozen backbone) against the oracle run live."""
    name = "model_r50c4_align_tiny"
    ocfg = G.MODEL_CASES[name]
    d = G.load(name)
    cfg, model = G.drn_model(ocfg, int(d["seed"]), "cuda", 5, "fp32")
    model.roi_heads.box_head.dropout_p = 0.0
    batch = G.batch_from(d)
    p = O.seeded_params(O.param_shapes(ocfg), int(d["seed"]))
    ocfg.dropout = 0.0
    ref = O.model_train_losses(p, batch, ocfg)
    model.train()
    losses = model(G.drn_inputs(batch))
    for k, v in losses.items():
        assert abs(float(v.detach()) - float(ref[k])) <= 1e-4 * max(abs(float(ref[k])), 1e-3), (k, float(v.detach()), float(ref[k]))


UNFROZEN = [("model_r50c4_tiny", 2), ("model_r50c4_tiny", 3), ("model_r50c4_align_tiny", 3), ("model_r50dc5_tiny", 3),
            ("model_r18dc5_tiny", 1), ("model_vgg16_small", 2), ("model_r50c4_tiny", 0), ("model_vgg16_small", 0)]


@pytest.mark.parametrize("name,freeze_at", UNFROZEN)
def test_unfrozen_backbone_train_step_fp32(name, freeze_at):
    """MODEL.BACKBONE.FREEZE_AT < 5: fc6 dX, RoIPool / ROIAlign backward, max-pool backward, conv dgrad / wgrad and
    the FrozenBN-affine / ReLU / residual backward of the trunk vs the oracle's autograd on the same seeded weights:
    losses <= 1e-4, every trainable gradient (trunk and heads), and the SGD update of the trunk."""
    from drn_wsod_pytorch_amd.engine import build_optimizer

    ocfg = G.MODEL_CASES[name]
    d = G.load(name)
    seed = int(d["seed"])
    batch = G.batch_from(d)
    ocfg.dropout = 0.0
    p = O.seeded_params(O.param_shapes(ocfg), seed)
    before = {n: t.clone() for n, t in p.items()}
    opt_o = O.SGDState(ocfg)
    ref_losses, ref_grads = O.train_step(p, batch, ocfg, opt_o, freeze_at=freeze_at)
    trunk = [n for n in ref_grads if n.startswith("backbone.")]
    assert trunk, "case must train part of the trunk"

    cfg, model = G.drn_model(ocfg, seed, "cuda", freeze_at, "fp32")
    model.roi_heads.box_head.dropout_p = 0.0
    model.train()
    opt = build_optimizer(cfg, model)
    got_names = sorted(n for n, q in model.named_parameters() if q.requires_grad and n.startswith("backbone."))
    assert got_names == sorted(trunk)
    opt.zero_grad()
    losses = model(G.drn_inputs(batch))
    sum(losses.values()).backward()
    for k, v in losses.items():
        assert abs(float(v.detach()) - ref_losses[k]) <= 1e-4 * max(abs(ref_losses[k]), 1e-3), (k, float(v.detach()), ref_losses[k])
    sd = dict(model.named_parameters())
    # VGG trained from conv1: in this fixture ONE pre-activation of plain2.0.conv1 (of 131072) lies within 1e-5 of zero
    # and lands on different sides of the ReLU in the two fp32 summation orders (measured: the only mask mismatch in
    # the whole trunk).  That moves one pixel's contribution in plain2.0.conv1 and everything below it, which a
    # max-norm comparison sees at the 1e-2 level; those tensors get the relaxed bound, all others the tight one.
    # Every conv / pool backward call inside this very backward was checked against CPU autograd on its own inputs
    # (<= 2e-6 relative).
    def loose(n):
        return (name, freeze_at) == ("model_vgg16_small", 0) and (".plain1." in n or ".plain2.0.conv1." in n)

    for n in trunk:
        g, rg = sd[n].grad.detach().cpu().numpy().astype(np.float64), ref_grads[n].numpy().astype(np.float64)
        assert g.shape == rg.shape, n
        if np.abs(rg).max() < 1e-7:
            assert np.abs(g).max() < 1e-5, n
        else:
            assert _relerr(g, rg) < (5e-2 if loose(n) else 2e-3), (n, _relerr(g, rg))
            l2 = float(np.linalg.norm(g - rg) / np.linalg.norm(rg))
            assert l2 < (2e-2 if loose(n) else 3e-3), (n, l2)
    for n in ("roi_heads.box_head.fc2.weight", "roi_heads.box_refinery_0.cls_score.weight"):
        assert _relerr(sd[n].grad.detach().cpu().numpy(), ref_grads[n].numpy()) < 2e-3, n
    opt.step()
    torch.cuda.synchronize()
    for n in trunk:
        delta, ref_delta = (sd[n].detach().cpu() - before[n]).numpy(), (p[n] - before[n]).numpy()
        if np.abs(ref_delta).max() > 1e-9:
            assert _relerr(delta, ref_delta) < (5e-2 if loose(n) else 2e-3), n
    # second step on the updated weights: packed conv copies must have been refreshed
    ref2, _ = O.train_step(p, batch, ocfg, opt_o, freeze_at=freeze_at)
    opt.zero_grad()
    losses2 = model(G.drn_inputs(batch))
    sum(losses2.values()).backward()
    opt.step()
    for k, v in losses2.items():
        assert abs(float(v.detach()) - ref2[k]) <= 2e-2 * max(abs(ref2[k]), 1e-3), (k, float(v.detach()), ref2[k])
    ocfg.dropout = 0.5


def test_unfrozen_backbone_bf16_and_pipelined_guard():
    """bf16 mode trains the trunk too (finite gradients, loss close to fp32); the pipelined optimizer refuses it"""
    from drn_wsod_pytorch_amd._cabi import DrnError
    from drn_wsod_pytorch_amd.engine import build_optimizer

    name = "model_r50c4_tiny"
    ocfg, d = G.MODEL_CASES[name], G.load(name)
    cfg, model = G.drn_model(ocfg, int(d["seed"]), "cuda", 2, "bf16")
    model.roi_heads.box_head.dropout_p = 0.0
    model.train()
    opt = build_optimizer(cfg, model)
    with pytest.raises(DrnError):
        opt.enable_pipelined(None)
    opt.zero_grad()
    losses = model(G.drn_inputs(G.batch_from(d)))
    sum(losses.values()).backward()
    for k, v in losses.items():
        ref = float(d["step0_%s" % k])
        assert abs(float(v.detach()) - ref) <= 3e-2 * max(abs(ref), 1e-2), (k, float(v.detach()), ref)
    for n, q in model.named_parameters():
        if q.requires_grad and q.grad is not None:
            assert torch.isfinite(q.grad).all(), n
    opt.step()
    load_package().set_precision("fp32")


@pytest.mark.parametrize("name", ["model_r50c4_tiny", "model_vgg16_small"])
def test_train_step_bf16(name):
    """bf16 fast mode on the same fixture: loss agreement within 3e-2 relative (bf16 operands), finite grads."""
    ocfg, d, cfg, model = _setup(name, "bf16")
    model.train()
    losses = model(G.drn_inputs(G.batch_from(d)))
    sum(losses.values()).backward()
    for k, v in losses.items():
        ref = float(d["step0_%s" % k])
        assert abs(float(v.detach()) - ref) <= 3e-2 * max(abs(ref), 1e-2), (k, float(v.detach()), ref)
    for n, p in model.named_parameters():
        if p.requires_grad and p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
    load_package().set_precision("fp32")


@pytest.mark.parametrize("case", ["tiny", "full"])
def test_fc6_weight_gradient_tn_equals_nt(case):
    """bf16 mode: the fc6 weight gradient read from the pooled matrix A itself (drn_gemm_tn; the pooling launch writes
    only the tail rows of A^T) against the NT form on the fully materialised A^T: every gradient bit for bit - with the
    benchmark's two row slabs + joint peel at full size (R50-C4, R = 2000), and on the tiny fixture (no peel at all)."""
    from drn_wsod_pytorch_amd.engine import build_optimizer

    if case == "tiny":
        ocfg, d, cfg, model = _setup("model_r50c4_tiny", "bf16")
        batch = G.drn_inputs(G.batch_from(d))
    else:
        kw = dict(arch="wsr50", out_feature="res4", res5_dilation=1, num_classes=20)
        ocfg = O.OracleCfg(dropout=0.0, base_lr=2e-4, **kw)
        cfg, model = G.drn_model(ocfg, 3, "cuda", 5, "bf16")
        model.roi_heads.box_head.dropout_p = 0.0
        b = O.synthetic_batch(1, 2000, ocfg, seed=77)
        batch = G.drn_inputs([dict(x, gt_boxes=torch.zeros(len(x["gt_classes"]), 4)) for x in b])
    model.train()
    eng = model.roi_heads._engine
    grads = {}
    for tn in (True, False):
        eng.fc1_tn = tn
        eng.fc1_grad_slabs = 2 if case == "full" else 1
        for p_ in model.parameters():
            p_.grad = None
        eng._grads_valid = False
        sum(model(batch).values()).backward()
        torch.cuda.synchronize()
        if tn and case == "full":
            assert eng._last_state["w"]["AT_row0"] > 40000  # the pooling launch really skipped most of A^T
        grads[tn] = {n: p_.grad.detach().clone() for n, p_ in model.named_parameters() if p_.grad is not None}
    eng.fc1_tn = True
    assert "roi_heads.box_head.fc1.weight" in grads[True]
    for n in grads[True]:
        assert torch.equal(grads[True][n], grads[False][n]), n
    load_package().set_precision("fp32")


@pytest.mark.parametrize("R", [2000, 1361])
def test_fc6_fused_tn_step_equals_unfused(R):
    """(R = 1361: fewer than 32 K slabs - the chunks of a tile's update that find no slab follow the mainloop.)  Round 4: `FusedSGD.enable_fused_fc1_tn()` - the fc6 weight gradient's main columns and their optimizer step in ONE launch
    (drn_gemm_tn_sgd: every tile's update inside the next tile's mainloop) - against the default pipelined step (two row slabs +
    sgd_kernel on the optimizer stream) at the bench shape: three SGD steps, weight / momentum / bf16-shadow arenas bit for bit."""
    from drn_wsod_pytorch_amd.engine import build_optimizer

    kw = dict(arch="wsr50", out_feature="res4", res5_dilation=1, num_classes=20)
    ocfg = O.OracleCfg(dropout=0.0, base_lr=2e-4, **kw)
    b = O.synthetic_batch(1, R, ocfg, seed=79)
    batch = G.drn_inputs([dict(x, gt_boxes=torch.zeros(len(x["gt_classes"]), 4)) for x in b])
    res = []
    for fused in (False, True):
        cfg, model = G.drn_model(ocfg, 3, "cuda", 5, "bf16")
        model.roi_heads.box_head.dropout_p = 0.0
        model.train()
        opt = build_optimizer(cfg, model)
        opt.enable_pipelined(None, fused_tn=fused)
        eng = model.roi_heads._engine
        from drn_wsod_pytorch_amd import ops

        ops.GEMM_TIMING = []
        for _ in range(3):
            opt.zero_grad()
            sum(model(batch).values()).backward()
            opt.step()
        torch.cuda.synchronize()
        kinds = {t[3][0] for t in ops.GEMM_TIMING}
        ops.GEMM_TIMING = None
        assert ("tn_sgd" in kinds) == fused  # the fused launch really ran (and only then)
        res.append(dict(w=eng.arena_w.clone(), m=opt._mom.clone(), s=eng.arena_s.clone()))
        del model, opt
    for k in ("w", "m", "s"):
        assert torch.equal(res[0][k], res[1][k]), k
    load_package().set_precision("fp32")


@pytest.mark.parametrize("n_img", [1, 2])
def test_fused_loss_tail_step_equals_separate_calls(n_img):
    """`_HeadEngine.fused_loss_tail` (default): the predictor's split-K reduce + bias, WSDDN and the refinement cascade as
    drn_mil_oicr_losses (six launches) against the nine separate launches: three SGD steps with dropout (the counter-based
    masks must stay in step too), every loss and the weight / momentum arenas bit for bit."""
    from drn_wsod_pytorch_amd.engine import build_optimizer

    kw = dict(arch="wsr50", out_feature="res4", res5_dilation=1, num_classes=20)
    ocfg = O.OracleCfg(dropout=0.5, base_lr=2e-4, **kw)
    b = O.synthetic_batch(n_img, 500, ocfg, seed=83)
    batch = G.drn_inputs([dict(x, gt_boxes=torch.zeros(len(x["gt_classes"]), 4)) for x in b])
    res = []
    for fused in (False, True):
        torch.manual_seed(11)
        cfg, model = G.drn_model(ocfg, 3, "cuda", 5, "bf16")
        model.train()
        opt = build_optimizer(cfg, model)
        eng = model.roi_heads._engine
        eng.fused_loss_tail = fused
        losses = []
        for _ in range(3):
            opt.zero_grad()
            out = model(batch)
            sum(out.values()).backward()
            opt.step()
            losses.append({k: float(v.detach()) for k, v in out.items()})
        torch.cuda.synchronize()
        res.append(dict(w=eng.arena_w.clone(), m=opt._mom.clone(), losses=losses, ctr=int(eng.seed_dev)))
        del model, opt
    assert res[0]["losses"] == res[1]["losses"], (res[0]["losses"], res[1]["losses"])
    assert res[0]["ctr"] == res[1]["ctr"] != 0
    for k in ("w", "m"):
        assert torch.equal(res[0][k], res[1][k]), k
    load_package().set_precision("fp32")


def test_grad_accumulation_iter_size():
    """WSL.ITER_SIZE semantics (train_net.py:100-113): two backward() calls accumulate before one step."""
    ocfg, d, cfg, model = _setup("model_r50c4_tiny", "fp32")
    batch = G.drn_inputs(G.batch_from(d))
    model.train()
    sum(model(batch).values()).backward()
    g1 = model.roi_heads.box_head.fc1.weight.grad.clone()
    b1 = model.roi_heads.box_refinery_0.cls_score.bias.grad.clone()
    (sum(model(batch).values()) * 0.5).backward()
    assert torch.allclose(model.roi_heads.box_head.fc1.weight.grad, 1.5 * g1, rtol=1e-4, atol=1e-7)
    assert torch.allclose(model.roi_heads.box_refinery_0.cls_score.bias.grad, 1.5 * b1, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("lookahead", [1, 2, 3, "pairs", "group3", "group4"])
def test_hipgraph_step_equals_eager(lookahead):
    """GraphedTrainStep (whole step captured into a hipGraph, next image's backbone forked onto a side stream) must
    reproduce the eager trainer step for step: same losses over the steps on a cycle of three different batches (with
    lookahead=2 the trunk runs two batches ahead into alternating feature buffers)."""
    from drn_wsod_pytorch_amd.engine import GraphedTrainStep, build_optimizer

    name = "model_r50c4_tiny"
    d = G.load(name)
    ocfg = G.MODEL_CASES[name]
    base = G.batch_from(d)
    # two different batches with identical shapes (graphs need static shapes): swap / perturb the fixture's images
    b0 = G.drn_inputs([base[0]])
    alt = dict(base[0])
    alt["image"] = (255.0 - base[0]["image"]).contiguous()
    alt["objectness_logits"] = base[0]["objectness_logits"].flip(0).contiguous()
    b1 = G.drn_inputs([alt])
    alt2 = dict(base[0])
    alt2["image"] = base[0]["image"].flip(2).contiguous()
    alt2["proposal_boxes"] = base[0]["proposal_boxes"].flip(0).contiguous()
    alt2["gt_classes"] = (base[0]["gt_classes"] + 1) % ocfg.num_classes
    b2 = G.drn_inputs([alt2])
    seq = [b0, b1, b2, b0, b1, b1, b0, b2, b1, b0, b2, b2, b0, b1, b0, b2]  # not periodic in 2, 3 or 4: a wrong slot shows
    results = []
    for graphed in (False, True):
        cfg, model = G.drn_model(ocfg, int(d["seed"]), "cuda", 5, "fp32")
        model.roi_heads.box_head.dropout_p = 0.0
        model.train()
        opt = build_optimizer(cfg, model)
        out = []
        if graphed:
            # one conv chain per two batches (t+2, t+3), or per group of G (round 4: batches t+G .. t+2G-1)
            grp = {"pairs": 2, "group3": 3, "group4": 4}.get(lookahead, 0)
            pairs = grp > 0
            stepper = GraphedTrainStep(model, opt, seq[0], lookahead=1 if pairs else lookahead,
                                       trunk_pairs=(True if grp == 2 else grp) if pairs else False)
            for i in range(6):
                losses = stepper.step(*seq[i: i + (2 * grp if pairs else max(lookahead, 2) + 1)])
                out.append({k: float(v.detach()) for k, v in losses.items()})
        else:
            for i in range(6):
                opt.zero_grad()
                losses = model(seq[i])
                sum(losses.values()).backward()
                opt.step()
                out.append({k: float(v.detach()) for k, v in losses.items()})
        results.append(out)
    for e, g in zip(*results):
        for k in e:
            assert abs(e[k] - g[k]) <= 1e-5 * max(abs(e[k]), 1e-3), (k, e[k], g[k])


@pytest.mark.parametrize("group", [2, 4])
def test_ring_schedule_equals_waiting_schedule(group):
    """Round 6: under the ring schedule (GraphedTrainStep(ring=True), bench.py's default) the side stream never waits for the
    main stream - RING_SETS staging sets and RING_SLOTS trunk-feature slots, and the host stays at most RING_LAG steps ahead of the
    heads.  Thirty steps enqueued WITHOUT a sync (the host runs ahead as far as the schedule lets it) over a batch sequence that is
    not periodic in 2, 3 or 4 must give the losses of the waiting schedule bit for bit, step by step: a set or slot overwritten
    before its reader ran pairs a step with another batch's proposals / labels / features.  (This tiny model is host-bound - it pins
    the slot / set arithmetic for groups of 2 and 4; the GPU-bound case is tests/test_bench_mode_gpu.py::test_ring_schedule_full_size_*.)"""
    from drn_wsod_pytorch_amd.engine import GraphedTrainStep, build_optimizer

    name = "model_r50c4_tiny"
    d = G.load(name)
    ocfg = G.MODEL_CASES[name]
    base = G.batch_from(d)
    b0 = G.drn_inputs([base[0]])
    alt = dict(base[0])
    alt["image"] = (255.0 - base[0]["image"]).contiguous()
    alt["objectness_logits"] = base[0]["objectness_logits"].flip(0).contiguous()
    b1 = G.drn_inputs([alt])
    alt2 = dict(base[0])
    alt2["image"] = base[0]["image"].flip(2).contiguous()
    alt2["proposal_boxes"] = base[0]["proposal_boxes"].flip(0).contiguous()
    alt2["gt_classes"] = (base[0]["gt_classes"] + 1) % ocfg.num_classes
    b2 = G.drn_inputs([alt2])
    pat = [b0, b1, b2, b0, b1, b1, b0, b2, b1, b0, b2, b2, b0, b1, b0, b2, b2, b1, b1, b0]
    steps = 30
    seq = [pat[(i * 7 + i // 5) % len(pat)] for i in range(steps + 2 * group)]
    results = []
    for ring in (False, True):
        cfg, model = G.drn_model(ocfg, int(d["seed"]), "cuda", 5, "fp32")
        model.roi_heads.box_head.dropout_p = 0.0
        model.train()
        opt = build_optimizer(cfg, model)
        opt.enable_pipelined()
        stepper = GraphedTrainStep(model, opt, seq[0], split_tail=True, trunk_pairs=(True if group == 2 else group), eager_fc6=True, ring=ring)
        out = []
        for i in range(steps):
            losses = stepper.step(*seq[i: i + 2 * group])
            out.append(torch.stack([losses[k].detach().clone().reshape(()) for k in sorted(losses)]))  # (no sync: a device copy)
        assert stepper._ring_on == ring
        torch.cuda.synchronize()
        results.append(torch.stack(out).cpu())
        stepper.release()
    assert torch.isfinite(results[0]).all()
    assert len({tuple(r.tolist()) for r in results[0]}) > steps // 2  # the sequence really changes the losses from step to step
    assert torch.equal(results[0], results[1])


@pytest.mark.parametrize("comm,lookahead", [("fp32", 1), ("bf16", 1), ("fp32", 2), ("fp32", "pairs")])
def test_split_tail_exchange_step_equals_eager(comm, lookahead):
    """The N>1 step on one GPU: a 1-rank RCCL group with the exchange forced on, GraphedTrainStep(split_tail=True)
    (captured heads graph + eager fc6-dW / all-reduce / SGD tail on the optimizer stream).  With fp32 buckets it must
    reproduce the plain eager trainer exactly (weights bit for bit after 4 steps); with bf16 fc6 buckets the
    gradient is rounded once to bf16 before the update, so weights agree to bf16 resolution of one lr-scaled step."""
    import socket

    import torch.distributed as dist
    from drn_wsod_pytorch_amd.engine import DataParallel, GraphedTrainStep, build_optimizer

    name = "model_r50c4_tiny"
    d = G.load(name)
    ocfg = G.MODEL_CASES[name]
    base = G.batch_from(d)
    b0 = G.drn_inputs([base[0]])
    alt = dict(base[0])
    alt["image"] = (255.0 - base[0]["image"]).contiguous()
    alt["objectness_logits"] = base[0]["objectness_logits"].flip(0).contiguous()
    b1 = G.drn_inputs([alt])
    seq = [b0, b1, b0, b1, b0, b1, b0, b1]
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port)
    try:
        res = []
        for mode in ("eager", "split"):
            cfg, model = G.drn_model(ocfg, int(d["seed"]), "cuda", 5, "fp32")
            model.roi_heads.box_head.dropout_p = 0.0
            model.train()
            opt = build_optimizer(cfg, model)
            out = []
            if mode == "split":
                dp = DataParallel(model, force_exchange=True)
                assert dp.exchange and dp.world == 1
                dp.broadcast_parameters(0)
                opt.enable_pipelined(dp, slab_rows=[16, 48],
                                     comm_dtype=torch.bfloat16 if comm == "bf16" else torch.float32)
                assert (model.roi_heads._engine.fc1_grad_bucket is not None) == (comm == "bf16")
                pairs = lookahead == "pairs"
                stepper = GraphedTrainStep(model, opt, seq[0], split_tail=True, lookahead=1 if pairs else lookahead,
                                           trunk_pairs=pairs)
                for i in range(4):
                    losses = stepper.step(*seq[i: i + (4 if pairs else 3)])
                    out.append({k: float(v.detach()) for k, v in losses.items()})
            else:
                for i in range(4):
                    opt.zero_grad()
                    losses = model(seq[i])
                    sum(losses.values()).backward()
                    opt.step()
                    out.append({k: float(v.detach()) for k, v in losses.items()})
            torch.cuda.synchronize()
            res.append((out, {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}))
    finally:
        dist.destroy_process_group()
    (le, pe), (ls, ps) = res
    if os.environ.get("DRN_TEST_DEBUG"):
        for i, (e, g) in enumerate(zip(le, ls)):
            print("step", i, {k: (round(e[k], 5), round(g[k], 5)) for k in e})
        for n in pe:
            print(n, float((pe[n] - ps[n]).abs().max()), float(pe[n].abs().max()))
    # bf16 wire: every gradient is rounded once to bf16 (2^-9 relative) before the update.  This case trains at a step
    # size where the losses jump by 10x between steps, so the rounding is amplified step over step (measured: 4e-5,
    # 4e-4, 4e-2 relative on the losses of steps 1..3, a pseudo-GT flip in the last one): tight on the first steps,
    # loose on the last, weights to a few bf16 ulps of the accumulated update
    for i, (e, g) in enumerate(zip(le, ls)):
        for k in e:
            tol = 1e-5 if comm == "fp32" else (2e-3 if i < 3 else 1e-1)
            assert abs(e[k] - g[k]) <= tol * max(abs(e[k]), 1e-3), (i, k, e[k], g[k])
    for n in pe:
        if comm == "fp32":
            assert torch.equal(pe[n], ps[n]), n
        else:
            assert torch.allclose(pe[n], ps[n], rtol=0, atol=5e-3), (n, float((pe[n] - ps[n]).abs().max()))


def test_pipelined_sgd_equals_plain():
    """FusedSGD.enable_pipelined (per-bucket update on a second stream during backward) == plain step()."""
    from drn_wsod_pytorch_amd.engine import build_optimizer

    name = "model_r50c4_tiny"
    d = G.load(name)
    ocfg = G.MODEL_CASES[name]
    batch = G.drn_inputs(G.batch_from(d))
    params = []
    for pipelined in (False, True):
        cfg, model = G.drn_model(ocfg, int(d["seed"]), "cuda", 5, "fp32")
        model.roi_heads.box_head.dropout_p = 0.0
        model.train()
        opt = build_optimizer(cfg, model)
        if pipelined:
            opt.enable_pipelined(None, slab_rows=[16, 48])
        for _ in range(3):
            opt.zero_grad()
            sum(model(batch).values()).backward()
            opt.step()
        torch.cuda.synchronize()
        params.append({n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad})
    for n in params[0]:
        assert torch.equal(params[0][n], params[1][n]), n


# ---------------------------------------------------------------------------------------------------------------
# Full-size parity at BASELINE.json's own shapes (the tiny goldens above pin the semantics against the reference;
# these pin the HIP path against the oracle at the sizes bench.py runs, where tiling / split-K / LDS paths differ).
FULL_CASES = {
    # configs[1]: WS-R50-C4, 224x224, R=2000, K=20 (the bench workload)
    "r50c4_r2000_k20": (dict(arch="wsr50", out_feature="res4", res5_dilation=1, num_classes=20), 2000),
    # configs[3] shape: WS-R101-C4, COCO-shaped K=80
    "r101c4_r2000_k80": (dict(arch="wsr101", out_feature="res4", res5_dilation=1, num_classes=80), 2000),
    # configs[2] shape: WS-R50-DilatedC5, R=4000 (27x27x2048 map: the window-staged ROIPool path, fc6 K = 100352)
    "r50dc5_r4000_k20": (dict(arch="wsr50", out_feature="res5", res5_dilation=2, num_classes=20), 4000),
    # configs[0] at full size (round 3): VGG16 with dilated conv5 (28x28x512 map), fc6 [4096 x 25088], fc7 4096 -> 4096 - the
    # N = 4096 GEMM shapes of oicr_V_16_DC5_1x.yaml, which only the small V16 golden covered before
    "vgg16_r2000_k20": (dict(arch="vgg16", out_feature="plain5", res5_dilation=2, num_classes=20, dan_dim=(4096, 4096),
                             pixel_mean=(103.939, 116.779, 123.68), base_lr=0.001), 2000),
    # PCLROIHeads on the bench trunk: proposal clustering of 2000 boxes on the device
    "pcl_r50c4_r2000_k20": (dict(arch="wsr50", out_feature="res4", res5_dilation=1, num_classes=20, heads="pcl"), 2000),
}


# measured: the L1 norm of the full fc6 weight gradient (100-200 M entries) is within 9.7e-6 of the oracle's on every case,
# while single entries differ by up to 2.3e-3 of the largest one (RoIPool arg-max flips, below)
FULL_FC1_L1_TOL = 1e-4


@pytest.mark.parametrize("case", list(FULL_CASES))
def test_full_size_train_step_matches_oracle_fp32(case):
    """One full-size train step (fwd + bwd + SGD) in the fp32 parity mode vs the CPU oracle on the same seeded
    weights and SURVEY 8(d) synthetic inputs: every loss within 1e-4 relative (north-star bound), the image-level MIL
    scores, and the SGD update of the largest and the smallest trainable tensors."""
    from drn_wsod_pytorch_amd.engine import build_optimizer

    kw, R = FULL_CASES[case]
    torch.set_num_threads(min(32, os.cpu_count() or 1))  # the oracle's fastest setting on the GPU box's host
    ocfg = O.OracleCfg(dropout=0.0, **kw)
    p = O.init_params(ocfg, seed=3)
    batch = O.synthetic_batch(1, R, ocfg, seed=4321)
    opt_o = O.SGDState(ocfg)
    names = ["roi_heads.box_head.fc1.weight", "roi_heads.box_head.fc2.bias", "roi_heads.box_refinery_2.cls_score.weight",
             "roi_heads.box_predictor.det.weight"]
    before = {n: p[n].clone() for n in names}
    ref_losses, ref_grads = O.train_step(p, batch, ocfg, opt_o)

    cfg, model = G.drn_model(ocfg, 3, "cuda", 5, "fp32")
    model.roi_heads.box_head.dropout_p = 0.0
    model.train()
    opt = build_optimizer(cfg, model)
    opt.zero_grad()
    losses = model(G.drn_inputs([dict(b, gt_boxes=torch.zeros(len(b["gt_classes"]), 4)) for b in batch]))
    sum(losses.values()).backward()
    got = {k: float(v.detach()) for k, v in losses.items()}
    assert set(got) == set(ref_losses)
    for k in got:
        assert abs(got[k] - ref_losses[k]) <= 1e-4 * max(abs(ref_losses[k]), 1e-3), (k, got[k], ref_losses[k])
    sd = dict(model.named_parameters())
    for n in names:
        g, rg = sd[n].grad.detach().cpu(), ref_grads[n]
        if n.endswith("fc1.weight"):  # 100-200 M entries: compare a strided sample and the L1 norm
            l1, rl1 = float(g.double().abs().sum()), float(rg.double().abs().sum())
            _record("full_size.%s.fc1_l1" % case, abs(l1 - rl1) / rl1)
            assert abs(l1 - rl1) < FULL_FC1_L1_TOL * rl1, (case, l1, rl1)
            g, rg = g.reshape(-1)[::4099], rg.reshape(-1)[::4099]
        _record("full_size.%s.grad.%s" % (case, n), _relerr(g.numpy(), rg.numpy()))
        # (4e-3: fc1.weight's gradient holds single pooled activations - a RoIPool window whose two largest values differ
        # by less than the trunk's fp32 rounding picks the other one; 1.4e-3 .. 2.3e-3 depending on which conv kernel
        # (summation order) served the res4 layers, with all losses within 1e-4)
        assert _relerr(g.numpy(), rg.numpy()) < 4e-3, n
    opt.step()
    torch.cuda.synchronize()
    for n in names:
        new, ref_new = sd[n].detach().cpu(), p[n]
        delta, ref_delta = (new - before[n]).reshape(-1)[::4099 if new.numel() > 10 ** 7 else 1], \
            (ref_new - before[n]).reshape(-1)[::4099 if new.numel() > 10 ** 7 else 1]
        assert _relerr(delta.numpy(), ref_delta.numpy()) < 4e-3, n
    load_package().set_precision("fp32")


def test_pcl_heads_two_steps_vs_oracle_fp32():
    """PCLROIHeads (roi_heads_pcl.py) end to end on the reference-generated fixture's inputs: two training steps of the
    product against the oracle (whose PCL flow is pinned to the reference by tests/test_oracle_golden.py): losses,
    every trainable gradient, the updated weights; then the clusters of every branch, index for index."""
    from drn_wsod_pytorch_amd.engine import build_optimizer

    name = "model_pcl_r50c4_tiny"
    ocfg, d, cfg, model = _setup(name, "fp32")
    ocfg.dropout = 0.0
    assert type(model.roi_heads).__name__ == "PCLROIHeads"
    batch = G.batch_from(d)
    p = O.seeded_params(O.param_shapes(ocfg), int(d["seed"]))
    opt_o = O.SGDState(ocfg)
    model.train()
    opt = build_optimizer(cfg, model)
    for step in range(2):
        if step == 0:
            _, aux = O.model_train_losses({k: v.clone() for k, v in p.items()}, batch, ocfg, None, True)
        ref_losses, ref_grads = O.train_step(p, batch, ocfg, opt_o)
        opt.zero_grad()
        losses = model(G.drn_inputs(batch))
        sum(losses.values()).backward()
        got = {k: float(v.detach()) for k, v in losses.items()}
        assert set(got) == set(ref_losses)
        for k in got:
            tol = 1e-4 if step == 0 else 2e-2
            _record("pcl_two_steps.step%d.%s" % (step, k), abs(got[k] - float(ref_losses[k])) / max(abs(float(ref_losses[k])), 1e-3))
            assert abs(got[k] - float(ref_losses[k])) <= tol * max(abs(float(ref_losses[k])), 1e-3), (step, k, got[k])
        if step == 0:
            tg = model.roi_heads._last_state["aux"]["targets"]
            for k in range(ocfg.refine_num):
                t = aux["pcl"][k]
                n = int(tg[k]["n_pc"].item())
                assert n == len(t["pc_labels"])
                assert np.array_equal(tg[k]["labels"].cpu().numpy(), t["labels"]), k
                assert np.array_equal(tg[k]["gt_assignment"].cpu().numpy(), t["gt_assignment"]), k
                assert np.array_equal(tg[k]["pc_rows"].cpu().numpy()[:n], t["centre_rows"]), k
                assert np.array_equal(tg[k]["pc_count"].cpu().numpy()[:n], t["pc_count"]), k
            for n_, prm in model.named_parameters():
                if prm.requires_grad and n_ in ref_grads:
                    rg = ref_grads[n_].numpy()
                    g = prm.grad.detach().cpu().numpy()
                    if np.abs(rg).max() < 1e-6:
                        assert np.abs(g).max() < 1e-5, n_
                    else:
                        assert _relerr(g, rg) < 2e-3, n_
        opt.step()
    torch.cuda.synchronize()
    for n_, prm in model.named_parameters():
        if prm.requires_grad and n_ in p and "bbox_pred" not in n_:
            assert _relerr(prm.detach().cpu().numpy(), p[n_].numpy()) < 5e-3, n_


def test_pcl_heads_reject_batches_and_graph_capture_ok():
    """PCL clusters one image per step (the reference asserts it): a 2-image batch fails loudly; the three PCL launches
    are capture-safe (no host round trip), so the hipGraph step equals the eager one"""
    from drn_wsod_pytorch_amd._cabi import DrnError
    from drn_wsod_pytorch_amd.engine import GraphedTrainStep, build_optimizer

    name = "model_pcl_r50c4_tiny"
    ocfg, d, cfg, model = _setup(name, "fp32")
    batch = G.drn_inputs(G.batch_from(d))
    model.train()
    with pytest.raises(DrnError):
        model(batch + batch)
    res = []
    for mode in ("eager", "graph"):
        ocfg, d, cfg, model = _setup(name, "fp32")
        model.train()
        opt = build_optimizer(cfg, model)
        out = []
        if mode == "graph":
            stepper = GraphedTrainStep(model, opt, batch)
            for _ in range(3):
                out.append({k: float(v.detach()) for k, v in stepper.step(batch, batch).items()})
        else:
            for _ in range(3):
                opt.zero_grad()
                losses = model(batch)
                sum(losses.values()).backward()
                opt.step()
                out.append({k: float(v.detach()) for k, v in losses.items()})
        torch.cuda.synchronize()
        res.append((out, {n: prm.detach().clone() for n, prm in model.named_parameters() if prm.requires_grad}))
    for a, b in zip(res[0][0], res[1][0]):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-6 * max(1.0, abs(a[k])), (k, a[k], b[k])
    for n_ in res[0][1]:
        assert torch.allclose(res[0][1][n_], res[1][1][n_], rtol=0, atol=1e-6), n_


def test_tta_matches_reference_golden():
    """GeneralizedRCNNWithTTAAVG (2 scales x flip) vs the golden produced by the reference's own class: the mapper's
    augmented images / proposals bit for bit, the device-side back-transformed box and score averages
    (drn_tta_accumulate), and the final detections (classes exact, scores / boxes within fp32 tolerance)."""
    from drn_wsod_pytorch_amd.modeling import GeneralizedRCNNWithTTAAVG
    from drn_wsod_pytorch_amd.structures import Boxes, Instances

    d = G.load("tta_r50c4_tiny")
    ocfg = G.MODEL_CASES["model_r50c4_tiny"]
    cfg, model = G.drn_model(ocfg, int(d["seed"]), "cuda", 5, "fp32")
    cfg.merge_from_list(["TEST.AUG.ENABLED", "True", "TEST.AUG.MIN_SIZES", str(tuple(int(x) for x in d["min_sizes"])),
                         "TEST.AUG.MAX_SIZE", str(int(d["max_size"])), "TEST.AUG.FLIP", "True",
                         "DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST", str(int(d["topk"]))])
    model.eval()
    img = torch.from_numpy(d["image_u8"])
    H, W = img.shape[1:]
    prop = Instances((H, W))
    prop.proposal_boxes = Boxes(torch.from_numpy(d["proposal_boxes"]))
    prop.objectness_logits = torch.from_numpy(d["objectness_logits"])
    inp = {"image": img, "proposals": prop, "height": H, "width": W}
    tta = GeneralizedRCNNWithTTAAVG(cfg, model)
    assert tta.tta_mapper.device is not None  # (round 5) resize + flip on the device: the same BYTES as the reference's PIL images
    from drn_wsod_pytorch_amd.modeling.tta import DatasetMapperTTAAVG

    for mapper in (tta.tta_mapper, DatasetMapperTTAAVG(cfg)):  # device path (default), host path
        augs = mapper(inp)
        assert len(augs) == int(d["n_aug"])
        for i, a in enumerate(augs):
            assert a["image"].is_cuda == (mapper.device is not None)
            assert np.array_equal(a["image"].cpu().numpy().astype(np.float32), d["aug%d_image" % i].astype(np.float32)), i
            # (device mapper: the proposals of all augmentations were uploaded in one pinned copy and are device views)
            assert a["proposals"].proposal_boxes.tensor.is_cuda == (mapper.device is not None)
            assert np.array_equal(a["proposals"].proposal_boxes.tensor.cpu().numpy(), d["aug%d_boxes" % i]), i
            assert np.array_equal(a["proposals"].objectness_logits.cpu().numpy(), d["aug%d_obj" % i]), i
    augs = tta.tta_mapper(inp)
    with torch.no_grad():
        avg_b, avg_s = tta._get_augmented_boxes(augs)
    assert np.allclose(avg_s.cpu().numpy(), d["avg_scores"], rtol=1e-4, atol=1e-6)
    assert np.allclose(avg_b.cpu().numpy(), d["avg_boxes"], rtol=1e-5, atol=1e-3)
    out = tta([inp])[0]["instances"]
    assert np.array_equal(out.pred_classes.cpu().numpy(), d["det_classes"])
    assert np.allclose(out.scores.cpu().numpy(), d["det_scores"], rtol=1e-4, atol=1e-6)
    assert np.allclose(out.pred_boxes.tensor.cpu().numpy(), d["det_boxes"], rtol=1e-5, atol=1e-3)
    # batch_size = 2 (test_time_augmentation_avg.py:200-225): each size's plain and flipped image in ONE model.inference - the same
    # averages and detections within the same bounds (the convs of a 2-image batch may take other tiles: not bit for bit)
    tta2 = GeneralizedRCNNWithTTAAVG(cfg, model, batch_size=2)
    with torch.no_grad():
        avg_b2, avg_s2 = tta2._get_augmented_boxes(tta2.tta_mapper(inp))
    assert np.allclose(avg_s2.cpu().numpy(), d["avg_scores"], rtol=1e-4, atol=1e-6)
    assert np.allclose(avg_b2.cpu().numpy(), d["avg_boxes"], rtol=1e-5, atol=1e-3)
    out2 = tta2([inp])[0]["instances"]
    assert np.array_equal(out2.pred_classes.cpu().numpy(), d["det_classes"])
    assert np.allclose(out2.scores.cpu().numpy(), d["det_scores"], rtol=1e-4, atol=1e-6)
    load_package().set_precision("fp32")


def test_tta_mapper_vectorised_equals_per_augmentation():
    """Round 5: the device mapper transforms the proposals of all augmentations in one [A, R, 4] numpy sweep and uploads them in one
    pinned copy.  Against the per-augmentation mapper (the reference's transform_proposals order: scale, flip, corner bounding box,
    clip, drop empty, top-k) bit for bit - boxes that leave the image, zero-width / zero-height boxes, a top-k that cuts."""
    import drn_wsod_pytorch_amd.modeling.tta as tta_mod
    from drn_wsod_pytorch_amd.structures import Boxes, Instances

    cfg = G.drn_model(G.MODEL_CASES["model_r50c4_tiny"], 0, "cuda", 5, "fp32")[0]
    cfg.merge_from_list(["TEST.AUG.ENABLED", "True", "TEST.AUG.MIN_SIZES", "(96, 120, 176)", "TEST.AUG.MAX_SIZE", "200", "TEST.AUG.FLIP", "True",
                         "DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST", "700"])
    rng = np.random.default_rng(5)
    H, W, R = 75, 100, 900
    b = np.stack([rng.uniform(-8, W, R), rng.uniform(-8, H, R), rng.uniform(0, W + 12, R), rng.uniform(0, H + 12, R)], 1).astype(np.float32)
    b[:40, 2] = b[:40, 0]  # zero width
    b[40:70, 3] = b[40:70, 1]  # zero height
    b[70:90] = np.array([W + 5, 3, W + 9, 9], np.float32)  # outside: empty after the clip
    prop = Instances((H, W))
    prop.proposal_boxes = Boxes(torch.from_numpy(b))
    prop.objectness_logits = torch.from_numpy(rng.standard_normal(R).astype(np.float32))
    inp = {"image": torch.from_numpy(rng.integers(0, 256, (3, H, W), dtype=np.uint8)), "proposals": prop, "height": H, "width": W}
    mapper = tta_mod.DatasetMapperTTAAVG(cfg, device=torch.device("cuda"))
    got = mapper(inp)
    tta_mod.VECTORISED_MAPPER = False
    try:
        ref = mapper(inp)
    finally:
        tta_mod.VECTORISED_MAPPER = True
    assert len(got) == len(ref) == 6
    kept = set()
    for a, r in zip(got, ref):
        assert a["tta"] == r["tta"] and a["proposals"].image_size == r["proposals"].image_size
        assert torch.equal(a["image"], r["image"])
        assert a["proposals"].proposal_boxes.tensor.is_cuda
        assert np.array_equal(a["proposals"].proposal_boxes.tensor.cpu().numpy(), r["proposals"].proposal_boxes.tensor.cpu().numpy())
        assert np.array_equal(a["proposals"].objectness_logits.cpu().numpy(), r["proposals"].objectness_logits.cpu().numpy())
        kept.add(len(a["proposals"]))
    assert max(kept) == 700 or min(kept) < R  # the top-k or the empty-box filter really cut


def test_tta_accumulate_matches_host_transform():
    """drn_tta_accumulate vs the oracle's float32 numpy restatement of TransformList.inverse().apply_box + mean:
    bit-exact for the back-transformed boxes of one augmentation, and for the sequential mean of four"""
    from drn_wsod_pytorch_amd import ops

    rs = np.random.RandomState(9)
    R, K = 333, 7
    params = [(1.25, 0.8, -1.0), (1.25, 0.8, 67.0), (0.875, 0.8695652, -1.0), (0.875, 0.8695652, 96.0)]
    boxes = [torch.from_numpy((rs.rand(R, 4 * K) * 90).astype(np.float32)) for _ in params]
    scores = [torch.from_numpy(rs.rand(R, K + 1).astype(np.float32)) for _ in params]
    acc_b = torch.empty((R, 4 * K), device="cuda")
    acc_s = torch.empty((R, K + 1), device="cuda")
    refs = []
    for i, ((sx, sy, fw), b, s) in enumerate(zip(params, boxes, scores)):
        ops.tta_accumulate(b.cuda(), s.cuda(), acc_b, acc_s, sx, sy, fw, i == 0, len(params) if i == len(params) - 1 else 0)
        steps = ([("hflip", int(fw))] if fw >= 0 else []) + [("scale", sx, sy)]
        refs.append(torch.from_numpy(O.tta_apply_box(b.reshape(-1, 4).numpy(), steps)).reshape(R, 4 * K))
        if i == 0:
            assert torch.equal(acc_b.cpu(), refs[0])
    ref_b = torch.mean(torch.stack(refs), dim=0)
    ref_s = torch.mean(torch.stack(scores), dim=0)
    assert torch.allclose(acc_b.cpu(), ref_b, rtol=0, atol=1e-5) and torch.allclose(acc_s.cpu(), ref_s, rtol=0, atol=1e-7)


def test_checkpoint_load_into_live_cuda_model(tmp_path):
    """DetectionCheckpointer.load into a model that has already run (packed conv weights cached, head parameters living
    in the flat arena with bf16 shadows): the next inference must equal a model built directly on those weights."""
    from drn_wsod_pytorch_amd.checkpoint import DetectionCheckpointer

    name = "model_r50c4_tiny"
    d = G.load(name)
    ocfg = G.MODEL_CASES[name]
    batch = G.drn_inputs(G.batch_from(d), with_gt=False)
    _, src = G.drn_model(ocfg, 123, "cuda", 5, "bf16")
    src.eval()
    with torch.no_grad():
        ref, ref_scores, _ = src.inference(batch, do_postprocess=False)
    DetectionCheckpointer(src, str(tmp_path)).save("w")
    _, model = G.drn_model(ocfg, 7, "cuda", 5, "bf16")
    model.eval()
    with torch.no_grad():
        _, s0, _ = model.inference(batch, do_postprocess=False)  # caches packs / shadows of the OLD weights
    assert not torch.equal(s0[0], ref_scores[0])
    missing, unexpected = DetectionCheckpointer(model, str(tmp_path)).resume_or_load("", resume=True) or ([], [])
    with torch.no_grad():
        got, got_scores, _ = model.inference(batch, do_postprocess=False)
    for a, b in zip(got_scores, ref_scores):
        assert torch.equal(a, b)
    for a, b in zip(got, ref):
        assert torch.equal(a.pred_classes, b.pred_classes) and torch.equal(a.scores, b.scores)
    load_package().set_precision("fp32")


def test_dataset_mapper_output_trains(tmp_path):
    """the data path's output is the model's input: a record goes through load_proposals_into_dataset + DatasetMapper
    (train mode: crop / resize / flip / colour) straight into a train step on the GPU, and through the test-mode
    mapper into inference"""
    import pickle

    from drn_wsod_pytorch_amd import data as D
    from drn_wsod_pytorch_amd.engine import build_optimizer

    d = G.load("data_mapper")
    ocfg = G.MODEL_CASES["model_r50c4_tiny"]
    cfg, model = G.drn_model(ocfg, 5, "cuda", 5, "bf16")
    cfg.merge_from_list(["INPUT.MIN_SIZE_TRAIN", "(48, 64, 80)", "INPUT.MAX_SIZE_TRAIN", "120", "INPUT.MIN_SIZE_TEST", "64",
                         "INPUT.MAX_SIZE_TEST", "100", "INPUT.CROP.ENABLED", "True",
                         "DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TRAIN", "30", "DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST", "25"])
    pf = str(tmp_path / "props.pkl")
    with open(pf, "wb") as f:
        pickle.dump({"ids": [123], "boxes": [d["boxes"]], "objectness_logits": [d["scores"]]}, f)
    H, W = d["rgb"].shape[:2]
    rec = {"image_array": d["rgb"][:, :, ::-1].copy(), "height": H, "width": W, "image_id": 123,
           "annotations": [{"bbox": [10.0, 8.0, 50.0, 40.0], "bbox_mode": 0, "category_id": 3},
                           {"bbox": [30.5, 20.25, 80.0, 58.0], "bbox_mode": 0, "category_id": 1}]}
    recs = D.load_proposals_into_dataset([rec], pf)
    np.random.seed(3)
    batch = [D.DatasetMapper(cfg, True)(recs[0]) for _ in range(2)]
    assert batch[0]["image"].dtype == torch.uint8
    model.train()
    opt = build_optimizer(cfg, model)
    opt.zero_grad()
    losses = model(batch)
    sum(losses.values()).backward()
    opt.step()
    assert all(torch.isfinite(v).all() for v in losses.values()) and set(losses) >= {"loss_cls", "loss_cls_r0"}
    # the batch loader (TrainingSampler -> MapDataset(DatasetMapper) -> aspect-ratio buckets) yields the same kind of batch
    many = [dict(recs[0], image_id=123 + i) for i in range(5)]
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", "2", "DATALOADER.NUM_WORKERS", "0"])
    it = iter(D.build_detection_train_loader(cfg, many, rank=0, world_size=1))
    for _ in range(2):
        b = next(it)
        assert len(b) == 2 and all("proposals" in x and "instances" in x for x in b)
        opt.zero_grad()
        losses = model(b)
        sum(losses.values()).backward()
        opt.step()
        assert all(torch.isfinite(v).all() for v in losses.values())
    model.eval()
    out = model([D.DatasetMapper(cfg, False)(recs[0])])
    assert len(out) == 1 and len(out[0]["instances"]) <= cfg.TEST.DETECTIONS_PER_IMAGE
    load_package().set_precision("fp32")


def test_tta_outputs_feed_voc_evaluator():
    """detector (with TTA) -> PascalVOCDetectionEvaluator: CUDA Instances go straight into process(); evaluate() returns
    the reference's result layout (bbox AP/AP50/AP75 + CorLoc)"""
    from drn_wsod_pytorch_amd.evaluation import PascalVOCDetectionEvaluator
    from drn_wsod_pytorch_amd.modeling import GeneralizedRCNNWithTTAAVG
    from drn_wsod_pytorch_amd.structures import Boxes, Instances

    d = G.load("tta_r50c4_tiny")
    ocfg = G.MODEL_CASES["model_r50c4_tiny"]
    cfg, model = G.drn_model(ocfg, int(d["seed"]), "cuda", 5, "fp32")
    cfg.merge_from_list(["TEST.AUG.MIN_SIZES", "(48, 72)", "TEST.AUG.MAX_SIZE", "96", "TEST.AUG.FLIP", "True",
                         "DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST", "40"])
    model.eval()
    img = torch.from_numpy(d["image_u8"])
    H, W = img.shape[1:]
    prop = Instances((H, W))
    prop.proposal_boxes = Boxes(torch.from_numpy(d["proposal_boxes"]))
    prop.objectness_logits = torch.from_numpy(d["objectness_logits"])
    inp = {"image": img, "proposals": prop, "height": H, "width": W, "image_id": "000001"}
    out = GeneralizedRCNNWithTTAAVG(cfg, model)([inp])
    names = ["c%d" % i for i in range(ocfg.num_classes)]
    top = out[0]["instances"]
    k = int(top.pred_classes[0])
    gt_box = [int(v) for v in (top.pred_boxes.tensor[0].cpu() + torch.tensor([1.0, 1.0, 0.0, 0.0])).round().tolist()]
    ev = PascalVOCDetectionEvaluator(names, annotations={"000001": [(names[k], 0, gt_box)]}, year=2007)
    ev.process([inp], out)
    res = ev.evaluate()
    assert set(res["bbox"]) == {"AP", "AP50", "AP75"} and set(res["bbox CorLoc"]) == {"CL", "CL50", "CL75"}
    assert res["per_class"]["CL50"][names[k]] == 100.0  # the top detection of that class is the annotated object
    assert res["per_class"]["AP50"][names[k]] > 0
    load_package().set_precision("fp32")


# ---------------------------------------------------------------------------------------------------------------
# Ragged batches: the goldens hold images of different sizes but EQUAL proposal counts.  Real proposal files differ per
# image (SelectiveSearch / MCG after de-duplication: a few hundred to a few thousand), so the per-image segments of
# every [sum R_i, ...] operand - ROI batch indices, the per-image WSDDN softmax over proposals, pseudo-GT mining,
# label_and_sample - are exercised here with very different counts, including an image with a single proposal.
RAGGED = [((96, 128), 37, 11), ((128, 96), 1, 12), ((64, 80), 130, 13), ((72, 72), 8, 14)]


@pytest.mark.parametrize("name", ["model_r50c4_tiny", "model_r18dc5_tiny"])
def test_ragged_batch_matches_oracle_fp32(name):
    """4 images per step with 37 / 1 / 130 / 8 proposals and four different sizes: every loss within 1e-4 of the
    oracle's (north-star bound), image-level MIL scores, gradients of the heads' tensors, and a second step after one
    SGD update (weights that moved by ragged-batch gradients)."""
    from drn_wsod_pytorch_amd.engine import build_optimizer

    ocfg = copy.deepcopy(G.MODEL_CASES[name])
    ocfg.dropout = 0.0
    p = O.init_params(ocfg, seed=5)
    batch = []
    for (h, w), r, seed in RAGGED:
        batch += O.synthetic_batch(1, r, ocfg, seed=seed, H=h, W=w)
    opt_o = O.SGDState(ocfg)
    cfg, model = G.drn_model(ocfg, 5, "cuda", 5, "fp32")
    model.roi_heads.box_head.dropout_p = 0.0
    model.train()
    opt = build_optimizer(cfg, model)
    sd = dict(model.named_parameters())
    names = [n for n in sd if n.startswith("roi_heads.") and sd[n].requires_grad and "bbox_pred" not in n]
    for step in range(2):
        ref_losses, ref_grads = O.train_step(p, batch, ocfg, opt_o)
        opt.zero_grad()
        losses = model(G.drn_inputs(batch))
        sum(losses.values()).backward()
        got = {k: float(v.detach()) for k, v in losses.items()}
        assert set(got) == set(ref_losses)
        tol = 1e-4 if step == 0 else 2e-2  # step 1 sits behind one SGD step of an ill-conditioned toy net (as in test_train_two_steps_fp32)
        for k in got:
            assert abs(got[k] - ref_losses[k]) <= tol * max(abs(ref_losses[k]), 1e-3), (step, k, got[k], ref_losses[k])
        if step == 0:
            for n in names:
                g, rg = sd[n].grad.detach().cpu().numpy(), ref_grads[n].numpy()
                if np.abs(rg).max() < 1e-6:  # analytically zero (det bias): both sides are rounding noise
                    assert np.abs(g).max() < 1e-5, n
                else:
                    assert _relerr(g, rg) < 2e-3, (n, _relerr(g, rg))
        opt.step()
    torch.cuda.synchronize()
    load_package().set_precision("fp32")
