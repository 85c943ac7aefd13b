"""fp8 MFMA conv path of the frozen trunk (BASELINE configs[4]).  The reference has no reduced precision at all (SURVEY
F5), so the definition of record is its fp32 trunk (resnet_ws.py:217-237, :405-416) and the tolerance is DERIVED, not
guessed: oracle.OracleCfg(fp8_scales=...) restates the fp32 algorithm with every value the product stores in fp8 (OCP
e4m3fn: 3 significand bits, 2^-4 relative rounding) rounded at the same point with the same scales, so |emu - fp32| is
the size of the fp8 effect on each quantity and the product must stay within a small multiple of it of BOTH oracles.

  * drn_conv2d_nhwc_q alone: fp8 x fp8 -> fp32 accumulate is EXACT per product (4-bit significands), so against a
    CPU conv of the dequantised operands the pre-rounding value agrees to fp32 summation order and the stored fp8 byte
    may differ only where that value sits on a rounding boundary: <= 1 code, on < 1 % of the elements;
  * fp8 max-pool: bit-exact;
  * the calibrated WS-R50-C4 trunk at 224 x 224 and a full train step at R = 2000 against both oracles."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as G
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
O = G.O
load_package()
DEV = "cuda"
FP8 = torch.float8_e4m3fn


@pytest.fixture(scope="module")
def drn():
    import importlib

    pkg = load_package()
    pkg._cabi.lib()
    return importlib.import_module("drn_wsod_pytorch_amd.ops")


def _quant(t, s):
    return (t * s).clamp(-448, 448).to(FP8)


@pytest.mark.parametrize("N,H,W,cin,cout,k,pad,dil,res,relu,out", [
    (1, 14, 14, 64, 64, 3, 1, 1, False, True, "fp8"), (2, 14, 14, 256, 1024, 1, 0, 1, True, True, "fp8"),
    (1, 28, 28, 128, 128, 3, 2, 2, False, True, "fp8"), (1, 56, 56, 64, 256, 1, 0, 1, False, False, "fp8"),
    (2, 14, 14, 256, 1024, 1, 0, 1, True, True, "bf16"), (1, 112, 112, 64, 64, 3, 1, 1, False, True, "fp8"),
    # round 3: the 128x128 / 128x64 tiles and the two-K-group kernel at real-size maps, fp8 residual through the vector
    # epilogue (8-byte loads / stores), a bf16 residual into an fp8 output
    (1, 100, 152, 64, 256, 1, 0, 1, True, True, "fp8"), (1, 100, 152, 256, 64, 1, 0, 1, False, True, "fp8"),
    (1, 50, 76, 128, 128, 3, 1, 1, False, True, "fp8"), (1, 50, 76, 512, 128, 1, 0, 1, True, False, "fp8bf16res")])
@pytest.mark.parametrize("k64", [1, 0])
def test_conv_fp8(drn, N, H, W, cin, cout, k, pad, dil, res, relu, out, k64):
    """k64 = 1: v_mfma_scale_f32_32x32x64_f8f6f4 (the fp8 rate, round 3); 0: the K = 16 non-scaled form.  Both multiply
    exactly, so both sit within fp32 summation order of the fp64-free reference."""
    bf_res = out == "fp8bf16res"
    out = "fp8" if bf_res else out
    old = drn.tune(drn.TUNE_FP8_K64, k64)
    try:
        _conv_fp8_case(drn, N, H, W, cin, cout, k, pad, dil, res, relu, out, bf_res)
    finally:
        drn.tune(drn.TUNE_FP8_K64, old)


def _conv_fp8_case(drn, N, H, W, cin, cout, k, pad, dil, res, relu, out, bf_res):
    rs = np.random.RandomState(3)
    x = torch.from_numpy(np.abs(rs.standard_normal((N, H, W, cin))).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    s_x = 448.0 / float(x.abs().max())
    s_w = 448.0 / w.abs().amax(dim=(1, 2, 3))
    xq = _quant(x, s_x)
    wq = _quant(w, s_w.view(-1, 1, 1, 1))
    bn_scale = torch.from_numpy((0.8 + 0.4 * rs.rand(cout)).astype(np.float32))
    bn_bias = torch.from_numpy((0.1 * rs.standard_normal(cout)).astype(np.float32))
    ref = F.conv2d((xq.float() / s_x).permute(0, 3, 1, 2), wq.float() / s_w.view(-1, 1, 1, 1), padding=pad, dilation=dil)
    ref = ref * bn_scale.view(1, -1, 1, 1) + bn_bias.view(1, -1, 1, 1)
    r_q = None
    if res:
        r = torch.from_numpy(rs.standard_normal(tuple(ref.permute(0, 2, 3, 1).shape)).astype(np.float32))
        s_r = 1.0 if bf_res else 448.0 / float(r.abs().max())
        r_q = r.to(torch.bfloat16) if bf_res else _quant(r, s_r)
        ref = ref + (r_q.float() / s_r).permute(0, 3, 1, 2)
    if relu:
        ref = F.relu(ref)
    ref = ref.permute(0, 2, 3, 1).contiguous()
    s_y = 448.0 / float(ref.abs().max()) if out == "fp8" else 1.0
    # packed operands: k = (kh*KW + kw)*Cin + ci, rows padded to 128 bytes
    wp = torch.zeros((cout, drn.kpad(k * k * cin, FP8)), dtype=torch.float32)
    wp[:, : k * k * cin] = wq.float().permute(0, 2, 3, 1).reshape(cout, -1)
    alpha = (bn_scale * s_y / (s_x * s_w)).float().contiguous().to(DEV)
    beta = (bn_bias * s_y).float().contiguous().to(DEV)
    y = drn.conv2d_nhwc_q(xq.to(DEV), wp.to(FP8).to(DEV), cout, k, k, 1, pad, dil, alpha, beta,
                          FP8 if out == "fp8" else torch.bfloat16, r_q.to(DEV) if res else None,
                          s_y / s_r if res else 1.0, relu)
    torch.cuda.synchronize()
    # the kernel's pre-rounding value equals `ref` up to fp32 summation order (the products are exact): eps below; the
    # stored value must then be a correct rounding of SOME value within eps of ref: |stored - ref| <= ulp(ref)/2 + eps
    eps = 2e-6 * float(ref.abs().max()) * np.sqrt(cin * k * k)
    r64 = ref.double().numpy()
    if out == "bf16":
        got = y.float().cpu().double().numpy()
        half_ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(r64), 1e-30))) - 8)
        assert (np.abs(got - r64) <= half_ulp * 1.001 + eps).all()
        return
    got = (y.float().cpu().double() / s_y).numpy()
    mag = np.maximum(np.abs(r64) * s_y, 2.0 ** -6)  # below 2^-6 (scaled) e4m3 is subnormal: fixed spacing 2^-9
    half_ulp = 2.0 ** (np.floor(np.log2(mag)) - 4) / s_y
    err = np.abs(got - r64)
    assert (err <= half_ulp * 1.001 + eps).all(), float((err - half_ulp).max())
    exp = _quant(ref, s_y)
    same = (y.cpu().view(torch.uint8) == exp.view(torch.uint8)).float().mean()
    assert float(same) > 0.99, float(same)  # only values on a rounding boundary may land on the neighbouring code


def test_maxpool_fp8(drn):
    rs = np.random.RandomState(5)
    for stride, (h, w) in ((2, (56, 56)), (1, (28, 28)), (2, (13, 9))):
        x = torch.from_numpy(np.abs(rs.standard_normal((2, h, w, 64))).astype(np.float32))
        xq = _quant(x, 448.0 / float(x.max()))
        y = drn.maxpool2x2_nhwc(xq.to(DEV), stride)
        ref = F.max_pool2d(xq.float().permute(0, 3, 1, 2), 2, stride).permute(0, 2, 3, 1)
        assert torch.equal(y.float().cpu(), ref)


# per-stage bounds of the fp8 trunk against its emulator (set from profiles/r3_08_fp8_by_stage.txt with ~2x margin):
# relative rms of a stage's output when the stage is fed the EMULATOR's input, the fraction of identical fp8 codes there,
# and the relative rms of the product's own chain (flips compound along the 45 convs)
# measured: forced 2.2e-3 / 1.09e-2 / 2.4e-2 / 3.9e-2, same code 99.82 / 94.1 / 77.9 %, own chain 2.2e-3 / 2.0e-2 / 5.0e-2 /
# 8.1e-2 (within a stage the flips of one conv perturb the next conv's sums and breed further flips); a wrong tap,
# channel, scale or residual path gives a relative rms of O(1) and a same-code fraction near the chance level
FP8_STAGE_FORCED_REL = {"stem": 5e-3, "res2": 2.5e-2, "res3": 5e-2, "res4": 8e-2}
FP8_STAGE_FORCED_SAME = {"stem": 0.995, "res2": 0.88, "res3": 0.60, "res4": 0.0}
FP8_STAGE_CHAIN_REL = {"stem": 5e-3, "res2": 4e-2, "res3": 1e-1, "res4": 1.6e-1}
# relative bound of a refinement loss against an oracle that mined the SAME pseudo-GT rows (set from
# profiles/r3_07_fp8_by_stage.txt with ~2x margin): vs the fp8 emulator / vs the fp32 oracle
FP8_LOSS_SAME_ROWS = {"emu": 0.10, "fp32": 0.25}


def _within(v, e, r, floor, k=5.0, rel=2e-2):
    tol = k * abs(e - r) + rel * max(abs(r), floor)
    return abs(v - e) <= tol and abs(v - r) <= tol, tol


def test_fp8_trunk_and_train_step_full_size():
    """WS-R50-C4 (the benchmark trunk), 224 x 224, R = 2000: calibrate on two images, then (a) the res4 map of a third
    image and (b) the losses, image scores and fc6 weight update of one train step (fp8 trunk, bf16 heads = bench.py
    --workload r50c4_fp8) against the fp32 oracle and the fp8-emulating oracle fed with the product's own scales."""
    from drn_wsod_pytorch_amd.engine import build_optimizer

    torch.set_num_threads(min(32, os.cpu_count() or 1))
    kw = dict(arch="wsr50", out_feature="res4", res5_dilation=1, num_classes=20)
    ocfg = O.OracleCfg(dropout=0.0, base_lr=2e-4, **kw)
    cfg, model = G.drn_model(ocfg, 3, "cuda", 5, "bf16")
    model.roi_heads.box_head.dropout_p = 0.0
    calib = [O.synthetic_batch(1, 8, ocfg, seed=100 + i) for i in range(2)]
    batch = O.synthetic_batch(1, 2000, ocfg, seed=4321)
    ins = lambda b: G.drn_inputs([dict(x, gt_boxes=torch.zeros(len(x["gt_classes"]), 4)) for x in b])
    model.eval()
    with torch.no_grad():
        bf16_feat = model.backbone(model.preprocess_image(ins(batch)).tensor)["res4"].float().cpu()
        scales = model.backbone.calibrate_fp8([model.preprocess_image(ins(b)).tensor for b in calib])
        feat = model.backbone(model.preprocess_image(ins(batch)).tensor)["res4"].float().cpu()
        # the launch plan of the fp8 trunk (csrc/executor.hip) against the per-layer walk it replaces: bit-identical
        model.backbone.use_plan = False
        walk = model.backbone(model.preprocess_image(ins(batch)).tensor)["res4"].float().cpu()
        model.backbone.use_plan = True
        assert torch.equal(feat, walk)
        # the plan is keyed on the fp8 configuration's VALUES - a conv re-enabled with another output scale
        # (disable_fp8 + enable_fp8 hands out a new dict, possibly at the old address) must rebuild it
        c2 = model.backbone.res2[0].conv2
        old_q = dict(c2._fp8)
        c2.disable_fp8()
        c2.enable_fp8(old_q["out_scale"] * 0.75, old_q["out_dtype"])  # (not a power of two: other fp8 codes, another res4 map)
        moved = model.backbone(model.preprocess_image(ins(batch)).tensor)["res4"].float().cpu()
        model.backbone.use_plan = False
        moved_walk = model.backbone(model.preprocess_image(ins(batch)).tensor)["res4"].float().cpu()
        model.backbone.use_plan = True
        assert torch.equal(moved, moved_walk) and not torch.equal(moved, feat)
        c2.disable_fp8()
        c2.enable_fp8(old_q["out_scale"], old_q["out_dtype"])
        assert torch.equal(model.backbone(model.preprocess_image(ins(batch)).tensor)["res4"].float().cpu(), feat)
    assert all(np.isfinite(v) and v > 0 for v in scales.values()) and len(scales) == 45  # 3 stem + 13 blocks x 3 + 3 shortcuts
    last = [n for n, m in model.backbone.named_modules() if getattr(m, "_fp8", None) and m._fp8["out_dtype"] == torch.bfloat16]
    assert last == ["res4.5.conv3"]
    p = O.init_params(ocfg, seed=3)
    emu_cfg = copy.deepcopy(ocfg)
    emu_cfg.emulate_bf16, emu_cfg.fp8_scales, emu_cfg.fp8_last = True, dict(scales), last[0]
    x32, _ = O.preprocess_image([b["image"] for b in batch], ocfg)
    xe, _ = O.preprocess_image([b["image"] for b in batch], emu_cfg)
    f32 = O.backbone_forward(p, x32, ocfg)
    fe = O.backbone_forward(p, xe, emu_cfg)
    rms = lambda a: float(np.sqrt((np.asarray(a, np.float64) ** 2).mean()))
    d_pe, d_pr, d_er = rms(feat - fe), rms(feat - f32), rms(fe - f32)
    d_bf = rms(bf16_feat - f32)
    print("[fp8 trunk] res4 rms %.3f: |p-emu| %.2e |p-fp32| %.2e |emu-fp32| %.2e  (bf16 trunk vs fp32: %.2e)" % (
        rms(f32), d_pe, d_pr, d_er, d_bf))
    assert d_er > d_bf, "fp8 rounding must cost more than bf16 rounding, or the fp8 path is not running"
    assert max(d_pe, d_pr) <= 3.0 * d_er + 1e-2 * rms(f32)
    # (a') round 3: the trunk stage by stage against its emulator, in units of the fp8 spacing.  Both
    # run their own chain from the same image, so what separates them is rounding flips (a 1e-6 summation-order difference
    # lands an element on the neighbouring fp8 code) and their propagation: few flips after the 3-conv stem, more after
    # each stage.  An indexing / scale / layout defect in any fp8 layer moves whole channels by many codes and shows at
    # the first stage it touches, long before the flips of the 45-conv chain have compounded.
    bb = model.backbone
    with torch.no_grad():
        y = bb.stem.forward_nhwc(bb._input_nhwc(model.preprocess_image(ins(batch)).tensor))
        prod = {"stem": y}
        for stage, name in bb.stages_and_names:
            for block in stage:
                y = block.forward_nhwc(y)
            prod[name] = y
    femu = O.resnet_ws_forward(p, xe, emu_cfg)
    last_conv = {"stem": "stem.conv3", "res2": "res2.2.conv3", "res3": "res3.3.conv3", "res4": "res4.5.conv3"}

    def compare(t, ev, name, tag):
        sc = getattr(t, "_drn_scale", 1.0)
        pv = (t.float() / sc).permute(0, 3, 1, 2).cpu().double().numpy()
        assert pv.shape == ev.shape, (name, pv.shape, ev.shape)
        rel = rms(pv - ev) / rms(ev)
        same = None
        if t.dtype == FP8:
            assert abs(sc - scales[last_conv[name]]) <= 1e-6 * sc
            same = float((np.abs(pv - ev) * sc < 2.0 ** -10).mean())  # identical fp8 codes (spacing >= 2^-9 in code units)
        print("[fp8 %-14s %-5s] rel rms %.2e%s" % (tag, name, rel, "" if same is None else ", %.2f %% of the elements on the same fp8 code" % (100 * same)))
        return rel, same

    # (i) the product's own chain against the emulator's own chain (flips compound from stage to stage)
    chain = {name: compare(prod[name], femu[name].double().numpy(), name, "own chain") for name in ("stem", "res2", "res3", "res4")}
    # (ii) every stage FROM THE EMULATOR'S INPUT of that stage: only the flips of the stage's own 9-18 convs remain, so the
    # bound can be tight - a wrong tap, channel, scale or residual path in any fp8 layer fails here by orders of magnitude
    forced = {"stem": chain["stem"]}  # the stem's input is the bf16 image in both
    prev = "stem"
    with torch.no_grad():
        for stage, name in bb.stages_and_names:
            src = femu[prev]
            s_in = scales[last_conv[prev]]
            xin = _quant(src, s_in).permute(0, 2, 3, 1).contiguous().to(DEV)  # exact: the emulator's values are on the fp8 grid
            assert torch.equal(xin.float().cpu() / s_in, src.permute(0, 2, 3, 1)), "emulator output off the fp8 grid"
            xin._drn_scale = s_in
            y = xin
            for block in stage:
                y = block.forward_nhwc(y)
            forced[name] = compare(y, femu[name].double().numpy(), name, "emulator input")
            prev = name
    for name in ("stem", "res2", "res3", "res4"):
        rel, same = forced[name]
        assert rel <= FP8_STAGE_FORCED_REL[name], (name, rel)
        if same is not None:
            assert same >= FP8_STAGE_FORCED_SAME[name], (name, same)
        assert chain[name][0] <= FP8_STAGE_CHAIN_REL[name], (name, chain[name][0])
    # (b) one train step
    model.train()
    opt = build_optimizer(cfg, model)
    opt.zero_grad()
    # (b') round 3: the HEADS on the emulator's res4 map.  With the trunk pinned stage by stage above, what is left of the
    # step is the bf16 heads; fed the emulator's own feature map they differ from the emulating oracle by bf16 rounding
    # flips only - the regime tests/test_bench_mode_gpu.py measured at <= 3.6 % on every loss - so the 2x loss error
    # the end-to-end comparison below lets through (its pseudo-GT rows differ: loss_cls_r1 0.178 vs 0.089) cannot hide a
    # defect here: every loss within 1e-3 of the emulating oracle (measured 1.7e-5), image scores within 3 x |emu - fp32| + 1 %, and a
    # pseudo-GT row the emulator did not pick must be a near-tie in its scores
    emu_l, _, emu_aux = O.train_step(O.init_params(emu_cfg, seed=3), batch, emu_cfg, O.SGDState(emu_cfg), return_aux=True)
    f32_l, _, f32_aux = O.train_step(O.init_params(ocfg, seed=3), batch, ocfg, O.SGDState(ocfg), return_aux=True)
    fe_dev = fe.to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    assert torch.equal(fe_dev.float().cpu(), fe), "the emulator's res4 map is stored in bf16"
    model.backbone.forward = lambda x: {"res4": fe_dev}
    try:
        forced = {k: float(v.detach()) for k, v in model(ins(batch)).items()}
    finally:
        del model.backbone.forward
    st = model.roi_heads._last_state
    img_f = st["aux"]["img_scores"].cpu().numpy().astype(np.float64)
    e_img, r_img = emu_aux["img_scores"].numpy().astype(np.float64), f32_aux["img_scores"].numpy().astype(np.float64)
    tol = 3.0 * np.abs(e_img - r_img).max() + 1e-2 * np.abs(r_img).max()
    print("[fp8 heads on the emulator's res4] image scores |p-emu| %.2e (bound %.2e)" % (np.abs(img_f - e_img).max(), tol))
    assert np.abs(img_f - e_img).max() <= tol
    prev = [emu_aux["scores"].detach()] + [torch.softmax(l.detach(), dim=-1) for l in emu_aux["logits"][:-1]]
    for k in range(ocfg.refine_num):
        mine = st["aux"]["targets"][k]["pgt_idx"].cpu().numpy()[0]
        (_, pc, _, _, idx) = emu_aux["pgt"][k][0]
        for g, c in enumerate(pc.numpy()):
            if int(mine[g]) != int(idx[g]):
                col = prev[k][:, int(c)].numpy()
                ratio = float(col[int(mine[g])] / col.max())
                print("   pgt branch %d class %d: product row %d, emu %d, score ratio %.4f" % (k, int(c), int(mine[g]), int(idx[g]), ratio))
                assert ratio >= 0.9, ("pgt row not a near-tie", k, g, ratio)
    for k_, v in forced.items():
        rel = abs(v - emu_l[k_]) / max(abs(emu_l[k_]), 1e-2)
        print("   %-12s %.6f  emu %.6f  fp32 %.6f  rel(p, emu) %.2e" % (k_, v, emu_l[k_], f32_l[k_], rel))
        assert rel <= 1e-3, (k_, v, emu_l[k_])  # measured <= 1.7e-5 (dropout off in this test: no {0, 2} mask amplifies the flips)
    opt.zero_grad()
    w0 = model.roi_heads.box_head.fc1.weight.detach().reshape(-1)[::4099].cpu().clone()
    losses = model(ins(batch))
    sum(losses.values()).backward()
    opt.step()
    torch.cuda.synchronize()
    got = {k: float(v.detach()) for k, v in losses.items()}
    img = model.roi_heads._last_state["aux"]["img_scores"].cpu().numpy().astype(np.float64)
    rows_p = [tg["pgt_idx"].cpu().numpy().copy() for tg in model.roi_heads._last_state["aux"]["targets"]]
    dw = (model.roi_heads.box_head.fc1.weight.detach().reshape(-1)[::4099].cpu() - w0).numpy()
    res = {}
    for tag, c in (("emu", emu_cfg), ("fp32", ocfg)):
        pp = O.init_params(c, seed=3)
        w_before = pp["roi_heads.box_head.fc1.weight"].reshape(-1)[::4099].clone()
        l, _, aux = O.train_step(pp, batch, c, O.SGDState(c), return_aux=True)
        res[tag] = (l, aux["img_scores"].numpy().astype(np.float64),
                    (pp["roi_heads.box_head.fc1.weight"].reshape(-1)[::4099] - w_before).numpy(),
                    [[idx.numpy().copy() for (_, _, _, _, idx) in aux["pgt"][k]] for k in range(c.refine_num)])
    bad = []
    # round 3: a refinement loss is a CONTINUOUS function of the logits only for fixed pseudo-GT rows (get_pgt's arg-max,
    # roi_heads_oicr.py:504-506, is the discontinuity).  Where the product mined the same rows as an oracle, its loss is
    # held to that oracle tightly (FP8_LOSS_SAME_ROWS); the group bound below remains for branches whose rows differ.
    for k in range(ocfg.refine_num):
        name = "loss_cls_r%d" % k
        for tag in ("emu", "fp32"):
            o_idx = np.asarray(res[tag][3][k][0]).reshape(-1)  # image 0 (the batch holds one image)
            same = np.array_equal(np.asarray(rows_p[k][0]).reshape(-1)[: len(o_idx)], o_idx)
            v, o = got[name], res[tag][0][name]
            rel = abs(v - o) / max(abs(o), 1e-2)
            print("   %-12s rows %s as %-4s: product %.6f  %s %.6f  rel %.3e" % (name, "SAME" if same else "differ", tag, v, tag, o, rel))
            if same and rel > FP8_LOSS_SAME_ROWS[tag]:
                bad.append((name, tag, v, o, rel))
    # the refinement losses share one mechanism (pseudo-GT mining on noisy scores): their bf/fp8 effect is estimated as
    # a group - the largest |emu - fp32| among them - because a single pair of draws can be close by chance
    branch = [k for k in got if k != "loss_cls"]
    grp = max(abs(res["emu"][0][k] - res["fp32"][0][k]) for k in branch) if branch else 0.0
    for k, v in got.items():
        e, r = res["emu"][0][k], res["fp32"][0][k]
        tol = 5.0 * (abs(e - r) if k == "loss_cls" else grp) + 2e-2 * max(abs(r), 1e-2)
        print("   %-12s %.6f  emu %.6f  fp32 %.6f  bound %.2e" % (k, v, e, r, tol))
        if not (abs(v - e) <= tol and abs(v - r) <= tol):
            bad.append(k)
    d_er = np.abs(res["emu"][1] - res["fp32"][1]).max()
    tol = 5.0 * d_er + 2e-2 * np.abs(res["fp32"][1]).max()
    assert np.abs(img - res["emu"][1]).max() <= tol and np.abs(img - res["fp32"][1]).max() <= tol
    d_er = np.abs(res["emu"][2] - res["fp32"][2]).max()
    tol = 5.0 * d_er + 2e-2 * np.abs(res["fp32"][2]).max()
    assert np.abs(dw - res["emu"][2]).max() <= tol and np.abs(dw - res["fp32"][2]).max() <= tol
    assert not bad, bad
    model.backbone.disable_fp8()
    load_package().set_precision("fp32")
