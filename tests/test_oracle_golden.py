"""CPU tests (no GPU): pin the ORACLE against golden vectors produced by the unmodified reference
(tests/golden/gen_golden.py) and against the reference ROIAlign C++ compiled in place (oracle/_ref).
fp32 everywhere; tolerances are written next to each check."""
import os

import numpy as np
import pytest
import torch

import golden_util as G

O = G.O
RTOL = 2e-5  # oracle and reference run the same torch CPU ops: agreement is ~1e-7, allow 2e-5


def _close(a, b, rtol=RTOL, atol=1e-7):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------ op-level vectors
def test_roi_align_ramp_kat():
    """tests/layers/test_roi_align.py:13-45 goldens (values typed from the reference test's
    expected tensors; the fixture holds what the compiled reference produced)."""
    d = G.load("ops")
    ramp = torch.arange(25, dtype=torch.float32).reshape(1, 1, 5, 5)
    rois = torch.tensor([[0, 1, 1, 3, 3]], dtype=torch.float32)
    old = [[7.5, 8, 8.5, 9], [10, 10.5, 11, 11.5], [12.5, 13, 13.5, 14], [15, 15.5, 16, 16.5]]
    new = [[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0], [12.0, 12.5, 13.0, 13.5]]
    for al, kat in ((False, old), (True, new)):
        out = O.roi_align_forward(ramp, rois, 4, 1.0, 0, al)
        _close(out, d["ra_ramp_aligned%d" % int(al)], atol=1e-6)
        _close(out[0, 0], np.array(kat), atol=1e-6)


@pytest.mark.parametrize("al", [False, True])
@pytest.mark.parametrize("sr", [0, 2])
def test_roi_align_random_fwd_bwd(al, sr):
    d = G.load("ops")
    feat = torch.from_numpy(d["ra_feat"])
    rois = torch.from_numpy(d["ra_rois"])
    key = "ra_al%d_sr%d" % (int(al), sr)
    out = O.roi_align_forward(feat, rois, 7, 0.125, sr, al)
    _close(out, d[key + "_out"], atol=1e-6)
    gin = O.roi_align_backward(torch.from_numpy(d[key + "_gout"]), rois, tuple(feat.shape), 7, 0.125, sr, al)
    _close(gin, d[key + "_gin"], rtol=1e-4, atol=1e-5)  # sequential fp32 scatter sums


def test_roi_align_vs_compiled_reference():
    """oracle/_ref = detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp compiled as-is."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_ob", os.path.join(G.ROOT, "oracle", "build.py"))
    ob = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ob)
    so = ob.build_ref()
    if so is None:
        pytest.skip("oracle/_ref not built and /root/reference absent")
    spec = importlib.util.spec_from_file_location("d2_roialign_ref", so)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(2, 7, 19, 23, generator=g)
    R = 64
    x0 = torch.rand(R, generator=g) * 150
    y0 = torch.rand(R, generator=g) * 120
    rois = torch.stack([torch.randint(0, 2, (R,), generator=g).float(), x0, y0, x0 + torch.rand(R, generator=g) * 100,
                        y0 + torch.rand(R, generator=g) * 90], 1)
    for al in (False, True):
        for sr in (0, 1, 3):
            a = O.roi_align_forward(feat, rois, 7, 1 / 8.0, sr, al)
            b = ref.roi_align_forward(feat, rois, 1 / 8.0, 7, 7, sr, al)
            assert torch.equal(a, b), (al, sr, float((a - b).abs().max()))
            go = torch.randn(a.shape, generator=g)
            ga = O.roi_align_backward(go, rois, tuple(feat.shape), 7, 1 / 8.0, sr, al)
            gb = ref.roi_align_backward(go, rois, 1 / 8.0, 7, 7, 2, 7, 19, 23, sr, al)
            _close(ga, gb, rtol=1e-5, atol=1e-5)


def test_iou_matcher_box2box_fbn():
    d = G.load("ops")
    iou = O.pairwise_iou(torch.from_numpy(d["iou_b1"]), torch.from_numpy(d["iou_b2"]))
    assert np.array_equal(iou.numpy(), d["iou"])  # bit-exact: same fp32 ops
    m, l = O.matcher(iou)
    assert np.array_equal(m.numpy(), d["match_idx"]) and np.array_equal(l.numpy(), d["match_label"])
    # tests/modeling/test_matcher.py:14-28 golden (argmax path only; low-quality rule is off on this path)
    m2, _ = O.matcher(torch.from_numpy(d["mq"]), (0.3, 0.7), (0, -1, 1))
    assert m2.tolist() == [1, 1, 2, 0] == d["mq_idx"].tolist()
    dl = O.get_deltas(torch.from_numpy(d["b2b_src"]), torch.from_numpy(d["b2b_dst"]))
    assert np.array_equal(dl.numpy(), d["b2b_deltas"])
    out = O.apply_deltas(torch.from_numpy(d["b2b_apply_in"]), torch.from_numpy(d["b2b_src"]))
    assert np.array_equal(out.numpy(), d["b2b_apply_out"])
    z = O.apply_deltas(torch.zeros(20, 8), torch.from_numpy(d["b2b_src"]))
    assert np.array_equal(z.numpy(), d["b2b_apply_zero"])
    # tests/modeling/test_box2box_transform.py:16-29 round trip
    rt = O.apply_deltas(dl, torch.from_numpy(d["b2b_src"]))
    _close(rt, d["b2b_dst"], rtol=1e-4, atol=1e-4)
    p = {"x.norm." + k: O.seeded_tensor("x.norm." + k, (5,), 11) for k in ("weight", "bias", "running_mean", "running_var")}
    _close(O._bn(torch.from_numpy(d["ra_feat"]), p, "x.norm"), d["fbn_out"], atol=1e-6)


def test_inference_tail_indices_bit_exact():
    d = G.load("ops")
    b, s, c, rows = O.fast_rcnn_inference_single_image(torch.from_numpy(d["inf_boxes"].copy()),
                                                       torch.from_numpy(d["inf_scores"].copy()), (120, 200), 1e-5, 0.3,
                                                       100)
    assert np.array_equal(rows.numpy(), d["inf_out_rows"])
    assert np.array_equal(c.numpy(), d["inf_out_classes"])
    assert np.array_equal(s.numpy(), d["inf_out_scores"])
    assert np.array_equal(b.numpy(), d["inf_out_boxes"])


def _roi_pool_bruteforce(feat, rois, P, scale):
    """Independent pure-Python restatement of SURVEY Appendix C.1 (small cases only)."""
    import math

    N, C, H, W = feat.shape
    out = np.zeros((len(rois), C, P, P), np.float32)
    arg = -np.ones((len(rois), C, P, P), np.int32)

    def rnd(v):  # C round(): half away from zero, on the fp32 product
        v = float(np.float32(v))
        return int(math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1))

    for n, r in enumerate(rois):
        b = int(r[0])
        x1, y1, x2, y2 = (rnd(np.float32(r[i]) * np.float32(scale)) for i in (1, 2, 3, 4))
        rw, rh = max(x2 - x1 + 1, 1), max(y2 - y1 + 1, 1)
        bw, bh = np.float32(rw) / np.float32(P), np.float32(rh) / np.float32(P)
        for ph in range(P):
            hs = min(max(int(math.floor(np.float32(ph) * bh)) + y1, 0), H)
            he = min(max(int(math.ceil(np.float32(ph + 1) * bh)) + y1, 0), H)
            for pw in range(P):
                ws = min(max(int(math.floor(np.float32(pw) * bw)) + x1, 0), W)
                we = min(max(int(math.ceil(np.float32(pw + 1) * bw)) + x1, 0), W)
                if he <= hs or we <= ws:
                    continue
                win = feat[b, :, hs:he, ws:we].reshape(C, -1)
                k = win.argmax(1)  # first maximum
                out[n, :, ph, pw] = win[np.arange(C), k]
                arg[n, :, ph, pw] = (hs + k // (we - ws)) * W + ws + k % (we - ws)
    return out, arg


def test_roi_pool_vs_bruteforce_and_appendix_d():
    """RoIPool is 'parity unpinned' (torchvision absent): cross-check the C restatement against an
    independent brute-force, and against the hand-checked values of SURVEY Appendix D."""
    rs = np.random.RandomState(3)
    feat = rs.standard_normal((2, 4, 11, 9)).astype(np.float32)
    feat[0, 0, 2, 3] = feat[0, 0, 2, 4] = 9.0  # tie: first (row-major) maximum must win
    R = 40
    x0 = rs.rand(R) * 60
    y0 = rs.rand(R) * 80
    rois = np.stack([rs.randint(0, 2, R), x0, y0, x0 + rs.rand(R) * 50, y0 + rs.rand(R) * 60], 1).astype(np.float32)
    rois[0, 1:] = [-30, -30, -20, -20]  # fully outside -> empty bins -> 0 / argmax -1
    rois[1, 1:] = [12.0, 20.0, 12.0, 20.0]  # *0.125 = 1.5, 2.5 : half-away-from-zero rounding
    out, arg = O.roi_pool_forward(torch.from_numpy(feat), torch.from_numpy(rois), 7, 0.125)
    bo, ba = _roi_pool_bruteforce(feat, rois, 7, 0.125)
    assert np.array_equal(out.numpy(), bo) and np.array_equal(arg.numpy(), ba)
    # SURVEY Appendix D pooled values (P=1, scale 1)
    f = (torch.arange(32, dtype=torch.float32).view(1, 2, 4, 4) / 16)
    props = torch.tensor([[0, 0, 1, 1], [0, 0, 3, 3], [1, 1, 3, 3], [2, 0, 3, 2], [0, 2, 1, 3]], dtype=torch.float32)
    o, _ = O.roi_pool_forward(f, O.boxes_to_rois([props]), 1, 1.0)
    exp = [[.3125, 1.3125], [.9375, 1.9375], [.9375, 1.9375], [.6875, 1.6875], [.8125, 1.8125]]
    _close(o.view(5, 2), np.array(exp), atol=0)
    # backward: scatter-add by argmax == autograd of a gather
    go = torch.from_numpy(rs.standard_normal(out.shape).astype(np.float32))
    gi = O.roi_pool_backward(go, torch.from_numpy(rois), arg, feat.shape)
    exp_g = np.zeros(feat.shape, np.float64)
    a = arg.numpy()
    for n in range(R):
        for c in range(4):
            for p in range(49):
                k = a[n, c].reshape(-1)[p]
                if k >= 0:
                    exp_g[int(rois[n, 0]), c].reshape(-1)[k] += go[n, c].reshape(-1)[p].item()
    _close(gi, exp_g, rtol=1e-5, atol=1e-5)


def test_nms_vs_bruteforce():
    """nms/batched_nms are 'parity unpinned': check against an independent O(n^2) greedy restatement
    on tests/layers/test_nms.py:11-19-style inputs."""
    rs = np.random.RandomState(0)
    N = 600
    boxes = rs.rand(N, 4).astype(np.float32) * 100
    boxes[:, 2:] += boxes[:, :2]
    scores = rs.rand(N).astype(np.float32)
    scores[10] = scores[11]  # tie -> lower index first (stable sort)
    idxs = rs.randint(0, 7, N).astype(np.int64)
    for thr in (0.2, 0.5, 0.8):
        keep = O.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).numpy()
        order = np.argsort(-scores, kind="stable")
        alive = np.ones(N, bool)
        exp = []
        area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
        for i in order:
            if not alive[i]:
                continue
            exp.append(i)
            xx1 = np.maximum(boxes[i, 0], boxes[:, 0]); yy1 = np.maximum(boxes[i, 1], boxes[:, 1])
            xx2 = np.minimum(boxes[i, 2], boxes[:, 2]); yy2 = np.minimum(boxes[i, 3], boxes[:, 3])
            inter = np.maximum(np.float32(0), xx2 - xx1) * np.maximum(np.float32(0), yy2 - yy1)
            iou = inter / (area[i] + area - inter)
            alive &= ~(iou > np.float32(thr))
        assert keep.tolist() == exp
        kb = O.batched_nms(torch.from_numpy(boxes), torch.from_numpy(scores), torch.from_numpy(idxs), thr)
        off = idxs.astype(np.float32) * (boxes.max() + np.float32(1))
        kb2 = O.nms(torch.from_numpy(boxes + off[:, None]), torch.from_numpy(scores), thr)
        assert kb.tolist() == kb2.tolist()
    assert O.batched_nms(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0, dtype=torch.int64), 0.5).numel() == 0


# ------------------------------------------------------------------ head internals
@pytest.mark.parametrize("reg", [0, 1])
def test_heads_detail(reg):
    d = G.load("heads_reg%d" % reg)
    K = int(d["K"])
    cfg = O.OracleCfg(num_classes=K, refine_num=3, refine_reg=tuple(bool(x) for x in d["refine_reg"]), dan_dim=(24, 32),
                      pooler_res=3, dropout=0.0)
    sd_shapes = {"roi_heads.box_head.fc1.weight": (24, 54), "roi_heads.box_head.fc1.bias": (24,),
                 "roi_heads.box_head.fc2.weight": (32, 24), "roi_heads.box_head.fc2.bias": (32,)}
    for n in ("cls", "det"):
        sd_shapes["roi_heads.box_predictor.%s.weight" % n] = (K, 32)
        sd_shapes["roi_heads.box_predictor.%s.bias" % n] = (K,)
    for k in range(3):
        pre = "roi_heads.box_refinery_%d." % k
        sd_shapes.update({pre + "cls_score.weight": (K + 1, 32), pre + "cls_score.bias": (K + 1,),
                          pre + "bbox_pred.weight": (4 * K, 32), pre + "bbox_pred.bias": (4 * K,)})
    p = {n: t.requires_grad_(True) for n, t in O.seeded_params(sd_shapes, int(d["seed"])).items()}
    batch = G.batch_from(d)
    cfg_stride = 8

    # the heads fixture pools a raw feature map at scale 1/8: emulate with a vgg-like stride lookup
    class _C(O.OracleCfg):
        pass

    feat = torch.from_numpy(d["feat"])
    old = O.backbone_stride
    O.backbone_stride = lambda c: cfg_stride
    try:
        losses, aux = O.roi_heads_train(p, feat, [b["proposal_boxes"] for b in batch],
                                        [b["objectness_logits"] for b in batch], [b["gt_classes"] for b in batch], cfg,
                                        None, True)
    finally:
        O.backbone_stride = old
    for k, v in losses.items():
        _close(v.item(), float(d[k]))
    _close(aux["scores"].detach(), d["wsddn_scores"], atol=1e-9)
    _close(aux["img_scores"], d["img_scores"], atol=1e-8)
    for k in range(3):
        _close(aux["logits"][k].detach(), d["logits_r%d" % k], atol=1e-6)
    # forced tie (rows 3 and 7 identical): first index must win in pgt mining
    for pg in aux["pgt"][0]:
        assert 7 not in pg[4].tolist()
    total = sum(losses.values())
    names = [n for n in p if ("gradnone." + n) not in d]
    gs = torch.autograd.grad(total, [p[n] for n in names], allow_unused=True)
    for n, g in zip(names, gs):
        _close(g, d["grad." + n], rtol=1e-4, atol=1e-6)
    for n in p:
        if ("gradnone." + n) in d:
            assert "bbox_pred" in n  # unused parameters (F10)


def _replay_pcl_decisions(d):
    """PCL model golden: the reference's k-means draw and equal-degree picks are not functions of the inputs; the
    generator recorded the decisions it took (tests/golden/gen_golden.py `pclmodel`).  Replaying them in the two places
    of oracle/pcl_oracle.py that restate those steps pins everything else of the PCL flow (cascade wiring, clusters,
    loss, gradient, SGD) to the reference.  Returns the undo function."""
    from oracle import pcl_oracle as PO

    km, picks = iter(d["pcl_km_log"].tolist()), iter(d["pcl_pick_log"].tolist())
    keep = PO.kmeans_top_threshold, PO._argmax_last
    PO.kmeans_top_threshold = lambda v: np.float32(next(km))
    PO._argmax_last = lambda x: int(next(picks))

    def undo():
        PO.kmeans_top_threshold, PO._argmax_last = keep

    return undo


# ------------------------------------------------------------------ whole model, 2 SGD steps
@pytest.mark.parametrize("name", sorted(G.MODEL_CASES))
def test_full_model_two_steps(name):
    cfg = G.MODEL_CASES[name]
    cfg.dropout = 0.0
    d = G.load(name)
    seed = int(d["seed"])
    shapes = O.param_shapes(cfg)
    p = O.seeded_params(shapes, seed)
    batch = G.batch_from(d)
    masks = G.dropmasks_from(d)
    fz = G.FREEZE_AT.get(name, 5)
    tn = set(d["trainable"].tolist())
    on = set(O.trainable_names(p, cfg, fz))
    assert {n for n in tn if not ("bbox_pred" in n and n not in on)} == on
    opt = O.SGDState(cfg)
    undo = _replay_pcl_decisions(d) if cfg.heads == "pcl" else (lambda: None)
    for step in range(2):
        try:
            losses, grads = O.train_step(p, batch, cfg, opt, masks, fz)
        except Exception:
            undo()
            raise
        if step == 1:
            undo()
        for k, v in losses.items():
            _close(v, float(d["step%d_%s" % (step, k)]), rtol=1e-4)
        if step == 0:
            for n, g in grads.items():
                if "grad0." + n in d:
                    _close(g, d["grad0." + n], rtol=2e-3, atol=2e-6)
                elif "gradhead0." + n in d:
                    _close(g.reshape(-1)[:4096], d["gradhead0." + n], rtol=2e-3, atol=2e-6)
                    _close(g.double().abs().sum().item(), float(d["gradabs0." + n]), rtol=1e-4)
    for n in on:
        _close(p[n].reshape(-1)[:2048], d["after2.head." + n], rtol=1e-4, atol=1e-6)
    res, all_scores, all_boxes = O.model_inference(p, batch, cfg)
    x, _ = O.preprocess_image([b["image"] for b in batch], cfg)
    _close(O.backbone_forward(p, x, cfg), d["feat"], rtol=1e-4, atol=1e-5)
    for i, (b, s, c, rows) in enumerate(res):
        assert np.array_equal(c.numpy(), d["det%d_classes" % i])
        _close(s, d["det%d_scores" % i], rtol=1e-4)
        _close(b, d["det%d_boxes" % i], rtol=5e-4)  # after 2 SGD steps through exp() box decoding
        _close(all_scores[i], d["all_scores%d" % i][0], rtol=1e-4, atol=1e-7)


def test_appendix_d_known_answer():
    """tiny KAT produced by the real reference OICRROIHeads."""
    K = 3
    cfg = O.OracleCfg(num_classes=K, refine_num=2, refine_reg=(False, False), dan_dim=(4, 4), pooler_res=1, dropout=0.0)
    names = ["box_head.fc1.weight", "box_head.fc1.bias", "box_head.fc2.weight", "box_head.fc2.bias",
             "box_predictor.cls.weight", "box_predictor.cls.bias", "box_predictor.det.weight", "box_predictor.det.bias"]
    shapes = [(4, 2), (4,), (4, 4), (4,), (3, 4), (3,), (3, 4), (3,)]
    for k in range(2):
        names += ["box_refinery_%d.%s" % (k, s) for s in ("cls_score.weight", "cls_score.bias", "bbox_pred.weight",
                                                          "bbox_pred.bias")]
        shapes += [(4, 4), (4,), (12, 4), (12,)]
    p = {}
    for i, (n, sh) in enumerate(zip(names, shapes)):
        numel = int(np.prod(sh))
        p["roi_heads." + n] = (0.5 * torch.sin(torch.arange(numel, dtype=torch.float32) * (0.7 + 0.1 * i) + 0.3 * i)
                               ).view(sh).requires_grad_(True)
    feat = torch.arange(32, dtype=torch.float32).view(1, 2, 4, 4) / 16
    props = torch.tensor([[0, 0, 1, 1], [0, 0, 3, 3], [1, 1, 3, 3], [2, 0, 3, 2], [0, 2, 1, 3]], dtype=torch.float32)
    obj = torch.tensor([0.9, 0.5, 0.4, 0.2, 0.1])
    old = O.backbone_stride
    O.backbone_stride = lambda c: 1
    try:
        losses, aux = O.roi_heads_train(p, feat, [props], [obj], [torch.tensor([2, 0])], cfg, None, True)
    finally:
        O.backbone_stride = old
    _close(losses["loss_cls"].item(), 1.117370, rtol=1e-5)
    _close(losses["loss_cls_r0"].item(), 0.562272, rtol=1e-5)
    _close(losses["loss_cls_r1"].item(), 0.258158, rtol=1e-5)
    _close(aux["img_scores"][0], [0.650765, 0.274785, 0.074183], rtol=1e-5)
    _close(aux["scores"][0].detach(), [0.122995, 0.055225, 0.015751], rtol=1e-4)
    g = torch.autograd.grad(sum(losses.values()), [p["roi_heads.box_head.fc1.bias"],
                                                   p["roi_heads.box_predictor.cls.bias"]])
    _close(g[0], [0.072416, 0.048081, -0.012642, 0], rtol=1e-4, atol=1e-6)
    _close(g[1], [0.017981, 0.274978, -0.292959], rtol=1e-4, atol=1e-6)


def test_tta_golden():
    """oracle.tta_* vs the reference's own GeneralizedRCNNWithTTAAVG (tests/golden/gen_golden.py `tta`): augmented
    images (PIL resize + flip) and proposals bit for bit, averaged boxes / scores, final detections."""
    d = G.load("tta_r50c4_tiny")
    ocfg = G.MODEL_CASES["model_r50c4_tiny"]
    p = O.seeded_params(O.param_shapes(ocfg), int(d["seed"]))
    img = torch.from_numpy(d["image_u8"])
    boxes, obj = torch.from_numpy(d["proposal_boxes"]), torch.from_numpy(d["objectness_logits"])
    det, augs, avg_boxes, avg_scores = O.tta_inference(p, img, boxes, obj, tuple(img.shape[1:]), ocfg,
                                                       [int(x) for x in d["min_sizes"]], int(d["max_size"]), True,
                                                       int(d["topk"]), return_aux=True)
    assert len(augs) == int(d["n_aug"])
    for i, a in enumerate(augs):
        assert np.array_equal(a["image"].numpy(), d["aug%d_image" % i]), i
        assert np.array_equal(a["proposal_boxes"].numpy(), d["aug%d_boxes" % i]), i
        assert np.array_equal(a["objectness_logits"].numpy(), d["aug%d_obj" % i]), i
    assert np.allclose(avg_scores.numpy(), d["avg_scores"], rtol=1e-5, atol=1e-7)
    assert np.allclose(avg_boxes.numpy(), d["avg_boxes"], rtol=1e-6, atol=1e-4)
    b, s, c, _ = det
    assert np.array_equal(c.numpy(), d["det_classes"])
    assert np.allclose(s.numpy(), d["det_scores"], rtol=1e-5, atol=1e-7)
    assert np.allclose(b.numpy(), d["det_boxes"], rtol=1e-6, atol=1e-4)


def test_pil_bilinear_restatement():
    """oracle.pil_bilinear_resize_u8 (Pillow's integer BILINEAR resample, restated for the device-side TTA mapper) == the
    installed Pillow bit for bit (up- and down-scaling, one direction unchanged), == the augmented images the REFERENCE's own
    mapper produced (tests/golden/tta_r50c4_tiny.npz), and the product's vectorised coefficient tables
    (ops.pil_bilinear_coeffs) == the oracle's loop-for-loop ones."""
    from PIL import Image

    from __graft_entry__ import load_package

    load_package()
    from drn_wsod_pytorch_amd import ops

    rs = np.random.RandomState(0)
    for h, w, nh, nw in [(375, 500, 480, 640), (333, 500, 864, 1297), (500, 375, 240, 180), (37, 53, 37, 90), (64, 48, 21, 48),
                         (50, 60, 173, 60)]:
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        assert np.array_equal(O.pil_bilinear_resize_u8(img, nh, nw), ref), (h, w, nh, nw)
    for n_in, n_out in [(375, 480), (500, 1536), (500, 180), (97, 96), (64, 21)]:
        b, k, ks = ops.pil_bilinear_coeffs(n_in, n_out)
        ob, ok = O.pil_bilinear_coeffs(n_in, n_out)
        assert ks == len(ok[0]) and np.array_equal(b, np.asarray(ob)) and np.array_equal(k, np.asarray(ok)), (n_in, n_out)
    d = G.load("tta_r50c4_tiny")
    img = np.ascontiguousarray(d["image_u8"].transpose(1, 2, 0))
    n = 0
    for i in range(int(d["n_aug"])):
        a = d["aug%d_image" % i]  # [3, nh, nw] uint8, even i: not flipped (FLIP = True: pairs of (plain, mirrored))
        if i % 2 == 0:
            got = O.pil_bilinear_resize_u8(img, a.shape[1], a.shape[2]).transpose(2, 0, 1)
            assert np.array_equal(got, a), i
            n += 1
    assert n >= 1


def test_pcl_targets_and_loss_golden():
    """PCL: oracle/pcl_oracle.py against the reference's own PCL() (third_party/pcl.py) and its
    pcl_loss_cpu.cpp, on the golden cases where the reference's scikit-learn draw and numpy tie order coincide with
    the restated definitions (the generator counts the others; see the oracle's header)."""
    from oracle import pcl_oracle as PO

    d = G.load("pcl_unit")
    n = int(d["n_cases"])
    assert n >= 20 and n + int(d["n_kmeans_draw_differs"]) + int(d["n_tie_order_differs"]) == int(d["n_tried"])
    for i in range(n):
        g = lambda k: d["c%d_%s" % (i, k)]
        loss, dl, probs, t = PO.pcl_refine_loss(g("logits"), g("boxes"), g("last"), g("im_labels"))
        # integer / index outputs: bit-exact
        assert np.array_equal(t["labels"].astype(np.float32), g("labels")), i
        assert np.array_equal(t["gt_assignment"].astype(np.float32), g("gt_assignment")), i
        assert np.array_equal(t["pc_labels"].astype(np.float32), g("pc_labels")), i
        assert np.array_equal(t["pc_count"].astype(np.float32), g("pc_count")), i
        assert np.array_equal(t["cls_loss_weights"], g("cls_loss_weights")), i  # gathered values: exact
        np.testing.assert_allclose(t["pc_probs"], g("pc_probs"), rtol=1e-6)
        np.testing.assert_allclose(t["img_cls_loss_weights"], g("img_cls_loss_weights"), rtol=1e-6)
        assert abs(float(loss) - float(g("loss"))) <= 1e-5 * max(1.0, abs(float(g("loss")))), i
        gp = PO.pcl_loss_backward(probs, t, g("im_labels"))
        np.testing.assert_allclose(gp, g("dprobs"), rtol=1e-5, atol=1e-9)


def test_pcl_kmeans_divide_and_conquer_equals_all_pairs():
    """the O(n log n) search of kmeans_top_threshold returns the cut the all-pairs search returns (skewed MIL-like
    scores, softmax-like scores, duplicates, sizes up to 2000)"""
    from oracle import pcl_oracle as PO

    rs = np.random.RandomState(11)
    for n in (3, 4, 5, 9, 33, 64, 65, 200, 777, 2000):
        for rep in range(4 if n < 1000 else 2):
            if rep == 0:
                v = rs.rand(n) ** 6
            elif rep == 1:
                a = torch.softmax(torch.from_numpy(rs.randn(n, 5) * 3), 1)[:, 0].numpy()
                b = torch.softmax(torch.from_numpy(rs.randn(n) * 3), 0).numpy()
                v = a * b
            elif rep == 2:
                v = np.round(rs.rand(n) ** 2 * 16) / 16
            else:
                v = np.exp(rs.randn(n) * 3) * 1e-4
            v = np.maximum(v.astype(np.float32), np.float32(1e-9))
            t_all, _ = PO.kmeans_top_threshold_all_pairs(v)
            assert PO.kmeans_top_threshold(v) == t_all, (n, rep)


def test_pcl_kmeans_exact_optimum_bruteforce():
    """the k-means restatement is the global optimum: brute force over all cut pairs on small inputs, incl. duplicates,
    fewer distinct values than clusters, n < 3"""
    from oracle import pcl_oracle as PO

    rs = np.random.RandomState(5)
    for n in (1, 2, 3, 4, 7, 12, 30):
        for rep in range(6):
            v = (rs.rand(n) ** 3).astype(np.float32)
            if rep % 3 == 1:
                v = np.round(v * 4) / 4  # many duplicates
            if rep % 3 == 2:
                v[:] = v[0]
            thr = PO.kmeans_top_threshold(v)
            s = np.sort(v).astype(np.float64)
            cuts = [c for c in range(1, n) if s[c - 1] < s[c]]
            k = min(3, n, len(cuts) + 1)

            def sse(a, b):
                return ((s[a:b] - s[a:b].mean()) ** 2).sum()

            if k == 1:
                assert thr == s[0]
                continue
            best = None
            if k == 2:
                for c in cuts:
                    e = sse(0, c) + sse(c, n)
                    if best is None or e < best[0] - 1e-15:
                        best = (e, c)
            else:
                for i, c1 in enumerate(cuts):
                    for c2 in cuts[i + 1:]:
                        e = sse(0, c1) + sse(c1, c2) + sse(c2, n)
                        if best is None or e < best[0] - 1e-15:
                            best = (e, c2)
            top = int((v >= thr).sum())
            # same objective value (cut positions may differ only on exact ties of the objective)
            c_my = n - top
            if k == 2:
                e_my = sse(0, c_my) + sse(c_my, n)
            else:
                e_my = min(sse(0, c1) + sse(c1, c_my) + sse(c_my, n) for c1 in cuts if c1 < c_my)
            assert e_my <= best[0] + 1e-12, (n, rep)


# ------------------------------------------------------------------ CSCROIHeads
def test_csc_pool_table_sums_equal_direct_counts():
    """oracle/csc_ops.c is parity-unpinned (the reference's op is CUDA-only): pin the summed-area arithmetic by brute
    force - for random maps and boxes, every box sum read from the table equals the directly counted foreground pixels,
    and the score is the stated frame / context contrast of those counts."""
    import ctypes

    lib = O._lib()
    rs = np.random.RandomState(5)
    for trial in range(6):
        H, W = int(rs.randint(9, 70)), int(rs.randint(9, 90))
        m = torch.from_numpy(rs.rand(H, W).astype(np.float32))
        thr = np.float32(0.1 + 0.6 * rs.rand())
        table = torch.empty((H, W), dtype=torch.float32)
        lib.oracle_csc_integral(O._fp(m), O._fp(table), H, W, ctypes.c_float(thr))
        binm = (m.numpy() >= thr).astype(np.float64)
        assert np.array_equal(table.numpy().astype(np.float64), binm.cumsum(0).cumsum(1))
        for _ in range(60):
            x0, y0 = rs.rand() * (W + 6) - 3, rs.rand() * (H + 6) - 3
            roi = torch.tensor([0.0, x0, y0, x0 + rs.rand() * W, y0 + rs.rand() * H], dtype=torch.float32)
            boxes = (ctypes.c_int * 12)()
            sc = lib.oracle_csc_pool_one(O._fp(table), H, W, O._fp(roi), 1, ctypes.c_float(1.8), boxes)
            b = list(boxes)
            cnt = [binm[b[4 * i]: b[4 * i + 2] + 1, b[4 * i + 1]: b[4 * i + 3] + 1].sum() for i in range(3)]
            area = [(b[4 * i + 2] - b[4 * i] + 1) * (b[4 * i + 3] - b[4 * i + 1] + 1) for i in range(3)]
            assert b[0] <= b[4] and b[5] >= b[1] and b[8] <= b[0] and b[11] >= b[3]  # inner inside roi inside outer
            frame, ctx = cnt[0] - cnt[1], cnt[2] - cnt[0]
            want = frame / np.sqrt(max(area[0] - area[1], 1)) - ctx / np.sqrt(max(area[2] - area[0], 1))
            assert abs(sc - want) <= 1e-5 * max(1.0, abs(want)), (sc, want)


def test_csc_model_three_steps():
    """the unmodified reference CSCROIHeads (tests/golden/gen_golden.py `csc`; its CUDA-only csc_forward call served by
    oracle/csc_ops.c): image-gradient maps, weights, both losses, gradients and three SGD steps, the third one past
    WSL.CSC_MAX_ITER"""
    cfg = G.csc_case()
    d = G.load("model_csc_r18dc5_tiny")
    assert cfg.csc_max_iter == int(d["csc_max_iter"]) and cfg.csc_iter == int(d["iter0"]) and cfg.csc_tau == float(d["tau"])
    seed = int(d["seed"])
    p = O.seeded_params(O.param_shapes(cfg), seed)
    batch = G.batch_from(d)
    assert set(d["trainable"].tolist()) == set(O.trainable_names(p, cfg, 5))
    opt = O.SGDState(cfg)
    for step in range(3):
        losses, grads, aux = O.train_step(p, batch, cfg, opt, None, 5, return_aux=True)
        for k, v in losses.items():
            _close(v, float(d["step%d_%s" % (step, k)]), rtol=1e-4, atol=1e-9)
        _close(aux["scores"].detach(), d["step%d_scores" % step], rtol=1e-4, atol=1e-9)
        if step < 2:
            got, want = aux["cpgs"].numpy(), d["step%d_cpgs" % step]
            assert [c for c in range(4) if got[0, c].max() > 0] == [c for c in range(4) if want[0, c].max() > 0]
            assert np.abs(got - want).max() <= 1e-4
            # the weights count pixels over a threshold: equal unless a pixel sits within 1e-4 of it
            assert np.abs(aux["W_pos"].numpy() - d["step%d_W_pos" % step]).max() <= 2e-3
            assert np.abs(aux["W_neg"].numpy() - d["step%d_W_neg" % step]).max() <= 2e-3
        else:
            assert aux["cpgs"] is None and float(aux["W_pos"].min()) == 1.0 and float(aux["W_neg"].max()) == 0.0
        if step == 0:
            for n, g in grads.items():
                if "grad0." + n in d:
                    _close(g, d["grad0." + n], rtol=2e-3, atol=2e-6)
    for n in grads:
        _close(p[n].reshape(-1)[:2048], d["after3.head." + n], rtol=1e-4, atol=1e-6)
