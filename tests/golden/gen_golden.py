"""Generate golden vectors by running the UNMODIFIED reference (build container only).

    cd /root/repo && python tests/golden/gen_golden.py

Imports /root/reference through tests/golden/ref_harness.py (stand-ins for absent third-party
deps only), fills the reference model with name-seeded weights
(oracle.seeded_tensor — a numpy legacy-RandomState stream, reproducible anywhere), runs the
reference and stores inputs + expected outputs as small .npz fixtures in tests/golden/.
Fixtures are data only: no reference source text is stored.

Dropout (box_head.py:88-90) is patched to identity, or to a recorded mask sequence for the
`*_dropmask` case, because torch's global RNG cannot be matched on the GPU.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

import ref_harness as rh  # noqa: E402

rh.install()
from oracle import wsod_oracle as O  # noqa: E402

import torch.nn.functional as F  # noqa: E402
from detectron2.structures import Boxes, Instances  # noqa: E402
from detectron2.utils.events import EventStorage  # noqa: E402

_real_dropout = F.dropout


class DropoutPatch:
    def __init__(self, masks=None):
        self.masks = masks
        self.i = 0

    def __call__(self, x, p=0.5, training=True, inplace=False):
        if not training:
            return x
        if self.masks is None:
            return x
        m = self.masks[self.i % len(self.masks)]
        self.i += 1
        return x * m

    def __enter__(self):
        F.dropout = self
        torch.nn.functional.dropout = self
        return self

    def __exit__(self, *a):
        F.dropout = _real_dropout
        torch.nn.functional.dropout = _real_dropout


def fill_reference(model, seed):
    sd = model.state_dict()
    new = {}
    for n, t in sd.items():
        if n in ("pixel_mean", "pixel_std"):
            new[n] = t
            continue
        new[n] = O.seeded_tensor(n, tuple(t.shape), seed)
    model.load_state_dict(new)
    return {n: tuple(t.shape) for n, t in sd.items() if n not in ("pixel_mean", "pixel_std")}


def make_inputs(n_img, R, K, H, W, seed, n_gt=None):
    rs = np.random.RandomState(seed)
    batch = []
    for i in range(n_img):
        h = H - 8 * i  # ragged image sizes exercise the zero padding of ImageList.from_tensors
        w = W - 4 * i
        img = rs.randint(0, 256, size=(3, h, w)).astype(np.float32)
        x0 = rs.rand(R) * (w - 24)
        y0 = rs.rand(R) * (h - 24)
        bw = 12 + rs.rand(R) * (w - x0 - 12)
        bh = 12 + rs.rand(R) * (h - y0 - 12)
        boxes = np.stack([x0, y0, np.minimum(x0 + bw, w), np.minimum(y0 + bh, h)], 1).astype(np.float32)
        obj = np.sort(rs.rand(R).astype(np.float32))[::-1].copy()
        G = n_gt or rs.randint(1, 4)
        cls = rs.permutation(K)[:G].astype(np.int64)
        # gt boxes only feed logging in the reference (first label_and_sample_proposals call)
        gtb = boxes[rs.permutation(R)[:G]].copy()
        batch.append({"image": img, "proposal_boxes": boxes, "objectness_logits": obj, "gt_classes": cls,
                      "gt_boxes": gtb})
    return batch


def to_ref_inputs(batch):
    out = []
    for b in batch:
        h, w = b["image"].shape[1:]
        prop = Instances((h, w))
        prop.proposal_boxes = Boxes(torch.from_numpy(b["proposal_boxes"]))
        prop.objectness_logits = torch.from_numpy(b["objectness_logits"])
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(torch.from_numpy(b["gt_boxes"]))
        inst.gt_classes = torch.from_numpy(b["gt_classes"])
        out.append({"image": torch.from_numpy(b["image"]), "proposals": prop, "instances": inst, "height": h,
                    "width": w})
    return out


def flat_batch(batch, d):
    d["n_img"] = np.int64(len(batch))
    for i, b in enumerate(batch):
        for k, v in b.items():
            d["in%d_%s" % (i, k)] = v


def case_full_model(name, yaml_rel, opts, seed, n_img, R, H, W, dropmask=False, also_infer=True, steps=2):
    """Whole GeneralizedRCNNWSL: losses, grads of trainable params, 2 SGD steps, inference."""
    from detectron2.solver import build_optimizer

    cfg, model = rh.build_reference_model(yaml_rel, opts)
    shapes = fill_reference(model, seed)
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    batch = make_inputs(n_img, R, K, H, W, seed + 17)
    d = {"seed": np.int64(seed)}
    flat_batch(batch, d)
    d1, d2 = cfg.MODEL.ROI_BOX_HEAD.DAN_DIM
    masks = None
    if dropmask:
        rs = np.random.RandomState(seed + 5)
        masks = [torch.from_numpy((rs.rand(n_img * R, dd) > 0.5).astype(np.float32) * 2.0) for dd in (d1, d2)]
        d["dropmask0"] = masks[0].numpy()
        d["dropmask1"] = masks[1].numpy()
    model.train()
    opt = build_optimizer(cfg, model)
    tnames = [n for n, p in model.named_parameters() if p.requires_grad]
    d["trainable"] = np.array(tnames)
    with EventStorage() as storage, DropoutPatch(masks):
        for step in range(steps):
            opt.zero_grad()
            losses = model(to_ref_inputs(batch))
            total = sum(losses.values())
            total.backward()
            for k, v in losses.items():
                d["step%d_%s" % (step, k)] = np.float64(v.item())
            if step == 0:
                # intermediate activations for op-level parity
                for n, p in model.named_parameters():
                    if p.requires_grad and p.grad is not None and p.numel() <= 70000:
                        d["grad0." + n] = p.grad.detach().numpy().copy()
                    elif p.requires_grad and p.grad is not None:
                        g = p.grad.detach().reshape(-1)
                        d["gradhead0." + n] = g[:4096].numpy().copy()
                        d["gradsum0." + n] = np.float64(g.double().sum().item())
                        d["gradabs0." + n] = np.float64(g.double().abs().sum().item())
                    elif p.requires_grad:
                        d["gradnone0." + n] = np.int64(1)
            opt.step()
        for n, p in model.named_parameters():
            if p.requires_grad:
                f = p.detach().reshape(-1)
                d["after%d.head." % steps + n] = f[:2048].numpy().copy()
                d["after%d.sum." % steps + n] = np.float64(f.double().sum().item())
    # features / inference with the UPDATED weights
    if also_infer:
        model.eval()
        with torch.no_grad(), EventStorage():
            ins = to_ref_inputs(batch)
            for x in ins:
                x.pop("instances")
            results, all_scores, all_boxes = model.inference(ins, do_postprocess=False)
            images = model.preprocess_image(ins)
            feats = model.backbone(images.tensor)
            fk = list(feats.keys())[0]
            d["feat_name"] = np.array(fk)
            d["feat"] = feats[fk].numpy().copy()
        for i, r in enumerate(results):
            d["det%d_boxes" % i] = r.pred_boxes.tensor.numpy().copy()
            d["det%d_scores" % i] = r.scores.numpy().copy()
            d["det%d_classes" % i] = r.pred_classes.numpy().copy()
            d["all_scores%d" % i] = all_scores[i].numpy().copy()
        # the reference returns all_scores with a leading unsqueeze(0) per image (fast_rcnn.py:104-107)
    d["cfg_opts"] = np.array([yaml_rel] + list(opts))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", {k: v for k, v in d.items() if k.startswith("step")})


def case_csc(name, yaml_rel, opts, seed, R, H, W, tau, steps=3):
    """CSCROIHeads (roi_heads_csc.py) end to end: image-gradient maps (_forward_cpg), CSC weights, the two weighted BCE
    losses, SGD steps on both sides of WSL.CSC_MAX_ITER.

    The reference implements `_C.csc_forward` for CUDA only (wsl/layers/csrc/csc/csc.h), so it cannot execute here; the
    oracle's restatement (oracle/csc_ops.c) stands in for that ONE call.  Everything around it - the autograd maps, the
    clamps, the losses, the optimizer - is the unmodified reference, and the maps / weights it saw are recorded so the
    oracle and the HIP path are compared stage by stage.  Instance attributes set here (not code): `tau` (0.7 is never
    reached by a random-init model), `iter` = 1 (iteration 0 dumps debug PNGs through cv2)."""
    import tempfile

    import wsl._C as wsl_c
    from detectron2.solver import build_optimizer

    ocfg_like = O.OracleCfg()

    def csc_forward(cpgs, labels, preds, rois, tau_, debug, fg_threshold, mass_threshold, density_threshold, area_sqrt,
                    context_scale):
        ocfg_like.csc_fg_threshold, ocfg_like.csc_area_sqrt, ocfg_like.csc_context_scale = fg_threshold, area_sqrt, context_scale
        return O.csc_forward(cpgs, labels, preds, rois, ocfg_like)[0]

    wsl_c.csc_forward = csc_forward
    out_dir = tempfile.mkdtemp(prefix="csc_golden_")
    cfg, model = rh.build_reference_model(yaml_rel, list(opts) + ["OUTPUT_DIR", out_dir])
    fill_reference(model, seed)
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    batch = make_inputs(1, R, K, H, W, seed + 17, n_gt=2)
    d = {"seed": np.int64(seed), "tau": np.float64(tau), "csc_max_iter": np.int64(cfg.WSL.CSC_MAX_ITER),
         "iter0": np.int64(1)}
    flat_batch(batch, d)
    heads = model.roi_heads
    heads.tau = tau
    heads.iter = 1
    log = []
    orig = heads._forward_csc

    def spy(masks, pred_class_logits, proposals):
        res = orig(masks, pred_class_logits, proposals)
        log.append((None if masks is None else masks.detach().clone(), res[0].detach().clone(), res[1].detach().clone(),
                    pred_class_logits.detach().clone()))
        return res

    heads._forward_csc = spy
    model.train()
    opt = build_optimizer(cfg, model)
    tnames = [n for n, p in model.named_parameters() if p.requires_grad]
    d["trainable"] = np.array(tnames)
    with EventStorage(), DropoutPatch(None):
        for step in range(steps):
            opt.zero_grad()
            losses = model(to_ref_inputs(batch))
            sum(losses.values()).backward()
            for k, v in losses.items():
                d["step%d_%s" % (step, k)] = np.float64(v.item())
            masks, wpos, wneg, scores = log[-1]
            d["step%d_scores" % step] = scores.numpy()
            d["step%d_W_pos" % step] = wpos.numpy()
            d["step%d_W_neg" % step] = wneg.numpy()
            if masks is not None:
                d["step%d_cpgs" % step] = masks.numpy()
            if step == 0:
                for n, p in model.named_parameters():
                    if p.requires_grad and p.grad is not None and p.numel() <= 70000:
                        d["grad0." + n] = p.grad.detach().numpy().copy()
            opt.step()
        for n, p in model.named_parameters():
            if p.requires_grad:
                f = p.detach().reshape(-1)
                d["after%d.head." % steps + n] = f[:2048].numpy().copy()
                d["after%d.sum." % steps + n] = np.float64(f.double().sum().item())
    d["cfg_opts"] = np.array([yaml_rel] + list(opts))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", {k: v for k, v in d.items() if k.startswith("step") and np.ndim(v) == 0})
    for step in range(steps):
        sc = d["step%d_scores" % step].sum(0)
        print(" step", step, "image scores", sc, "labels", batch[0]["gt_classes"],
              "cpg classes", [int(c) for c in range(K) if "step%d_cpgs" % step in d and d["step%d_cpgs" % step][0, c].max() > 0],
              "W range", float((d["step%d_W_pos" % step] - d["step%d_W_neg" % step]).min()),
              float((d["step%d_W_pos" % step] - d["step%d_W_neg" % step]).max()))


def case_samplers(name):
    """the data-parallel partition: the reference's TrainingSampler / InferenceSampler per rank (this
    container runs one process: comm.get_rank / get_world_size are pointed at the rank being recorded),
    AspectRatioGroupedDataset batches and MapDataset's fallback draws"""
    rh.install()
    import importlib

    ds = importlib.import_module("detectron2.data.samplers.distributed_sampler")
    common = importlib.import_module("detectron2.data.common")
    import itertools

    d = {}
    for tag, size, seed, world, shuffle, n in (("a", 37, 11, 4, True, 25), ("b", 5, 3, 2, False, 12), ("c", 8, 5, 1, True, 20)):
        for r in range(world):
            ds.comm.get_rank, ds.comm.get_world_size = (lambda r=r: r), (lambda world=world: world)
            smp = ds.TrainingSampler(size, shuffle=shuffle, seed=seed)
            d["train_%s_r%d" % (tag, r)] = np.array([int(x) for x in itertools.islice(iter(smp), n)], dtype=np.int64)
        d["train_%s_cfg" % tag] = np.array([size, seed, world, int(shuffle), n], dtype=np.int64)
    for tag, size, world in (("a", 10, 4), ("b", 3, 4), ("c", 8, 8), ("d", 1, 1), ("e", 4952, 8)):
        for r in range(world):
            ds.comm.get_rank, ds.comm.get_world_size = (lambda r=r: r), (lambda world=world: world)
            smp = ds.InferenceSampler(size)
            d["infer_%s_r%d" % (tag, r)] = np.array(list(smp), dtype=np.int64)
        d["infer_%s_cfg" % tag] = np.array([size, world], dtype=np.int64)
    rs = np.random.RandomState(7)
    wh = rs.randint(100, 500, size=(23, 2))
    wh[5] = (300, 300)
    items = [{"width": int(w), "height": int(h), "id": i} for i, (w, h) in enumerate(wh)]
    batches = [[x["id"] for x in b] for b in common.AspectRatioGroupedDataset(items, 3)]
    d["group_wh"] = wh.astype(np.int64)
    d["group_batches"] = np.array(batches, dtype=np.int64)
    md = common.MapDataset(list(range(10)), lambda x: None if x % 3 == 0 else x * 10)
    d["map_out"] = np.array([md[i] for i in range(10)] + [md[i] for i in range(10)], dtype=np.int64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, {k: v.tolist() for k, v in d.items() if k in ("train_a_r1", "infer_a_r3", "map_out")})


def case_pcl_unit(name, seed, n_try=40):
    """PCL targets and loss, op level: the reference's own PCL() (third_party/pcl.py, with the scikit-learn installed
    here) and its pcl_loss_cpu.cpp (compiled in place).  Two steps of PCL() are not functions of their inputs
    (oracle/pcl_oracle.py header: sklearn's seeded k-means, numpy's unstable argsort on ties); cases where the
    reference's draw differs from the restated definition are counted and dropped, the rest are the golden."""
    import sklearn
    from oracle import pcl_oracle as PO
    from wsl.modeling.roi_heads.third_party import pcl as ref_pcl
    import wsl._C as wsl_c

    rs = np.random.RandomState(seed)
    d = {"sklearn": np.array(sklearn.__version__), "numpy": np.array(np.__version__)}
    kept = n_km_diff = n_other_diff = 0
    for t in range(n_try):
        R = int(rs.choice([24, 60, 150, 400]))
        K = int(rs.choice([4, 20]))
        W, H = 200.0, 160.0
        # clustered boxes (jittered copies of a few seeds) so that the IoU graph has real cliques
        nseed = max(3, R // 12)
        sx0 = rs.rand(nseed) * (W - 60)
        sy0 = rs.rand(nseed) * (H - 60)
        sw = 30 + rs.rand(nseed) * (W - sx0 - 30)
        sh = 30 + rs.rand(nseed) * (H - sy0 - 30)
        pick = rs.randint(0, nseed, size=R)
        jit = rs.randn(R, 4) * 6.0
        x0 = np.clip(sx0[pick] + jit[:, 0], 0, W - 21)
        y0 = np.clip(sy0[pick] + jit[:, 1], 0, H - 21)
        x1 = np.clip(sx0[pick] + sw[pick] + jit[:, 2], x0 + 20, W)
        y1 = np.clip(sy0[pick] + sh[pick] + jit[:, 3], y0 + 20, H)
        boxes = np.stack([x0, y0, x1, y1], 1).astype(np.float32)
        G = int(rs.randint(1, 4))
        im_labels = np.zeros((1, K), dtype=np.float32)
        im_labels[0, rs.permutation(K)[:G]] = 1
        first = (t % 2 == 0)
        if first:  # WSDDN-shaped scores [R, K]: softmax over classes x softmax over proposals
            a = torch.from_numpy(rs.randn(R, K).astype(np.float32) * 2.0)
            b = torch.from_numpy(rs.randn(R, K).astype(np.float32) * 3.0)
            last = (torch.softmax(a, 1) * torch.softmax(b, 0)).numpy()
        else:  # a previous refinement's softmax [R, K+1]
            last = torch.softmax(torch.from_numpy(rs.randn(R, K + 1).astype(np.float32) * 3.0), 1).numpy()
        logits = rs.randn(R, K + 1).astype(np.float32) * 2.0
        probs = torch.softmax(torch.from_numpy(logits), 1)
        ref = ref_pcl.PCL(boxes.copy(), torch.from_numpy(last.copy()), im_labels.copy(), probs.clone())
        mine = PO.pcl_targets(boxes, last, im_labels[0], probs.numpy())
        def same_as_ref(m):
            return (np.array_equal(ref["labels"][0], m["labels"].astype(np.float32)) and
                    np.array_equal(ref["gt_assignment"][0], m["gt_assignment"].astype(np.float32)) and
                    np.array_equal(ref["pc_labels"][0], m["pc_labels"].astype(np.float32)) and
                    np.array_equal(ref["pc_count"][0], m["pc_count"].astype(np.float32)) and
                    np.array_equal(ref["cls_loss_weights"][0], m["cls_loss_weights"]))

        if not same_as_ref(mine):
            # classify the difference by substituting the two non-functional steps with what THIS machine's numpy and
            # scikit-learn do: with both substituted the restatement must reproduce the reference on every case
            keep_pick, keep_km = PO._argmax_last, PO.kmeans_top_threshold
            try:
                PO._argmax_last = lambda x: int(np.asarray(x, dtype=np.float32).argsort()[::-1][0])
                tie_only = same_as_ref(PO.pcl_targets(boxes, last, im_labels[0], probs.numpy()))
                PO.kmeans_top_threshold = lambda v: v[ref_pcl._get_top_ranking_propoals(
                    np.asarray(v).reshape(-1, 1).copy())].min()
                both = same_as_ref(PO.pcl_targets(boxes, last, im_labels[0], probs.numpy()))
            finally:
                PO._argmax_last, PO.kmeans_top_threshold = keep_pick, keep_km
            assert both, "restatement differs from the reference beyond k-means draws / tie order (case %d)" % t
            if tie_only:
                n_other_diff += 1
            else:
                n_km_diff += 1
            continue
        # loss + gradient from the reference's C++
        tt = {k: torch.from_numpy(v) for k, v in ref.items()}
        out = torch.zeros(1, K + 1)
        pp = probs.clone()
        wsl_c.pcl_loss_forward(pp, tt["labels"], tt["cls_loss_weights"], tt["pc_labels"], tt["pc_probs"],
                               tt["img_cls_loss_weights"], tt["im_labels_real"], out)
        loss = out.sum() / R  # pcl_loss.py:51
        gin = torch.zeros(R, K + 1)
        wsl_c.pcl_loss_backward(pp, tt["labels"], tt["cls_loss_weights"], tt["gt_assignment"], tt["pc_labels"],
                                tt["pc_probs"], tt["pc_count"], tt["img_cls_loss_weights"], tt["im_labels_real"],
                                torch.ones(()), gin)
        gin /= R  # pcl_loss.py:89
        pre = "c%d_" % kept
        d[pre + "boxes"], d[pre + "last"], d[pre + "im_labels"], d[pre + "logits"] = boxes, last, im_labels[0], logits
        for k in ("labels", "cls_loss_weights", "gt_assignment", "pc_labels", "pc_probs", "pc_count",
                  "img_cls_loss_weights"):
            d[pre + k] = ref[k][0]
        d[pre + "loss"] = np.float32(loss.item())
        d[pre + "dprobs"] = gin.numpy()
        kept += 1
    d["n_cases"] = np.int64(kept)
    d["n_tried"] = np.int64(n_try)
    d["n_kmeans_draw_differs"] = np.int64(n_km_diff)
    d["n_tie_order_differs"] = np.int64(n_other_diff)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; kept", kept, "of", n_try, "| sklearn draw differs:",
          n_km_diff, "| other (tie order):", n_other_diff)


def case_heads_detail(name, seed, K=4, R=40, n_img=2):
    """OICRROIHeads through explicit kwargs (every class is @configurable): op-level intermediates —
    WSDDN scores, pgt indices incl. a forced TIE, labels, per-refinement losses, grads wrt logits."""
    from detectron2.layers import ShapeSpec
    from detectron2.modeling.box_regression import Box2BoxTransform
    from detectron2.modeling.matcher import Matcher
    from detectron2.modeling.poolers import ROIPooler
    from wsl.modeling.roi_heads.box_head import DiscriminativeAdaptionNeck
    from wsl.modeling.roi_heads.fast_rcnn import OICROutputLayers, WSDDNOutputLayers
    from wsl.modeling.roi_heads.roi_heads_oicr import OICRROIHeads

    C, P, D1, D2 = 6, 3, 24, 32
    for reg in (False, True):
        refine_reg = [False, False, reg]
        b2b = Box2BoxTransform((10.0, 10.0, 5.0, 5.0))
        neck = DiscriminativeAdaptionNeck(ShapeSpec(channels=C, height=P, width=P), conv_dims=[], fc_dims=[D1, D2])
        pred = WSDDNOutputLayers(ShapeSpec(channels=D2), box2box_transform=b2b, num_classes=K, mean_loss=True)
        refs = [OICROutputLayers(ShapeSpec(channels=D2), box2box_transform=b2b, num_classes=K, refine_k=k,
                                 refine_reg=refine_reg) for k in range(3)]
        heads = OICRROIHeads(box_in_features=["f"], box_pooler=ROIPooler(P, (0.125,), 0, "ROIPool"), box_head=neck,
                             box_predictor=pred, refine_K=3, refine_reg=refine_reg, box_refinery=refs,
                             num_classes=K, batch_size_per_image=4096, positive_fraction=1.0,
                             proposal_matcher=Matcher([0.5], [0, 1], False), proposal_append_gt=False)
        sd = heads.state_dict()
        heads.load_state_dict({n: O.seeded_tensor("roi_heads." + n, tuple(t.shape), seed) for n, t in sd.items()})
        heads.train()
        H = W = 96
        batch = make_inputs(n_img, R, K, H, W, seed + 3, n_gt=2)
        # force an exact tie for pgt mining: duplicate a proposal row (same box => same features => same score)
        for b in batch:
            b["image"] = b["image"][:, :H, :W] if b["image"].shape[1] >= H else b["image"]
            b["proposal_boxes"][7] = b["proposal_boxes"][3]
            b["objectness_logits"][7] = b["objectness_logits"][3]
        rs = np.random.RandomState(seed + 9)
        feat = rs.standard_normal((n_img, C, H // 8, W // 8)).astype(np.float32)
        ins = to_ref_inputs(batch)
        props = [x["proposals"] for x in ins]
        tg = [x["instances"] for x in ins]

        class _IL:
            pass

        d = {"seed": np.int64(seed), "feat": feat, "K": np.int64(K), "refine_reg": np.array(refine_reg)}
        flat_batch(batch, d)
        grads = {}
        with EventStorage(), DropoutPatch(None):
            logits_store = []
            hooks = []
            for k in range(3):
                def mk(k):
                    def hook(mod, inp, out):
                        out[0].retain_grad()
                        logits_store.append(out)
                    return hook
                hooks.append(refs[k].register_forward_hook(mk(k)))
            wsc = []
            def wsddn_hook(m, i, o):
                o[0].retain_grad()
                wsc.append(o[0])

            hooks.append(pred.register_forward_hook(wsddn_hook))
            _, losses = heads(None, {"f": torch.from_numpy(feat)}, props, tg)
            total = sum(losses.values())
            total.backward()
        for k, v in losses.items():
            d[k] = np.float64(v.item())
        d["wsddn_scores"] = wsc[0].detach().numpy().copy()
        d["wsddn_scores_grad"] = wsc[0].grad.numpy().copy()
        for k in range(3):
            d["logits_r%d" % k] = logits_store[k][0].detach().numpy().copy()
            d["logits_r%d_grad" % k] = logits_store[k][0].grad.numpy().copy()
            if refine_reg[k]:
                d["deltas_r%d" % k] = logits_store[k][1].detach().numpy().copy()
        d["img_scores"] = heads.pred_class_img_logits.numpy().copy()
        for n, p in heads.named_parameters():
            if p.grad is not None:
                d["grad.roi_heads." + n] = p.grad.numpy().copy()
            else:
                d["gradnone.roi_heads." + n] = np.int64(1)
        path = os.path.join(HERE, "%s_reg%d.npz" % (name, int(reg)))
        np.savez_compressed(path, **d)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB", {k: float(v) for k, v in losses.items()})


def case_ops(name, seed):
    """Op-level vectors from reference code that needs no model: ROIAlign (compiled reference C++,
    incl. tests/layers/test_roi_align.py ramp case), pairwise_iou, Matcher, Box2BoxTransform,
    FrozenBatchNorm2d, fast_rcnn_inference_single_image, apply_deltas with zero deltas."""
    from detectron2.layers import FrozenBatchNorm2d
    from detectron2.layers.roi_align import ROIAlign
    from detectron2.modeling.box_regression import Box2BoxTransform
    from detectron2.modeling.matcher import Matcher
    from detectron2.structures import pairwise_iou
    from wsl.modeling.roi_heads.fast_rcnn import fast_rcnn_inference_single_image

    rs = np.random.RandomState(seed)
    d = {}
    # ROIAlign ramp KAT (tests/layers/test_roi_align.py:13-45)
    ramp = np.arange(25, dtype=np.float32).reshape(1, 1, 5, 5)
    rois = np.array([[0, 1, 1, 3, 3]], dtype=np.float32)
    for al in (False, True):
        out = ROIAlign((4, 4), 1.0, 0, aligned=al)(torch.from_numpy(ramp), torch.from_numpy(rois))
        d["ra_ramp_aligned%d" % int(al)] = out.numpy().copy()
    # random ROIAlign cases fwd + bwd
    feat = rs.standard_normal((2, 5, 13, 17)).astype(np.float32)
    R = 24
    x0 = rs.rand(R) * 100
    y0 = rs.rand(R) * 70
    rois = np.stack([rs.randint(0, 2, R).astype(np.float32), x0, y0, x0 + 2 + rs.rand(R) * 120, y0 + 2 + rs.rand(R) * 90],
                    1).astype(np.float32)
    rois[0, 1:] = [-20, -10, 400, 300]  # far outside the map
    rois[1, 1:] = [30, 30, 30, 30]  # empty box
    d["ra_feat"] = feat
    d["ra_rois"] = rois
    for al in (False, True):
        for sr in (0, 2):
            ft = torch.from_numpy(feat).clone().requires_grad_(True)
            out = ROIAlign((7, 7), 0.125, sr, aligned=al)(ft, torch.from_numpy(rois))
            gout = torch.from_numpy(rs.standard_normal(tuple(out.shape)).astype(np.float32))
            out.backward(gout)
            key = "ra_al%d_sr%d" % (int(al), sr)
            d[key + "_out"] = out.detach().numpy().copy()
            d[key + "_gout"] = gout.numpy().copy()
            d[key + "_gin"] = ft.grad.numpy().copy()
    # pairwise_iou / matcher
    b1 = np.array([[0, 0, 10, 10], [5, 5, 20, 25], [30, 30, 31, 31], [0, 0, 0, 0]], dtype=np.float32)
    x0 = rs.rand(50) * 30
    y0 = rs.rand(50) * 30
    b2 = np.stack([x0, y0, x0 + rs.rand(50) * 20, y0 + rs.rand(50) * 20], 1).astype(np.float32)
    b2[0] = b1[0]
    b2[1] = [100, 100, 110, 110]  # zero-IoU column
    iou = pairwise_iou(Boxes(torch.from_numpy(b1)), Boxes(torch.from_numpy(b2)))
    m, l = Matcher([0.5], [0, 1], False)(iou)
    d.update(iou_b1=b1, iou_b2=b2, iou=iou.numpy().copy(), match_idx=m.numpy().copy(), match_label=l.numpy().copy())
    # tests/modeling/test_matcher.py:14-28 golden (RPN thresholds, low-quality on) kept as a KAT of the argmax path
    mq = torch.tensor([[0.15, 0.45, 0.2, 0.6], [0.3, 0.65, 0.05, 0.1], [0.05, 0.4, 0.25, 0.4]])
    m2, l2 = Matcher([0.3, 0.7], [0, -1, 1], True)(mq)
    d.update(mq=mq.numpy(), mq_idx=m2.numpy().copy(), mq_label=l2.numpy().copy())
    # Box2BoxTransform
    t = Box2BoxTransform((10.0, 10.0, 5.0, 5.0))
    src = b2[:20].copy()
    src[:, 2:] += 1.0
    dst = b2[20:40].copy()
    dst[:, 2:] += 1.0
    deltas = t.get_deltas(torch.from_numpy(src), torch.from_numpy(dst))
    d.update(b2b_src=src, b2b_dst=dst, b2b_deltas=deltas.numpy().copy())
    dd = torch.from_numpy(rs.standard_normal((20, 12)).astype(np.float32) * 2)
    dd[0, 2] = 50.0  # hits the scale clamp
    d["b2b_apply_in"] = dd.numpy().copy()
    d["b2b_apply_out"] = t.apply_deltas(dd, torch.from_numpy(src)).numpy().copy()
    d["b2b_apply_zero"] = t.apply_deltas(torch.zeros(20, 8), torch.from_numpy(src)).numpy().copy()
    # FrozenBN
    bn = FrozenBatchNorm2d(5)
    bn.load_state_dict({k: O.seeded_tensor("x.norm." + k, (5,), seed) for k in ("weight", "bias", "running_mean", "running_var")})
    d["fbn_out"] = bn(torch.from_numpy(feat)).numpy().copy()
    # inference tail incl. non-finite rows
    R, K = 300, 6
    x0 = rs.rand(R) * 150
    y0 = rs.rand(R) * 100
    pb = np.stack([x0, y0, x0 + 5 + rs.rand(R) * 80, y0 + 5 + rs.rand(R) * 60], 1).astype(np.float32)
    boxes = np.tile(pb, (1, K)).astype(np.float32) + rs.standard_normal((R, 4 * K)).astype(np.float32)
    sc = torch.softmax(torch.from_numpy(rs.standard_normal((R, K + 1)).astype(np.float32) * 3), 1).numpy()
    sc[5, 2] = np.nan
    boxes[9, 1] = np.inf
    sc[20] = sc[21]  # tied scores
    boxes[20] = boxes[21]
    res, kept, _, _ = fast_rcnn_inference_single_image(torch.from_numpy(boxes.copy()), torch.from_numpy(sc.copy()),
                                                       (120, 200), 1e-5, 0.3, 100)
    d.update(inf_boxes=boxes, inf_scores=sc, inf_out_boxes=res.pred_boxes.tensor.numpy().copy(),
             inf_out_scores=res.scores.numpy().copy(), inf_out_classes=res.pred_classes.numpy().copy(),
             inf_out_rows=kept.numpy().copy())
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


TINY_R50 = ["MODEL.RESNETS.STEM_OUT_CHANNELS", "8", "MODEL.RESNETS.RES2_OUT_CHANNELS", "32",
            "MODEL.RESNETS.WIDTH_PER_GROUP", "8", "MODEL.ROI_BOX_HEAD.DAN_DIM", "[48, 64]",
            "MODEL.ROI_HEADS.NUM_CLASSES", "5"]
C4 = ["MODEL.RESNETS.OUT_FEATURES", "['res4']", "MODEL.ROI_HEADS.IN_FEATURES", "['res4']",
      "MODEL.RESNETS.RES5_DILATION", "1"]

def case_tta(name, yaml_rel, opts, seed, R, H, W, min_sizes, max_size, topk):
    """GeneralizedRCNNWithTTAAVG (projects/WSL/wsl/modeling/test_time_augmentation_avg.py:139-321) on one uint8 image:
    the reference's own mapper (ResizeShortestEdge via PIL + horizontal flip, transformed/clipped/top-k proposals), one
    inference per augmentation, boxes mapped back and averaged, scores averaged, final NMS / top-k."""
    from wsl.modeling.test_time_augmentation_avg import DatasetMapperTTAAVG, GeneralizedRCNNWithTTAAVG

    o = list(opts) + ["TEST.AUG.ENABLED", "True", "TEST.AUG.MIN_SIZES", str(tuple(min_sizes)), "TEST.AUG.MAX_SIZE",
                      str(max_size), "TEST.AUG.FLIP", "True", "DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST", str(topk)]
    cfg, model = rh.build_reference_model(yaml_rel, o)
    fill_reference(model, seed)
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    b = make_inputs(1, R, K, H, W, seed + 17)[0]
    img8 = b["image"].astype(np.uint8)
    d = {"seed": np.int64(seed), "image_u8": img8, "proposal_boxes": b["proposal_boxes"],
         "objectness_logits": b["objectness_logits"], "min_sizes": np.array(min_sizes), "max_size": np.int64(max_size),
         "topk": np.int64(topk)}
    prop = Instances((H, W))
    prop.proposal_boxes = Boxes(torch.from_numpy(b["proposal_boxes"]))
    prop.objectness_logits = torch.from_numpy(b["objectness_logits"])
    inp = {"image": torch.from_numpy(img8), "proposals": prop, "height": H, "width": W}
    model.eval()
    tta = GeneralizedRCNNWithTTAAVG(cfg, model)
    with torch.no_grad(), EventStorage():
        aug, tfms = tta._get_augmented_inputs(dict(inp))
        d["n_aug"] = np.int64(len(aug))
        for i, a in enumerate(aug):
            d["aug%d_image" % i] = a["image"].numpy().copy()
            d["aug%d_boxes" % i] = a["proposals"].proposal_boxes.tensor.numpy().copy()
            d["aug%d_obj" % i] = a["proposals"].objectness_logits.numpy().copy()
        all_boxes, all_scores, _ = tta._get_augmented_boxes(aug, tfms)
        d["avg_boxes"] = all_boxes.numpy().copy()
        d["avg_scores"] = all_scores.numpy().copy()
        out = tta([dict(inp)])[0]["instances"]
    d["det_boxes"] = out.pred_boxes.tensor.numpy().copy()
    d["det_scores"] = out.scores.numpy().copy()
    d["det_classes"] = out.pred_classes.numpy().copy()
    d["cfg_opts"] = np.array([yaml_rel] + list(o))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "n_aug", len(aug), "dets", len(out),
          [tuple(a["image"].shape) for a in aug])


def case_data(name, yaml_rel, opts, seed):
    """The data path either side of the model: load_proposals_into_dataset
    (detectron2/data/build.py:102-153) on a Detectron1-keyed proposal pickle, then the reference DatasetMapper
    (detectron2/data/dataset_mapper.py + detection_utils.py: read_image, ResizeShortestEdge / RandomFlip /
    RandomBrightness / RandomSaturation of this fork's build_augmentation, transform_instance_annotations,
    transform_proposals with unique_boxes) in train mode under a fixed numpy seed and in test mode."""
    import pickle
    import tempfile

    from PIL import Image

    import detectron2.data.build as B
    import detectron2.data.dataset_mapper as DM

    o = list(opts) + ["INPUT.MIN_SIZE_TRAIN", "(48, 64, 80)", "INPUT.MAX_SIZE_TRAIN", "120", "INPUT.MIN_SIZE_TEST", "64",
                      "INPUT.MAX_SIZE_TEST", "100", "DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TRAIN", "30",
                      "DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST", "25"]
    cfg, _ = rh.build_reference_model(yaml_rel, o)
    rs = np.random.RandomState(seed)
    H, W, R = 60, 84, 40
    rgb = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    tmp = tempfile.mkdtemp()
    fn = os.path.join(tmp, "000123.png")
    Image.fromarray(rgb).save(fn)
    x0, y0 = rs.rand(R) * (W - 20), rs.rand(R) * (H - 20)
    boxes = np.stack([x0, y0, x0 + 4 + rs.rand(R) * (W - x0 - 4), y0 + 4 + rs.rand(R) * (H - y0 - 4)], 1).astype(np.float32)
    boxes[7] = boxes[3]          # exact duplicate -> unique_boxes
    boxes[11] = boxes[5] + 0.2   # duplicate after rounding
    boxes[13, 2] = boxes[13, 0]  # empty box
    scores = rs.rand(R).astype(np.float32)
    other = np.zeros((3, 4), dtype=np.float32)
    pf = os.path.join(tmp, "props.pkl")
    with open(pf, "wb") as f:  # Detectron1 key names + an image that is not in the dataset
        pickle.dump({"indexes": [7, 123], "boxes": [other, boxes], "scores": [np.zeros(3, np.float32), scores]}, f)
    annos = [{"bbox": [10.0, 8.0, 50.0, 40.0], "bbox_mode": 0, "category_id": 3},
             {"bbox": [30.5, 20.25, 80.0, 58.0], "bbox_mode": 0, "category_id": 1},
             {"bbox": [5.0, 5.0, 20.0, 20.0], "bbox_mode": 0, "category_id": 2, "iscrowd": 1}]
    rec = {"file_name": fn, "height": H, "width": W, "image_id": 123, "annotations": annos}
    d = {"rgb": rgb, "boxes": boxes, "scores": scores, "seed": np.int64(seed)}
    recs = B.load_proposals_into_dataset([dict(rec)], pf)
    d["loaded_boxes"] = recs[0]["proposal_boxes"]
    d["loaded_logits"] = recs[0]["proposal_objectness_logits"]
    for tag, is_train, nrep in (("train", True, 4), ("test", False, 1)):
        mapper = DM.DatasetMapper(cfg, is_train)
        np.random.seed(seed)
        for rep in range(nrep):
            out = mapper(recs[0])
            k = "%s%d_" % (tag, rep)
            d[k + "image"] = out["image"].numpy()
            d[k + "prop_boxes"] = out["proposals"].proposal_boxes.tensor.numpy()
            d[k + "prop_logits"] = out["proposals"].objectness_logits.numpy()
            if is_train:
                d[k + "gt_boxes"] = out["instances"].gt_boxes.tensor.numpy()
                d[k + "gt_classes"] = out["instances"].gt_classes.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", [tuple(d["train%d_image" % i].shape) for i in range(4)],
          len(d["train0_prop_boxes"]), len(d["test0_prop_boxes"]))


sys.path.insert(0, os.path.join(HERE, ".."))
from golden_util import voc_fixture  # noqa: E402  (shared with the tests)


def case_voc_eval(name, seed):
    """detectron2/evaluation/pascal_voc_evaluation.py: the text lines of PascalVOCDetectionEvaluator.process, then
    voc_eval / voc_eval_corloc per class and IoU threshold, then the aggregation of evaluate()"""
    import tempfile

    import detectron2.evaluation.pascal_voc_evaluation as V

    classes, annos, dets = voc_fixture(seed)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "Annotations"))
    for iid, objs in annos.items():
        body = "".join("<object><name>%s</name><pose>Unspecified</pose><truncated>0</truncated><difficult>%d</difficult>"
                       "<bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>"
                       % (n, df, b[0], b[1], b[2], b[3]) for n, df, b in objs)
        with open(os.path.join(tmp, "Annotations", iid + ".xml"), "w") as f:
            f.write("<annotation><filename>%s.jpg</filename>%s</annotation>" % (iid, body))
    with open(os.path.join(tmp, "test.txt"), "w") as f:
        f.write("\n".join(annos.keys()) + "\n")
    lines = {c: [] for c in range(len(classes))}
    for c, iid, score, box in dets:  # PascalVOCDetectionEvaluator.process (:52-66)
        xmin, ymin, xmax, ymax = np.array(box, dtype=np.float32)
        xmin += 1
        ymin += 1
        lines[c].append(f"{iid} {score:.3f} {xmin:.1f} {ymin:.1f} {xmax:.1f} {ymax:.1f}")
    d = {"seed": np.int64(seed)}
    for year07 in (True, False):
        aps, cls_ = {}, {}
        for ci, cname in enumerate(classes):
            with open(os.path.join(tmp, cname + ".txt"), "w") as f:
                f.write("\n".join(lines[ci] or [""]))
            for thr in range(50, 100, 5):
                rec, prec, ap = V.voc_eval(os.path.join(tmp, "{}.txt"), os.path.join(tmp, "Annotations", "{}.xml"),
                                           os.path.join(tmp, "test.txt"), cname, ovthresh=thr / 100.0,
                                           use_07_metric=year07)
                aps.setdefault(thr, []).append(ap * 100)
                cl = V.voc_eval_corloc(os.path.join(tmp, "{}.txt"), os.path.join(tmp, "Annotations", "{}.xml"),
                                       os.path.join(tmp, "test.txt"), cname, ovthresh=thr / 100.0, use_07_metric=year07)
                cls_.setdefault(thr, []).append(cl * 100)
                if thr == 50:
                    d["rec_%d_%s" % (year07, cname)], d["prec_%d_%s" % (year07, cname)] = rec, prec
        tag = "y07" if year07 else "y12"
        d["ap_" + tag] = np.array([aps[t] for t in range(50, 100, 5)])
        d["corloc_" + tag] = np.array([cls_[t] for t in range(50, 100, 5)])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, "AP50 (07):", d["ap_y07"][0], "CorLoc50:", d["corloc_y07"][0])


def c2_style_name(model_key):
    """inverse of the naming the released WSL checkpoints use (projects/WSL/tools/convert_resnet_ws_c2.py output, i.e.
    Caffe2 blob names with the stem renamed to stem_convN and fc6/fc7 to fc1/fc2); None for keys such files lack"""
    k = model_key
    if k.startswith("backbone."):
        k = k[len("backbone."):]
    elif k.startswith("roi_heads.box_head."):
        k = k[len("roi_heads.box_head."):]
    else:
        return None
    parts = k.split(".")
    leaf = {"weight": "w", "bias": "b"}[parts[-1]] if parts[-2] != "norm" else \
        {"weight": "bn_s", "bias": "bn_b", "running_mean": "bn_rm", "running_var": "bn_riv"}[parts[-1]]
    body = parts[:-1] if parts[-2] != "norm" else parts[:-2]
    body = [{"conv1": "branch2a", "conv2": "branch2b", "conv3": "branch2c", "shortcut": "branch1"}.get(x, x)
            if body[0].startswith("res") else x for x in body]
    if body[0] == "stem":
        body = ["stem_" + body[1]]
    return "_".join(body + [leaf])


def case_checkpoint(name, yaml_rel, opts):
    """detectron2/checkpoint/c2_model_loading.py: convert_c2_detectron_names + align_and_update_state_dicts run on a
    synthetic WSL-style Caffe2 checkpoint (every tensor filled with its own index) for the tiny R50-C4 model, plus
    decoys: a *_momentum blob, an unrelated blob, a shape mismatch, and a d2-format (no conversion) pass."""
    import importlib.util

    cfg, model = rh.build_reference_model(yaml_rel, opts)
    spec = importlib.util.spec_from_file_location("ref_c2_model_loading",
                                                  os.path.join(rh.REF, "detectron2", "checkpoint", "c2_model_loading.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sd = model.state_dict()
    ckpt, ckeys, cshapes = {}, [], []
    for i, (k, t) in enumerate(sd.items()):
        c = c2_style_name(k)
        if c is None:
            continue
        shape = tuple(t.shape)
        if k.endswith("res3.1.conv2.weight"):
            shape = shape[:-1] + (shape[-1] + 1,)  # shape mismatch: must be skipped with a warning
        ckpt[c] = torch.full(shape, float(len(ckeys) + 1))
        ckeys.append(c)
        cshapes.append(np.array(shape))
    for extra, shape in (("res2_0_branch2a_w_momentum", (3,)), ("pred_w", (7, 5)), ("conv5_mask_w", (2, 2))):
        ckpt[extra] = torch.full(shape, float(len(ckeys) + 1))
        ckeys.append(extra)
        cshapes.append(np.array(shape))
    d = {"ckpt_keys": np.array(ckeys), "model_keys": np.array(list(sd.keys()))}
    for i, sh in enumerate(cshapes):
        d["ckpt_shape%d" % i] = sh
    for tag, c2 in (("c2", True), ("d2", False)):
        msd = {k: torch.zeros_like(v) for k, v in sd.items()}
        src = {k: v for k, v in ckpt.items() if not k.endswith("_momentum")} if c2 else \
            {k.replace("backbone.", ""): torch.full(tuple(v.shape), float(j + 1)) for j, (k, v) in enumerate(sd.items())
             if k.startswith("backbone.")}
        if not c2:
            d["d2_keys"] = np.array(list(src.keys()))
        mod.align_and_update_state_dicts(msd, src, c2_conversion=c2)
        # every loaded tensor is constant = 1-based index of its source key
        d["map_" + tag] = np.array([int(v.reshape(-1)[0].item()) for v in msd.values()], dtype=np.int64)
    new_w, new_to_orig = mod.convert_c2_detectron_names({k: v for k, v in ckpt.items() if not k.endswith("_momentum")})
    d["renamed"] = np.array(sorted(new_w.keys()))
    d["renamed_orig"] = np.array([new_to_orig[k] for k in sorted(new_w.keys())])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, "loaded c2:", int((d["map_c2"] > 0).sum()), "of", len(sd), " d2:", int((d["map_d2"] > 0).sum()))


def case_convert(name, seed):
    """The reference's offline converters run UNMODIFIED on synthetic files (projects/WSL/tools/): proposal_convert.py's
    convert_ss_box / convert_mcg_box on .mat files written here with scipy (1-indexed (y1, x1, y2, x2) boxes; an image
    with a single proposal; the dataset catalogue replaced by a list of image ids), and the three checkpoint key
    renaming scripts - convert_resnet_ws_pth.py, convert_resnet_ws_c2.py, convert_vgg.py (module-level scripts: run
    with runpy under their own sys.argv).  Recorded: inputs and the files the scripts wrote."""
    import importlib.util
    import pickle
    import runpy
    import tempfile

    import scipy.io

    tools = os.path.join(rh.REF, "projects", "WSL", "tools")
    rng = np.random.RandomState(seed)
    d = {}
    with tempfile.TemporaryDirectory() as tmp:
        # ---- proposals -------------------------------------------------------------------------------------------
        spec = importlib.util.spec_from_file_location("ref_proposal_convert", os.path.join(tools, "proposal_convert.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        ids = ["000005", "000007", "000012", "2008_000002"]
        counts = [7, 1, 12, 5]
        dicts = [{"image_id": i, "file_name": "/x/JPEGImages/%s.jpg" % i} for i in ids]

        class _Catalog:
            @staticmethod
            def get(_name):
                return dicts

        mod.DatasetCatalog = _Catalog
        raw = []
        for n in counts:
            y1, x1 = rng.randint(1, 200, n), rng.randint(1, 300, n)
            raw.append(np.stack([y1, x1, y1 + rng.randint(0, 150, n), x1 + rng.randint(0, 180, n)], 1).astype(np.float64))
        cell = np.empty((len(raw),), dtype=object)
        for i, r in enumerate(raw):
            cell[i] = r
        ss_mat = os.path.join(tmp, "ss_boxes.mat")
        scipy.io.savemat(ss_mat, {"boxes": cell})
        mcg_dir = os.path.join(tmp, "mcg")
        os.makedirs(mcg_dir)
        mcg_scores = [rng.rand(n, 1).astype(np.float64) for n in counts]
        for i, r, sc in zip(ids, raw, mcg_scores):
            scipy.io.savemat(os.path.join(mcg_dir, i + ".mat"), {"boxes": r, "scores": sc})
        argv = sys.argv
        try:
            sys.argv = ["proposal_convert.py", "voc_2007_train", ss_mat, os.path.join(tmp, "ss_out.pkl")]
            mod.convert_ss_box()
            sys.argv = ["proposal_convert.py", "voc_2007_train", mcg_dir, os.path.join(tmp, "mcg_out.pkl")]
            mod.convert_mcg_box()
        finally:
            sys.argv = argv
        d["prop_ids"] = np.array(ids)
        for tag in ("ss", "mcg"):
            with open(os.path.join(tmp, tag + "_out.pkl"), "rb") as f:
                out = pickle.load(f)
            assert sorted(out) == ["boxes", "indexes", "scores"]
            d[tag + "_indexes"] = np.array(out["indexes"])
            for i in range(len(ids)):
                d["%s_boxes%d" % (tag, i)] = out["boxes"][i]
                d["%s_scores%d" % (tag, i)] = np.asarray(out["scores"][i])
        for i in range(len(ids)):
            d["raw_boxes%d" % i] = raw[i]
            d["raw_scores%d" % i] = mcg_scores[i]

        # ---- checkpoint keys -------------------------------------------------------------------------------------
        def run(script, src, dst):
            try:
                sys.argv = [script, src, dst]
                runpy.run_path(os.path.join(tools, script), run_name="__main__")
            finally:
                sys.argv = argv

        pth_keys = ["module.backbone.stem.conv1.weight", "module.backbone.res2.0.conv1.norm.running_mean",
                    "module.backbone.res4.5.conv3.weight", "module.neck.fc1.weight", "module.neck.fc2.bias",
                    "module.neck.pool.weight", "module.head.fc_cls.weight", "epoch_marker"]
        src = os.path.join(tmp, "ws.pth")
        torch.save({"state_dict": {k: torch.full((2,), float(i)) for i, k in enumerate(pth_keys)}, "epoch": 120}, src)
        run("convert_resnet_ws_pth.py", src, os.path.join(tmp, "ws_out.pth"))
        out = torch.load(os.path.join(tmp, "ws_out.pth"))
        d["pth_in"] = np.array(pth_keys)
        d["pth_out"] = np.array([k for k, _ in sorted(out.items(), key=lambda kv: float(kv[1][0]))])

        c2_keys = ["conv1_1_w", "conv1_1_bn_s", "conv1_2_w", "conv1_3_bn_b", "res2_0_branch2a_w", "res2_0_branch2a_bn_s",
                   "res3_0_branch1_w", "res_conv1_bn_s", "fc6_w", "fc6_b", "fc7_w", "fc7_b", "pred_w",
                   "res2_0_branch2a_w_momentum"]
        src = os.path.join(tmp, "ws_c2.pkl")
        with open(src, "wb") as f:
            pickle.dump({"blobs": {k: np.full((2,), float(i), np.float32) for i, k in enumerate(c2_keys)}}, f, 2)
        run("convert_resnet_ws_c2.py", src, os.path.join(tmp, "ws_c2_out.pkl"))
        with open(os.path.join(tmp, "ws_c2_out.pkl"), "rb") as f:
            out = pickle.load(f)
        d["c2_in"] = np.array(c2_keys)
        d["c2_out_keys"] = np.array(list(out.keys()))
        d["c2_out_src"] = np.array([int(v[0]) for v in out.values()], dtype=np.int64)

        vgg_keys = ["conv1_1_w", "conv1_1_b", "conv3_2_w", "conv5_3_b", "fc6_w", "fc6_b", "fc7_w", "fc7_b", "pred_w"]
        src = os.path.join(tmp, "vgg.pkl")
        with open(src, "wb") as f:
            pickle.dump({k: np.full((2,), float(i), np.float32) for i, k in enumerate(vgg_keys)}, f, 2)
        run("convert_vgg.py", src, os.path.join(tmp, "vgg_out.pkl"))
        with open(os.path.join(tmp, "vgg_out.pkl"), "rb") as f:
            out = pickle.load(f)
        d["vgg_in"] = np.array(vgg_keys)
        d["vgg_out_keys"] = np.array(list(out.keys()))
        d["vgg_out_src"] = np.array([int(v[0]) for v in out.values()], dtype=np.int64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["ops", "heads", "r50dc5", "r50c4", "r50c4_align", "r18", "vgg", "r50c4_drop", "r50c4_reg", "tta", "ckpt", "data", "voc"]
    if "ops" in which:
        case_ops("ops", 11)
    if "heads" in which:
        case_heads_detail("heads", 21)
    if "r50dc5" in which:
        case_full_model("model_r50dc5_tiny", "PascalVOC-Detection/oicr_WSR_50_DC5_1x.yaml", TINY_R50, 31, 2, 48, 96, 80)
    if "r50c4" in which:
        case_full_model("model_r50c4_tiny", "PascalVOC-Detection/oicr_WSR_50_DC5_1x.yaml", TINY_R50 + C4, 32, 2, 48,
                        128, 96)
    if "r50c4_align" in which:
        case_full_model("model_r50c4_align_tiny", "PascalVOC-Detection/oicr_WSR_50_DC5_1x.yaml",
                        TINY_R50 + C4 + ["MODEL.ROI_BOX_HEAD.POOLER_TYPE", "ROIAlignV2", "MODEL.BACKBONE.FREEZE_AT", "3"],
                        33, 1, 40, 128, 96)
    if "r18" in which:
        case_full_model("model_r18dc5_tiny", "PascalVOC-Detection/oicr_WSR_18_DC5_1x.yaml",
                        ["MODEL.RESNETS.STEM_OUT_CHANNELS", "8", "MODEL.ROI_BOX_HEAD.DAN_DIM", "[48, 64]",
                         "MODEL.ROI_HEADS.NUM_CLASSES", "5"], 34, 1, 40, 96, 96)
    if "vgg" in which:
        case_full_model("model_vgg16_small", "PascalVOC-Detection/oicr_V_16_DC5_1x.yaml",
                        ["MODEL.ROI_BOX_HEAD.DAN_DIM", "[64, 64]", "MODEL.ROI_HEADS.NUM_CLASSES", "5"], 35, 1, 32, 64, 64)
    if "r50c4_drop" in which:
        case_full_model("model_r50c4_dropmask_tiny", "PascalVOC-Detection/oicr_WSR_50_DC5_1x.yaml", TINY_R50 + C4, 36,
                        1, 40, 96, 96, dropmask=True)
    if "pcl" in which:
        case_pcl_unit("pcl_unit", 61)
    if "voc" in which:
        case_voc_eval("voc_eval", 51)
    if "data" in which:
        case_data("data_mapper", "PascalVOC-Detection/oicr_WSR_50_DC5_1x.yaml", TINY_R50 + C4, 41)
    if "convert" in which:
        case_convert("convert", seed=31)
    if "ckpt" in which:
        case_checkpoint("ckpt_r50c4_tiny", "PascalVOC-Detection/oicr_WSR_50_DC5_1x.yaml", TINY_R50 + C4)
    if "tta" in which:
        case_tta("tta_r50c4_tiny", "PascalVOC-Detection/oicr_WSR_50_DC5_1x.yaml", TINY_R50 + C4, 38, 48, 60, 84,
                 (48, 72), 96, 40)
    if "samplers" in which:
        case_samplers("samplers")
    if "wsddn" in which:
        case_full_model("model_wsddn_r50c4_tiny", "PascalVOC-Detection/wsddn_WSR_50_DC5_1x.yaml", TINY_R50 + C4, 40, 2,
                        48, 128, 96)
    if "pclmodel" in which:
        # the reference's _PCLLoss moves its output with .cuda(device_id) (wsl/layers/pcl_loss.py:51,91): without a
        # GPU in this container that call is made the identity for this case (environment shim, like PIL.Image.LINEAR)
        torch.Tensor.cuda = lambda self, *a, **k: self
        # The reference's scikit-learn draw / numpy tie order are not functions of the inputs (oracle/pcl_oracle.py
        # header), and in a model this small (top clusters of ~5 boxes) equal degrees are the rule.  So the golden also
        # records the DECISIONS the reference took - per k-means call the top-cluster threshold, per greedy iteration
        # the picked node - obtained by running the oracle with this machine's scikit-learn / numpy in those two places
        # and checking that it then reproduces the reference's two training steps.  tests/test_oracle_golden.py replays
        # the recorded decisions, which pins everything else of the PCL model flow to the reference on any machine.
        import golden_util as GU
        from oracle import pcl_oracle as PO
        from wsl.modeling.roi_heads.third_party import pcl as ref_pcl

        seed = 39
        case_full_model("model_pcl_r50c4_tiny", "PascalVOC-Detection/pcl_WSR_50_DC5_1x.yaml", TINY_R50 + C4, seed, 1, 60,
                        128, 96)
        ocfg = GU.MODEL_CASES["model_pcl_r50c4_tiny"]
        ocfg.dropout = 0.0
        dd = GU.load("model_pcl_r50c4_tiny")
        km_log, pick_log = [], []

        def km_hook(v):
            v = np.asarray(v)
            t = v[ref_pcl._get_top_ranking_propoals(v.reshape(-1, 1).copy())].min()
            km_log.append(np.float32(t))
            return t

        def pick_hook(x):
            i = int(np.asarray(x, dtype=np.float32).argsort()[::-1][0])
            pick_log.append(i)
            return i

        keep = PO.kmeans_top_threshold, PO._argmax_last
        PO.kmeans_top_threshold, PO._argmax_last = km_hook, pick_hook
        try:
            pp = O.seeded_params(O.param_shapes(ocfg), seed)
            opt = O.SGDState(ocfg)
            for step in range(2):
                ls, _ = O.train_step(pp, GU.batch_from(dd), ocfg, opt, None, 5)
                for k, v in ls.items():
                    ref_v = float(dd["step%d_%s" % (step, k)])
                    assert abs(float(v) - ref_v) <= 1e-4 * max(1.0, abs(ref_v)), (step, k, float(v), ref_v)
        finally:
            PO.kmeans_top_threshold, PO._argmax_last = keep
        dd["pcl_km_log"] = np.array(km_log, dtype=np.float32)
        dd["pcl_pick_log"] = np.array(pick_log, dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, "model_pcl_r50c4_tiny.npz"), **dd)
        print("PCL model golden: oracle with the reference's recorded decisions reproduces both steps;",
              len(km_log), "k-means calls,", len(pick_log), "picks")
    if "csc" in which:
        case_csc("model_csc_r18dc5_tiny", "PascalVOC-Detection/csc_WSR_18_DC5_1x.yaml",
                 ["MODEL.RESNETS.STEM_OUT_CHANNELS", "8", "MODEL.ROI_BOX_HEAD.DAN_DIM", "[48, 64]",
                  "MODEL.ROI_HEADS.NUM_CLASSES", "4", "WSL.CSC_MAX_ITER", "2", "SOLVER.BASE_LR", "0.00002"], 43, 48, 96, 80, tau=0.15)
    if "r50c4_reg" in which:
        case_full_model("model_r50c4_reg_tiny", "PascalVOC-Detection/reg/oicr_WSR_50_DC5_1x.yaml", TINY_R50 + C4, 37,
                        1, 40, 96, 96)
