"""Throughput of one FULL test-time-augmented image with the shipped TEST.AUG settings
(projects/WSL/configs/PascalVOC-Detection/oicr_WSR_50_DC5_1x.yaml:49-54: MIN_SIZES 480 .. 1152 (8 sizes), MAX_SIZE 4000, FLIP ->
16 passes per image; projects/WSL/wsl/modeling/test_time_augmentation_avg.py:139-321) on a VOC-sized image (375 x 500, 2000
proposals), for the constructed R50-C4 model of the bench workload and for the shipped R50-DC5 recipe.
Reports: the mapper (host: 8 resizes + flips + proposal transforms, as in the reference's loader), the 16 device passes + the
averaging + the final NMS / top-k, and images/s of the whole call.  TTA_WORKLOADS=r50c4,r50dc5 selects."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench
from __graft_entry__ import load_package

pkg = load_package()
pkg.set_precision("bf16")
from drn_wsod_pytorch_amd.modeling import build_model
from drn_wsod_pytorch_amd.modeling.tta import GeneralizedRCNNWithTTAAVG
from drn_wsod_pytorch_amd.structures import Boxes, Instances

H, W, R = 375, 500, 2000
for wl in os.environ.get("TTA_WORKLOADS", "r50c4,r50dc5").split(","):
    cfg = bench.build_cfg(pkg, "cuda")
    extra = ["TEST.AUG.ENABLED", "True", "TEST.AUG.MIN_SIZES", "(480, 576, 672, 768, 864, 960, 1056, 1152)", "TEST.AUG.MAX_SIZE", "4000",
             "TEST.AUG.FLIP", "True"]
    if wl == "r50dc5":
        extra += ["MODEL.RESNETS.OUT_FEATURES", "['res5']", "MODEL.ROI_HEADS.IN_FEATURES", "['res5']", "MODEL.RESNETS.RES5_DILATION", "2",
                  "MODEL.ROI_BOX_HEAD.DAN_DIM", "[2048, 4096]"]
    cfg.merge_from_list(extra)
    model = build_model(cfg)
    bench.init_weights(model, seed=0)
    model.eval()
    g = torch.Generator().manual_seed(1)
    img = torch.randint(0, 256, (3, H, W), generator=g).to(torch.uint8)  # (uint8 like a decoded image: the mapper resizes it with PIL, as the reference does)
    x0 = torch.rand(R, generator=g) * (W - 60)
    y0 = torch.rand(R, generator=g) * (H - 60)
    bw = 20 + torch.rand(R, generator=g) * (W - x0 - 20) * 0.6
    bh = 20 + torch.rand(R, generator=g) * (H - y0 - 20) * 0.6
    p = Instances((H, W))
    p.proposal_boxes = Boxes(torch.stack([x0, y0, x0 + bw, y0 + bh], 1))
    p.objectness_logits = torch.rand(R, generator=g)
    inp = {"image": img, "proposals": p, "height": H, "width": W}
    tta = GeneralizedRCNNWithTTAAVG(cfg, model, batch_size=int(os.environ.get("TTA_BATCH", "1")))
    if os.environ.get("TTA_SYNC_AB", "0") == "1":  # same-box A/B of the per-pass host syncs removed in round 5
        import drn_wsod_pytorch_amd.modeling.tta as tta_mod
        import drn_wsod_pytorch_amd.ops as ops_mod
        for flag in (False, True, False, True):
            tta_mod.VECTORISED_MAPPER = ops_mod.COL0_CACHE = flag
            for _ in range(2):
                tta([inp])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                tta([inp])
            torch.cuda.synchronize()
            print("%s: per-pass uploads %s: whole call %.1f ms" % (wl, "batched / cached" if flag else "per pass (syncs)",
                                                                   (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    for _ in range(2):
        out = tta([inp])
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        out = tta([inp])
    torch.cuda.synchronize()
    full = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        aug = tta.tta_mapper(inp)
    mapper = (time.perf_counter() - t0) / n
    with torch.no_grad():
        for _ in range(2):
            tta._get_augmented_boxes(aug)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            tta._get_augmented_boxes(aug)
        torch.cuda.synchronize()
    passes = (time.perf_counter() - t0) / n
    sizes = sorted({tuple(a["image"].shape[1:]) for a in aug})
    print("%s (batch_size %d): %d augmentations per image (%s .. %s), R = %d: whole call %.1f ms = %.2f img/s; mapper (host) %.1f ms; the %d device passes + "
          "averaging %.1f ms = %.2f ms per augmentation; %d detections" % (wl, tta.batch_size, len(aug), "%dx%d" % sizes[0], "%dx%d" % sizes[-1], R, full * 1e3, 1.0 / full,
                                                          mapper * 1e3, len(aug), passes * 1e3, passes * 1e3 / len(aug),
                                                          len(out[0]["instances"])), flush=True)
    del model, tta
