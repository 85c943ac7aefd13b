"""Where does the HIP path's bf16 mode leave the bf16-emulating oracle?  Stage-by-stage comparison of one forward
(features, pooled fc6 operand, fc6 / fc7 activations, logits, losses) on SURVEY 8(d) inputs.
  python tools/debug_bf16_parity.py [r50c4|r50dc5] [R]"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as G  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

O = G.O
pkg = load_package()
case = sys.argv[1] if len(sys.argv) > 1 else "r50c4"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
kw = dict(r50c4=dict(arch="wsr50", out_feature="res4", res5_dilation=1, num_classes=20),
          r50dc5=dict(arch="wsr50", out_feature="res5", res5_dilation=2, num_classes=20))[case]
torch.set_num_threads(min(32, os.cpu_count() or 1))
ocfg = O.OracleCfg(dropout=0.0, **kw)
batch = O.synthetic_batch(1, R, ocfg, seed=4321)


def stats(name, a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b)
    print("%-28s max|d| %.3e  rms d %.3e  rms ref %.3e  rel(rms) %.3e  frac(d>0) %.4f" % (
        name, d.max(), np.sqrt((d ** 2).mean()), np.sqrt((b ** 2).mean()),
        np.sqrt((d ** 2).mean()) / max(np.sqrt((b ** 2).mean()), 1e-30), float((d > 0).mean())))


cfg, model = G.drn_model(ocfg, 3, "cuda", 5, "bf16")
model.roi_heads.box_head.dropout_p = 0.0
model.train()
ins = G.drn_inputs([dict(b, gt_boxes=torch.zeros(len(b["gt_classes"]), 4)) for b in batch])
with torch.no_grad():
    images = model.preprocess_image(ins)
    feats = model.backbone(images.tensor)
feat = feats[ocfg.out_feature].float().cpu()
losses = model(ins)
st = model.roi_heads._last_state
w = st["w"]
torch.cuda.synchronize()

for emu in (True, False):
    c = copy.deepcopy(ocfg)
    c.emulate_bf16 = emu
    p = O.init_params(c, seed=3)
    x, _ = O.preprocess_image([b["image"] for b in batch], c)
    print("==== oracle emulate_bf16=%s" % emu)
    stats("preprocessed image", images.tensor.float().cpu().numpy(), x.numpy())
    f = O.backbone_forward(p, x, c)
    stats("feature map " + ocfg.out_feature, feat.numpy(), f.numpy())
    ref_losses, aux = O.model_train_losses(p, batch, c, None, True)
    K1 = aux["pooled"][0].numel()
    stats("pooled A", w["A"][:, :K1].float().cpu().numpy(), aux["pooled"].reshape(R, -1).numpy())
    stats("fc7 out H2", w["H2"][:, : aux["fc7"].shape[1]].float().cpu().numpy(), aux["fc7"].detach().numpy())
    stats("MIL scores", st["aux"]["scores"].cpu().numpy(), aux["scores"].detach().numpy())
    stats("image scores", st["aux"]["img_scores"].cpu().numpy(), aux["img_scores"].numpy())
    for k in range(c.refine_num):
        col = {n: c0 for n, _, c0, _ in model.roi_heads._engine.cols}["r%d" % k]
        stats("refine logits %d" % k, w["logits"][:, col: col + c.num_classes + 1].cpu().numpy(),
              aux["logits"][k].detach().numpy())
        mine = st["aux"]["targets"][k]["pgt_idx"].cpu().numpy()[0]
        theirs = aux["pgt"][k][0][4].numpy()
        lab = st["aux"]["targets"][k]["labels"].cpu().numpy()
        print("   pgt rows  product %s  oracle %s   labels differing: %d" % (
            mine[: len(theirs)].tolist(), theirs.tolist(), int((lab != aux["labels"][k].numpy()).sum())))
    for k, v in losses.items():
        print("   %-14s product %.6f  oracle %.6f  rel %.3e" % (k, float(v.detach()), float(ref_losses[k]),
                                                               abs(float(v.detach()) - float(ref_losses[k])) / max(abs(float(ref_losses[k])), 1e-6)))
