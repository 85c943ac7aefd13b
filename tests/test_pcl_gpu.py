"""PCL refinement on the device through the C-ABI (drn_pcl_adjacency, drn_pcl_refine) against
oracle/pcl_oracle.py on the same inputs and against the golden cases produced by the reference's own PCL() and
pcl_loss_cpu.cpp.  Integer / index outputs bit-exact; float sums to 1e-5 relative (float32 summation order)."""
import numpy as np
import pytest
import torch

import golden_util as G
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def drn():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    pkg = load_package()
    pkg._cabi.lib()  # raises if the HIP library is missing: no fallback
    import importlib

    return importlib.import_module("drn_wsod_pytorch_amd.ops")


def _run(drn, logits_list, last, boxes, im_labels):
    """logits_list: per-branch [R, K+1] numpy; last: WSDDN scores [R, K].  Returns the per-branch dicts + dlogits."""
    ops = drn
    R, K = last.shape
    nb = len(logits_list)
    pad = 3  # branch columns do not start at 0 and are not adjacent: exercises cols[] / ld
    ld = pad + nb * (K + 1 + 2)
    lg = torch.zeros((R, ld), dtype=torch.float32)
    cols = []
    for b, l in enumerate(logits_list):
        c0 = pad + b * (K + 3)
        lg[:, c0: c0 + K + 1] = torch.from_numpy(l)
        cols.append(c0)
    lg = lg.cuda()
    bx = torch.from_numpy(boxes).cuda()
    adj = ops.pcl_adjacency(bx, 0.4)
    dl = torch.full((R, ld), 7.0, dtype=torch.float32, device="cuda")
    out = ops.pcl_refine(lg, cols, K, torch.from_numpy(last).cuda().contiguous(), bx, adj,
                         torch.from_numpy(im_labels.astype(np.float32)).cuda(), dl)
    torch.cuda.synchronize()
    return out, dl.cpu().numpy(), cols, adj.cpu().numpy()


def _check_branch(o, t, loss, dlog, dl_dev, c0, K, tag):
    n = int(o["n_pc"].item())
    assert n == len(t["pc_labels"]), tag
    assert np.array_equal(o["labels"].cpu().numpy(), t["labels"]), tag
    assert np.array_equal(o["gt_assignment"].cpu().numpy(), t["gt_assignment"]), tag
    assert np.array_equal(o["pc_labels"].cpu().numpy()[:n], t["pc_labels"]), tag
    assert np.array_equal(o["pc_count"].cpu().numpy()[:n], t["pc_count"]), tag
    if "centre_rows" in t:
        assert np.array_equal(o["pc_rows"].cpu().numpy()[:n], t["centre_rows"]), tag
    assert np.array_equal(o["cls_loss_weights"].cpu().numpy(), t["cls_loss_weights"]), tag  # gathered scores: exact
    np.testing.assert_allclose(o["pc_probs"].cpu().numpy()[:n], t["pc_probs"], rtol=1e-5)
    np.testing.assert_allclose(o["img_cls_loss_weights"].cpu().numpy()[:n], t["img_cls_loss_weights"], rtol=1e-5)
    assert abs(float(o["loss"].item()) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss))), tag
    if dlog is not None:
        np.testing.assert_allclose(dl_dev[:, c0: c0 + K + 1], dlog, rtol=2e-4, atol=1e-8)


def test_pcl_golden_cases(drn):
    """the reference's own outputs (single branch per case; WSDDN-shaped or softmax-shaped last scores)"""
    from oracle import pcl_oracle as PO

    d = G.load("pcl_unit")
    for i in range(int(d["n_cases"])):
        g = lambda k: d["c%d_%s" % (i, k)]
        last, K = g("last"), len(g("im_labels"))
        if last.shape[1] != K:
            last = np.ascontiguousarray(last[:, 1:])  # PCL() drops the background column itself (pcl.py:31-32)
        out, dl, cols, _ = _run(drn, [g("logits")], last, g("boxes"), g("im_labels"))
        t = dict(labels=g("labels").astype(np.int32), gt_assignment=g("gt_assignment").astype(np.int32),
                 pc_labels=g("pc_labels").astype(np.int32), pc_count=g("pc_count").astype(np.int32),
                 cls_loss_weights=g("cls_loss_weights"), pc_probs=g("pc_probs"),
                 img_cls_loss_weights=g("img_cls_loss_weights"))
        probs = torch.softmax(torch.from_numpy(g("logits")), 1).numpy()
        gp = g("dprobs")
        dlog = probs * (gp - (gp * probs).sum(axis=1, keepdims=True))
        _check_branch(out[0], t, g("loss"), dlog, dl, cols[0], K, "golden %d" % i)
        assert np.all(dl[:, : cols[0]] == 7.0) and np.all(dl[:, cols[0] + K + 1:] == 7.0)  # nothing else touched


@pytest.mark.parametrize("R,K,nb,seed", [(40, 5, 3, 0), (333, 20, 3, 1), (2000, 20, 3, 2), (4096, 20, 2, 3),
                                         (31, 4, 1, 4), (5, 3, 2, 5), (2000, 20, 2, 106), (3000, 8, 2, 207)])
def test_pcl_refine_cascade_vs_oracle(drn, R, K, nb, seed):
    """branch b clusters on branch b-1's softmax: the whole cascade of one image against the oracle"""
    from oracle import pcl_oracle as PO

    rs = np.random.RandomState(100 + seed)
    W, H = 320.0, 240.0
    nseed = max(2, R // 15)
    sx0, sy0 = rs.rand(nseed) * (W - 80), rs.rand(nseed) * (H - 80)
    sw, sh = 40 + rs.rand(nseed) * (W - sx0 - 40), 40 + rs.rand(nseed) * (H - sy0 - 40)
    pick = rs.randint(0, nseed, size=R)
    jit = rs.randn(R, 4) * 8.0
    x0 = np.clip(sx0[pick] + jit[:, 0], 0, W - 21)
    y0 = np.clip(sy0[pick] + jit[:, 1], 0, H - 21)
    x1 = np.clip(sx0[pick] + sw[pick] + jit[:, 2], x0 + 20, W)
    y1 = np.clip(sy0[pick] + sh[pick] + jit[:, 3], y0 + 20, H)
    boxes = np.stack([x0, y0, x1, y1], 1).astype(np.float32)
    im = np.zeros(K, dtype=np.float32)
    im[rs.permutation(K)[: rs.randint(1, 4)]] = 1
    a = torch.from_numpy(rs.randn(R, K).astype(np.float32) * 2.0)
    b = torch.from_numpy(rs.randn(R, K).astype(np.float32) * 3.0)
    last = (torch.softmax(a, 1) * torch.softmax(b, 0)).numpy()
    logits = [rs.randn(R, K + 1).astype(np.float32) * 2.0 for _ in range(nb)]
    if seed >= 200:  # heavy-tailed scores: a handful of dominant proposals, a long flat tail
        last = (last ** 3 / (last ** 3).sum(0, keepdims=True)).astype(np.float32)
    elif seed >= 100:  # quantised scores and logits: thousands of duplicates, few distinct cut positions
        last = (np.round(last * 2000.0) / 2000.0).astype(np.float32)
        logits = [np.round(l * 2.0) / 2.0 for l in logits]
    out, dl, cols, adj = _run(drn, logits, last, boxes, im)
    # adjacency bits == IoU > 0.4 of the oracle
    ref_adj = PO._iou_np(boxes, boxes) > np.float32(0.4)
    bits = ((adj.view(np.uint32)[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(R, -1)[:, :R].astype(bool)
    assert np.array_equal(bits, ref_adj)
    prev = last
    for k in range(nb):
        loss, dlog, probs, t = PO.pcl_refine_loss(logits[k], boxes, prev, im)
        # the device softmax (expf) and torch's differ by an ulp; feed the oracle's next branch the device's own
        # probabilities so that both sides cluster the same numbers
        np.testing.assert_allclose(out[k]["probs"].cpu().numpy(), probs, rtol=2e-6, atol=1e-9)
        _check_branch(out[k], t, loss, dlog, dl, cols[k], K, "branch %d" % k)
        prev = out[k]["probs"].cpu().numpy()


@pytest.mark.parametrize("case", ["one_proposal", "constant_scores", "zero_area_boxes", "no_labels", "duplicates"])
def test_pcl_edge_cases_vs_oracle(drn, case):
    """degenerate inputs: a single proposal; all scores equal (one k-means cluster = every proposal); zero-area boxes
    (no self edge in the IoU graph: the greedy loop ends on an empty clique and the centre gets score 0, its cluster
    is empty -> NaN mean probability that fmaxf turns into eps exactly like pcl_loss_cpu.cpp); an image without labels
    (no centres, loss 0); duplicated proposals (a later centre with an empty cluster)"""
    from oracle import pcl_oracle as PO

    rs = np.random.RandomState(9)
    K, R = 4, 40
    x0, y0 = rs.rand(R) * 100, rs.rand(R) * 80
    boxes = np.stack([x0, y0, x0 + 20 + rs.rand(R) * 60, y0 + 20 + rs.rand(R) * 50], 1).astype(np.float32)
    im = np.array([1, 0, 1, 0], dtype=np.float32)
    last = (torch.softmax(torch.from_numpy(rs.randn(R, K).astype(np.float32) * 2), 1) *
            torch.softmax(torch.from_numpy(rs.randn(R, K).astype(np.float32) * 3), 0)).numpy()
    if case == "one_proposal":
        boxes, last, R = boxes[:1], last[:1], 1
    elif case == "constant_scores":
        last = np.full_like(last, 0.01)
    elif case == "zero_area_boxes":
        boxes[:, 2] = boxes[:, 0]
    elif case == "no_labels":
        im = np.zeros(K, dtype=np.float32)
    elif case == "duplicates":
        boxes[1::2] = boxes[0::2]
        last[1::2] = last[0::2]
    logits = [rs.randn(R, K + 1).astype(np.float32) * 2.0 for _ in range(2)]
    out, dl, cols, _ = _run(drn, logits, np.ascontiguousarray(last), np.ascontiguousarray(boxes), im)
    prev = last
    for k in range(2):
        with np.errstate(all="ignore"):
            loss, dlog, probs, t = PO.pcl_refine_loss(logits[k], boxes, prev, im)
        n = int(out[k]["n_pc"].item())
        assert n == len(t["pc_labels"]), (case, k)
        assert np.array_equal(out[k]["labels"].cpu().numpy(), t["labels"]), (case, k)
        assert np.array_equal(out[k]["gt_assignment"].cpu().numpy(), t["gt_assignment"]), (case, k)
        assert np.array_equal(out[k]["pc_rows"].cpu().numpy()[:n], t["centre_rows"]), (case, k)
        assert np.array_equal(out[k]["pc_count"].cpu().numpy()[:n], t["pc_count"]), (case, k)
        np.testing.assert_allclose(out[k]["pc_probs"].cpu().numpy()[:n], t["pc_probs"], rtol=1e-5, equal_nan=True)
        got = float(out[k]["loss"].item())
        assert np.isfinite(got) and abs(got - float(loss)) <= 1e-5 * max(1.0, abs(float(loss))), (case, k, got, loss)
        np.testing.assert_allclose(dl[:, cols[k]: cols[k] + K + 1], dlog, rtol=2e-4, atol=1e-8)
        prev = out[k]["probs"].cpu().numpy()


def test_pcl_limits(drn):
    from drn_wsod_pytorch_amd._cabi import DrnError

    ops = drn
    with pytest.raises(DrnError):
        ops.pcl_adjacency(torch.zeros((4097, 4), device="cuda"))
