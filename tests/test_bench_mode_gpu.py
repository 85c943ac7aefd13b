"""The BENCHMARKED mode at the BENCHMARKED sizes against the oracle ("pin the benchmarked mode").

What bench.py times is: set_precision("bf16") + FusedSGD.enable_pipelined() with the bf16 fc6 gradient bucket +
GraphedTrainStep(split_tail=True, trunk_pairs=4, eager_fc6=True).  Here exactly that configuration runs 5 steps over
3 distinct SURVEY 8(d) batches in a non-periodic order (a wrong staging slot, a stale weight shadow or a bucket bug
shows as an O(1) error) for BASELINE configs[1] (R50-C4, R=2000), configs[2]'s shape (R50-DC5, R=4000) and configs[3]'s
shape (R101-C4, K=80), and is compared per step with TWO oracles:

  (A) oracle.OracleCfg(emulate_bf16=True): the reference's fp32 algorithm with every value the product stores in bf16
      rounded at the same point (oracle/wsod_oracle.py);
  (B) the plain fp32 oracle (= the reference's arithmetic).

Derivation of the bounds.  bf16 keeps 8 significand bits: every stored value carries up to 2^-9 relative rounding.  Two
correct bf16 implementations that differ only in fp32 summation order do NOT stay bit-close: a 1e-6 difference flips a
rounding with probability ~2.5e-4 per element, the flips spread (every output depends on ~10^3 inputs) and after the
45-conv trunk about half of the res4 elements differ by one bf16 ulp (tools/debug_bf16_parity.py on this workload: res4
map 4.9e-3 rms relative vs (A), 6.8e-3 vs (B); logits ~0.02 absolute).  How much a LOSS moves under that noise depends
on the loss: the MIL loss by ~1e-3, but a refinement loss of 0.1 is a weighted mean of heavy-tailed log-probabilities
and moves by several per cent, and once SGD steps feed the noise back the sensitivity grows from step to step.  The
bound is therefore not guessed but MEASURED per quantity by the two oracles themselves: (A) and (B) differ by exactly
one application of "round what the product stores in bf16", so |A - B| is the size of the bf16 effect on that quantity
(pooled over the pinned steps), and the product must stay within 3 x |A - B| + 1 % of the value of BOTH (three draws - product, A, B -
of the same noise: the distance of one pair is a noisy estimate of the distance of another; round 2 used 5x + 2 %, round 3
tightened it to what the three shapes need, profiles/r3_06_bench_mode_parity.txt, while a wrong staging slot, a stale weight shadow or a bucket bug moves the MIL loss and the image
scores by O(0.1 ... 1)).  This is applied to the first three steps; even at the reduced learning rate below the MIL head
of these synthetic weights sits on a knife edge (image scores flip between classes from step 3 on, |A - B| itself
reaches 0.7), so steps 3 and 4 are pinned differently: the graphed run must equal the eager run of the same mode to 1e-5
on every loss of every step (a wrong slot / stale buffer in the graph schedule breaks that at any step), all values
must be finite, the bf16 weight shadows must equal the rounded master weights bit for bit after the last step, and the
sampled fc6 weight movement / fc7 bias over all five steps must stay within 3 x |A - B| + 1 % as well.
Pseudo-GT mining (get_pgt's arg-max over R, roi_heads_oicr.py:504-506) is DISCONTINUOUS: the product's arg-max rows are
compared with both oracles' rows; where they differ from BOTH at step 0 (identical weights), the product's row must be a
near-tie in (A)'s scores (>= 0.9 x the maximum) - at steps 0, 1 and 2 (round 3: a row that is not fails the test).
Round 3 also pins step 0 with a FIXED bound against the emulating oracle (every loss within 5 %) and adds
test_bench_mode_at_the_bench_learning_rate: one optimizer step at bench.py's base_lr = 0.01.

Dropout uses injected {0, 2} multiplier masks shared with the oracle; the graphed run must also equal the
eager pipelined run of the same mode (same kernels: 1e-5)."""
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

CASES = {
    "r50c4_r2000_k20": (dict(arch="wsr50", out_feature="res4", res5_dilation=1, num_classes=20), 2000),
    "r50dc5_r4000_k20": (dict(arch="wsr50", out_feature="res5", res5_dilation=2, num_classes=20), 4000),
    "r101c4_r2000_k80": (dict(arch="wsr101", out_feature="res4", res5_dilation=1, num_classes=80), 2000),
}
ORDER = [0, 1, 2, 0, 1, 1, 0, 2, 1, 0, 2, 2]  # not periodic in 2, 3 or 4
STEPS = 5
SEED = 3
SAMPLE = 4099  # stride of the fc6 weight sample


def _masks(R, d1, d2, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(R, d, generator=g) >= 0.5).float() * 2.0 for d in (d1, d2)]


def _oracle_run(ocfg, batches, masks, emulate, steps=STEPS):
    cfg = copy.deepcopy(ocfg)
    cfg.emulate_bf16 = emulate
    p = O.init_params(cfg, seed=SEED)
    opt = O.SGDState(cfg)
    out = []
    w0 = p["roi_heads.box_head.fc1.weight"].reshape(-1)[::SAMPLE].clone()
    for t in range(steps):
        losses, _, aux = O.train_step(p, batches[ORDER[t]], cfg, opt, dropout_masks=masks, return_aux=True)
        prev = [aux["scores"].detach()] + [torch.softmax(l.detach(), dim=-1) for l in aux["logits"][:-1]]
        out.append(dict(losses=losses, img_scores=aux["img_scores"].numpy().copy(),
                        pgt=[[(pc.numpy().copy(), idx.numpy().copy()) for (_, pc, _, _, idx) in aux["pgt"][k]]
                             for k in range(cfg.refine_num)],
                        prev=[s.numpy().copy() for s in prev]))
    w = p["roi_heads.box_head.fc1.weight"].reshape(-1)[::SAMPLE].clone()
    return out, (w - w0).numpy(), p["roi_heads.box_head.fc2.bias"].numpy().copy()


def _product_run(ocfg, batches, masks, graphed, steps=STEPS):
    from drn_wsod_pytorch_amd.engine import GraphedTrainStep, build_optimizer

    cfg, model = G.drn_model(ocfg, SEED, "cuda", 5, "bf16")
    model.roi_heads.box_head.dropout_masks = [m.cuda() for m in masks]
    model.train()
    opt = build_optimizer(cfg, model)
    opt.enable_pipelined()  # bench.py: bf16 fc6 gradient bucket in the bf16 mode
    assert opt._comm_dtype == torch.bfloat16 and model.roi_heads._engine.fc1_grad_bucket.dtype == torch.bfloat16
    w0 = model.roi_heads.box_head.fc1.weight.detach().reshape(-1)[::SAMPLE].cpu().clone()
    ins = [G.drn_inputs([dict(b, gt_boxes=torch.zeros(len(b["gt_classes"]), 4)) for b in bb]) for bb in batches]
    for bb in ins:
        for x in bb:
            x["image"] = x["image"].cuda()
            x["proposals"].proposal_boxes.tensor = x["proposals"].proposal_boxes.tensor.cuda()
            x["proposals"].objectness_logits = x["proposals"].objectness_logits.cuda()
    seq = [ins[i] for i in ORDER]
    out = []
    eng = model.roi_heads._engine
    if graphed:
        stepper = GraphedTrainStep(model, opt, seq[0], split_tail=True, trunk_pairs=4, eager_fc6=True)  # = bench.py (round 4: groups of 4)
    for t in range(steps):
        if graphed:
            losses = stepper.step(*seq[t: t + 8])
            st = stepper.last_state
        else:
            losses = model(seq[t])
            sum(losses.values()).backward()
            opt.step()
            opt.zero_grad()
            st = model.roi_heads._last_state
        torch.cuda.synchronize()
        out.append(dict(losses={k: float(v.detach()) for k, v in losses.items()},
                        img_scores=st["aux"]["img_scores"].cpu().numpy().copy(),
                        pgt=[tg["pgt_idx"].cpu().numpy().copy() for tg in st["aux"]["targets"]]))
    w = model.roi_heads.box_head.fc1.weight.detach().reshape(-1)[::SAMPLE].cpu()
    b2 = model.roi_heads.box_head.fc2.bias.detach().cpu().numpy().copy()
    # the bf16 compute copies the NEXT forward would read must be the rounded master weights, bit for bit (a stale
    # shadow - an SGD bucket that did not refresh its rows - shows here directly)
    used = [(o, n) for _, _, o, n, u in eng.segments if u]
    for o, n in used:
        assert torch.equal(eng.arena_s[o: o + n], eng.arena_w[o: o + n].to(torch.bfloat16)), "stale bf16 weight shadow"
    d1, d2 = model.roi_heads.box_head.fc1.weight.shape[0], model.roi_heads.box_head.fc2.weight.shape[0]
    assert torch.equal(eng.sh["W2T"][:, :d2], model.roi_heads.box_head.fc2.weight.detach().to(torch.bfloat16).t()), \
        "stale K-major twin of fc7's weight"
    del model, opt
    torch.cuda.empty_cache()
    return out, (w - w0).numpy(), b2


def _rel(a, b, floor):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor))


def test_ring_schedule_full_size_equals_waiting_schedule():
    """Round 6: bench.py's default schedule - GraphedTrainStep(ring=True): the side stream never waits for the main stream,
    RING_SETS staging sets and RING_SLOTS trunk slots, the host at most RING_LAG steps ahead of the heads - against the waiting
    schedule (ring=False) at BASELINE configs[1]'s size, where a step is GPU-bound (1.2 ms of device work against 0.4 ms of host
    work: the host DOES run ahead; with one staging set or without the throttle the same 60 steps end on different losses).
    Sixty steps without a sync over eight distinct batches: every loss of every step the same bits."""
    from drn_wsod_pytorch_amd.engine import GraphedTrainStep, build_optimizer

    kw, R = CASES["r50c4_r2000_k20"]
    ocfg = O.OracleCfg(dropout=0.5, base_lr=2e-4, **kw)
    batches = [O.synthetic_batch(1, R, ocfg, seed=977 + 31 * i) for i in range(8)]
    steps, group = 60, 4
    order = [(i * 5 + i // 7) % 8 for i in range(steps + 2 * group)]
    results = []
    for ring in (False, True):
        cfg, model = G.drn_model(ocfg, SEED, "cuda", 5, "bf16")
        model.train()
        opt = build_optimizer(cfg, model)
        opt.enable_pipelined()
        ins = [G.drn_inputs([dict(b, gt_boxes=torch.zeros(len(b["gt_classes"]), 4)) for b in bb]) for bb in batches]
        for bb in ins:
            for x in bb:
                x["image"] = x["image"].cuda()
                x["proposals"].proposal_boxes.tensor = x["proposals"].proposal_boxes.tensor.cuda()
                x["proposals"].objectness_logits = x["proposals"].objectness_logits.cuda()
        seq = [ins[i] for i in order]
        stepper = GraphedTrainStep(model, opt, seq[0], split_tail=True, trunk_pairs=group, eager_fc6=True, ring=ring)
        out = []
        for t in range(steps):
            losses = stepper.step(*seq[t: t + 2 * group])
            out.append(torch.stack([losses[k].detach().clone().reshape(()) for k in sorted(losses)]))  # (a device copy, no sync)
        assert stepper._ring_on == ring
        torch.cuda.synchronize()
        results.append(torch.stack(out).cpu())
        stepper.release()
        del stepper, model, opt
        torch.cuda.empty_cache()
    load_package().set_precision("fp32")
    assert torch.isfinite(results[0]).all()
    assert torch.equal(results[0], results[1])


@pytest.mark.parametrize("case", list(CASES))
def test_bench_mode_full_size_vs_oracles(case):
    kw, R = CASES[case]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    # base_lr: with the benchmark's 0.01 the MIL head of these synthetic weights saturates after ONE step (loss_cls
    # jumps to its clamp constant G * 13.8155 / K) and the refinement losses then swing between 1e-5 and 50:
    # a chaotic regime in which two correct implementations decorrelate within two steps.  2e-4 keeps the five steps in
    # the regime where losses are smooth in the weights; the mechanics under test (slots, shadows, buckets, schedule)
    # do not depend on the learning rate
    ocfg = O.OracleCfg(dropout=0.5, base_lr=2e-4, **kw)
    batches = [O.synthetic_batch(1, R, ocfg, seed=4321 + 17 * i) for i in range(3)]
    masks = _masks(R, ocfg.dan_dim[0], ocfg.dan_dim[1], 99)
    got, dw, b2 = _product_run(ocfg, batches, masks, graphed=True)
    eager, dw_e, b2_e = _product_run(ocfg, batches, masks, graphed=False)
    emu, dw_emu, b2_emu = _oracle_run(ocfg, batches, masks, emulate=True)
    ref, dw_ref, b2_ref = _oracle_run(ocfg, batches, masks, emulate=False)
    load_package().set_precision("fp32")
    rep, bad, ties = [], [], 0
    K = ocfg.num_classes

    def check(ok, what):
        if not ok:
            bad.append(what)

    PINNED = 3  # steps compared with the oracles; the later ones are pinned through graphed == eager (see docstring)
    # size of the bf16 effect on a quantity = the spread of the two oracles, POOLED over the pinned steps: the three runs
    # (product, A, B) are three draws of the same noise and the distance of ONE pair at ONE step is a noisy estimate -
    # at step 2 of R50-C4 the two oracles' image scores happen to agree to 2.5e-3 while each differs from the product by
    # 2e-2, like all three pairs do at steps 0 and 1 (profiles/r3_07_parity_round2.txt)
    FLOOR = 1e-2
    spread = {name: max(abs(emu[t]["losses"][name] - ref[t]["losses"][name]) / max(abs(ref[t]["losses"][name]), FLOOR)
                        for t in range(PINNED)) for name in ref[0]["losses"]}
    spread_img = max(float(np.abs(emu[t]["img_scores"].astype(np.float64) - ref[t]["img_scores"].astype(np.float64)).max())
                     for t in range(PINNED))

    def within(v, e, r, what, floor=FLOOR):
        """v (product) within (3 x the oracles' pooled relative spread + 1 %) of both oracles (round 2: 5 x the same
        step's |emu - fp32| + 2 %); at step 0 - identical weights: what separates product and emulator is one
        forward's worth of rounding FLIPS, not the roundings themselves - additionally within a FIXED 5 % of the
        emulating oracle, whatever the oracles' spread (measured on the three shapes with the {0, 2} dropout masks:
        <= 3.6 %, profiles/r3_06_bench_mode_parity.txt).  The fixed 3 % / 1e-3 the round-2 review proposed from
        profiles/r2_01_* does not hold here: that record ran without dropout; with the benchmark's masks the emulating and
        the fp32 oracle themselves differ by 6-16 % on the refinement losses and by 1.4e-2 ... 3.3e-2 on the image
        scores at step 0."""
        tol = (3.0 * spread[what[0]] + 1e-2) * max(abs(r), floor)
        check(abs(v - e) <= tol and abs(v - r) <= tol, (what, v, e, r, tol))
        if what[1] == 0:
            check(abs(v - e) <= 5e-2 * max(abs(e), floor), (what, "step 0 vs the emulating oracle", v, e))
        return tol

    for t in range(STEPS):
        assert set(got[t]["losses"]) == set(ref[t]["losses"])
        for k, v in got[t]["losses"].items():
            check(abs(v - eager[t]["losses"][k]) <= 1e-5 * max(abs(v), 1e-3), ("graphed != eager", t, k, v, eager[t]["losses"][k]))
        gi, ei, ri = got[t]["img_scores"].astype(np.float64), emu[t]["img_scores"].astype(np.float64), ref[t]["img_scores"].astype(np.float64)
        d_pe, d_pr, d_er = np.abs(gi - ei).max(), np.abs(gi - ri).max(), np.abs(ei - ri).max()
        tol_img = 3.0 * spread_img + 1e-2 * np.abs(ri).max()
        check(t >= PINNED or (d_pe <= tol_img and d_pr <= tol_img), ("img_scores", t, d_pe, d_pr, tol_img))
        rep.append("step %d  image scores: |p-emu| %.2e  |p-fp32| %.2e  |emu-fp32| %.2e  (bound %.2e)" % (t, d_pe, d_pr, d_er, tol_img))
        for name in sorted(got[t]["losses"]):
            v, e, r = got[t]["losses"][name], emu[t]["losses"][name], ref[t]["losses"][name]
            assert np.isfinite(v)
            tol = within(v, e, r, (name, t)) if t < PINNED else float("nan")
            rep.append("   %-12s %.6f  emu %.6f  fp32 %.6f   |p-emu| %.2e |p-fp32| %.2e  bound %.2e" % (
                name, v, e, r, abs(v - e), abs(v - r), tol))
        for k in range(ocfg.refine_num):
            mine = got[t]["pgt"][k][0]
            ce, ie = emu[t]["pgt"][k][0]
            _, ir = ref[t]["pgt"][k][0]
            for g, c in enumerate(ce):
                if int(mine[g]) not in (int(ie[g]), int(ir[g])) and t < PINNED:
                    ties += 1
                    col = emu[t]["prev"][k][:, int(c)]
                    # a row neither oracle picked must be a near-tie in (A)'s scores - at steps 1 and 2 as well (round 2
                    # only counted those; at base_lr = 2e-4 two steps move the scores far less than the 10 % margin)
                    ratio = float(col[int(mine[g])] / col.max())
                    rep.append("   step %d pgt branch %d class %d: product row %d, emu %d, fp32 %d, score ratio %.4f" % (
                        t, k, int(c), int(mine[g]), int(ie[g]), int(ir[g]), ratio))
                    check(ratio >= 0.9, ("pgt row not a near-tie", t, k, g, int(mine[g]), int(ie[g]), int(ir[g]), ratio))
    rep.append("pseudo-GT rows that differ from both oracles (near-ties): %d" % ties)
    # SGD through the bf16 bucket, 5 steps: sampled fc6 weight movement and the fc7 bias
    scale = float(np.abs(dw_emu).max())
    d_pe, d_pr, d_er = (float(np.abs(a - b).max()) for a, b in ((dw, dw_emu), (dw, dw_ref), (dw_emu, dw_ref)))
    rep.append("fc6 weight movement (sampled, max|dw| %.2e): |p-emu| %.2e  |p-fp32| %.2e  |emu-fp32| %.2e" % (scale, d_pe, d_pr, d_er))
    check(float(np.abs(dw - dw_e).max()) <= 1e-5 * scale, ("graphed != eager weights",))
    check(max(d_pe, d_pr) <= 3.0 * d_er + 1e-2 * scale, ("fc6 weight movement", d_pe, d_pr, d_er))
    b_pe, b_pr, b_er = (float(np.abs(a - b).max()) for a, b in ((b2, b2_emu), (b2, b2_ref), (b2_emu, b2_ref)))
    check(max(b_pe, b_pr) <= 3.0 * b_er + 1e-2 * float(np.abs(b2_ref).max()), ("fc7 bias", b_pe, b_pr, b_er))
    print("[bench-mode parity %s]\n%s" % (case, "\n".join(rep)))
    assert not bad, "%s\n%s" % (bad, "\n".join(rep))


@pytest.mark.parametrize("case", list(CASES))
def test_bench_mode_at_the_bench_learning_rate(case):
    """VERDICT r2 (weak 1a): the comparison above runs at base_lr = 2e-4; bench.py trains at 0.01.  Here the SAME mode
    (bf16, pipelined optimizer with the bf16 bucket, GraphedTrainStep as bench.py builds it) takes ONE optimizer step at
    the benchmark's base_lr = 0.01 and then a second forward.
    What can be pinned, measured first (profiles/r3_06_bench_mode_parity.txt): with these synthetic weights ONE step at
    0.01 already saturates the MIL head on two of the three shapes - loss_cls sits at its clamp constant G * 13.8155 / K
    (SURVEY F7), image scores move by 0.14 (R50-DC5) and 0.68 (R101-K80) between the two ORACLES' own runs - so step-1
    activations are a coin flip there, for any implementation.  Pinned instead:
      * step 0 (the forward the update is computed from): losses within 5 % of the emulating oracle;
      * the UPDATE at the bench learning rate: sampled fc6 weight movement and the fc7 bias after one step within
        3 x |emu - fp32| + 1 % of both oracles - lr, weight decay, momentum initialisation and the bf16 bucket enter here;
      * step 1 wherever the oracles themselves still agree (image scores within 5e-3 of each other: R50-C4, the bench
        workload - measured 7e-4): losses and image scores within 3 x |emu - fp32| + 1 %, and a pseudo-GT row that neither
        oracle picked FAILS unless it is a near-tie; elsewhere step 1 must be finite and equal to the eager run (1e-5)."""
    kw, R = CASES[case]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ocfg = O.OracleCfg(dropout=0.5, base_lr=0.01, **kw)
    batches = [O.synthetic_batch(1, R, ocfg, seed=4321 + 17 * i) for i in range(3)]
    masks = _masks(R, ocfg.dan_dim[0], ocfg.dan_dim[1], 99)
    # two runs of one step each give the update; a two-step run gives step 1
    got, _, _ = _product_run(ocfg, batches, masks, graphed=True, steps=2)
    eager, _, _ = _product_run(ocfg, batches, masks, graphed=False, steps=2)
    got1, dw, b2 = _product_run(ocfg, batches, masks, graphed=True, steps=1)
    emu, _, _ = _oracle_run(ocfg, batches, masks, emulate=True, steps=2)
    ref, _, _ = _oracle_run(ocfg, batches, masks, emulate=False, steps=2)
    _, dw_emu, b2_emu = _oracle_run(ocfg, batches, masks, emulate=True, steps=1)
    _, dw_ref, b2_ref = _oracle_run(ocfg, batches, masks, emulate=False, steps=1)
    load_package().set_precision("fp32")
    rep, bad = [], []
    for name in sorted(got[0]["losses"]):
        v, e = got[0]["losses"][name], emu[0]["losses"][name]
        assert got1[0]["losses"][name] == v, "the one-step and the two-step run must start identically"
        if abs(v - e) > 5e-2 * max(abs(e), 1e-2):
            bad.append((name, 0, v, e))
    scale = float(np.abs(dw_emu).max())
    d_pe, d_pr, d_er = (float(np.abs(a - b).max()) for a, b in ((dw, dw_emu), (dw, dw_ref), (dw_emu, dw_ref)))
    rep.append("one step at base_lr 0.01: fc6 weight movement (sampled, max|dw| %.2e): |p-emu| %.2e  |p-fp32| %.2e  |emu-fp32| %.2e"
               % (scale, d_pe, d_pr, d_er))
    if max(d_pe, d_pr) > 3.0 * d_er + 1e-2 * scale:
        bad.append(("fc6 weight movement", d_pe, d_pr, d_er))
    b_pe, b_pr, b_er = (float(np.abs(a - b).max()) for a, b in ((b2, b2_emu), (b2, b2_ref), (b2_emu, b2_ref)))
    rep.append("   fc7 bias: |p-emu| %.2e  |p-fp32| %.2e  |emu-fp32| %.2e" % (b_pe, b_pr, b_er))
    if max(b_pe, b_pr) > 3.0 * b_er + 1e-2 * float(np.abs(b2_ref).max()):
        bad.append(("fc7 bias", b_pe, b_pr, b_er))
    t = 1
    gi, ei, ri = (x[t]["img_scores"].astype(np.float64) for x in (got, emu, ref))
    d_pe, d_pr, d_er = np.abs(gi - ei).max(), np.abs(gi - ri).max(), np.abs(ei - ri).max()
    agree = d_er <= 5e-3
    rep.append("step 1  image scores: |p-emu| %.2e  |p-fp32| %.2e  |emu-fp32| %.2e  -> oracles %s" % (
        d_pe, d_pr, d_er, "agree: step 1 is pinned" if agree else "decorrelated (saturated MIL head): step 1 pinned through graphed == eager only"))
    if agree and max(d_pe, d_pr) > 3.0 * d_er + 1e-2 * np.abs(ri).max():
        bad.append(("img_scores", t, d_pe, d_pr, d_er))
    for name in sorted(got[t]["losses"]):
        v, e, r = got[t]["losses"][name], emu[t]["losses"][name], ref[t]["losses"][name]
        rep.append("   %-12s %.6f  emu %.6f  fp32 %.6f" % (name, v, e, r))
        if not np.isfinite(v) or abs(v - eager[t]["losses"][name]) > 1e-5 * max(abs(v), 1e-3):
            bad.append(("graphed != eager / not finite", name, v, eager[t]["losses"][name]))
        tol = 3.0 * abs(e - r) + 1e-2 * max(abs(r), 1e-2)
        if agree and (abs(v - e) > tol or abs(v - r) > tol):
            bad.append((name, t, v, e, r, tol))
    if agree:
        for k in range(ocfg.refine_num):
            mine = got[t]["pgt"][k][0]
            ce, ie = emu[t]["pgt"][k][0]
            _, ir = ref[t]["pgt"][k][0]
            for g, c in enumerate(ce):
                if int(mine[g]) not in (int(ie[g]), int(ir[g])):
                    col = emu[t]["prev"][k][:, int(c)]
                    ratio = float(col[int(mine[g])] / col.max())
                    rep.append("   pgt branch %d class %d: product row %d, emu %d, fp32 %d, score ratio %.4f" % (
                        k, int(c), int(mine[g]), int(ie[g]), int(ir[g]), ratio))
                    if ratio < 0.9:
                        bad.append(("pgt row not a near-tie", t, k, g, int(mine[g]), int(ie[g]), int(ir[g]), ratio))
    print("[bench-LR parity %s]\n%s" % (case, "\n".join(rep)))
    assert not bad, "%s\n%s" % (bad, "\n".join(rep))
