"""The training step as replayed hipGraphs (what sits around the hot path in projects/WSL/tools/train_net.py:65-117).

  GraphedTrainStep   frozen trunk (FREEZE_AT = 5, every shipped config): heads graph on the main stream, the conv chain of FUTURE
                     batches on a side stream, pooling piece + fc6 forward + fc6 weight-gradient tail issued eagerly around it
                     (bench.py's step)
  GraphedFullStep    trainable trunk: forward, losses, backward, optimizer as one graph (two with a gradient exchange between)

Split out of engine.py in round 5; `engine` re-exports both names."""
import os

import torch
import torch.distributed as dist

from . import ops
from ._cabi import DrnError


class GraphedTrainStep:
    """One training step as three hipGraphs captured once and replayed: a step is ~100 kernel launches of 2-450 us, so
    the eager Python host needs ~2.4 ms to enqueue what the GPU executes in ~2.1 ms; a replay costs the host ~9 us per
    node and nothing else.

      g_main (main stream) : heads forward (fc6 GEMM first, on the pooled operand prepared by the PREVIOUS step), MIL /
                             OICR losses, explicit backward up to the fc6 weight gradient
      tail   (eager)       : fc6 dW row-slab GEMMs on the main stream; per bucket, (all-reduce +) SGD on the optimizer
                             stream (`split_tail`, the default of bench.py; with split_tail=False the tail is part of
                             g_main - measured 1.3 % slower, the executor schedules forked branches late)
      g_bb   (side stream) : preprocess + frozen backbone of the NEXT batch's image (latency-bound convs under the GEMMs)
      g_pool (main stream) : ROIPool(+objectness) -> A and A^T of the next batch, behind this step's last reader of A^T
    One buffer set suffices.  Legal because every shipped config freezes the whole backbone (FREEZE_AT=5).

    lookahead=2 (bench.py's default): measured with HIP events, the conv chain of ONE image takes ~1.7 ms beside the
    GEMMs (0.6 ms alone) and the main stream idled 0.26 ms per step waiting for it.  With two feature buffers / image
    staging sets the trunk of batch t+2 runs during step t (two graphs per piece, one per slot), so the pooling graph of
    batch t+1 never waits; `step()` then takes the batch two steps ahead as a third argument (only its images are read).

    Static shapes only (fixed image size, proposals per image, images per GPU: the benchmark's case and the common
    fixed-R training case); anything else runs the eager path.

    N > 1 (`split_tail`): the captured heads graph stops in front of the fc6 weight-gradient GEMMs.  That tail - the
    announcement of the small gradients, the dW row-slab GEMMs and, from the optimizer's hooks, one RCCL all-reduce +
    SGD launch per bucket on the optimizer stream - is issued eagerly after the replay (~10 launches), so no
    collective is ever captured into a hipGraph, while the next image's backbone graph and pooling graph run under
    the exchange.  The optimizer stream is joined at the start of the next step."""

    _needs_frozen_trunk = True

    def __init__(self, model, optimizer, example_batch, split_tail=False, lookahead=1, trunk_pairs=False,
                 eager_fc6=False, stage_ahead=True, ring=True):
        if getattr(model, "cpg", False):
            raise DrnError("CSCROIHeads decides per step, on the host, which class maps to compute: eager steps only")
        if self._needs_frozen_trunk and any(p.requires_grad for p in model.backbone.parameters()):
            raise DrnError("GraphedTrainStep runs the trunk of FUTURE batches on a side stream, which needs a frozen "
                           "backbone (FREEZE_AT = 5); use GraphedFullStep for a trainable trunk")
        assert 1 <= lookahead <= 4
        self.lookahead = lookahead
        # eager_fc6 (bench.py): the fc6 forward GEMM - first launch of the heads and the step's dominant kernel - is
        # issued eagerly in front of the captured heads graph, like the fc6 dW tail behind it, so HIP events on the
        # launch stream can bracket it inside the timed region (ops.GEMM_TIMING)
        self.eager_fc6 = bool(eager_fc6)
        # stage_ahead: the next batch's labels are staged behind this step's heads graph and its proposals on the side
        # stream (False = the round-2 order, kept for A/B runs: proposals in front of the pooling graph on the main
        # stream, labels at the start of their own step, i.e. between the pooling kernel and the fc6 forward)
        self.stage_ahead = bool(stage_ahead)
        # (lookahead >= 2 / trunk groups) the pooling piece - one staging launch + the pooling kernel - is issued eagerly: a
        # graph's hand-over to the next launch costs more than an eager launch gap (+0.65 %, profiles/r2_39_eager_pool_ab.txt;
        # the graphed form of the piece was removed in round 5)
        # trunk_pairs: ONE conv chain per TWO batches (t+2 and t+3, launched on even steps): the chain is latency-bound,
        # so two images cost what one costs and the per-image chain time halves - for trunks whose chain is as long as
        # the step (WS-R101).  step() then takes (batch, next, t+2, t+3).
        self.trunk_pairs = bool(trunk_pairs)
        # (round 4) an int G > 2 generalises the pair to a GROUP of G batches per conv chain, launched every G-th step for
        # batches t+G .. t+2G-1: trunks whose chain is longer than TWO steps (WS-R101: ~100 launches, 0.7 ms of every 1.5-ms
        # step spent waiting for it) get G steps per chain.  step() then takes batches t .. t+2G-1.
        self.G = (2 if trunk_pairs is True else int(trunk_pairs)) if trunk_pairs else 0
        if self.trunk_pairs and self.G < 2:
            raise DrnError("trunk_pairs: True (two batches per conv chain) or an int >= 2")
        self.model, self.opt = model, optimizer
        self.heads = model.roi_heads
        self.engine = self.heads._engine
        dev = model.device
        K = self.heads.num_classes
        self.nper = [len(x["proposals"]) for x in example_batch]
        n_img, M = len(example_batch), sum(self.nper)
        self.n_img, self.K = n_img, K
        self.image = [x["image"].to(dev).float().clone() for x in example_batch]
        # lookahead 2: the trunk of batch t+2 runs during step t into the feature buffer batch t no longer needs, so the
        # conv chain (0.6 ms alone, ~1.7 ms beside the GEMMs) has two steps to finish instead of one
        # lookahead L >= 2: L slots; with L >= 3 the chains alternate over L-1 side streams, i.e. L-1 conv chains are in
        # flight at once and each may take L-1 steps - for trunks whose chain is longer than the step (WS-R101: ~100 launches)
        self._images = [self.image] + [[im.clone() for im in self.image] for _ in range(lookahead - 1)]
        self._feats = [None] * max(lookahead, 2)
        self._bb_done = [None] * max(lookahead, 2)
        self._t = 0
        off = [0]
        for n in self.nper:
            off.append(off[-1] + n)
        mk = lambda: torch.zeros((M, 5), dtype=torch.float32, device=dev)
        # Proposals / objectness / labels of the NEXT batch are staged on the side stream and read by the pooling piece at the end
        # of the step.  RING (round 6, trunk groups on one GPU): RING_SETS staging sets and RING_SLOTS trunk-feature slots, and
        # the side stream does NOT wait for the main stream at all - any such wait, even on an event that completed two steps
        # ago, costs ~1 % of the step (profiles/r6_40_side_wait_experiments.txt).  What a wait guaranteed - the previous reader of
        # a set / slot is done before the side stream overwrites it - the HOST guarantees instead: before it enqueues step t it
        # waits (a no-op in steady state: it runs ~1 step ahead) for the event the optimizer stream already waits on, recorded
        # behind the heads of step t - RING_LAG.  Set (t + 1) % RING_SETS was last read at the end of step t - RING_SETS, trunk slot
        # q % RING_SLOTS by the pooling launches of group q - RING_SLOTS (the last at the end of step t + 2G - 2 - G RING_SLOTS):
        # both in front of heads(t - RING_LAG) when RING_SETS >= RING_LAG + 1 and G RING_SLOTS >= 2G - 1 + RING_LAG.
        self.RING_LAG, self.RING_SETS, self.RING_SLOTS = 3, 4, 3
        self.rois, self._rois_nb = mk(), [mk() for _ in range(self.RING_SETS)]
        for t in [self.rois] + self._rois_nb:
            for i in range(n_img):
                t[off[i]: off[i + 1], 0] = float(i)
        self.obj = torch.zeros((M,), dtype=torch.float32, device=dev)
        self._obj_nb = [torch.zeros((M,), dtype=torch.float32, device=dev) for _ in range(self.RING_SETS)]
        self._ring_want = bool(ring)  # (False: the side stream waits for the main stream at the start of every step, as before round 6)
        self._ring_on = False  # (set by _prime_pairs)
        self._nb = 0          # the staging set in use (0 for every schedule but the ring)
        self._ring_evs = []   # events behind the heads of the last steps (oldest first)
        self.props = torch.zeros((M, 4), dtype=torch.float32, device=dev)
        # image-level labels live in ONE device block (one H2D copy per step): [onehot f32 | classes i32 | count i32]
        self._gt_block = torch.zeros((2 * n_img * K + n_img,), dtype=torch.int32, device=dev)
        # labels of the NEXT batch land here (H2D on the side stream); the pooling graph moves them into the block the
        # heads graph reads - a node inside a graph instead of an eager copy with two launch gaps on the main stream
        self._gt_nb = [torch.zeros_like(self._gt_block) for _ in range(self.RING_SETS)]
        nk = n_img * K
        self.gt = dict(onehot=self._gt_block[:nk].view(torch.float32).view(n_img, K),
                       classes=self._gt_block[nk: 2 * nk].view(n_img, K), count=self._gt_block[2 * nk:],
                       props=self.props, max_rows=max(self.nper))
        self.img_off = torch.tensor(off, dtype=torch.int32, device=dev)
        self.losses = None
        self._side = torch.cuda.Stream()
        self._primed = False
        self.split_tail = bool(split_tail)
        if self.engine.kshard is not None and not (self.split_tail and self.eager_fc6 and (lookahead >= 2 or trunk_pairs)):
            raise DrnError("K-sharded fc6 holds collectives in the pooling piece, behind the fc6 GEMM and in the dW tail: "
                           "GraphedTrainStep(split_tail=True, eager_fc6=True, lookahead >= 2 or trunk_pairs)")
        self.engine.defer_fc1_tail = self.split_tail
        self.engine.pool_sets_pinned = None  # a new step captures anew: the previous owner's pin (if any) is void
        self.engine.pool_sets_pin_owner = None

    def release(self):
        """give the fc6 operand sets back (they may grow again); the captured graphs of this object must not be replayed
        afterwards"""
        own = getattr(self.engine, "pool_sets_pin_owner", None)
        if own is not None and own() is self:
            self.engine.pool_sets_pinned = None
            self.engine.pool_sets_pin_owner = None
        self._primed = False

    # ---- host side of one step: stage inputs into the static buffers (tiny async copies) -----------------------
    @property
    def rois_next(self):
        return self._rois_nb[self._nb]

    @property
    def obj_next(self):
        return self._obj_nb[self._nb]

    @property
    def _gt_stage(self):
        return self._gt_nb[self._nb]

    def _stage_labels(self, batch, dst=None):
        """Image-level labels of the CURRENT batch.  They are built on the host and go through a ring of PINNED
        staging buffers so the H2D copies are truly asynchronous - a pageable source would block the host until the
        previous replay has drained and leave the GPU idle between replays."""
        ints = [torch.unique(x["instances"].gt_classes.cpu(), sorted=True) for x in batch]
        nk = self.n_img * self.K
        if not hasattr(self, "_ring"):
            self._ring = [dict(buf=torch.zeros_like(self._gt_block, device="cpu").pin_memory(), ev=None) for _ in range(8)]
            self._ring_i = 0
        slot = self._ring[self._ring_i]
        self._ring_i = (self._ring_i + 1) % len(self._ring)
        if slot["ev"] is not None:
            slot["ev"].synchronize()  # only blocks when the host is a full ring ahead of the GPU
        buf = slot["buf"]
        buf.zero_()
        oh = buf[:nk].view(torch.float32).view(self.n_img, self.K)
        cl = buf[nk: 2 * nk].view(self.n_img, self.K)
        for i, g in enumerate(ints):
            oh[i, g] = 1
            cl[i, : len(g)] = g.to(torch.int32)
            buf[2 * nk + i] = len(g)
        (self._gt_block if dst is None else dst).copy_(buf, non_blocking=True)
        slot["ev"] = torch.cuda.Event()
        slot["ev"].record()

    def _stage_labels_ahead(self, next_batch, via_stage=False):
        """Labels of the NEXT batch, issued during this step instead of at the start of their own (where a 4-us H2D copy
        and its two launch gaps sat between the pooling kernel and the fc6 forward).  via_stage (lookahead >= 2 / pairs,
        called on the side stream next to the proposals): into `_gt_stage`, from where the pooling graph - which runs
        behind this step's heads graph, the label block's reader - copies them into the block; otherwise straight into
        the block on the current stream, behind the heads graph.  step() skips its own staging when it is handed this
        very batch."""
        if next_batch is not None and self.stage_ahead:
            self._stage_labels(next_batch, self._gt_stage if via_stage else None)
            self._labels_for = next_batch

    def _stage_labels_now(self, batch):
        if getattr(self, "_labels_for", None) is not batch:
            self._stage_labels(batch)
        self._labels_for = None

    def _stage_next(self, batch):
        """image + proposals of the NEXT batch (device tensors: async D2D copies)"""
        self._stage_props(batch)
        self._stage_image(batch, 0)

    def _stage_props(self, batch):
        off = 0
        for i, x in enumerate(batch):
            n = self.nper[i]
            assert len(x["proposals"]) == n, "graphed step: proposals per image must stay fixed"
            bx, ob = x["proposals"].proposal_boxes.tensor, x["proposals"].objectness_logits
            self.rois_next[off: off + n, 1:].copy_(bx, non_blocking=True)
            self.obj_next[off: off + n].copy_(ob, non_blocking=True)
            for src in (bx, ob):
                if src.is_cuda:  # the caller's tensors may be read on a side stream: keep the allocator from reusing them
                    src.record_stream(torch.cuda.current_stream())
            off += n

    def _stage_image(self, batch, slot):
        for i, x in enumerate(batch):
            self._images[slot][i].copy_(x["image"], non_blocking=True)

    def _backbone(self, slot=0):
        m = self.model
        imgs = m.preprocess_image([{"image": im} for im in self._images[slot]])
        feats = m.backbone(imgs.tensor)
        f = feats[self.heads.box_in_features[0]].permute(0, 2, 3, 1)
        assert f.is_contiguous()
        return f

    def _pool_next(self, slot=None):
        """pooled fc6 operand (A, A^T) of the staged next batch + hand its proposals over to the heads"""
        feat = self.feat_next if slot is None else self._feats[slot]
        # the heads only need the pooled operand and the proposal boxes (pseudo-GT mining / IoU labelling) of a batch; the
        # small copy goes IN FRONT of the pooling kernel (its last reader, the previous heads graph, is long done) so that
        # nothing sits between the pooling and the fc6 forward
        # one launch: proposal boxes -> props, and (lookahead >= 2) the next batch's labels, staged on the side stream
        # (_stage_labels_ahead), -> the label block
        via = slot is not None and self.stage_ahead
        ops.stage_heads_inputs(self.rois_next, self.props, self._gt_stage if via else None, self._gt_block if via else None)
        if self.engine.kshard is not None:  # K-sharded fc6: this rank's channel slice of every rank's image (collectives, eager)
            self.pooled = self.engine.pool_kshard(feat, self.rois_next, self.obj_next)
            return
        self.pooled = self.engine.pool(feat, self.rois_next, self.obj_next, True, slot=0)
        self._check_pooled()

    def _check_pooled(self):
        """The captured heads graph reads the fc6 operand pair (A, A^T) at the addresses it saw when it was captured; the
        pooling piece is issued eagerly (eager_pool) and asks the head engine for its buffers on every step.  Once primed,
        the two must agree - a re-allocation in between (an inference pass used to replace the sets) would
        make the replayed graph read freed memory without any error."""
        ptrs = (self.pooled["A"].data_ptr(), self.pooled["AT"].data_ptr())
        if not getattr(self, "_primed", False):
            self._pool_ptrs = ptrs
            self.engine.pool_sets_pinned = (self.pooled["A"].dtype, True)
            import weakref

            # the pin belongs to THIS step object: it lapses when the object dies or another step is built on the model
            # (it used to outlive its owner and made the re-creation the error message asks for fail)
            self.engine.pool_sets_pin_owner = weakref.ref(self)
        elif ptrs != self._pool_ptrs:
            raise DrnError("the fc6 operand buffers moved after the step was captured (%s -> %s): re-create the "
                           "GraphedTrainStep" % (self._pool_ptrs, ptrs))

    # ---- the three captured pieces ---------------------------------------------------------------------------
    def _bb_body(self, slot=None):
        with torch.no_grad():
            if slot is None:
                self.feat_next.copy_(self._backbone())
            else:
                self._feats[slot].copy_(self._backbone(slot))

    @property
    def last_state(self):
        """intermediate values (MIL scores, image scores, pseudo-GT rows, labels) of the step that ran last"""
        return self._captured_state if getattr(self, "_replayed", False) else self._eager_state

    def _heads(self, eager):
        """fc6 forward (eager, when timed) + the heads graph of the current batch on the current stream"""
        self._fc6_eager()
        return self._main_body() if eager else (self.g_main.replay(), self.losses)[1]

    def _fc6_eager(self):
        if self.eager_fc6:
            self._fc6_part = self.engine.fc6_partials(self.pooled, self.rois.shape[0], True)

    def _main_body(self):
        losses, st = self.engine.forward(None, self.rois, self.obj, True, self.img_off, self.n_img, self.gt,
                                         pooled=self.pooled, fc6_part=self._fc6_part if self.eager_fc6 else None)
        # static buffers: after a replay the CAPTURED state holds that step's MIL scores / pseudo-GT / labels; the priming
        # step ran eagerly (its state is the one built before the capture)
        if torch.cuda.is_current_stream_capturing():
            self._captured_state = st
        else:
            self._eager_state = st
        # = sum(losses.values()).backward() without autograd's scalar adds / ones / stack launches
        self.engine.backward(st, None)   # pipelined SGD buckets fork onto the optimizer stream in here
        if not self.split_tail:
            self.opt.step(1.0)           # joins the optimizer stream
        return losses

    def _pool_body(self, slot=None):
        with torch.no_grad():
            self._pool_next(slot)

    # ---- lookahead L >= 2 -----------------------------------------------------------------------------------
    def _run2(self, eager, next_batch, far_batch):
        """Step t with the trunk L batches ahead.  Batch j lives in slot j % L (image staging set + feature buffer).
          side : image of batch t+L -> its slot, backbone graph of that slot (the slot's last reader, the pooling of batch t,
                 finished with the previous call: one wait on the main stream orders it); L-1 side streams take turns
          main : heads graph of batch t (+ eager tail), proposals of batch t+1, wait for the trunk of batch t+1 (launched
                 L-1 calls ago), pooling graph of its slot"""
        main = torch.cuda.current_stream()
        L = self.lookahead
        t = self._t
        s1, sL = (t + 1) % L, (t + L) % L
        side = self._sides[t % (L - 1)]
        side.wait_stream(main)
        losses = self._heads(eager)
        if self.split_tail:
            self.engine.run_fc1_tail()
        with torch.cuda.stream(side):
            evp = None
            if self.stage_ahead:
                self._stage_props(next_batch)  # in front of the conv chain, off the main stream (see _run_pairs)
                self._stage_labels_ahead(next_batch, via_stage=True)
                evp = torch.cuda.Event()
                evp.record(side)
            self._stage_image(far_batch, sL)
            self._bb_body(sL) if eager else self.g_bb2[sL].replay()
            ev = torch.cuda.Event()
            ev.record(side)
        if evp is not None:
            main.wait_event(evp)
        else:
            self._stage_props(next_batch)
        main.wait_event(self._bb_done[s1])
        self._bb_done[sL] = ev
        self._pool_body(s1)
        if self.split_tail:
            self.opt.step(1.0)
        self._t = t + 1
        return losses

    def _prime2(self, first_batch, next_batch, upcoming):
        """upcoming = [batch t+2, ..., batch t+L] (only their images are read)"""
        self.heads.train()
        main = torch.cuda.current_stream()
        L = self.lookahead
        self._sides = [self._side] + [torch.cuda.Stream() for _ in range(L - 2)]
        ahead = [next_batch] + list(upcoming[:-1])  # batches 1 .. L-1: their trunks run here, eagerly
        with torch.no_grad():
            self._stage_image(first_batch, 0)
            self._feats[0] = self._backbone(0).clone()
            self._stage_props(first_batch)
            self._pool_next(0)
            for j, b in enumerate(ahead, start=1):
                self._stage_image(b, j)
                self._feats[j] = self._backbone(j).clone()
                self._bb_done[j] = torch.cuda.Event()
                self._bb_done[j].record(main)
        self._stage_labels(first_batch)
        self.opt.zero_grad()
        self._t = 0
        first = {k: v.detach().clone() for k, v in self._run2(True, next_batch, upcoming[-1]).items()}
        self.opt.zero_grad()
        torch.cuda.synchronize()
        self.g_main = torch.cuda.CUDAGraph()
        self.g_bb2 = [torch.cuda.CUDAGraph() for _ in range(L)]
        for sl in range(L):
            with torch.cuda.graph(self.g_bb2[sl], capture_error_mode="thread_local"):
                self._bb_body(sl)
        with torch.cuda.graph(self.g_main, capture_error_mode="thread_local"):
            self.losses = self._main_body()
        self._primed = True
        return first

    # ---- trunk in pairs ---------------------------------------------------------------------------------------
    def _pair_backbone(self, ps):
        m = self.model
        imgs = m.preprocess_image([{"image": im} for im in self._pimages[ps]])
        f = m.backbone(imgs.tensor)[self.heads.box_in_features[0]].permute(0, 2, 3, 1)
        assert f.is_contiguous()
        return f

    def _pair_stage(self, group, ps):
        for i, x in enumerate([x for b in group for x in b]):
            self._pimages[ps][i].copy_(x["image"], non_blocking=True)

    def _pair_bb_body(self, ps):
        with torch.no_grad():
            self._pfeats[ps].copy_(self._pair_backbone(ps))

    def _pair_pool_body(self, ps, half):
        with torch.no_grad():
            n = self.n_img
            # in front of the pooling kernel (see _pool_next)
            ops.stage_heads_inputs(self.rois_next, self.props, self._gt_stage if self.stage_ahead else None,
                                   self._gt_block if self.stage_ahead else None)
            if self.engine.kshard is not None:
                self.pooled = self.engine.pool_kshard(self._pfeats[ps][half * n: (half + 1) * n], self.rois_next, self.obj_next)
                return
            self.pooled = self.engine.pool(self._pfeats[ps][half * n: (half + 1) * n], self.rois_next, self.obj_next, True,
                                           slot=0)
            self._check_pooled()

    def _run_pairs(self, eager, next_batch, *ahead):
        """Step t.  Batches 2k and 2k+1 form pair k, living in pair slot k % P (P = 2, or RING_SLOTS under the ring schedule).
        Even t: the conv chain of pair t/2 + 1 (batches t+2, t+3) starts on the side stream - with P = 2 its slot was last read by
        the pooling of batch t-1.  Every t: the
        pooling of batch t+1 reads its half of its pair's features (pair (t+1)/2, launched at step 2 ((t+1)/2) - 2)."""
        G = self.G
        main = torch.cuda.current_stream()
        t = self._t
        ring = self._ring_on and not eager
        if ring:
            while len(self._ring_evs) > self.RING_LAG - 1:
                self._ring_evs.pop(0).synchronize()  # heads(t - RING_LAG) are done: every earlier reader of the set / slot written below
            self._nb = (t + 1) % self.RING_SETS
        else:
            self._side.wait_stream(main)
        P = self.RING_SLOTS if self._ring_on else 2
        losses = self._heads(eager)
        evp = None
        probe = os.environ.get("DRN_PROBE_TAIL_FILL")  # experiment (tools/README: tail window of the fused dW launch)
        if probe:
            if not hasattr(self, "_probe_s"):
                self._probe_s = torch.cuda.Stream()
                self._probe_buf = torch.empty((int(probe) << 20,), dtype=torch.uint8, device=self.props.device)
            pe = torch.cuda.Event()
            pe.record(main)
        if self.split_tail:
            self.engine.run_fc1_tail()
        if ring:
            ev = getattr(self.opt, "small_ready_event", None)  # FusedSGD._on_grad_ready("small"): between the heads graph and the dW launch
            if ev is None:
                raise DrnError("the ring schedule needs the pipelined optimizer's hook on the fc6 tail")
            self.opt.small_ready_event = None
            self._ring_evs.append(ev)
        if probe:
            with torch.cuda.stream(self._probe_s):
                self._probe_s.wait_event(pe)
                self._probe_buf.fill_(1)  # as many bytes as the pooling launch writes: does the dW launch's last round have room for them?
            main.wait_stream(self._probe_s)
        with torch.cuda.stream(self._side):
            # proposals of batch t+1 on the side stream (behind the last reader of their staging set: the wait above, or under
            # the ring schedule the host's throttle; in front of the conv chain): three small launches that sat between the last dW slab and the
            # pooling graph on the main stream (22 us in the timeline)
            if self.stage_ahead:
                self._stage_props(next_batch)
                self._stage_labels_ahead(next_batch, via_stage=True)  # -> _gt_stage; the pooling graph hands them on
                evp = torch.cuda.Event()
                evp.record(self._side)
            if t % G == 0:
                ps = (t // G + 1) % P
                self._pair_stage(ahead[G - 2: 2 * G - 2], ps)  # batches t+G .. t+2G-1 (ahead[0] is batch t+2)
                self._pair_bb_body(ps) if eager else self.g_pbb[ps].replay()
                ev = torch.cuda.Event()
                ev.record(self._side)
                self._pdone[ps] = ev
        if evp is not None:
            main.wait_event(evp)
        else:
            self._stage_props(next_batch)
        k1, h1 = ((t + 1) // G) % P, (t + 1) % G
        main.wait_event(self._pdone[k1])
        self._pair_pool_body(k1, h1)
        if self.split_tail:
            self.opt.step(1.0)
        self._t = t + 1
        return losses

    def _prime_pairs(self, b0, b1, *ahead):
        self.heads.train()
        main = torch.cuda.current_stream()
        G = self.G
        # the ring: one GPU, the eager fc6 tail with the pipelined optimizer's hook (its event is the throttle's)
        self._ring_on = (self.engine.kshard is None and self.split_tail and getattr(self.engine, "grad_ready_hook", None) is not None
                         and self._ring_want)
        P = self.RING_SLOTS if self._ring_on else 2
        self._pimages = [[im.clone() for _ in range(G) for im in self.image] for _ in range(P)]
        self._pfeats, self._pdone = [None] * P, [None] * P
        with torch.no_grad():
            self._pair_stage([b0, b1] + list(ahead[: G - 2]), 0)
            self._pfeats[0] = self._pair_backbone(0).clone()
            for ps in range(1, P):
                self._pfeats[ps] = torch.zeros_like(self._pfeats[0])
            self._stage_props(b0)
            self._pair_pool_body(0, 0)
        self._pdone[0] = torch.cuda.Event()
        self._pdone[0].record(main)
        self._stage_labels(b0)
        self.opt.zero_grad()
        self._t = 0
        first = {k: v.detach().clone() for k, v in self._run_pairs(True, b1, *ahead).items()}
        self.opt.zero_grad()
        torch.cuda.synchronize()
        self.g_main = torch.cuda.CUDAGraph()
        self.g_pbb = [torch.cuda.CUDAGraph() for _ in range(P)]
        for ps in range(P):
            with torch.cuda.graph(self.g_pbb[ps], capture_error_mode="thread_local"):
                self._pair_bb_body(ps)
        with torch.cuda.graph(self.g_main, capture_error_mode="thread_local"):
            self.losses = self._main_body()
        self._primed = True
        return first

    def _run(self, eager, next_batch=None):
        """One step = three pieces on two torch streams, ordered by events exactly like eager multi-stream code.  (A
        single graph with the backbone as an internal branch was measured first: the HIP graph executor starts that
        branch late whatever the capture order, so the 49 latency-bound conv nodes ended up on the critical path.)
          side : backbone graph of the NEXT image     - may start once the previous step's pooling has read feat_next
          main : heads / losses / backward / SGD graph - reads the operand pooled by the previous step
          main : pooling graph of the NEXT batch       - behind this step's last reader of A^T, after the backbone"""
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)  # staged image is in place; previous pooling has consumed feat_next
        # submit the heads graph FIRST: submitting a graph costs the host ~9 us per node, and the backbone graph has
        # 49 nodes - issued first it would leave the main stream idle for ~0.45 ms in front of the fc6 GEMM
        losses = self._heads(eager)
        if self.split_tail:
            self.engine.run_fc1_tail()  # eager: dW slabs on this stream, all-reduce + SGD per bucket on the optimizer stream
        with torch.cuda.stream(self._side):
            if next_batch is not None:
                # the next batch's image / proposals are consumed by this stream's backbone graph and by the pooling graph
                # behind it: staging them here keeps five small copies off the front of the heads graph
                self._stage_next(next_batch)
            self._bb_body() if eager else self.g_bb.replay()
            done = torch.cuda.Event()
            done.record(self._side)
        self._stage_labels_ahead(next_batch)
        main.wait_event(done)
        self._pool_body() if eager else self.g_pool.replay()
        if self.split_tail:
            self.opt.step(1.0)  # join the optimizer stream: the exchange ran under the backbone + pooling above
        return losses

    def prime(self, first_batch, next_batch):
        """Step 0, eagerly (so every workspace exists and the captured SGD is not the momentum-initialising first
        step), then the captures.  Returns step 0's losses."""
        self.heads.train()
        self._stage_next(first_batch)
        with torch.no_grad():
            self.feat_next = self._backbone().clone()
            self._pool_next()
        self._stage_labels(first_batch)
        self._stage_next(next_batch)
        self.opt.zero_grad()
        first = {k: v.detach().clone() for k, v in self._run(eager=True).items()}
        self.opt.zero_grad()
        torch.cuda.synchronize()
        self.g_bb, self.g_main, self.g_pool = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: with N > 1 the RCCL watchdog thread polls its events while we capture; only this thread's calls
        # belong to the capture
        with torch.cuda.graph(self.g_bb, capture_error_mode="thread_local"):
            self._bb_body()
        with torch.cuda.graph(self.g_main, capture_error_mode="thread_local"):
            self.losses = self._main_body()
        with torch.cuda.graph(self.g_pool, capture_error_mode="thread_local"):
            self._pool_body()
        self._primed = True
        return first

    def step(self, batch, next_batch, *upcoming):
        """run the step for `batch` (which must be the batch passed as `next_batch` to the previous call); the same
        step prepares `next_batch` (backbone on the side stream, pooling behind the last dW GEMM).  With lookahead=L >= 2
        the caller also hands over the L-1 batches after that (`upcoming` = batches t+2 .. t+L: only their images are
        read); the backbone of the last one runs now.  With trunk_pairs: step(batch, next_batch, batch t+2, batch t+3)."""
        self._replayed = self._primed
        if self._primed:
            # the captured (or eagerly issued) SGD launches read lr / weight decay from device tables: follow the schedule
            self.opt.refresh_tables()
        if self.trunk_pairs:
            if len(upcoming) != 2 * self.G - 2:
                raise DrnError("GraphedTrainStep(trunk_pairs=%d).step needs batches t+2 .. t+%d" % (self.G, 2 * self.G - 1))
            if not self._primed:
                return self._prime_pairs(batch, next_batch, *upcoming)
            self._stage_labels_now(batch)
            return self._run_pairs(False, next_batch, *upcoming)
        if self.lookahead >= 2:
            if len(upcoming) != self.lookahead - 1:
                raise DrnError("GraphedTrainStep(lookahead=%d).step needs the %d batches after next_batch"
                               % (self.lookahead, self.lookahead - 1))
            if not self._primed:
                return self._prime2(batch, next_batch, list(upcoming))
            self._stage_labels_now(batch)
            return self._run2(False, next_batch, upcoming[-1])
        if not self._primed:
            return self.prime(batch, next_batch)
        self._stage_labels_now(batch)
        return self._run(eager=False, next_batch=next_batch)


class GraphedFullStep(GraphedTrainStep):
    """The training step of a TRAINABLE trunk (MODEL.BACKBONE.FREEZE_AT < 5) as ONE hipGraph on one stream: preprocess,
    trunk forward with saved activations, ROIPool (with arg-max), heads, losses, the explicit backward through the
    heads, fc6 dX, RoIPool / ROIAlign backward and every trainable trunk block (conv dgrad / wgrad straight into the
    trunk's gradient arena), then the fused SGD step of both arenas and the re-pack of the updated conv weights at the
    top of the next replay.  ~250 launches of 5-60 us that the eager Python host cannot enqueue as fast as the GPU runs
    them; a replay costs one graph launch.  Nothing of a future batch can run ahead here - the trunk's weights change
    every step - so there is no side stream and no lookahead: step(batch) stages `batch` and replays.
    Static shapes (image size, proposals per image, images per GPU), ITER_SIZE 1.
    N > 1 (`parallel` = the model's DataParallel, round 3): no collective is ever captured - the step becomes TWO graphs
    around an eager exchange: [forward + backward] -> all-reduce of the head engine's gradient arena (everything up to the
    end of fc1.weight: the unused bbox_pred tail never travels) and of the trunk's flat gradient arena -> [SGD of both
    arenas with 1 / world + zero_grad].  DataParallel's per-bucket hooks stay silent (`sync_gradients` off): the buckets
    only pay when the optimizer can update one while the next is still being produced, which the plain optimizer step of a
    trainable trunk does not do."""

    _needs_frozen_trunk = False

    def __init__(self, model, optimizer, example_batch, parallel=None):
        if getattr(optimizer, "_pipelined", False):
            raise DrnError("GraphedFullStep uses the plain optimizer step (the pipelined mode assumes a frozen trunk)")
        super().__init__(model, optimizer, example_batch, split_tail=False, lookahead=1)
        self.g_step = None
        self.g_trunk = None
        self.g_opt = None
        if parallel is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            #  without the DataParallel object the step would run with NO gradient exchange and no error
            raise DrnError("GraphedFullStep in a %d-rank job needs parallel=DataParallel(model): without it the replicas "
                           "train on their local gradients and diverge" % dist.get_world_size())
        self.dp = parallel if (parallel is not None and parallel.exchange) else None
        self._comm = None

    def _stage(self, batch):
        self._stage_labels(batch)
        off = 0
        for i, x in enumerate(batch):
            n = self.nper[i]
            assert len(x["proposals"]) == n, "graphed step: proposals per image must stay fixed"
            self.rois[off: off + n, 1:].copy_(x["proposals"].proposal_boxes.tensor, non_blocking=True)
            self.obj[off: off + n].copy_(x["proposals"].objectness_logits, non_blocking=True)
            off += n
        self.props.copy_(self.rois[:, 1:])
        self._stage_image(batch, 0)

    def _fwd_bwd(self):
        m, eng = self.model, self.engine
        imgs = m.preprocess_image([{"image": im} for im in self._images[0]])
        feats = m.backbone(imgs.tensor)  # training mode + trainable blocks: activations are kept for backward_nhwc()
        f = feats[self.heads.box_in_features[0]].permute(0, 2, 3, 1)
        assert f.is_contiguous()
        # N > 1: the trunk's backward is a piece of its own (_trunk_bwd), so that the all-reduce of the heads' gradient arena -
        # final when the heads' backward ends - runs under it; the hook only keeps what that piece needs
        eng.feature_grad_hook = m._backbone_backward if self.dp is None else self._keep_feature_grad
        losses, st = eng.forward(f, self.rois, self.obj, True, self.img_off, self.n_img, self.gt)
        if torch.cuda.is_current_stream_capturing():
            self._captured_state = st
        else:
            self._eager_state = st
        # the per-bucket hooks of DataParallel (eager trainer) are replaced by _exchange_* below: off while THIS step's
        # backward runs, restored right after (the flag used to stay off on the shared object for good)
        flag = None if self.dp is None else self.dp.sync_gradients
        if self.dp is not None:
            self.dp.sync_gradients = False
        try:
            eng.backward(st, None)
        finally:
            if self.dp is not None:
                self.dp.sync_gradients = flag
        return losses

    def _keep_feature_grad(self, dfeat, acc):
        self._dfeat = (dfeat, acc)

    def _trunk_bwd(self):
        dfeat, acc = self._dfeat
        self.model._backbone_backward(dfeat, acc)

    def _opt_body(self):
        self.opt.step(1.0 if self.dp is None else self.dp.grad_scale)
        self.opt.zero_grad()

    def _exchange_heads(self):
        """all-reduce of the heads' gradient arena (the SGD kernels apply 1 / world) on the exchange stream, started the
        moment the heads' backward is queued: it runs under the trunk's backward (round 4; one eager call between two
        graphs with nothing beside it before)"""
        e = self.engine
        o, n = e._seg["fc1.weight"]
        main = torch.cuda.current_stream()
        if self._comm is None:
            self._comm = torch.cuda.Stream()
        self._comm.wait_stream(main)
        with torch.cuda.stream(self._comm):
            dist.all_reduce(e.arena_g[: o + n], group=self.dp.group)

    def _exchange_trunk(self):
        """all-reduce of the trunk's gradient arena behind the heads' one (same stream: RCCL runs one collective of a
        communicator at a time anyway), then the main stream joins"""
        main = torch.cuda.current_stream()
        bg = getattr(self.model, "_bb_grad_arena", None)
        if bg is not None:
            self._comm.wait_stream(main)
            with torch.cuda.stream(self._comm):
                dist.all_reduce(bg, group=self.dp.group)
        main.wait_stream(self._comm)

    def _full_body(self):
        losses = self._fwd_bwd()
        if self.dp is not None:
            self._exchange_heads()
            self._trunk_bwd()
            self._exchange_trunk()
        self._opt_body()
        return losses

    def step(self, batch):
        self.heads.train()
        self._replayed = self.g_step is not None
        self._stage(batch)
        if self.g_step is None:
            # step 0 eagerly (workspaces exist, momentum buffers initialised, packs invalidated by the update), then the
            # capture: the captured forward starts with the re-pack of the freshly updated conv weights
            self.opt.zero_grad()
            first = {k: v.detach().clone() for k, v in self._full_body().items()}
            torch.cuda.synchronize()
            self.g_step = torch.cuda.CUDAGraph()
            if self.dp is None:
                with torch.cuda.graph(self.g_step, capture_error_mode="thread_local"):
                    self.losses = self._full_body()
            else:
                # the capture pass RUNS nothing: the gradient arenas and weights are exactly as step 0 left them
                with torch.cuda.graph(self.g_step, capture_error_mode="thread_local"):
                    self.losses = self._fwd_bwd()
                self.g_trunk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_trunk, capture_error_mode="thread_local"):
                    self._trunk_bwd()
                self.g_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_opt, capture_error_mode="thread_local"):
                    self._opt_body()
            return first
        self.opt.refresh_tables()
        self.g_step.replay()
        if self.dp is not None:
            self._exchange_heads()   # exchange stream: under the trunk's backward
            self.g_trunk.replay()
            self._exchange_trunk()
            self.g_opt.replay()
        return self.losses
