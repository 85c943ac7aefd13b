"""Host logic of the trainer-side pieces (engine.py) that needs no GPU: LR-schedule stepping per micro-iteration
(detectron2/engine/hooks.py:232-235), DDP no_sync semantics for WSL.ITER_SIZE > 1, optimizer / scheduler checkpoint
state (detectron2/engine/defaults.py:304-319), in-place refresh of the device-side lr tables, hook ownership."""
import io

import pytest
import torch

import golden_util as G
from __graft_entry__ import load_package

load_package()
from drn_wsod_pytorch_amd._cabi import DrnError  # noqa: E402
from drn_wsod_pytorch_amd.engine import (DataParallel, FusedSGD, Trainer, WarmupMultiStepLR,  # noqa: E402
                                         build_optimizer)
from drn_wsod_pytorch_amd.modeling import build_model  # noqa: E402

NAME = "model_r50c4_tiny"


def _cpu_model():
    cfg = G.drn_cfg(G.MODEL_CASES[NAME], "cpu")
    return cfg, build_model(cfg)


class _StubOpt:
    def __init__(self):
        self.param_groups = [{"lr": 0.01, "initial_lr": 0.01}]
        self.steps, self.zeroed = 0, 0

    def step(self, scale=1.0):
        self.steps += 1

    def zero_grad(self):
        self.zeroed += 1


class _StubDP:
    grad_scale = 1.0

    def __init__(self):
        self.sync_gradients = True
        self.seen, self.finished = [], 0

    def finish(self):
        self.finished += 1


def _stub_trainer(iter_size, scheduler=True, start_iter=0):
    cfg, _ = _cpu_model()
    cfg.WSL.ITER_SIZE = iter_size
    dp = _StubDP()
    w = torch.zeros((), requires_grad=True)

    class M:
        training = True

        def __call__(self, data):
            return {"loss_cls": (w * 2.0).sum()}

    class Inst:
        def __len__(self):
            return 1

    def it():
        while True:
            yield [{"instances": Inst()}]

    opt = _StubOpt()
    sched = WarmupMultiStepLR(opt, [4, 8], gamma=0.1, warmup_factor=0.001, warmup_iters=3) if scheduler else None
    return Trainer(cfg, M(), it(), optimizer=opt, scheduler=sched, parallel=dp, start_iter=start_iter), opt, dp, sched


def test_scheduler_steps_every_micro_iteration_and_no_sync_window():
    tr, opt, dp, sched = _stub_trainer(iter_size=4)
    assert sched.last_epoch == 0
    syncs = []
    for i in range(9):
        tr.run_step()
        syncs.append(dp.sync_gradients)
        # hooks.LRScheduler.after_step: one scheduler step per iteration whatever ITER_SIZE is
        assert sched.last_epoch == i + 1
    # train_net.py:105: optimizer steps when iter % ITER_SIZE == 0 -> iterations 0, 4, 8; only those exchange gradients
    assert opt.steps == 3 and dp.finished == 3
    assert syncs == [True, False, False, False, True, False, False, False, True]
    # milestones are in micro-iteration units: after 9 iterations last_epoch = 9 >= 8 -> two decays
    assert opt.param_groups[0]["lr"] == pytest.approx(0.01 * 0.1 * 0.1)


def test_trainer_start_iter_and_pipelined_guard():
    tr, opt, dp, _ = _stub_trainer(iter_size=2, start_iter=7)
    assert tr.iter == 7 and tr.start_iter == 7
    tr.run_step()  # iteration 7: first iteration zeroes the gradients, 7 % 2 != 0 -> no optimizer step
    assert opt.zeroed == 1 and opt.steps == 0 and tr.iter == 8
    tr.run_step()
    assert opt.steps == 1
    cfg, model = _cpu_model()
    cfg.WSL.ITER_SIZE = 2
    o = build_optimizer(cfg, model)
    o.enable_pipelined()
    with pytest.raises(DrnError):
        Trainer(cfg, model, iter([]), optimizer=o)


def test_scheduler_state_dict_roundtrip():
    a, b = _StubOpt(), _StubOpt()
    s = WarmupMultiStepLR(a, [5, 9], gamma=0.5, warmup_factor=0.01, warmup_iters=4)
    for _ in range(7):
        s.step()
    buf = io.BytesIO()
    torch.save({"scheduler": s.state_dict()}, buf)
    buf.seek(0)
    t = WarmupMultiStepLR(b, [1], gamma=0.9, warmup_iters=0)
    t.load_state_dict(torch.load(buf)["scheduler"])
    assert t.last_epoch == s.last_epoch == 7 and t.milestones == [5, 9]
    assert b.param_groups[0]["lr"] == a.param_groups[0]["lr"]
    s.step(), t.step()
    assert b.param_groups[0]["lr"] == a.param_groups[0]["lr"]


def test_optimizer_state_dict_roundtrip_copies_into_device_arena():
    cfg, model = _cpu_model()
    opt = build_optimizer(cfg, model)
    e = opt.engine
    opt._mom = torch.randn_like(e.arena_w)
    opt._steps = 5
    opt.param_groups[0]["lr"] = 0.123
    buf = io.BytesIO()
    torch.save({"optimizer": opt.state_dict()}, buf)
    buf.seek(0)
    sd = torch.load(buf, map_location="cpu")["optimizer"]
    cfg2, model2 = _cpu_model()
    opt2 = build_optimizer(cfg2, model2)
    opt2._segs()  # a device table exists before the load: it must be rewritten in place
    table = opt2._segs_dev
    ptr = table.data_ptr()
    opt2.load_state_dict(sd)
    assert opt2._steps == 5 and opt2.param_groups[0]["lr"] == 0.123
    assert opt2._mom is not sd["momentum_buffer"], "the checkpoint tensor must be copied, not adopted"
    assert opt2._mom.device == opt2.engine.arena_w.device and opt2._mom.shape == opt2.engine.arena_w.shape
    assert torch.equal(opt2._mom, opt._mom)
    assert opt2._segs_dev.data_ptr() == ptr
    import numpy as np

    rows = np.frombuffer(opt2._segs_dev.cpu().numpy().tobytes(),
                         dtype=[("off", "<i8"), ("cnt", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
    assert rows[0]["lr"] == np.float32(0.123)
    # a checkpoint of a different model is refused, not silently adopted
    bad = dict(sd)
    bad["momentum_buffer"] = torch.zeros(3)
    with pytest.raises(DrnError):
        opt2.load_state_dict(bad)


def test_refresh_tables_in_place_after_lr_change():
    import numpy as np

    cfg, model = _cpu_model()
    opt = build_optimizer(cfg, model)
    opt.enable_pipelined()
    small, n_small = opt._bucket_table("small")
    d1 = model.roi_heads.box_head.fc1.weight.shape[0]
    slab, n_slab = opt._bucket_table(("fc1", 0, d1))
    ptrs = (small.data_ptr(), slab.data_ptr())
    dt = [("off", "<i8"), ("cnt", "<i8"), ("lr", "<f4"), ("wd", "<f4")]
    before = np.frombuffer(slab.numpy().tobytes(), dtype=dt)["lr"].copy()
    sched = WarmupMultiStepLR(opt, [2], gamma=0.1, warmup_iters=0)
    sched.step(), sched.step()  # last_epoch 2 -> decayed
    opt.refresh_tables()
    assert (opt._bucket_segs["small"][1].data_ptr(), opt._bucket_segs[("fc1", 0, d1)][1].data_ptr()) == ptrs
    after = np.frombuffer(slab.numpy().tobytes(), dtype=dt)["lr"]
    assert np.allclose(after, before * 0.1)
    lr_small = np.frombuffer(small.numpy().tobytes(), dtype=dt)["lr"]
    want = [g["lr"] for g in opt.param_groups if g["used"] and g["name"] != "fc1.weight"]
    assert np.allclose(lr_small, np.asarray(want, dtype=np.float32))


def test_single_process_dataparallel_keeps_the_pipelined_hook():
    cfg, model = _cpu_model()
    opt = build_optimizer(cfg, model)
    opt.enable_pipelined()
    hook = model.roi_heads._engine.grad_ready_hook
    assert hook is not None
    dp = DataParallel(model)  # what Trainer builds by default
    assert dp.world == 1 and model.roi_heads._engine.grad_ready_hook is hook
