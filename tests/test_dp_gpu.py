"""Two REAL ranks on one MI355X (gloo backend on CUDA tensors; RCCL refuses two ranks per device): the N>1 training
step - GraphedTrainStep(split_tail=True): captured heads graph, eager fc6-dW slabs, per-bucket all-reduce on the
optimizer stream, deferred per-bucket SGD; with the sharded exchange: reduce-scatter per fc6 slab, update of the owned
rows only, all-gather of the updated compute copy - must (a) not deadlock, (b) leave both ranks with identical weights, and
(c) equal single-process training on the mean gradient of the two ranks' batches (= DDP semantics)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as G
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
NAME = "model_r50c4_tiny"


def _batches():
    d = G.load(NAME)
    base = G.batch_from(d)
    a = dict(base[0])
    b = dict(base[0])
    b["image"] = (255.0 - base[0]["image"]).contiguous()
    b["objectness_logits"] = base[0]["objectness_logits"].flip(0).contiguous()
    return int(d["seed"]), [G.drn_inputs([a]), G.drn_inputs([b])]


def _worker(rank, world, port, comm, exchange, q, precision="fp32"):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        load_package()
        from drn_wsod_pytorch_amd.engine import DataParallel, GraphedTrainStep, build_optimizer

        seed, batches = _batches()
        cfg, model = G.drn_model(G.MODEL_CASES[NAME], seed + 10 * rank, "cuda", 5, precision)  # ranks start different
        model.roi_heads.box_head.dropout_p = 0.0
        model.train()
        opt = build_optimizer(cfg, model)
        dp = DataParallel(model)
        assert dp.world == 2 and dp.exchange
        dp.broadcast_parameters(0)
        graphed_ks = exchange.startswith("fc6_kshard+graph")
        wire16 = exchange.endswith("+wire16")
        exchange = exchange.split("+")[0]
        opt.enable_pipelined(dp, slab_rows=[16, 48], comm_dtype=torch.bfloat16 if comm == "bf16" else torch.float32,
                             exchange=exchange, kshard_wire=torch.bfloat16 if wire16 else None)
        assert opt._sharded == (exchange == "sharded")
        if exchange == "fc6_kshard":
            assert model.roi_heads._engine.kshard is not None
        mine = batches[rank]
        losses = []
        if graphed_ks:
            # the same inside bench.py's schedule: pooling piece, fc6 GEMM + reduce-scatter and the dW tail eager around the
            # captured heads graph
            stepper = GraphedTrainStep(model, opt, mine, split_tail=True, trunk_pairs=True, eager_fc6=True)
            for _ in range(3):
                out = stepper.step(mine, mine, mine, mine)
                losses.append({k: float(v.detach()) for k, v in out.items()})
        elif exchange == "fc6_kshard":
            # K-sharded fc6 (round 4): eager steps - the forward holds collectives (all-gather of the ranks' feature maps,
            # reduce-scatter of the partial H1), the backward an all-gather of dP1; no fc6 gradient exchange at all
            assert opt._kshard and model.roi_heads._engine.kshard["world"] == 2
            for _ in range(3):
                opt.zero_grad()
                out = model(mine)
                sum(out.values()).backward()
                dp.finish()
                opt.step(dp.grad_scale)
                losses.append({k: float(v.detach()) for k, v in out.items()})
        else:
            stepper = GraphedTrainStep(model, opt, mine, split_tail=True, trunk_pairs=True)  # bench.py's default schedule
            for _ in range(3):
                out = stepper.step(mine, mine, mine, mine)
                losses.append({k: float(v.detach()) for k, v in out.items()})
        ev_scores = None
        if exchange == "fc6_kshard" and not graphed_ks:
            #  an evaluation forward between K-sharded training steps reads every column of fc1.weight: it must
            # gather the other rank's columns first (a collective: both ranks evaluate) - both ranks then see the SAME scores
            assert opt._master_stale
            model.eval()
            with torch.no_grad():
                _, sc, _ = model.inference(batches[0], do_postprocess=False)
            ev_scores = sc[0].float().cpu().numpy().copy()
            assert not opt._master_stale
            model.train()
        opt.sync_master()  # sharded exchange: the fp32 master rows the other rank owns (a collective; no-op otherwise)
        torch.cuda.synchronize()
        sd = {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters() if p.requires_grad}  # by value
        e = model.roi_heads._engine
        if e.arena_s is not None:  # bf16 mode: the compute copy every rank reads must be the rounded master, everywhere
            for _, _, o, n, used in e.segments:
                if used:
                    assert torch.equal(e.arena_s[o: o + n], e.arena_w[o: o + n].to(torch.bfloat16)), "stale shadow rows"
        q.put((rank, "ok", sd, losses, ev_scores))
    except Exception as ex:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: %r\n%s" % (ex, traceback.format_exc()), None, None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("comm,exchange", [("fp32", "allreduce"), ("bf16", "allreduce"), ("fp32", "sharded"),
                                           ("bf16", "sharded"), ("fp32", "fc6_kshard"), ("bf16", "fc6_kshard"),
                                           ("bf16", "fc6_kshard+graph"), ("bf16", "fc6_kshard+graph+wire16")])
def test_two_rank_step_equals_mean_gradient_training(comm, exchange):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, comm, exchange, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[1]
    # (b) replicas stay identical
    import numpy as np

    for n in res[0][2]:
        assert np.array_equal(res[0][2][n], res[1][2][n]), n
    if res[0][4] is not None:  # evaluation between K-sharded steps: gathered weights -> the same scores on both ranks
        assert np.array_equal(res[0][4], res[1][4]) and np.isfinite(res[0][4]).all()
    if exchange.endswith("+wire16"):
        # bf16 partial pre-activations on the wire perturb H1 by 2^-9 relative - in this fp32-precision fixture (a tiny,
        # ill-conditioned net) that is a different trajectory after three steps (measured 2e-3 .. 4e-2 on fc2.weight), so
        # only (a) no deadlock and (b) identical replicas are asserted for the option; (c) holds for the fp32 wire above
        assert all(np.isfinite(v) for l in res[0][3] for v in l.values())
        return
    # (c) single process, same start (rank 0's weights), mean of the two batches' gradients per step
    load_package()
    from drn_wsod_pytorch_amd.engine import build_optimizer

    seed, batches = _batches()
    cfg, model = G.drn_model(G.MODEL_CASES[NAME], seed, "cuda", 5, "fp32")
    model.roi_heads.box_head.dropout_p = 0.0
    model.train()
    opt = build_optimizer(cfg, model)
    for _ in range(3):
        opt.zero_grad()
        for b in batches:  # WSL.ITER_SIZE-style accumulation of loss / 2 == mean gradient over the two ranks
            (sum(model(b).values()) * 0.5).backward()
        opt.step()
    torch.cuda.synchronize()
    # fp32 buckets: sum of two gradients scaled by 1/2 in the SGD kernel vs two half-scaled gradients accumulated by
    # the GEMMs - the same numbers up to fp32 rounding of the scaling; bf16 buckets round every gradient once per rank
    # and once in the sum (measured: 1.1e-3 on fc7's weight after 3 steps)
    tol = 2e-6 if comm == "fp32" else 2e-3  # bf16: 3 steps x lr x 2^-8 relative rounding of a gradient of O(10)
    for n, p in model.named_parameters():
        if not p.requires_grad or n not in res[0][2]:
            continue
        diff = float((p.detach().cpu() - torch.from_numpy(res[0][2][n])).abs().max())
        scale = max(float(p.detach().abs().max()), 1.0)
        assert diff <= tol * scale, (n, diff)


def test_two_rank_sharded_exchange_bf16_mode():
    """bf16 compute mode + sharded exchange: only the bf16 shadow rows travel every step (the fp32 master and the momentum
    of the other rank's rows are stale until sync_master()).  After three steps and a sync: both replicas hold identical
    fp32 weights, every shadow row equals its rounded master row on both ranks (asserted inside the workers), losses finite."""
    import numpy as np

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, "bf16", "sharded", q, "bf16")) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[1]
    for n in res[0][2]:
        assert np.array_equal(res[0][2][n], res[1][2][n]), n
    assert all(np.isfinite(v) for l in res[0][3] for v in l.values())


def _run_two(comm, exchange, precision):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, comm, exchange, q, precision)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[1]
    return res


def test_two_rank_kshard_bf16_wire_is_bounded_by_the_compute_dtype():
    """ bench.py's N > 1 default reduce-scatters fc6's partial pre-activations in bf16.  In the benchmarked mode
    (bf16 compute: H1 is stored in bf16 whatever the wire carries) the wire's extra rounding must stay inside what the
    compute dtype itself costs: over three steps of bench.py's schedule, every loss and the fc7 weight of the bf16-wire run
    sit no further from the fp32-wire run than 3 x the distance between the bf16-mode and the fp32-mode runs (both on the
    fp32 wire) + 1 % - the bound tests/test_bench_mode_gpu.py uses for the compute dtype."""
    import numpy as np

    w32 = _run_two("bf16", "fc6_kshard+graph", "bf16")
    w16 = _run_two("bf16", "fc6_kshard+graph+wire16", "bf16")
    f32 = _run_two("bf16", "fc6_kshard+graph", "fp32")
    for res in (w32, w16, f32):
        for n in res[0][2]:
            assert np.array_equal(res[0][2][n], res[1][2][n]), n  # replicas identical under either wire
    for step in range(3):
        for k in w32[0][3][step]:
            a, b, c = w16[0][3][step][k], w32[0][3][step][k], f32[0][3][step][k]
            assert np.isfinite(a) and abs(a - b) <= 3.0 * abs(b - c) + 0.01 * max(abs(a), abs(b), 1e-3), (step, k, a, b, c)
    for n in ("roi_heads.box_head.fc2.weight", "roi_heads.box_head.fc1.weight"):
        d16 = float(np.abs(w16[0][2][n] - w32[0][2][n]).max())
        dmode = float(np.abs(w32[0][2][n] - f32[0][2][n]).max())
        scale = max(float(np.abs(f32[0][2][n]).max()), 1.0)
        assert d16 <= 3.0 * dmode + 1e-3 * scale, (n, d16, dmode)


def _worker_full(rank, world, port, q):
    """GraphedFullStep(parallel=dp): trainable trunk (FREEZE_AT = 2), two graphs around the eager exchange"""
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        load_package()
        from drn_wsod_pytorch_amd.engine import DataParallel, GraphedFullStep, build_optimizer

        seed, batches = _batches()
        cfg, model = G.drn_model(G.MODEL_CASES[NAME], seed + 10 * rank, "cuda", 2, "fp32")
        model.roi_heads.box_head.dropout_p = 0.0
        model.train()
        opt = build_optimizer(cfg, model)
        dp = DataParallel(model)
        dp.broadcast_parameters(0)
        stepper = GraphedFullStep(model, opt, batches[rank], parallel=dp)
        assert stepper.dp is dp and dp.sync_gradients  #  the shared flag is only off WHILE the step's backward runs
        losses = []
        for _ in range(4):
            out = stepper.step(batches[rank])
            losses.append({k: float(v.detach()) for k, v in out.items()})
            assert dp.sync_gradients
        assert stepper.g_opt is not None and stepper.g_trunk is not None  # heads all-reduce runs under the trunk-backward graph
        torch.cuda.synchronize()
        sd = {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters() if p.requires_grad}
        q.put((rank, "ok", sd, losses, None))
    except Exception as ex:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: %r\n%s" % (ex, traceback.format_exc()), None, None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_two_rank_graphed_full_step_trainable_trunk():
    """N > 1 with a trainable trunk as graphs (round 3): [forward + backward] graph -> eager all-reduce of the head
    gradient arena and the trunk's gradient arena -> [SGD of both arenas] graph.  Replicas stay bit-identical (every rank
    applies the same summed gradient), trunk tensors move, and the result matches single-process training on the mean
    gradient within the run-to-run spread of the trainable-trunk step (float atomics in the RoIPool backward)."""
    import numpy as np

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_full, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[1]
    assert any(n.startswith("backbone.") for n in res[0][2])
    for n in res[0][2]:
        assert np.array_equal(res[0][2][n], res[1][2][n]), n
    load_package()
    from drn_wsod_pytorch_amd.engine import build_optimizer

    seed, batches = _batches()
    cfg, model = G.drn_model(G.MODEL_CASES[NAME], seed, "cuda", 2, "fp32")
    model.roi_heads.box_head.dropout_p = 0.0
    model.train()
    before = {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters() if p.requires_grad}
    opt = build_optimizer(cfg, model)
    for _ in range(4):
        opt.zero_grad()
        for b in batches:
            (sum(model(b).values()) * 0.5).backward()
        opt.step()
    torch.cuda.synchronize()
    moved = 0
    for n, p in model.named_parameters():
        if not p.requires_grad or n not in res[0][2]:
            continue
        ref, got = p.detach().cpu().numpy(), res[0][2][n]
        assert np.abs(ref - got).max() <= 1.2e-2 * max(np.abs(ref).max(), 1e-3), n
        if n.startswith("backbone.") and not np.array_equal(got, before[n]):
            moved += 1
    assert moved > 0, "no trunk tensor moved under the two-rank graphed step"
