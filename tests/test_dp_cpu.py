"""world_size-2 gloo test (CPU) of the data-parallel gradient exchange: the bucket schedule that follows the
explicit backward (small tensors first, then fc6-gradient row slabs) must leave every rank with the SUM of the
ranks' gradients for all used parameters, must never touch the unused bbox_pred tail, and the
parameter broadcast must make ranks identical.  No HIP kernel is involved: only the exchange logic runs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as G


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package

        load_package()
        from drn_wsod_pytorch_amd.engine import DataParallel
        from drn_wsod_pytorch_amd.modeling import build_model

        torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix that
        model = build_model(G.drn_cfg(G.MODEL_CASES["model_r50c4_tiny"], "cpu"))
        dp = DataParallel(model, slabs=3, backend_stream=False)
        assert dp.world == 2 and dp.grad_scale == 0.5
        dp.broadcast_parameters(0)
        e = model.roi_heads._engine
        w0 = e.arena_w.clone()
        gathered = [torch.zeros_like(w0) for _ in range(world)]
        dist.all_gather(gathered, w0)
        assert torch.equal(gathered[0], gathered[1]), "broadcast_parameters must make ranks identical"
        # emulate the explicit backward: fill gradients with rank-dependent values, fire the hooks in order
        n = e.arena_g.numel()
        base = torch.arange(n, dtype=torch.float32) * 1e-3
        e.arena_g.copy_(base * (rank + 1))
        d1, k1 = model.roi_heads.box_head.fc1.weight.shape
        e.grad_ready_hook("small")
        rows = (d1 + e.fc1_grad_slabs - 1) // e.fc1_grad_slabs
        for s in range(e.fc1_grad_slabs):
            r0, r1 = s * rows, min(d1, (s + 1) * rows)
            if r0 < r1:
                e.grad_ready_hook(("fc1", r0, r1))
        dp.finish()
        exp = base * 3.0  # ranks contribute 1x and 2x
        used_end = max(o + c for _, _, o, c, u in e.segments if u)
        o_fc1, c_fc1 = e._seg["fc1.weight"]
        assert torch.allclose(e.arena_g[:o_fc1], exp[:o_fc1])
        assert torch.allclose(e.arena_g[o_fc1: o_fc1 + c_fc1], exp[o_fc1: o_fc1 + c_fc1])
        for name, _, o, c, used in e.segments:
            if not used:  # unused bbox_pred: never reduced
                assert torch.equal(e.arena_g[o: o + c], base[o: o + c] * (rank + 1)), name
        assert used_end <= o_fc1 + c_fc1
        # trainable trunk (FREEZE_AT < 5): its flat gradient arena is one more bucket, announced after its backward
        model._bb_grad_arena = torch.arange(37, dtype=torch.float32) * (rank + 1)
        e.grad_ready_hook("backbone")
        dp.finish()
        assert torch.equal(model._bb_grad_arena, torch.arange(37, dtype=torch.float32) * 3.0)
        del model._bb_grad_arena
        # WSL.ITER_SIZE > 1 (DDP no_sync): micro-steps accumulate locally, only the window's last backward exchanges.
        # Two micro-steps per rank with gradients g1 = base*(rank+1), g2 = 2*base*(rank+1): after the window every rank
        # must hold sum_r (g1_r + g2_r) = 9*base - what a single process accumulating all four micro-batches holds
        def fire():
            e.grad_ready_hook("small")
            for s in range(e.fc1_grad_slabs):
                r0, r1 = s * rows, min(d1, (s + 1) * rows)
                if r0 < r1:
                    e.grad_ready_hook(("fc1", r0, r1))

        e.arena_g.copy_(base * (rank + 1))
        dp.sync_gradients = False
        fire()
        dp.finish()
        assert torch.equal(e.arena_g, base * (rank + 1)), "no exchange inside the accumulation window"
        e.arena_g.add_(2.0 * base * (rank + 1))  # the second micro-step's GEMMs accumulate on top
        dp.sync_gradients = True
        fire()
        dp.finish()
        assert torch.allclose(e.arena_g[:o_fc1 + c_fc1], (base * 9.0)[:o_fc1 + c_fc1])
        # the pipelined optimizer's own exchange (what the N>1 bench step uses): in bf16 mode the small bucket is cast
        # into a bf16 wire buffer (the fp32 arena keeps the local gradient), the fc6 row slabs come from the bf16
        # exchange buffer the dW GEMM writes into
        from drn_wsod_pytorch_amd.engine import build_optimizer

        opt = build_optimizer(G.drn_cfg(G.MODEL_CASES["model_r50c4_tiny"], "cpu"), model)
        opt.enable_pipelined(dp, slab_rows=[16, 40, d1], comm_dtype=torch.bfloat16, exchange="allreduce")
        assert opt._slab_ends == [16, 40, d1] and e.fc1_grad_bucket.dtype == torch.bfloat16
        e.arena_g.copy_(base * (rank + 1))
        vals = (torch.arange(d1 * k1, dtype=torch.float32).view(d1, k1) % 61) * 0.25  # exactly representable in bf16
        e.fc1_grad_bucket.copy_((vals * (rank + 1)).to(torch.bfloat16))
        small = opt._exchange("small")
        assert small is opt._small_bucket and small.dtype == torch.bfloat16 and small.numel() == o_fc1
        r0 = 0
        for r1 in opt._slab_ends:
            assert opt._exchange(("fc1", r0, r1)) is e.fc1_grad_bucket
            r0 = r1
        assert torch.allclose(small.float(), exp[:o_fc1], rtol=2.0 ** -7, atol=0)  # two bf16 roundings + a bf16 sum
        assert torch.equal(e.arena_g[:o_fc1], base[:o_fc1] * (rank + 1))  # local fp32 gradient untouched
        assert torch.equal(e.arena_g[o_fc1: o_fc1 + c_fc1], base[o_fc1: o_fc1 + c_fc1] * (rank + 1))  # arena fc6 slot unused
        assert torch.equal(e.fc1_grad_bucket.float(), vals * 3.0)
        # ---- sharded exchange (SURVEY 8(e) target): reduce-scatter of each slab, owned rows, all-gather -------------
        opt.enable_pipelined(dp, slab_rows=[16, 40, d1], comm_dtype=torch.bfloat16, exchange="sharded")
        assert opt._sharded
        e.fc1_grad_bucket.copy_((vals * (rank + 1)).to(torch.bfloat16))
        e.arena_s = torch.zeros_like(e.arena_w, dtype=torch.bfloat16)  # the bf16 shadow the forward reads
        opt._mom = torch.zeros_like(e.arena_w)
        r0 = 0
        for r1 in opt._slab_ends:
            what = ("fc1", r0, r1)
            out = opt._exchange(what)
            a, b = opt._own_rows(what)
            nq = (r1 - r0) // 2
            assert (a, b) == (r0 + rank * nq, r0 + (rank + 1) * nq) and out.shape == (nq, k1)
            assert torch.equal(out.float(), vals[a:b] * 3.0), "reduce-scatter: the summed gradient of this rank's rows"
            # stand-in for the owned-shard SGD kernel: mark the rows this rank updated, in every arena
            e.arena_s[o_fc1 + a * k1: o_fc1 + b * k1] = float(rank + 1)
            e.arena_w[o_fc1 + a * k1: o_fc1 + b * k1] = 10.0 * (rank + 1)
            opt._mom[o_fc1 + a * k1: o_fc1 + b * k1] = 100.0 * (rank + 1)
            opt._gather_rows(what)  # per step: only the compute copy travels
            sh = e.arena_s[o_fc1 + r0 * k1: o_fc1 + r1 * k1].view(r1 - r0, k1).float()
            assert torch.equal(sh[:nq], torch.ones(nq, k1)) and torch.equal(sh[nq:], 2.0 * torch.ones(nq, k1))
            r0 = r1
        opt._master_stale = True
        opt.sync_master()  # checkpoint time: fp32 master + momentum of the other rank's rows arrive too
        wv = e.arena_w[o_fc1: o_fc1 + c_fc1].view(d1, k1)
        mv = opt._mom[o_fc1: o_fc1 + c_fc1].view(d1, k1)
        r0 = 0
        for r1 in opt._slab_ends:
            nq = (r1 - r0) // 2
            assert torch.equal(wv[r0: r0 + nq], torch.full((nq, k1), 10.0)) and torch.equal(wv[r0 + nq: r1], torch.full((nq, k1), 20.0))
            assert torch.equal(mv[r0: r0 + nq], torch.full((nq, k1), 100.0)) and torch.equal(mv[r0 + nq: r1], torch.full((nq, k1), 200.0))
            r0 = r1
        with_odd = False
        try:
            opt.enable_pipelined(dp, slab_rows=[15, d1], comm_dtype=torch.float32, exchange="sharded")
        except Exception:  # noqa: BLE001 - a slab that does not split evenly over the ranks is refused loudly
            with_odd = True
        assert with_odd
        opt.enable_pipelined(dp, slab_rows=[15, d1], comm_dtype=torch.float32)  # default: falls back to the all-reduce
        assert not opt._sharded
        e.arena_s = None
        # fp32 wire: the small bucket is reduced in place in the arena
        opt.enable_pipelined(dp, slab_rows=[16, 40, d1], comm_dtype=torch.float32, exchange="allreduce")
        assert opt._exchange("small") is None
        assert torch.allclose(e.arena_g[:o_fc1], exp[:o_fc1])
        # the collective self-test bench.py runs in front of its warm-up (DataParallel.selftest): sizes of this tiny model
        res = dp.selftest({"small": o_fc1, "slabs": [(16, k1), (24, k1), (d1 - 40, k1)]}, iters=2, timeout=30.0,
                          wire_dtype=torch.float32)
        assert "all_reduce small bucket" in res and res["all_reduce small bucket"]["bytes"] == 4 * o_fc1
        for i, rows_ in enumerate((16, 24, d1 - 40)):
            if rows_ % 2 == 0:
                assert res["reduce_scatter fc6 slab %d" % i]["ms"] > 0 and res["all_gather fc6 slab %d" % i]["busbw_GBps"] > 0
            else:
                assert "reduce_scatter fc6 slab %d" % i not in res  # the step falls back to the all-reduce there
        # (round 5) ... and the K-sharded fc6's own collectives, which bench.py adds for N > 1: packed feature all-gather,
        # reduce-scatter of the [N*M x D1] partial pre-activations in the wire dtype, all-gather of dP1
        res = dp.selftest({"small": 16, "slabs": [], "kshard": {"pack_bytes": 1000, "M": 24, "D1": d1, "wire_dtype": torch.float32,
                                                                   "dp1_dtype": torch.bfloat16}}, iters=1, timeout=30.0)
        for key_ in ("all_gather feature pack (fc6_kshard)", "reduce_scatter H1 partials (fc6_kshard)", "all_gather dP1 (fc6_kshard)"):
            assert res[key_]["ms"] > 0, key_
        assert res["reduce_scatter H1 partials (fc6_kshard)"]["bytes"] == 2 * 24 * d1 * 4
        # re-entrant exchange selection (bench.py's fallback chain): K-sharded -> sharded -> all-reduce on one optimizer
        opt.enable_pipelined(dp, comm_dtype=torch.float32, exchange="fc6_kshard")
        assert opt._kshard and e.kshard is not None and e.kshard["world"] == 2 and callable(e.kshard["sync"])
        opt.enable_pipelined(dp, slab_rows=[16, 40, d1], comm_dtype=torch.float32, exchange="sharded")
        assert not opt._kshard and e.kshard is None and opt._sharded
        opt.enable_pipelined(dp, slab_rows=[16, 40, d1], comm_dtype=torch.float32, exchange="allreduce")
        assert not opt._kshard and not opt._sharded and e.kshard is None
        q.put((rank, "ok"))
    except Exception as ex:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: %r\n%s" % (ex, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def test_gradient_exchange_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _lone_saver(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package

        load_package()
        from drn_wsod_pytorch_amd._cabi import DrnError
        from drn_wsod_pytorch_amd.checkpoint import DetectionCheckpointer
        from drn_wsod_pytorch_amd.engine import DataParallel, build_optimizer
        from drn_wsod_pytorch_amd.modeling import build_model

        torch.manual_seed(5)
        cfg = G.drn_cfg(G.MODEL_CASES["model_r50c4_tiny"], "cpu")
        model = build_model(cfg)
        dp = DataParallel(model, slabs=3, backend_stream=False)
        opt = build_optimizer(cfg, model)
        d1, k1 = model.roi_heads.box_head.fc1.weight.shape
        opt.enable_pipelined(dp, slab_rows=[16, 40, d1], comm_dtype=torch.bfloat16, exchange="sharded")
        e = model.roi_heads._engine
        o, c = e._seg["fc1.weight"]
        opt._mom = torch.zeros_like(e.arena_w)
        # ---- 1. both ranks: a bare model.state_dict() gathers the owners' rows (no optimizer in sight) ----------------
        r0 = 0
        for r1 in opt._slab_ends:
            a, b = opt._own_rows(("fc1", r0, r1))
            e.arena_w[o + a * k1: o + b * k1] = 10.0 * (rank + 1)  # stand-in for the owned-shard SGD kernel
            r0 = r1
        opt._master_stale = True
        sd = model.state_dict()
        w = sd["roi_heads.box_head.fc1.weight"]
        r0 = 0
        for r1 in opt._slab_ends:
            nq = (r1 - r0) // 2
            assert torch.equal(w[r0: r0 + nq], torch.full((nq, k1), 10.0)), "rank 0's rows"
            assert torch.equal(w[r0 + nq: r1], torch.full((nq, k1), 20.0)), "rank 1's rows"
            r0 = r1
        assert not opt._master_stale
        dist.barrier()
        # ---- 2. the reference's pattern: rank 0 alone checkpoints -> a DrnError within the deadline, not a hang -----
        opt._master_stale = True
        opt.sync_timeout = 3.0
        if rank == 0:
            ck = DetectionCheckpointer(model, save_dir="", optimizer=opt)
            try:
                ck.save("model_final")
                q.put((rank, "FAIL: a lone save() returned"))
            except DrnError as ex:
                assert "EVERY" in str(ex) and "rank 0" in str(ex), str(ex)
                q.put((rank, "ok"))
        else:
            import time

            time.sleep(6.0)  # never joins the collective
            q.put((rank, "ok"))
    except Exception as ex:  # noqa: BLE001
        import traceback

        q.put((rank, "FAIL: %r\n%s" % (ex, traceback.format_exc())))
    finally:
        q.close()
        q.join_thread()  # flush the result before the hard exit
        os._exit(0)  # the group is poisoned by design (a pending all-reduce on rank 0): no orderly teardown


def test_sharded_state_dict_gathers_and_lone_save_fails_fast():
    """(1) model.state_dict() alone must not write stale rows in the sharded exchange; (2) a rank-0-only
    save() - what a driver ported from the reference does (defaults.py:352) - must raise instead of deadlocking."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lone_saver, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
