"""Collective self-test on N devices of one node, with the bucket sizes of the BASELINE configs[1] training step:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/rccl_selftest.py
  (options: --backend gloo --single-device to exercise the flow where only one GPU - or none - is available)
Every rank prints its device uuid; rank 0 prints one JSON line with the time and bus bandwidth of every collective the
step issues (all-reduce of the small bucket, reduce-scatter / all-gather per fc6 row slab).  A collective that does not
complete within --timeout raises instead of hanging (DataParallel.selftest).  bench.py --gpus N runs the same test
before its warm-up."""
import argparse
import json
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool: RCCL's peer mappings need it

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package  # noqa: E402


class _Eng:
    arena_w = None


class _Model:  # DataParallel only needs the head engine's device here
    class roi_heads:  # noqa: N801
        _engine = _Eng()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--single-device", action="store_true")
    ap.add_argument("--cpu", action="store_true", help="gloo on host tensors (no GPU needed)")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--timeout", type=float, default=120.0)
    ap.add_argument("--d1", type=int, default=2048)
    ap.add_argument("--k1", type=int, default=50176)
    ap.add_argument("--small", type=int, default=9_437_184 + 4096 + 2048 + 103 * 4097, help="elements of the small bucket")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if not args.cpu:
        torch.cuda.set_device(local)
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    load_package()
    from drn_wsod_pytorch_amd.engine import DataParallel

    dev = torch.device("cpu") if args.cpu else torch.device("cuda:%d" % local)
    _Eng.arena_w = torch.zeros(1, device=dev)
    dp = DataParallel.__new__(DataParallel)
    dp.model, dp.engine, dp.group, dp.world = _Model, _Eng, None, world
    if not args.cpu:
        props = torch.cuda.get_device_properties(local)
        print("[selftest] rank %d pid %d cuda:%d %s uuid %s backend %s" % (rank, os.getpid(), local, props.name,
                                                                         getattr(props, "uuid", ""), args.backend), file=sys.stderr, flush=True)
    half = ((args.d1 + 255) // 256 + 1) // 2 * 256
    slabs = [(half, args.k1), (args.d1 - half, args.k1)] if 0 < half < args.d1 else [(args.d1, args.k1)]
    res = dp.selftest({"small": args.small, "slabs": slabs}, iters=args.iters, timeout=args.timeout,
                      wire_dtype=torch.float32 if args.cpu else torch.bfloat16)
    if rank == 0:
        print(json.dumps({"world": world, "backend": args.backend, "collectives": res}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
