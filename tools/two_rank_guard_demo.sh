#!/bin/bash
# The driver's N > 1 launch line on ONE GPU over gloo (RCCL refuses two ranks per device): the K-sharded default, then the same
# with a failure injected into the K-shard collectives' self-test / into its warm-up (DRN_BENCH_FAIL) - the run must fall back
# to the sharded gradient exchange and still print its JSON line.  usage: tools/two_rank_guard_demo.sh <out-prefix> [ranks]
out=$1; n=${2:-2}
run() {
  tag=$1; shift
  env "$@" timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus $n --steps 10 --warmup 2 --backend gloo --single-device --no-cpu-baseline > ${out}_${tag}.json 2> ${out}_${tag}.err
  echo "== $tag rc=$?"
  grep "\[bench\] fc6_kshard\|\[bench\] exchange" ${out}_${tag}.err | head -5
  python - <<PY
import json
try:
    j = json.loads(open("${out}_${tag}.json").read().strip().splitlines()[-1])
    g = j["grad_exchange"]
    print("value %.1f img/s  ms/step %.2f  exchange %s  fallback_from %s  other %s" % (j["value"], j["ms_per_step"], g["exchange"], g["fallback_from"], g["other_exchange"]))
    print("selftest:", sorted(g["selftest"].keys()) if g["selftest"] else None)
except Exception as e:
    print("no JSON line:", repr(e))
PY
}
run default DRN_X=1
run fail_selftest DRN_BENCH_FAIL=kshard_selftest
run fail_warmup DRN_BENCH_FAIL=fc6_kshard_warmup
