#!/bin/bash
# The ring schedule of the graphed step (default) against the side stream's wait on the main stream (bench.py --no-ring): the
# losses of the last step must be the same bits (same kernels on the same batches in the same order), then the step rate,
# interleaved on one box.   usage: tools/ring_ab.sh <out.txt> [rounds] [steps]
out=$1; n=${2:-4}; steps=${3:-300}
R=${GRAFT_REPO_ROOT:-/root/repo}
: > $out
get() { python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.1f img/s  %.4f ms  losses %s' % (d['value'], d['ms_per_step'], ' '.join('%s=%.9g' % (k, v) for k, v in sorted(d['losses_last_step'].items()))))"; }
for i in $(seq 1 $n); do
  for ring in "--no-ring" ""; do
    echo "[$i] ${ring:-ring}: $(timeout 300 python $R/bench.py --no-cpu-baseline --no-side --steps $steps $ring 2>/dev/null | get)" | tee -a $out
  done
done
