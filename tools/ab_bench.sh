#!/bin/bash
# Same-box A/B of two bench.py argument sets, interleaved: tools/ab_bench.sh <tag> "<args A>" "<args B>" [rounds]
tag=$1; A=$2; B=$3; n=${4:-3}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${tag}_ab.txt
mkdir -p $R/gpurun_out
: > $O
get() { python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.1f img/s  %.4f ms' % (d['value'], d['ms_per_step']))"; }
for i in $(seq 1 $n); do
  echo "A[$i] ($A): $(timeout 300 python $R/bench.py --no-cpu-baseline $A 2>/dev/null | get)" | tee -a $O
  echo "B[$i] ($B): $(timeout 300 python $R/bench.py --no-cpu-baseline $B 2>/dev/null | get)" | tee -a $O
done
