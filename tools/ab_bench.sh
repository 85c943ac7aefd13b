#!/bin/bash
# interleaved A/B of bench.py argument sets on one box: tools/ab_bench.sh <rounds> <steps> "<args A>" "<args B>" ...
rounds=$1; steps=$2; shift 2
for r in $(seq $rounds); do
  i=0
  for a in "$@"; do
    python bench.py --steps $steps --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); rs=d['roofline_step']
print('[%d] %-40s %.1f img/s  %.4f ms  dominant %.0f TF' % ($i, '''$a''', d['value'], d['ms_per_step'], rs.get('dominant_kernel_tflops_in_step', 0)))"
    i=$((i+1))
  done
done
