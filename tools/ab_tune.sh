#!/bin/bash
# interleaved A/B of DRN_TUNE settings: ab_env.sh <rounds> <steps> "<tune A>" "<tune B>" ...
rounds=$1; steps=$2; shift 2
for r in $(seq $rounds); do
  for a in "$@"; do
    DRN_TUNE="$a" python bench.py --steps $steps --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); rs=d['roofline_step']
print('DRN_TUNE=%-12s %.1f img/s  %.4f ms  dominant %.0f TF' % ('''$a''', d['value'], d['ms_per_step'], rs.get('dominant_kernel_tflops_in_step', 0)))"
  done
done
