#!/bin/bash
# A/B of bench.py variants on ONE box (same process sequence, 300 steps each): tools/ab_bench.sh out.txt "args1" "args2" ...
out=$1; shift
: > $out
for a in "$@"; do
  python bench.py --no-cpu-baseline --steps 300 --no-launch-timing $a 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s %7.1f img/s  %.4f ms/step  host %.3f (unblocked %.3f)' % ('$a', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d.get('host_ms_per_step_unblocked') or 0))" >> $out 2>&1
done
cat $out
