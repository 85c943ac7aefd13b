#!/bin/bash
# BASELINE's other configs and the larger per-GPU batch as side measurements on one box (bench.py argument sets, 100 steps)
for a in "" "--workload r101c4_k80" "--workload r50dc5 --proposals 4000" "--workload r50c4_fp8" "--workload v16" "--ims-per-gpu 4" "--heads pcl" ""; do
  python bench.py --steps ${STEPS:-100} --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); rs=d['roofline_step']
print('%-44s %7.1f img/s  %.4f ms/step  whole step %.3f of peak, dominant %.0f TF' % ('''$a''', d['value'], d['ms_per_step'], rs['frac'], rs.get('dominant_kernel_tflops_in_step', 0) or 0))"
done
