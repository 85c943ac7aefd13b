#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "pp8 or linear_act or bias_act or conv" > $O/r6_05_tests.log 2>&1; tail -3 $O/r6_05_tests.log
rm -f $O/r6_05_conv_800.txt
for wl in r50dc5 r50c4; do
  for t in "25=0" "25=1" "25=2"; do
    echo "== $wl DRN_TUNE=$t" >> $O/r6_05_conv_800.txt
    DRN_TUNE=$t timeout 300 python tools/conv_bench.py 800 1216 --workload $wl 2>&1 | grep -v amdgpu.ids >> $O/r6_05_conv_800.txt
  done
done
PROF=1 VARIANTS=1 timeout 300 python tools/pp8_probe.py fc7 res5_3x3 2>&1 | grep -v amdgpu.ids > $O/r6_05_pp8_probe.txt
timeout 300 python tools/linear_bench.py 2>&1 | grep -v amdgpu.ids > $O/r6_05_linear.txt
cat $O/r6_05_linear.txt
timeout 600 python bench.py --no-side > $O/r6_05_bench.json 2> $O/r6_05_bench.err; tail -c 1500 $O/r6_05_bench.json
