#!/bin/bash
# round 6, first GPU session: the eight-wave kernel's parity tests, the per-layer conv tables with / without it, fc7
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "pp8 or linear_act or conv_ring or conv1x1_pp" > $O/r6_01_tests.log 2>&1; tail -5 $O/r6_01_tests.log
for wl in r50dc5 r50c4; do
  for t in "25=0" "25=2,26=3" "25=2,26=4" "25=2,26=5" "25=1"; do
    echo "== $wl DRN_TUNE=$t" >> $O/r6_01_conv_800.txt
    DRN_TUNE=$t timeout 300 python tools/conv_bench.py 800 1216 --workload $wl >> $O/r6_01_conv_800.txt 2>&1
  done
done
timeout 300 python tools/linear_bench.py > $O/r6_01_linear.txt 2>&1
cat $O/r6_01_linear.txt
