#!/bin/bash
# the 4x-expansion 1x1 convs (+ residual) of the DC5 trunk under every kernel that can run them
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/r6_33_conv3.txt
for t in "" "24=0" "24=0,25=0" "24=0,25=0,23=64" "24=0,25=0,23=128" "24=0,25=0,23=0" "24=0,25=2"; do
  echo "== DRN_TUNE=$t" >> $O/r6_33_conv3.txt
  DRN_TUNE="$t" CONV_ONLY="conv3" timeout 300 python tools/conv_bench.py 800 1216 --workload r50dc5 2>&1 | grep -v "amdgpu.ids\|^sum\|pixels" >> $O/r6_33_conv3.txt
done
cat $O/r6_33_conv3.txt
