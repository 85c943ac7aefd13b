#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "pp8 or linear_act or bias_act" > $O/r6_07_tests.log 2>&1; tail -3 $O/r6_07_tests.log
timeout 300 python tools/graph_gap_probe.py 2>&1 | grep -v amdgpu.ids > $O/r6_07_graph_gap.txt; cat $O/r6_07_graph_gap.txt
PROF=1 VARIANTS=1 timeout 300 python tools/pp8_probe.py fc7 2>&1 | grep -v amdgpu.ids | grep -v "wave [1-35-7]" > $O/r6_07_pp8_probe.txt; cat $O/r6_07_pp8_probe.txt
timeout 300 python tools/linear_bench.py 2>&1 | grep -v amdgpu.ids > $O/r6_07_linear.txt; grep "5 stages" $O/r6_07_linear.txt
